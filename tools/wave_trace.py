"""parse `WTRACE block wave xcd t0 t1` lines (TDTK_WAVE_TRACE=<launch index>, 100 MHz clock) from stdin: when do the
waves of one k_search launch end, per XCD"""
import re, sys
import numpy as np
pat = re.compile(r"^WTRACE (\d+) (\d+) (\d+) (\d+) (\d+)$")
rows = [list(map(int, m.groups())) for m in (pat.match(l.strip()) for l in sys.stdin) if m]
a = np.array(rows, dtype=np.int64)
if len(a) == 0:
    print("no trace"); sys.exit()
t0, t1 = a[:, 3].min(), a[:, 4].max()
span = (t1 - t0) / 100.0
print("%d waves, launch span %.1f us; wave lifetime mean %.1f us (%.0f %% of the span); starts within %.1f us" %
      (len(a), span, (a[:, 4] - a[:, 3]).mean() / 100.0, 100.0 * (a[:, 4] - a[:, 3]).mean() / (t1 - t0), (a[:, 3].max() - t0) / 100.0))
end = (a[:, 4] - t0) / 100.0
print("wave end time percentiles (us): " + "  ".join("p%d %.1f" % (p, np.percentile(end, p)) for p in (1, 10, 25, 50, 75, 90, 99, 100)))
for x in range(8):
    e = end[(a[:, 2] & 7) == x]
    if len(e): print("XCD %d: %4d waves, last ends %.1f us, median %.1f us" % (x, len(e), e.max(), np.median(e)))
