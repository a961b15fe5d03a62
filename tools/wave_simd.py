"""stdin: the stderr of `TDTK_WAVE_TRACE=<launch index> python tools/icp_probe.py ...` -- the trace carries HW_REG_HW_ID, so:
how many waves of the launch each SIMD / CU hosted, and when the waves of the fuller ones end"""
import re, sys, collections
import numpy as np
pat = re.compile(r"^WTRACE (\d+) (\d+) (\d+) (\d+) (\d+)$")
rows = [list(map(int, m.groups())) for m in (pat.match(l.strip()) for l in sys.stdin) if m]
a = np.array(rows, dtype=np.int64)
t0 = a[:, 3].min(); span = a[:, 4].max() - t0
xcd = a[:, 2] & 7; hw = a[:, 2] >> 8
# gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = [(int(x), int(s_), int(h), int(c), int(d)) for x, s_, h, c, d in zip(xcd, se, sh, cu, simd)]
cnt = collections.Counter(key)
end = (a[:, 4] - t0) / span
start = (a[:, 3] - t0) / span
print("waves", len(a), "distinct SIMD keys", len(cnt), "waves per SIMD histogram", sorted(collections.Counter(cnt.values()).items()))
by = collections.defaultdict(list)
for k, e in zip(key, end): by[cnt[k]].append(e)
for n in sorted(by): print("SIMDs hosting %d waves: %5d waves, end mean %.3f median %.3f max %.3f" % (n, len(by[n]), np.mean(by[n]), np.median(by[n]), np.max(by[n])))
cuk = collections.Counter([(k[0], k[1], k[2], k[3]) for k in key])
print("waves per CU histogram", sorted(collections.Counter(cuk.values()).items()), "distinct CUs", len(cuk))
byc = collections.defaultdict(list)
for k, e in zip(key, end): byc[cuk[(k[0], k[1], k[2], k[3])]].append(e)
for n in sorted(byc): print("CUs hosting %2d waves: %5d waves, end mean %.3f median %.3f max %.3f" % (n, len(byc[n]), np.mean(byc[n]), np.median(byc[n]), np.max(byc[n])))
print("late starters (start > 5 %% of span): %d" % int((start > 0.05).sum()))
