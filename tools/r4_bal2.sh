#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4g
for fx in 6 30 100 400 2000; do
  TDTK_SB_FIX=$fx python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals > gpurun_out/r4g/b20_$fx.json 2>gpurun_out/r4g/b20_$fx.err
  python -c "import json;d=json.load(open('gpurun_out/r4g/b20_$fx.json'));print('fix=$fx s20 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
done
TDTK_BALANCE=0 python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals > gpurun_out/r4g/b20_off.json 2>gpurun_out/r4g/b20_off.err
python -c "import json;d=json.load(open('gpurun_out/r4g/b20_off.json'));print('balance off s20 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
TDTK_WAVE_TRACE=30 TDTK_BALANCE=0 python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals 2>&1 >/dev/null | grep WTRACE > gpurun_out/r4g/wtrace_off.txt
TDTK_WAVE_TRACE=30 TDTK_SB_FIX=30 python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals 2>&1 >/dev/null | grep WTRACE > gpurun_out/r4g/wtrace_on.txt
python - <<'PY'
import numpy as np
for nm in ("off", "on"):
    rows = [l.split() for l in open("gpurun_out/r4g/wtrace_%s.txt" % nm)]
    if not rows: print(nm, "no trace"); continue
    st = np.array([int(r[4]) for r in rows], float); en = np.array([int(r[5]) for r in rows], float)
    ok = en > 0
    st, en = st[ok], en[ok]
    t0 = st.min(); dur = en.max() - t0
    print(nm, "waves", len(st), "launch %.1f us" % (dur / 100.0), "wave end quantiles (share of launch): 5%% %.2f 25%% %.2f 50%% %.2f 75%% %.2f 95%% %.2f" % tuple(np.quantile((en - t0) / dur, [0.05, 0.25, 0.5, 0.75, 0.95])),
          "mean wave duration %.2f of launch" % ((en - st).mean() / dur))
PY
