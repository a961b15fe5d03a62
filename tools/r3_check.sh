#!/bin/bash
# GPU box: parity suite, the ICP probe at the usual three sizes, then the counters of the driver's command (summarised on the box)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/keep/tests.log
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
{ N=1000000 K=20 W=5 run A=1; N=1000000 K=100 W=10 run A=1; N=4000000 K=30 W=5 run A=1; N=300000 K=100 W=10 run A=1; } | tee gpurun_out/keep/probe.log
if [ "${1:-}" != "noprof" ]; then
  set -- 20 5
  tag="prof_s$1_w$2"
  bash tools/profile_bench.sh $tag $1 $2 > gpurun_out/keep/$tag.log 2>&1
  python tools/summarize_profiles.py $tag r03 > gpurun_out/keep/$tag.summary.txt 2>&1
  rm -rf gpurun_out/$tag
  cp profiles/r03_*s20_w5* gpurun_out/keep/
  python - <<'PY'
import json
d=json.load(open('profiles/r03_pmc_bench_s20_w5.json')); k=d['kernels']['k_search [timed region]']
cyc=k['GRBM_GUI_ACTIVE']/8
print('k_search timed region: VALU insts %.2fM  lane eff %.3f  valu busy %.3f  vmem %.2fM  tcp acc %.1fM  kernel cycles %.0f' % (k['SQ_INSTS_VALU']/1e6, k['SQ_THREAD_CYCLES_VALU']/(k['SQ_ACTIVE_INST_VALU']*64), k['SQ_ACTIVE_INST_VALU']*4/(1024*cyc), k['SQ_INSTS_VMEM_RD']/1e6, k.get('TCP_TOTAL_CACHE_ACCESSES_sum',0)/1e6, cyc))
PY
  grep "timed region" profiles/r03_kernel_stats_s20_w5.csv
fi
