#!/usr/bin/env bash
# GPU box: graph-SLAM rounds with every link's searches started from the previous round's hits (product) against cold rounds
# (lab library, TDTK_LINK_WARM=0), and the lab library with the switch on as the control
for v in "product" "lab 1" "lab 0"; do set -- $v; echo "TDTK_LIB=$1 TDTK_LINK_WARM=${2:-}"; TDTK_LIB=$1 TDTK_LINK_WARM=${2:-1} python bench.py --workload graphslam --steps 10 --warmup 3 --no-rehearsal --no-cpu 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:d.get(k) for k in ('link_launch_ms','rest_ms')}, d.get('roofline',{}).get('kernel_ms'), d.get('roofline',{}).get('visits_per_query'))
print({k:d[k] for k in d if 'x_hash' in k or 'pose' in k or 'check' in k})"; done
