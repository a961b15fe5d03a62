#!/usr/bin/env bash
# GPU box: the configs[4]-shape leg (10M-point scans) with the single-pass kernel at six waves per SIMD (product) against five
# (lab library built with LABFLAGS=-DTDTK_REFILL_WPS=1), with and without the deferred quick check
for v in "product 1" "lab 1" "product 0" "lab 0"; do set -- $v; echo "TDTK_LIB=$1 TDTK_DEFER_CHECK=$2"; TDTK_LIB=$1 TDTK_DEFER_CHECK=$2 python bench.py --workload c5 --no-cpu --c5-scans 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d.get('c5_shape_1gpu',d)
for k in ('whole_scan_pass','icp_10M','lum_round'):
    v=c.get(k,{}); print('  ',k,{x:v.get(x) for x in ('ms','k_search_ms','ms_per_iteration','link_launch_ms') if v.get(x) is not None})"; done
