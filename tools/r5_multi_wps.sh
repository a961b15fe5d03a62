#!/usr/bin/env bash
# GPU box: the several-links launch (graph-SLAM) compiled for more waves per SIMD in the lab library (LABFLAGS=-DTDTK_MULTI_WPS=5|6) against the product's four
for v in product lab product lab; do echo "TDTK_LIB=$v"; TDTK_LIB=$v python bench.py --workload graphslam --steps 10 --warmup 3 --no-rehearsal --no-cpu 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms'))"; done
