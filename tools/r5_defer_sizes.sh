#!/usr/bin/env bash
# GPU box: where the deferred quick check stops paying -- ICP pairs of 2M / 4M points (trees of 115 / 230 MB), lab library with the
# size limit lifted (TDTK_DEFER_MAX_MB), check deferred against made
for n in 2000000 4000000; do for dc in 1 0; do echo "points $n TDTK_DEFER_CHECK=$dc"; TDTK_LIB=lab TDTK_DEFER_MAX_MB=100000 TDTK_DEFER_CHECK=$dc python bench.py --points $n --steps 20 --warmup 5 --no-c5 --no-cpu --no-normals --no-small-scans --no-graphslam-base 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('  ', d['ms_per_step'], r.get('kernel_ms'), r.get('visits_per_query'), d['config']['tree'])"; done; done
