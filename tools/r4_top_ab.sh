#!/bin/bash
# A/B: upper tree levels in LDS (TDTK_TOP_BLOCK=0 / 512 / 1024), lab library; parity first
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4top; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "k5 or float_range or far_from or pair_sums_inside or full_size_icp or lattice or alternative or big_clouds" > $O/parity.log 2>&1
tail -3 $O/parity.log
for tb in 0 1024 512 0 1024 512; do
TDTK_LIB=lab TDTK_TOP_BLOCK=$tb python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/b_$tb.json 2>$O/b_$tb.err
python -c "import json;d=json.load(open('$O/b_$tb.json'));print('top $tb s20 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
done
TDTK_LIB=lab TDTK_TOP_BLOCK=0 python bench.py --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/c_0.json 2>$O/c_0.err
TDTK_LIB=lab TDTK_TOP_BLOCK=1024 python bench.py --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/c_1024.json 2>$O/c_1024.err
for tb in 0 1024; do python -c "import json;d=json.load(open('$O/c_$tb.json'));print('top $tb s100 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"; done
