#!/bin/bash
# A/B: slabs handed out by the workgroup's waves together (TDTK_SHARE_BLOCK=0 / 256 / 512 / 1024), lab library; parity first
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4share; mkdir -p $O
TDTK_LIB=lab TDTK_SHARE_BLOCK=512 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "k5 or float_range or far_from or pair_sums_inside or full_size_icp or lattice or big_clouds" > $O/parity.log 2>&1
tail -3 $O/parity.log
for tb in 0 256 512 1024 0 256 512 1024; do
TDTK_LIB=lab TDTK_SHARE_BLOCK=$tb python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/b_$tb.json 2>$O/b_$tb.err
python -c "import json;d=json.load(open('$O/b_$tb.json'));print('share $tb s20 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
done
for tb in 0 512; do
TDTK_LIB=lab TDTK_SHARE_BLOCK=$tb python bench.py --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/c_$tb.json 2>$O/c_$tb.err
python -c "import json;d=json.load(open('$O/c_$tb.json'));print('share $tb s100 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"; done
