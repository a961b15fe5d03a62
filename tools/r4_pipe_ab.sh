#!/bin/bash
# A/B: pipelined hand-out (TDTK_PIPE=0 / 1), lab library; parity first
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4pipe; mkdir -p $O
TDTK_LIB=lab TDTK_PIPE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "k5 or float_range or far_from or pair_sums_inside or full_size_icp or lattice or big_clouds or lum_links_fused or company" > $O/parity.log 2>&1
tail -3 $O/parity.log
for tb in 0 1 0 1; do
TDTK_LIB=lab TDTK_PIPE=$tb python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/b_$tb.json 2>$O/b_$tb.err
python -c "import json;d=json.load(open('$O/b_$tb.json'));print('pipe $tb s20 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
done
for tb in 0 1; do
TDTK_LIB=lab TDTK_PIPE=$tb python bench.py --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/c_$tb.json 2>$O/c_$tb.err
python -c "import json;d=json.load(open('$O/c_$tb.json'));print('pipe $tb s100 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
TDTK_LIB=lab TDTK_PIPE=$tb python bench.py --workload graphslam --no-cpu > $O/g_$tb.json 2>$O/g_$tb.err
python -c "import json;d=json.load(open('$O/g_$tb.json'));print('pipe $tb gs ms_per_step %.4f value %.3e' % (d['ms_per_step'], d['value']))"
done
