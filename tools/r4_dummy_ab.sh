#!/bin/bash
# the group no bucket owns (+inf points): parity subset, then the bench three times (product library)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4dummy; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "k5 or float_range or far_from or pair_sums_inside or full_size_icp or lattice or alternative or big_clouds or duplicate or bucket" > $O/parity.log 2>&1
tail -3 $O/parity.log
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/b_$i.json 2>$O/b_$i.err
python -c "import json;d=json.load(open('$O/b_$i.json'));print('s20 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
done
python bench.py --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal > $O/c.json 2>$O/c.err
python -c "import json;d=json.load(open('$O/c.json'));print('s100 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
python bench.py --workload graphslam --no-cpu > $O/g.json 2>$O/g.err
python -c "import json;d=json.load(open('$O/g.json'));print('gs ms_per_step %.4f value %.3e' % (d['ms_per_step'], d['value']))"
