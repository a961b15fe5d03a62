mkdir -p gpurun_out/r6b
for v in 1 0; do
  TDTK_ICP_DEVICE_LOOP=$v python bench.py --steps 20 --warmup 5 --no-cpu --no-normals --no-graphslam-base --no-c5 2> gpurun_out/r6b/small_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loop=$v', d['legs'], d['ms_per_step'])"
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
