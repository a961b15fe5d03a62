# GPU box: the several-links launch -- configs[3] (84 links of 1M points) and the configs[4] shape (links of 10M points)
python bench.py --workload graphslam --gpus 1 --steps 6 --warmup 2 --no-rehearsal --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graphslam 84 links: ms_per_step %.3f link launch %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
python bench.py --workload c5 --no-cpu --c5-scans 3 2>/dev/null > /dev/null
python - <<PY
import json
d=json.load(open("bench_legs.json"))["c5_shape_1gpu"]
print("c5 (3 scans): pass %.4f icp_10M k_search %.4f  lum round %.3f ms, link launch %.4f (%d links)" % (d["whole_scan_pass"]["k_search_ms"], d["icp_10M"]["k_search_ms"], d["lum_round"]["ms"], d["lum_round"]["link_launch_ms"], d["lum_round"]["links"]))
PY
