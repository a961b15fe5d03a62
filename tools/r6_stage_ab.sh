# GPU box: the headline loop and the 10M legs with the results staged in LDS / stored at retire -- both in the lab library
# (TDTK_STAGE_RESULTS is a lab switch), and the product library beside them
run() {
  for rep in 1 2; do
  env $2 python bench.py --steps 20 --warmup 5 --no-cpu --no-normals --no-small-scans --no-graphslam-base --no-c5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 icp 1M: ms_per_step %.4f k_search %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
  env $2 python bench.py --workload c5 --no-cpu --c5-scans 2 --c5-links 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 c5: pass k_search %.4f ms' % d['roofline']['kernel_ms'])"
  python - <<PY
import json
d=json.load(open("bench_legs.json"))["c5_shape_1gpu"]
print("   icp_10M k_search %.4f ms/iter %.4f  link launch %.4f" % (d["icp_10M"]["k_search_ms"], d["icp_10M"]["ms_per_iteration"], d["lum_round"]["link_launch_ms"]))
PY
}
run "lab staged " "TDTK_LIB=lab"
run "lab at-retire" "TDTK_LIB=lab TDTK_STAGE_RESULTS=0"
run "product    " "X=1"
