# GPU box: counters of the 10M-query whole-scan pass with the results staged in LDS / stored at retire (lab library both)
cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload c5 --no-cpu --c5-scans 2 --c5-links 1"
for mode in 1 0; do
  for c in "WRITE_SIZE FETCH_SIZE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    rm -rf /tmp/sp; TDTK_LIB=lab TDTK_STAGE_RESULTS=$mode timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/sp -o p -- $CMD > /dev/null 2>&1
    MODE=$mode python - <<'PY'
import csv, collections, os
rows = sorted(csv.DictReader(open("/tmp/sp/p_counter_collection.csv")), key=lambda r: int(r["Dispatch_Id"]))
# the whole-scan passes: the first five dispatches of the un-instrumented single-pass kernel
seen = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "k_search_refill<" not in n: continue
    a = [t.strip() for t in n.split("k_search_refill<")[1].split(">")[0].split(",")]
    if a[4] == "true": continue
    seen.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
first = list(seen.values())[:5]
out = collections.defaultdict(list)
for d in first:
    for k, v in d.items(): out[k].append(v)
print("staged=%s" % os.environ["MODE"], {k: round(sum(v) / len(v), 1) for k, v in out.items()})
PY
  done
done
