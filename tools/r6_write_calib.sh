cd /tmp; export TMPDIR=/tmp
for c in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/wc; timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/wc -o p -- $GRAFT_REPO_ROOT/tools/micro/write_calib > /dev/null 2>&1
  python - <<'PY'
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/wc/p_counter_collection.csv")):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-14s %-24s %14.1f per launch" % (k[0], k[1], sum(v) / len(v)))
PY
done
