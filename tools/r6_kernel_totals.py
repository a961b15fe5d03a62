"""rocprofv3 kernel trace (a directory that holds *_kernel_trace.csv) -> per-kernel totals as CSV on stdout:
kernel,calls,total_us,avg_us,min_us,max_us,percent   (template arguments kept short; python tools/r6_kernel_totals.py <dir>)"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tdtk::", "")
    if n.startswith("k_search_refill_multi<"): n = "k_search_count(instrumented)" if n.split(",")[4].strip() == "true" else "k_search (several links per launch)"
    elif n.startswith("k_search_refill<"): n = "k_search_count(instrumented)" if n.split(",")[4].strip() == "true" else "k_search_refill"
    elif n.startswith("rocprim"): n = "rocprim::" + ("init_lookback_scan_state" if "init_lookback" in n else "scan / sort kernel")
    agg[n[:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("kernel,calls,total_us,avg_us,min_us,max_us,percent")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.2f' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
