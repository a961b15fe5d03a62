"""gpurun_out/<tag>/{gs,gs_fetch,gs_write}/ (tools/profile_graphslam.sh) -> profiles/<prefix>_graphslam_kernel_stats.csv and
profiles/<prefix>_graphslam_pmc.json:  python tools/summarize_graphslam_profile.py <tag> <round-prefix>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("tdtk::", "")
    if n.startswith("k_search_refill<"):
        a = [t.strip() for t in n[len("k_search_refill<"):].split(">")[0].split(",")]
        return "k_search_count(instrumented, not timed)" if len(a) >= 5 and a[4] == "true" else "k_search"
    if n.startswith("k_search_refill_multi<"):
        a = [t.strip() for t in n[len("k_search_refill_multi<"):].split(">")[0].split(",")]
        return "k_search_count(instrumented, not timed)" if len(a) >= 5 and a[4] == "true" else "k_search (several links per launch)"
    if n.startswith("k_search<"):
        return "k_search(one query per lane)"
    return n[:80]

rows = list(csv.DictReader(open(os.path.join(src, "gs", "p_kernel_trace.csv"))))
agg = collections.defaultdict(list)
for r in rows:
    agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
with open(os.path.join("profiles", pre + "_graphslam_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%.3f,%.3f,%.3f,%.3f,%.2f\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
pmc = {"command": "python bench.py --workload graphslam --steps 10 --warmup 3", "kernels": {},
       "note": "per-dispatch averages over all dispatches of the run (a search dispatch covers up to 64 link passes of 1M queries each: 84 links = one of 64 + one of 20 per step); FETCH_SIZE / WRITE_SIZE in KiB"}
for p in ("gs_fetch", "gs_write"):
    fn = os.path.join(src, p, "p_counter_collection.csv")
    if not os.path.exists(fn):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if not k.startswith("k_"):
            continue
        name = c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")
        pmc["kernels"].setdefault(k, {})[name] = sum(v) / len(v)
        pmc["kernels"][k]["dispatches_" + p] = len(v)
json.dump(pmc, open(os.path.join("profiles", pre + "_graphslam_pmc.json"), "w"), indent=1, sort_keys=True)
fn = os.path.join(src, "gs.json")
if os.path.exists(fn):
    open(os.path.join("profiles", pre + "_graphslam_bench_under_rocprof.json"), "w").write(open(fn).read())
print(open(os.path.join("profiles", pre + "_graphslam_kernel_stats.csv")).read()[:1500])
print(json.dumps(pmc["kernels"].get("k_search (several links per launch)", pmc["kernels"].get("k_search", {})), indent=1))
