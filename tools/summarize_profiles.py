"""Turn the rocprofv3 CSVs merged back under gpurun_out/<tag>/ into the small, tracked
summaries under profiles/:  python tools/summarize_profiles.py <tag> <round-prefix>
The summary carries the command, steps and warmup of the profiled run; its name does too (<prefix>_pmc_bench_s<steps>_w<warmup>.json):
bench.py only takes counters from a summary of ITS OWN steps / warmup."""
import collections, csv, json, os, sys
tag, pre = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)

def short(n):
    n = n.split("(")[0].replace("void ", "").replace("tdtk::", "")
    if n.startswith("k_ann_normals<"):
        return "k_ann_normals_count(instrumented, not timed)" if "true" in n else "k_ann_normals"
    if n.startswith("k_search_refill<"):
        a = [t.strip() for t in n[len("k_search_refill<"):].split(">")[0].split(",")]
        # <BLOCK, SD, THRESH, WPS, COUNT, FUSE, DYN>
        if len(a) >= 5 and a[4] == "true":
            return "k_search_count(instrumented, not timed)"
        return "k_search"
    if n.startswith("k_search<"):
        a = [t.strip() for t in n[len("k_search<"):-1].split(",")]
        # <BLOCK, SD, COUNT, DIRMODE, UNI, WPS>
        if a[2] == "true":
            return "k_search_count(instrumented, not timed)"
        if a[3] != "0":
            return "k_search_dir"
        return "k_search"
    return n

def timed_region(passname):
    """(first, count) of the bench's timed search launches within a pass: bench.py prints how many search launches
    precede its timed region (settle rounds + warm-up) and how many steps it timed"""
    try:
        d = json.loads(open(os.path.join(src, passname + ".json")).read().strip().splitlines()[-1])
        return int(d["search_launches_before_timed_region"]), int(d["steps"])
    except Exception:
        return 10, 100

def run_args():
    d = json.loads(open(os.path.join(src, "stats.json")).read().strip().splitlines()[-1])
    return int(d["steps"]), int(d["warmup"])
STEPS, WARM = run_args()
SUF = "_s%d_w%d" % (STEPS, WARM)
try:
    COMMAND = open(os.path.join(src, "command.txt")).read().strip().replace(os.environ.get("GRAFT_REPO_ROOT", "/nonexistent") + "/", "")
except Exception:
    COMMAND = "python bench.py --steps %d --warmup %d --no-cpu --no-graphslam-base --no-normals --no-small-scans" % (STEPS, WARM)

# --- kernel stats (from the kernel trace of the --stats run)
rows = list(csv.DictReader(open(os.path.join(src, "stats", "p_kernel_trace.csv"))))
agg = collections.defaultdict(list)
for r in rows:
    agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
# the bench's timed region = launches 11..110 of the resident-loop search kernel (10 warm-up launches before,
# the host-buffer legs after); listed separately so it can be compared with bench.py's HIP-event average
loop = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows
        if short(r["Kernel_Name"]) == "k_search" and "k_search_refill<" in r["Kernel_Name"]]
t_first, t_n = timed_region("stats")
if len(loop) >= t_first + t_n:
    agg["k_search [timed region: launches %d-%d]" % (t_first + 1, t_first + t_n)] = loop[t_first:t_first + t_n]
tot = sum(sum(v) for k, v in agg.items() if not k.startswith("k_search ["))
with open(os.path.join("profiles", pre + "_kernel_stats" + SUF + ".csv"), "w") as f:
    f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%.3f,%.3f,%.3f,%.3f,%.2f\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
stats_file = os.path.join(src, "stats", "p_kernel_stats.csv")
if os.path.exists(stats_file):
    open(os.path.join("profiles", pre + "_rocprofv3_kernel_stats_raw" + SUF + ".csv"), "w").write(open(stats_file).read())

# --- PMC passes
pmc = {"command": COMMAND, "steps": STEPS, "warmup": WARM, "kernels": {},
       "note": "per-dispatch averages; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE "
               "reads 1/2 of the streamed bytes (calibrated: k_transform reads 3 x 8 MB = 23437.5 KiB, reports ~11738)"}
for p in ("fetch", "write", "sq1", "sq2", "sq3", "tcc", "tcp", "tcp2", "tcp3"):
    fn = os.path.join(src, p, "p_counter_collection.csv")
    if not os.path.exists(fn):
        continue
    acc = collections.defaultdict(list)
    seen = collections.defaultdict(dict)      # the timed region of the bench = the 11th..110th k_search dispatch
    for r in sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Dispatch_Id"])):
        k = short(r["Kernel_Name"])
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        if k == "k_search" and "k_search_refill<" in r["Kernel_Name"]:
            seen[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    for (k, c), v in acc.items():
        name = c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")
        pmc["kernels"].setdefault(k, {})[name] = sum(v) / len(v)
        pmc["kernels"][k]["dispatches_" + p] = len(v)
    p_first, p_n = timed_region(p)
    for c, d in seen.items():
        vals = [d[i] for i in sorted(d)]
        if len(vals) >= p_first + p_n:
            name = c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")
            pmc["kernels"].setdefault("k_search [timed region]", {})[name] = sum(vals[p_first:p_first + p_n]) / float(p_n)
json.dump(pmc, open(os.path.join("profiles", pre + "_pmc_bench" + SUF + ".json"), "w"), indent=1, sort_keys=True)
for p in ("stats",):
    fn = os.path.join(src, p + ".json")
    if os.path.exists(fn):
        open(os.path.join("profiles", pre + "_bench_under_rocprof" + SUF + ".json"), "w").write(open(fn).read())
print(open(os.path.join("profiles", pre + "_kernel_stats" + SUF + ".csv")).read())
print(json.dumps(pmc["kernels"].get("k_search", {}), indent=1))
