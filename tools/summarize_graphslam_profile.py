"""gpurun_out/<tag>/{gs,gs_fetch,gs_write}/ (tools/profile_graphslam.sh) -> profiles/<prefix>_graphslam_kernel_stats.csv and
profiles/<prefix>_graphslam_pmc.json:  python tools/summarize_graphslam_profile.py <tag> <round-prefix>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
suffix = os.environ.get("GS_SUFFIX", "")


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
with open(os.path.join("profiles", pre + "_graphslam" + suffix + "_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%.3f,%.3f,%.3f,%.3f,%.2f\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
cmd = os.environ.get("GS_CMD_LABEL", "python bench.py --workload graphslam --steps 10 --warmup 3 --no-rehearsal")
pmc = {"command": cmd, "kernels": {},
       "note": "per-dispatch averages over all dispatches of the run; one search dispatch covers ALL link passes of the step "
               "(84 links of 1M queries each in the full graph, a rank's 11 in the share run: up to 128 links per launch); "
               "FETCH_SIZE / WRITE_SIZE in KiB (gfx950: FETCH_SIZE reads half the streamed bytes, see bench.py pmc_traffic_bytes); "
               "SQ_* / GRBM_GUI_ACTIVE as rocprofv3 reports them (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles; "
               "GRBM_GUI_ACTIVE summed over the 8 XCDs)"}
for p in ("gs_fetch", "gs_write", "gs_sq1", "gs_sq2", "gs_tcp", "gs_tcp2", "gs_tcp3"):
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
# derived figures of the link launch, from the counters above
k = pmc["kernels"].get("k_search (several links per launch)")
if k and k.get("GRBM_GUI_ACTIVE") and k.get("SQ_ACTIVE_INST_VALU"):
    cyc = k["GRBM_GUI_ACTIVE"] / 8.0
    d = {"kernel_cycles": cyc, "valu_busy": k["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * cyc),
         "what": "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); lane_efficiency = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU); "
                 "wave_wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES; fabric = 2 FETCH_SIZE + WRITE_SIZE"}
    if k.get("SQ_THREAD_CYCLES_VALU"):
        d["lane_efficiency"] = k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_ACTIVE_INST_VALU"])
    if k.get("SQ_WAIT_ANY") and k.get("SQ_WAVE_CYCLES"):
        d["wave_wait_share"] = k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]
    if k.get("FETCH_SIZE_KiB") is not None and k.get("WRITE_SIZE_KiB") is not None:
        d["fabric_bytes_per_launch"] = (2.0 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0
    if k.get("TCC_HIT_sum") is not None and k.get("TCC_MISS_sum") is not None:
        d["l2_hit_rate"] = k["TCC_HIT_sum"] / max(1.0, k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
    if k.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        d["l1_tag_accesses_per_cu_cycle"] = k["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cyc
    links = int(os.environ.get("GS_LINKS", "84"))
    d["links_per_launch"] = links
    if "fabric_bytes_per_launch" in d:
        d["fabric_GB_per_link"] = d["fabric_bytes_per_launch"] / links / 1e9
    pmc["derived"] = d
json.dump(pmc, open(os.path.join("profiles", pre + "_graphslam" + suffix + "_pmc.json"), "w"), indent=1, sort_keys=True)
fn = os.path.join(src, "gs.json")
if os.path.exists(fn):
    open(os.path.join("profiles", pre + "_graphslam" + suffix + "_bench_under_rocprof.json"), "w").write(open(fn).read())
print(open(os.path.join("profiles", pre + "_graphslam" + suffix + "_kernel_stats.csv")).read()[:1500])
print(json.dumps(pmc.get("derived"), indent=1))
print(json.dumps(pmc["kernels"].get("k_search (several links per launch)", pmc["kernels"].get("k_search", {})), indent=1))
