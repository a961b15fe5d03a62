"""gpurun_out/<tag>/{c5,c5_*}/ (tools/profile_c5.sh) -> profiles/<prefix>_c5_kernel_stats.csv, profiles/<prefix>_c5_pmc.json,
profiles/<prefix>_c5_bench_under_rocprof.json:   python tools/summarize_c5_profile.py <tag> <round-prefix>

The command runs, in this order: whole-scan passes (k_search_refill), one instrumented replay, K ICP iterations (k_search_refill
again), K instrumented ones, calcNormals, lum6DEuler rounds (k_search_refill_multi).  The instrumented launches (COUNT = true in
the kernel's template arguments) separate the first two groups of k_search_refill launches."""
import collections, csv, json, os, sys
tag, pre = sys.argv[1], sys.argv[2]
src = tag if os.path.isdir(tag) else os.path.join("gpurun_out", tag)      # (a directory, or a tag under gpurun_out/)
OUTDIR = sys.argv[3] if len(sys.argv) > 3 else "profiles"                   # on the GPU box: gpurun_out/<something> (the raw CSVs are too big to travel)
os.makedirs(OUTDIR, exist_ok=True)


def targs(n, head):
    return [t.strip() for t in n[len(head):].split(">")[0].split(",")]


def base(n):
    return n.split("(")[0].replace("void ", "").replace("tdtk::", "")


def label_rows(rows):
    """rows in dispatch order -> list of labels"""
    out = []
    seen_count = 0          # instrumented k_search_refill launches seen so far
    in_count_run = False
    groups = 0
    for r in rows:
        n = base(r["Kernel_Name"])
        if n.startswith("k_search_refill_multi<"):
            a = targs(n, "k_search_refill_multi<")
            out.append("k_search_count(instrumented, not timed)" if len(a) >= 5 and a[4] == "true" else "k_search (several links per launch)")
        elif n.startswith("k_search_refill<"):
            a = targs(n, "k_search_refill<")
            if len(a) >= 5 and a[4] == "true":
                if not in_count_run:
                    groups += 1
                in_count_run = True
                out.append("k_search_count(instrumented, not timed)")
            else:
                in_count_run = False
                out.append("k_search [whole-scan pass, 10M queries]" if groups == 0 else
                           ("k_search [icp6D::match at 10M]" if groups == 1 else "k_search (other)"))
        elif n.startswith("k_ann_normals<"):
            out.append("k_ann_normals_count(instrumented, not timed)" if "true" in n else "k_ann_normals<10>")
        elif n.startswith("k_search"):
            out.append("k_search (small batches)")
        else:
            out.append(n[:90])
    return out


rows = sorted(csv.DictReader(open(os.path.join(src, "c5", "p_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
labs = label_rows(rows)
agg = collections.defaultdict(list)
for r, l in zip(rows, labs):
    agg[l].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
with open(os.path.join(OUTDIR, pre + "_c5_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%.3f,%.3f,%.3f,%.3f,%.2f\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
try:
    cmd = open(os.path.join(src, "command.txt")).read().strip()
except Exception:
    cmd = "python bench.py --workload c5 --no-cpu"
pmc = {"command": cmd, "kernels": {},
       "note": "per-dispatch averages by kernel and phase of the command; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: "
               "FETCH_SIZE reads half the streamed bytes, bench.py pmc_traffic_bytes); SQ_* in quad-cycles where rocprofv3 says so; "
               "GRBM_GUI_ACTIVE summed over the 8 XCDs"}
for p in sorted(os.listdir(src)):
    fn = os.path.join(src, p, "p_counter_collection.csv")
    if not p.startswith("c5_") or not os.path.exists(fn):
        continue
    rr = list(csv.DictReader(open(fn)))
    # one row per (dispatch, counter): label the dispatches in order
    disp = {}
    for r in rr:
        disp.setdefault(int(r["Dispatch_Id"]), r)
    order = [disp[i] for i in sorted(disp)]
    lab = dict(zip(sorted(disp), label_rows(order)))
    acc = collections.defaultdict(list)
    for r in rr:
        acc[(lab[int(r["Dispatch_Id"])], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if not (k.startswith("k_search") or k.startswith("k_ann") or k.startswith("k_accum")):
            continue
        name = c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")
        pmc["kernels"].setdefault(k, {})[name] = sum(v) / len(v)
        pmc["kernels"][k]["dispatches_" + p] = len(v)
for k, d in pmc["kernels"].items():
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_ACTIVE_INST_VALU"):
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        e = {"kernel_cycles": cyc, "valu_busy": d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * cyc)}
        if d.get("SQ_THREAD_CYCLES_VALU"): e["lane_efficiency"] = d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])
        if d.get("SQ_WAIT_ANY") and d.get("SQ_WAVE_CYCLES"): e["wave_wait_share"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
        if d.get("FETCH_SIZE_KiB") is not None and d.get("WRITE_SIZE_KiB") is not None:
            e["fabric_bytes_per_launch"] = (2.0 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024.0
        if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None:
            e["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        if d.get("TCP_GATE_EN2_sum"): e["vector_l1_busy"] = d["TCP_GATE_EN2_sum"] / 256.0 / cyc
        if d.get("TCP_PENDING_STALL_CYCLES_sum"): e["vector_l1_stalled_on_pending_fills"] = d["TCP_PENDING_STALL_CYCLES_sum"] / 256.0 / cyc
        if d.get("TCP_TCC_READ_REQ_LATENCY_sum") and d.get("TCP_TCC_READ_REQ_sum"):
            e["l1_miss_latency_cycles"] = d["TCP_TCC_READ_REQ_LATENCY_sum"] / d["TCP_TCC_READ_REQ_sum"]
        d["derived"] = e
json.dump(pmc, open(os.path.join(OUTDIR, pre + "_c5_pmc.json"), "w"), indent=1, sort_keys=True)
fn = os.path.join(src, "c5.json")
if os.path.exists(fn):
    open(os.path.join(OUTDIR, pre + "_c5_bench_under_rocprof.json"), "w").write(open(fn).read())
raw = os.path.join(src, "c5", "p_kernel_stats.csv")
if os.path.exists(raw):
    open(os.path.join(OUTDIR, pre + "_c5_rocprofv3_kernel_stats_raw.csv"), "w").write(open(raw).read())
print(open(os.path.join(OUTDIR, pre + "_c5_kernel_stats.csv")).read()[:3000])
for k, d in pmc["kernels"].items():
    if "derived" in d:
        print(k, json.dumps(d["derived"], indent=1))
