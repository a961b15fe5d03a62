"""Turn the rocprofv3 kernel trace of tools/profile_normals.sh (gpurun_out/<tag>/normals/) into the tracked summary
profiles/<round-prefix>_normals_kernel_stats.csv:  python tools/summarize_normals_profile.py <tag> <round-prefix>"""
import collections, csv, os, sys
tag, pre = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def short(n):
    return n.split("(")[0].replace("void ", "").replace("tdtk::", "")

# --- Scan::calcNormals (tools/profile_normals.sh): per-kernel totals over the traced runs, per run in the last column
nsrc = os.path.join(src, "normals", "p_kernel_trace.csv")
if os.path.exists(nsrc):
    rows = list(csv.DictReader(open(nsrc)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[short(r["Kernel_Name"])[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    runs = max(1, len(agg.get("k_ann_normals<10>", [])))
    tot = sum(sum(v) for v in agg.values())
    with open(os.path.join("profiles", pre + "_normals_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent,us_per_calcNormals(%d runs of 1M points)\n" % runs)
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write("%s,%d,%.3f,%.3f,%.3f,%.3f,%.2f,%.1f\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                          100 * sum(v) / tot, sum(v) / runs))
    log = os.path.join(src, "normals.log")
    if os.path.exists(log):
        open(os.path.join("profiles", pre + "_normals_probe_under_rocprof.txt"), "w").write(open(log).read())

# --- PMC passes of the same command: per-dispatch averages of the largest dispatches (the 1M-point runs)
import json
pmc = {"command": "python tools/normals_probe.py --reps 5", "kernels": {},
       "note": "per-dispatch averages over the 1M-point runs (the 1000-point warm-up dispatch is dropped); FETCH_SIZE / "
               "WRITE_SIZE in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE reads 1/2 of the streamed bytes "
               "(calibration in r01_pmc_bench.json)"}
for pp in ("normals_fetch", "normals_write", "normals_sq"):
    fn = os.path.join(src, pp, "p_counter_collection.csv")
    if not os.path.exists(fn):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        acc[(short(r["Kernel_Name"])[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if not (k.startswith("k_ann_normals") or k.startswith("k_ann_small") or k.startswith("k_ann_measure")):
            continue
        if k.startswith("k_ann_normals") or k.startswith("k_ann_small"):
            v = sorted(v)[1:] if len(v) > 1 else v          # drop the warm-up call
        name = c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")
        pmc["kernels"].setdefault(k, {})[name] = sum(v) / len(v)
        pmc["kernels"][k]["dispatches_" + pp] = len(v)
if pmc["kernels"]:
    json.dump(pmc, open(os.path.join("profiles", pre + "_normals_pmc.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(pmc["kernels"].get("k_ann_normals<10>", {}), indent=1))
