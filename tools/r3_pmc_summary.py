"""GPU box: per-kernel averages of every counter found under <dir>/*/ (rocprofv3 --pmc csv outputs), search kernels only.
usage: python tools/r3_pmc_summary.py <dir>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "search" not in k:
            continue
        short = k.split("(")[0][:70]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name in sorted(c):
        v = c[name]
        # the timed region of icp_probe: the last 100 dispatches
        tail = v[-100:]
        print("   %-44s n=%4d  mean(all) %.6g   mean(last 100) %.6g" % (name, len(v), sum(v) / len(v), sum(tail) / len(tail)))
