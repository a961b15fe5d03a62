"""Average the counters of tools/r4_tcp_diag.sh over the k_search_refill dispatches (all of them: warm-up and timed)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "k_search_refill" not in r["Kernel_Name"]:
                continue
            k = r["Counter_Name"]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k in sorted(acc):
    s, n = acc[k]
    print("%-44s %16.1f  (per launch, %d dispatches)" % (k, s / n, n))
