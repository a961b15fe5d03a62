"""print a rocprofv3 kernel_stats.csv: name (shortened), calls, total ms, average us, max us.  usage: python tools/kstats.py <dir>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
        print("%-60s calls %5s  total %9.3f ms  avg %9.2f us  max %9.2f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
              float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
