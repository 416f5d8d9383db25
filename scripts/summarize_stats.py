"""Prints a rocprofv3 *kernel_stats.csv as a short table: kernel | calls | avg us | total ms."""
import csv, glob, sys
f = sys.argv[1]
if not f.endswith(".csv"):
    f = glob.glob(f + "/*/*kernel_stats.csv")[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-46s calls %5s avg %8.1f us total %8.3f ms" % (name[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
    tot += float(r["TotalDurationNs"])
print("total ms %.3f" % (tot / 1e6))
