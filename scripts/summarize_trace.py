"""Summarises a rocprofv3 --kernel-trace CSV per (kernel, grid): launches, avg/min/max microseconds."""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|a3d::|void ", "", r["Kernel_Name"])
    name = name.split("(")[0]
    d[(name, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
print("# kernel | grid_x (threads) | grid_y | launches | avg us | min us | max us | total ms")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%s | %d | %d | %d | %.1f | %.1f | %.1f | %.2f" % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1000))
