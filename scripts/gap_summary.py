"""Kernel durations and the gaps between consecutive launches of one kernel in a rocprofv3 --kernel-trace CSV."""
import csv, sys, collections
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else None
    by[int(r["Grid_Size_X"])].append(((e - s) / 1e3, gap))
    prev_end = e
for g, v in sorted(by.items()):
    d = np.array([x[0] for x in v]); gp = np.array([x[1] for x in v if x[1] is not None and x[1] < 100])
    print(f"grid_x {g}: launches {len(v)}, duration median {np.median(d):.2f} us, gap before launch median {np.median(gp):.2f} us (p10 {np.percentile(gp,10):.2f}, p90 {np.percentile(gp,90):.2f})")
