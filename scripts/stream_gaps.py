"""Per-stream launch gaps of one kernel in a rocprofv3 --kernel-trace CSV: for every stream, the idle time between
the end of a launch and the start of the next one on the same stream, grouped by the next launch's grid.
usage: stream_gaps.py kernel_trace.csv kernel_substring"""
import collections, csv, sys
import numpy as np

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r["Stream_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])))
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for s, v in by_stream.items():
    v.sort()
    for (s0, e0, g0), (s1, e1, g1) in zip(v, v[1:]):
        if g0 == g1 and s1 - e0 < 200_000:  # same level, same alignment
            gaps[g1].append((s1 - e0) / 1e3)
        durs[g1].append((e1 - s1) / 1e3)
for g in sorted(durs):
    d, gp = np.array(durs[g]), np.array(gaps[g])
    print(f"grid_x {g}: {len(d)} launches on {len(by_stream)} streams, duration median {np.median(d):.1f} us "
          f"(p10 {np.percentile(d,10):.1f}, p90 {np.percentile(d,90):.1f}); gap to the previous launch of the stream "
          f"median {np.median(gp):.2f} us (p10 {np.percentile(gp,10):.2f}, p90 {np.percentile(gp,90):.2f})")
