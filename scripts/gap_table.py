"""rocprofv3 --kernel-trace CSV -> the end -> start gap table of a dependent launch chain.
    python3 scripts/gap_table.py kernel_trace.csv [kernel substring]
For every launch of a matching kernel whose predecessor ON THE SAME QUEUE ended less than 50 us before it started (i.e.
inside one chain, not across host round trips): gap = its Start - the predecessor's End.  Grouped by (kernel, grid)."""
import collections, csv, re, sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else ""
by_queue = collections.defaultdict(list)
for r in rows:
    by_queue[r.get("Queue_Id", "0")].append(r)
table = collections.defaultdict(lambda: {"dur": [], "gap": [], "pitch": []})
for q, rs in by_queue.items():
    rs.sort(key=lambda r: int(r["Start_Timestamp"]))
    for prev, cur in zip([None] + rs[:-1], rs):
        name = re.sub(r"\(anonymous namespace\)::|a3d::|void ", "", cur["Kernel_Name"]).split("(")[0]
        if want not in name:
            continue
        key = (name, int(cur["Grid_Size_X"]) // max(1, int(cur["Workgroup_Size_X"])), int(cur["Grid_Size_Y"]))
        s, e = int(cur["Start_Timestamp"]), int(cur["End_Timestamp"])
        table[key]["dur"].append((e - s) / 1e3)
        if prev is not None:
            gap = (s - int(prev["End_Timestamp"])) / 1e3
            if gap < 50.0:
                table[key]["gap"].append(gap)
                table[key]["pitch"].append((s - int(prev["Start_Timestamp"])) / 1e3)
print("# kernel | blocks_x | grid_y | launches | duration us: median (min) | gap to predecessor's end us: median (p10, p90) | start-to-start us: median")
for k, v in sorted(table.items(), key=lambda kv: -len(kv[1]["dur"])):
    d, g, p = np.array(v["dur"]), np.array(v["gap"] or [np.nan]), np.array(v["pitch"] or [np.nan])
    print(f"{k[0]} | {k[1]} | {k[2]} | {len(d)} | {np.median(d):.2f} ({d.min():.2f}) | "
          f"{np.median(g):.2f} ({np.percentile(g, 10):.2f}, {np.percentile(g, 90):.2f}) | {np.median(p):.2f}")
