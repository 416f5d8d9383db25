"""Device occupancy of a window of a rocprofv3 --kernel-trace CSV: how much of the window some kernel was running
(union of the intervals), the kernel time summed (concurrency = sum / union), the idle time by the kernel that
ends it, and per-kernel totals.   usage: busy_timeline.py kernel_trace.csv [from_frac to_frac] [memory_copy_trace.csv]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 0.95)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             re.sub(r"\(anonymous namespace\)::|a3d::|void ", "", r["Kernel_Name"]).split("(")[0]) for r in rows)
t_lo, t_hi = iv[0][0], max(e for _, e, _ in iv)
w0, w1 = t_lo + (t_hi - t_lo) * f0, t_lo + (t_hi - t_lo) * f1
iv = [(max(s, w0), min(e, w1), n) for s, e, n in iv if e > w0 and s < w1]
union, total, idle_by = 0.0, 0.0, collections.Counter()
per = collections.defaultdict(lambda: [0, 0.0])
cur_end = w0
for s, e, n in iv:
    total += e - s
    per[n][0] += 1
    per[n][1] += e - s
    if s > cur_end:
        idle_by[n] += s - cur_end
        union += e - s
        cur_end = e
    elif e > cur_end:
        union += e - cur_end
        cur_end = e
span = w1 - w0
print(f"window {span/1e6:.3f} ms: some kernel running {union/span*100:.1f} %, kernel time summed {total/1e6:.3f} ms "
      f"(concurrency {total/max(union,1):.2f}), idle {(span-union)/1e6:.3f} ms")
print("idle time by the kernel that ends the gap (ms):")
for n, v in idle_by.most_common(12):
    print(f"  {n:40s} {v/1e6:8.3f}")
print("kernel time (ms, launches):")
for n, (c, v) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {n:40s} {v/1e6:8.3f}  {c}")
if len(sys.argv) > 4:
    cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in csv.DictReader(open(sys.argv[4]))]
    cp = [(max(s, w0), min(e, w1), d) for s, e, d in cp if e > w0 and s < w1]
    by = collections.Counter()
    for s, e, d in cp:
        by[d] += e - s
    print("copies in the window (ms):", {k: round(v / 1e6, 3) for k, v in by.items()})
