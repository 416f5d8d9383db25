"""FETCH_SIZE / WRITE_SIZE counter CSVs (rocprofv3 --pmc, one counter per pass) -> per-launch HBM bytes of one kernel,
averaged over its launches.  FETCH_SIZE is doubled (gfx950 counts a 128-byte fabric read request as 64 bytes;
MI355X_MICROARCH.md, HBM section); the rocprofv3 derived metrics report kilobytes (1024 B).

    python3 scripts/summarize_traffic.py FETCH_DIR WRITE_DIR KERNEL_SUBSTRING [key=value ...]

Extra key=value pairs are copied into the JSON (bench.py matches a profile to its workload through them)."""
import csv
import glob
import json
import sys


def per_launch(directory, counter, kernel):
    files = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)
    vals = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                vals[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    return [vals[k] for k in sorted(vals)]


kernel = sys.argv[3]
fetch = per_launch(sys.argv[1], "FETCH_SIZE", kernel)
write = per_launch(sys.argv[2], "WRITE_SIZE", kernel)
assert fetch and len(fetch) == len(write), (len(fetch), len(write))
n = len(fetch)
fetch_b = sum(fetch) * 1024.0 * 2.0 / n
write_b = sum(write) * 1024.0 / n
out = {
    "kernel": kernel, "launches_averaged": n,
    "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
    "traffic_bytes_per_launch": fetch_b + write_b,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate counter-only passes; KiB -> bytes; "
              "FETCH_SIZE x2 (gfx950: every fabric read request is 128 B and is tallied as 64 B — validated for 16 / 12 / 8 / 4 / 1 byte-per-lane reads in profiles/round3_fetch_calibration.json); mean over every launch of the kernel in the run",
}
for kv in sys.argv[4:]:
    k, v = kv.split("=", 1)
    try:
        out[k] = int(v)
    except ValueError:
        out[k] = v
print(json.dumps(out))
