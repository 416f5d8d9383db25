"""FETCH_SIZE / WRITE_SIZE counter CSVs (rocprofv3 --pmc, one counter per pass) -> per-launch HBM bytes of the ICP
kernel, averaged over the launches of the timed bench steps.  FETCH_SIZE is doubled (gfx950 counts a 128-byte
fabric read request as 64 bytes; MI355X_MICROARCH.md, HBM section); both counters are in KiB... the rocprofv3
derived metrics report kilobytes (1024 B)."""
import csv
import glob
import json
import sys

KERNEL = "image_icp_kernel"


def per_launch(directory, counter):
    files = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)
    vals = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and KERNEL in r["Kernel_Name"]:
                vals[int(r["Dispatch_Id"])] = (float(r["Counter_Value"]), int(r["Grid_Size"]))
    return [vals[k] for k in sorted(vals)]


fetch = per_launch(sys.argv[1], "FETCH_SIZE")
write = per_launch(sys.argv[2], "WRITE_SIZE")
pairs = int(sys.argv[3])
conc = int(sys.argv[4]) if len(sys.argv) > 4 else 1
assert fetch and len(fetch) == len(write), (len(fetch), len(write))
n = len(fetch)
fetch_b = sum(v for v, _ in fetch) * 1024.0 * 2.0 / n
write_b = sum(v for v, _ in write) * 1024.0 / n
print(json.dumps({
    "kernel": KERNEL, "launches_averaged": n, "pairs_per_gpu": pairs, "concurrent_launches": conc,
    "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
    "traffic_bytes_per_launch": fetch_b + write_b,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 2 --warmup 1; "
              "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read correction); mean over every launch of the kernel "
              "(warm-up, timed steps and the final-cost pass alike: all stream the same pyramids)",
}))
