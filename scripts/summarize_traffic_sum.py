"""FETCH_SIZE / WRITE_SIZE counter CSVs (rocprofv3 --pmc, one counter per pass) -> HBM bytes of a SET of kernels summed,
per unit of work (e.g. the frame builder's kernels per frame).  FETCH_SIZE doubled on gfx950 (summarize_traffic.py).

    python3 scripts/summarize_traffic_sum.py FETCH_DIR WRITE_DIR KERNEL_REGEX UNITS [key=value ...]

UNITS = how many units (frames) the profiled run processed with these kernels."""
import collections, csv, glob, json, re, sys


def totals(directory, counter, rx):
    out = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                m = rx.search(r["Kernel_Name"])
                if m:
                    out[m.group(0)] += float(r["Counter_Value"])
                    n[m.group(0)] += 1
    return out, n


rx = re.compile(sys.argv[3])
units = float(sys.argv[4])
fetch, nf = totals(sys.argv[1], "FETCH_SIZE", rx)
write, nw = totals(sys.argv[2], "WRITE_SIZE", rx)
assert fetch and nf == nw, (nf, nw)
per_kernel = {k: {"fetch_bytes_per_unit": fetch[k] * 1024.0 * 2.0 / units, "write_bytes_per_unit": write[k] * 1024.0 / units,
                  "launches": nf[k]} for k in sorted(fetch)}
out = {"kernels": sys.argv[3], "units": units, "per_kernel": per_kernel,
       "fetch_bytes_per_unit": sum(v["fetch_bytes_per_unit"] for v in per_kernel.values()),
       "write_bytes_per_unit": sum(v["write_bytes_per_unit"] for v in per_kernel.values()),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate counter-only passes; KiB -> bytes; FETCH_SIZE x2 "
                 "(gfx950, profiles/round3_fetch_calibration.json); summed over every launch of the matching kernels in the run, divided by the units processed"}
out["traffic_bytes_per_launch"] = out["fetch_bytes_per_unit"] + out["write_bytes_per_unit"]  # (bench.py's key: per unit here)
for kv in sys.argv[5:]:
    k, v = kv.split("=", 1)
    try:
        out[k] = int(v)
    except ValueError:
        out[k] = v
print(json.dumps(out))
