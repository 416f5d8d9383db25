"""FETCH_SIZE (and, when collected, the TCC_EA0_RDREQ counters) per calibration kernel -> the factor that turns the
reported figure into bytes.   python3 scripts/fetch_calib_summary.py <rocprofv3 out dir> [<second dir> ...] < expected"""
import csv, glob, json, sys
expected = {"calib_b16": 805306368, "calib_b12": 805306368, "calib_b8": 805306368, "calib_b4": 805306368,
            "calib_b1": 201326592, "calib_mix": 48 * 640 * 480 * 39 + 4 * (48 * 480 + 2) * 642}
vals = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in expected:
                if k in r["Kernel_Name"]:
                    vals.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
out = {}
for (k, c), v in sorted(vals.items()):
    mean = sum(v) / len(v)
    e = out.setdefault(k, {"expected_bytes": expected[k]})
    e[c] = mean
    if c == "FETCH_SIZE":
        e["bytes_over_FETCH_SIZE_KiB"] = expected[k] / (mean * 1024.0)
    elif "RDREQ" in c:
        e["bytes_per_" + c] = (expected[k] / mean) if mean else None
print(json.dumps(out, indent=1))
