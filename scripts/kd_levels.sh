# Per-launch durations of one kd-tree selection build (500 k uniform points; KD_LEVELS_PROG=scripts/pcl_probe.py: the depth-image
# cloud of configs[2]): rocprofv3 --kernel-trace of scripts/kd_probe.py, the
# launches of the LAST build in stream order.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kd_levels
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 ${KD_LEVELS_PROG:-scripts/kd_probe.py} > $OUT/probe.out 2> $OUT/probe.err || exit 1
python3 - $(ls $OUT/t/*/*kernel_trace.csv | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sel_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last build = the launches from the last sel_pack_kernel on
last = max(i for i, r in enumerate(rows) if "sel_pack" in r["Kernel_Name"])
build = rows[last:]
t0 = int(build[0]["Start_Timestamp"])
prev_end = t0
for r in build:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][r['Kernel_Name'].index('sel_'):].split('(')[0][:32]:32s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d} blocks  start +{(s - t0) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:5.1f}  runs {(e - s) / 1e3:6.1f} us")
    prev_end = e
print(f"first start to last end: {(prev_end - t0) / 1e3:.1f} us")
PY
rm -rf $OUT/t
