# The frame builder's launch chain: end -> start gaps between consecutive kernels of a 32-frame launch sequence
# (rocprofv3 --kernel-trace of scripts/build_trace_probe.py, scripts/gap_table.py) -> gpurun_out/builder_gaps.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/builder_gaps
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 scripts/build_trace_probe.py ${FRAMES:-32} -1 > $OUT/probe.out 2> $OUT/probe.err &&
python3 scripts/gap_table.py $(ls $OUT/t/*/*kernel_trace.csv | head -1) > gpurun_out/builder_gaps.txt && cat gpurun_out/builder_gaps.txt
