# rocprofv3 --kernel-trace --stats of ten 16-frame builds -> gpurun_out/builder_trace/summary.txt (per-kernel average and
# total; the builder's kernel breakdown of DESIGN §5).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/builder_trace
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 scripts/build_trace_probe.py ${FRAMES:-32} ${PRIO:-0} > $OUT/probe.out 2> $OUT/probe.err &&
python3 scripts/summarize_stats.py $(ls $OUT/t/*/*kernel_stats.csv | head -1) > $OUT/summary.txt &&
python3 scripts/summarize_trace.py $(ls $OUT/t/*/*kernel_trace.csv | head -1) | sed -n 1,30p > $OUT/per_grid.txt &&
cat $OUT/summary.txt
