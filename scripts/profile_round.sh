# Round profile of the default bench: (1) rocprofv3 --kernel-trace --stats, (2) per-grid summary,
# (3) FETCH_SIZE / WRITE_SIZE passes -> HBM bytes per launch.  Outputs under gpurun_out/profile_<round>/;
# copy the summaries into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-round1}
OUT=gpurun_out/profile_$ROUND
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-extras --cpu-pairs 0 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err &&
K=$(ls $OUT/trace/*/*kernel_trace.csv | head -1) && S=$(ls $OUT/trace/*/*kernel_stats.csv | head -1) &&
cp $S $OUT/${ROUND}_bench_kernel_stats.csv &&
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-extras --cpu-pairs 0   ($ROUND, MI355X)"; python3 scripts/summarize_trace.py $K | sed -n 1,40p; } > $OUT/${ROUND}_bench_per_grid.txt &&
ROUND=$ROUND CONC=${CONC:-3} bash scripts/traffic_pmc.sh > /dev/null &&
cp gpurun_out/${ROUND}_hbm_traffic.json $OUT/ &&
python3 bench.py > $OUT/${ROUND}_bench.json 2> $OUT/bench.err &&
head -12 $OUT/${ROUND}_bench_per_grid.txt && cat $OUT/${ROUND}_hbm_traffic.json && tail -c 1500 $OUT/${ROUND}_bench.json
