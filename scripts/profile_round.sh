# The round's committed evidence: for the headline bench and for each secondary workload (kd-tree queries, Icp,
# frame build) (1) rocprofv3 --kernel-trace --stats, (2) a per-grid summary, (3) FETCH_SIZE / WRITE_SIZE passes ->
# HBM bytes per launch of its dominant kernel.  Outputs under gpurun_out/profile_<round>/; copy them into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-round6}
OUT=gpurun_out/profile_$ROUND
rm -rf $OUT && mkdir -p $OUT
trace() {  # NAME program args...
  local NAME=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- "$@" > $OUT/${NAME}_under_rocprof.out 2> $OUT/trace_$NAME.err &&
  cp $(ls $OUT/trace_$NAME/*/*kernel_stats.csv | head -1) $OUT/${ROUND}_${NAME}_kernel_stats.csv &&
  { echo "# rocprofv3 --kernel-trace --stats -- $*   ($ROUND, MI355X)"; python3 scripts/summarize_trace.py $(ls $OUT/trace_$NAME/*/*kernel_trace.csv | head -1) | sed -n 1,40p; } > $OUT/${ROUND}_${NAME}_per_grid.txt
}
trace bench python3 bench.py --steps 5 --warmup 2 --no-extras --cpu-pairs 0 &&
trace kdtree python3 scripts/kd_probe.py &&
trace pcl_icp python3 scripts/pcl_probe.py &&
trace frame_build python3 scripts/build_trace_probe.py 32 -1 &&
ROUND=$ROUND bash scripts/traffic_pmc.sh bench image_icp_head_kernel "pairs_per_gpu=64 concurrent_launches=${CONC:-3} distinct_frames=1" python3 bench.py --steps 2 --warmup 1 --no-extras --cpu-pairs 0 > /dev/null &&
mv gpurun_out/${ROUND}_bench_traffic.json gpurun_out/${ROUND}_bench_traffic_distinct.json &&
ROUND=$ROUND bash scripts/traffic_pmc.sh bench image_icp_head_kernel "pairs_per_gpu=64 concurrent_launches=${CONC:-3} distinct_frames=0" python3 bench.py --shared-frames --steps 2 --warmup 1 --no-extras --cpu-pairs 0 > /dev/null &&
mv gpurun_out/${ROUND}_bench_traffic.json gpurun_out/${ROUND}_bench_traffic_shared.json &&
ROUND=$ROUND bash scripts/traffic_pmc.sh kdtree kdtree_nearest_kernel "queries=500000 points=500000" python3 scripts/kd_probe.py > /dev/null &&
ROUND=$ROUND bash scripts/traffic_pmc.sh pcl_icp pcl_icp_head_kernel "source_points=500000 target_points=500000" python3 scripts/pcl_probe.py > /dev/null &&
rm -rf gpurun_out/traffic_fetch_fb gpurun_out/traffic_write_fb &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic_fetch_fb -- python3 scripts/build_trace_probe.py 32 -1 > /dev/null 2> gpurun_out/traffic_fetch_fb.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic_write_fb -- python3 scripts/build_trace_probe.py 32 -1 > /dev/null 2> gpurun_out/traffic_write_fb.err &&
python3 scripts/summarize_traffic_sum.py gpurun_out/traffic_fetch_fb gpurun_out/traffic_write_fb "level0_quad_kernel|level0_kernel|blur_fused_kernel|blur_halve_words_kernel|splat_packed_kernel|resize_pick_kernel|luma_imap_kernel|unsplat_kernel|minmax_u16_kernel|dims_table_kernel" 320 frames_per_build=32 width=640 height=480 > gpurun_out/${ROUND}_frame_build_traffic.json &&
cp gpurun_out/${ROUND}_*_traffic*.json $OUT/ &&
python3 bench.py > $OUT/${ROUND}_bench.json 2> $OUT/bench.err &&
cp gpurun_out/bench_detail_n1.json $OUT/${ROUND}_bench_detail.json &&
head -12 $OUT/${ROUND}_bench_per_grid.txt && cat $OUT/${ROUND}_*_traffic.json && tail -c 2500 $OUT/${ROUND}_bench.json
