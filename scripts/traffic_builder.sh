# FETCH_SIZE / WRITE_SIZE of the frame builder's kernels (two counter-only passes), per frame and per kernel ->
# gpurun_out/${ROUND}_frame_build_traffic.json   (ROUND=round6 bash scripts/traffic_builder.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-round6}
rm -rf gpurun_out/traffic_fetch_fb gpurun_out/traffic_write_fb
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic_fetch_fb -- python3 scripts/build_trace_probe.py 32 -1 > /dev/null 2> gpurun_out/traffic_fetch_fb.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic_write_fb -- python3 scripts/build_trace_probe.py 32 -1 > /dev/null 2> gpurun_out/traffic_write_fb.err &&
python3 scripts/summarize_traffic_sum.py gpurun_out/traffic_fetch_fb gpurun_out/traffic_write_fb "level0_quad_kernel|level0_kernel|blur_fused_kernel|blur_halve_words_kernel|splat_packed_kernel|resize_pick_kernel|luma_imap_kernel|unsplat_kernel|minmax_u16_kernel|dims_table_kernel" 320 frames_per_build=32 width=640 height=480 > gpurun_out/${ROUND}_frame_build_traffic.json &&
cat gpurun_out/${ROUND}_frame_build_traffic.json
