# rocprofv3 --kernel-trace --stats of ten 32-frame builds for each variant library named on the command line
# ("product" = the product library, "diag" = the diagnostics build with whatever A3D_* knobs the caller exported): per-kernel average -> gpurun_out/variant_trace/<name>.txt
#   bash scripts/variant_trace.sh product l0p1 l0p2 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/variant_trace
mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = product ]; then unset A3D_LIBRARY;
  elif [ "$v" = diag ]; then export A3D_LIBRARY=$GRAFT_REPO_ROOT/align3d_amd/csrc/libalign3d_hip_diag.so;  # (with the caller's A3D_* knobs)
  else export A3D_LIBRARY=$GRAFT_REPO_ROOT/scripts/variantbuild_$v/libalign3d_hip_variant.so; fi
  rm -rf $OUT/t_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$v -- python3 scripts/build_trace_probe.py ${FRAMES:-32} -1 > $OUT/$v.out 2> $OUT/$v.err || exit 1
  python3 scripts/summarize_stats.py $(ls $OUT/t_$v/*/*kernel_stats.csv | head -1) > $OUT/$v.txt
  rm -rf $OUT/t_$v
  echo "== $v"; grep -i "level0\|blur_fused\|splat\|resize_pick\|halve\|luma\|minmax\|dims" $OUT/$v.txt
done
