# The kd-tree selection build's kernels for in-block entry lengths 2048 / 1024 / 512 (A3D_KDTREE_NARROW_LEN, diagnostics
# build) and with the wide placement launches forced on: rocprofv3 --kernel-trace --stats of scripts/kd_probe.py each.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kd_narrow_sweep
rm -rf $OUT && mkdir -p $OUT
export A3D_LIBRARY=$GRAFT_REPO_ROOT/align3d_amd/csrc/libalign3d_hip_diag.so
for cfg in "len2048:" "len1024:A3D_KDTREE_NARROW_LEN=1024" "len512:A3D_KDTREE_NARROW_LEN=512" "place1:A3D_KDTREE_WIDE_PLACE=1"; do
  name=${cfg%%:*}; kv=${cfg#*:}
  unset A3D_KDTREE_NARROW_LEN A3D_KDTREE_WIDE_PLACE
  [ -n "$kv" ] && export "$kv"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$name -- python3 scripts/kd_probe.py > $OUT/$name.out 2> $OUT/$name.err || exit 1
  python3 scripts/summarize_stats.py $(ls $OUT/t_$name/*/*kernel_stats.csv | head -1) > $OUT/$name.txt
  rm -rf $OUT/t_$name
  echo "== $name ($kv)"; grep "^build" $OUT/$name.out; grep "sel_" $OUT/$name.txt
done
