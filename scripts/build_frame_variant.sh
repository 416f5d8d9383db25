# A diagnostics-build variant that differs from libalign3d_hip_diag.so in frame.hip / bilateral.hip only (extra -D flags):
#   VARIANT=name bash scripts/build_frame_variant.sh -DA3D_L0_PROBE=2   ->  scripts/variantbuild_name/libalign3d_hip_variant.so
# (the other objects are taken from align3d_amd/csrc/diag: run `make -C align3d_amd/csrc diag` first)
set -e
cd "$(dirname "$0")/../align3d_amd/csrc"
OUT=../../scripts/variantbuild${VARIANT:+_$VARIANT}
mkdir -p $OUT
cp diag/*.o $OUT/
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w --offload-arch=gfx950 -DA3D_DIAGNOSTICS $*"
for f in frame bilateral; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libalign3d_hip_variant.so $OUT/*.o
rm -f $OUT/*.o
