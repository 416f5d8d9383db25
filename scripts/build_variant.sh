# Diagnostic build of the library with extra -D flags into scripts/variantbuild/ (use with A3D_LIBRARY=...)
set -e
cd "$(dirname "$0")/../align3d_amd/csrc"
OUT=../../scripts/variantbuild${VARIANT:+_$VARIANT}
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w --offload-arch=gfx950 -DA3D_DIAGNOSTICS $*"
for f in context image frame icp_engine image_icp kdtree kdtree_build kdtree_sort kdtree_select bilateral multi; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libalign3d_hip_variant.so $OUT/*.o
