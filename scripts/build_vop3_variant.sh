# The library with the VOP3-select post-pass on every source (scripts/hipcc_vop3.sh) into scripts/variantbuild_vop3/,
# diagnostics flags as scripts/build_variant.sh uses them: A/B against scripts/variantbuild/ inside one gpurun call.
set -e
cd "$(dirname "$0")/../align3d_amd/csrc"
OUT=../../scripts/variantbuild_vop3
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w --offload-arch=gfx950 -DA3D_DIAGNOSTICS $*"
for f in context image frame icp_engine image_icp kdtree kdtree_build kdtree_sort kdtree_select bilateral multi; do
  ../../scripts/hipcc_vop3.sh $f.hip $OUT/$f.o $FLAGS &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libalign3d_hip_variant.so $OUT/*.o
