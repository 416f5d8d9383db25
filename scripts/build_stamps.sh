# Diagnostic build of the library with s_memrealtime stamps in the iteration tail (read by scripts/tail_stamps.py).
set -e
cd "$(dirname "$0")/../align3d_amd/csrc"
OUT=../../scripts/${STAMP_OUT:-stampbuild}   # STAMP_OUT=name: a second build beside the first (A/B inside one gpurun call)
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w --offload-arch=gfx950 -DA3D_DIAGNOSTICS -DA3D_TAIL_STAMPS -DA3D_HEAD_ROTATE=0 $*"
for f in context image frame icp_engine image_icp kdtree kdtree_build kdtree_sort kdtree_select bilateral multi; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libalign3d_hip_stamps.so $OUT/*.o
