# Launch-geometry sweep of the two kd-tree query kernels (block size, LDS levels, blocks per CU).
# Usage on an MI355X: bash scripts/kd_sweep.sh
export A3D_LIBRARY=${A3D_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/align3d_amd/csrc/libalign3d_hip_diag.so}  # the knobs exist in the diagnostics build only
for cfg in "256 12 8" "256 13 5" "512 13 4" "512 14 2" "1024 14 2" "1024 15 1" "1024 13 2" "512 12 4"; do
  set -- $cfg
  echo "== block=$1 lds_levels=$2 blocks_per_cu=$3"
  A3D_KD_BLOCK=$1 A3D_KD_LDS_LEVELS=$2 A3D_KD_BLOCKS_PER_CU=$3 timeout -k 10 120 python3 scripts/kd_probe.py 2>&1 | tail -1
  A3D_PCL_BLOCK=$1 A3D_PCL_LDS_LEVELS=$2 A3D_PCL_BLOCKS_PER_CU=$3 timeout -k 10 120 python3 scripts/pcl_probe.py 2>&1 | tail -1
done
