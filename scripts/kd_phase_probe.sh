for lib in "" scripts/variantbuild/lib_probe1.so scripts/variantbuild/lib_probe2.so; do
for cfg in "512 12 4" "512 12 2" "512 12 1" "1024 14 2" "256 12 8" "256 12 4"; do
  set -- $cfg
  echo "== lib=$lib block=$1 lds_levels=$2 blocks_per_cu=$3"
  A3D_LIBRARY=$lib A3D_KD_BLOCK=$1 A3D_KD_LDS_LEVELS=$2 A3D_KD_BLOCKS_PER_CU=$3 timeout -k 10 120 python3 scripts/kd_probe.py 2>&1 | tail -1
done; done
