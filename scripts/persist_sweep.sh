# bash scripts/persist_sweep.sh  (on the GPU box): lone pair and batches, persistent kernel (A3D_ICP_PERSIST=mask, diagnostics
# build) against per-iteration launches (the default)
export A3D_LIBRARY=${A3D_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/align3d_amd/csrc/libalign3d_hip_diag.so}  # the knobs exist in the diagnostics build only
set -e
cd "$(dirname "$0")/.."
for w in 0.125 0.25 0.375 0.5; do A3D_ICP_WAVES=$w python scripts/persist_probe.py lone; done
for w in 0.125 0.25; do A3D_ICP_PERSIST=7 A3D_ICP_WAVES=$w python scripts/persist_probe.py lone; done
for p in 0 4 6 7; do A3D_ICP_PERSIST=$p python scripts/persist_probe.py batch 64; done
for p in 0 6; do A3D_ICP_PERSIST=$p python scripts/persist_probe.py batch 16; done
for p in 0 7; do A3D_ICP_PERSIST=$p python scripts/persist_probe.py batch 4; done
