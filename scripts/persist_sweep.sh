# bash scripts/persist_sweep.sh  (on the GPU box): lone pair and 64-pair batch, persistent kernel against per-iteration launches
set -e
cd "$(dirname "$0")/.."
for p in "" 0; do for w in 0.0625 0.125 0.25 0.5; do A3D_ICP_PERSIST=$p A3D_ICP_WAVES=$w python scripts/persist_probe.py lone; done; done
A3D_ICP_PERSIST= python scripts/persist_probe.py lone
for p in "" 0 4 6 7; do A3D_ICP_PERSIST=$p python scripts/persist_probe.py batch 64; done
for p in "" 0; do A3D_ICP_PERSIST=$p python scripts/persist_probe.py batch 16; A3D_ICP_PERSIST=$p python scripts/persist_probe.py batch 4; done
