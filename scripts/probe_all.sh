for acc in valu mfma; do for v in 8,2 16,2 32,2; do for lvl in 0 1 2; do
A3D_ICP_ACCUM=$acc A3D_ICP_VARIANT=$v timeout -k 10 120 python scripts/level_probe.py --level $lvl 2>&1 | tail -1
done; done; done
