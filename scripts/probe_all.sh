for ns in 0 1; do for acc in valu mfma; do for lvl in 0 1 2; do
if [ $ns = 1 ]; then export A3D_ICP_NOSOLVE=1; else unset A3D_ICP_NOSOLVE; fi
echo -n "nosolve=$ns "; A3D_ICP_ACCUM=$acc A3D_ICP_VARIANT=8,1 timeout -k 10 120 python scripts/level_probe.py --level $lvl 2>&1 | tail -1 | cut -c1-150
done; done; done
