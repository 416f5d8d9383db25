python scripts/kd_build_probe.py 2>&1 | grep "n=500000 build=device sort=own wide_len=None\|n=270213 build=device sort=own wide_len=None"
for e in "X=1" "A3D_ICP_PERSISTENT=1" "A3D_ICP_WAVES=0.125" "A3D_ICP_WAVES=0.5" "A3D_ICP_WAVES=1" "A3D_ICP_PERSISTENT=1 A3D_ICP_WAVES=0.5" "A3D_ICP_PERSISTENT=1 A3D_ICP_WAVES=0.125" "A3D_ICP_NOSOLVE=1"; do
  env $e python scripts/single_pair_probe.py 2>&1 | tail -2
done
