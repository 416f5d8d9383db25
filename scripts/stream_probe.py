import sys, os
sys.path.insert(0, os.getcwd())
import bench
from align3d_amd import Context, MsIcpParams, IcpParams
ctx = Context(0)
prm = MsIcpParams.repeat(3, IcpParams.default())
for b in (1, 2, 3, 4):
    for pinned in (True, False):
        r = bench.streaming_bench(ctx, prm, 64, 640, 480, rounds=6, builders=b, pinned=pinned)
        print(b, pinned, round(r["pairs_per_s"]), flush=True)
