import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context, R3dTree, _abi
import bench
ctx = Context(0, library=_abi.DIAG_LIB_PATH)
(tgt, src), _ = bench.pcl_clouds(ctx, 500000)
print("cloud", tgt.points.shape, flush=True)
t = R3dTree.new(ctx, tgt.points)
ctx.synchronize()
