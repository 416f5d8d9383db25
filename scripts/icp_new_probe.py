import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from align3d_amd import Context, Icp, IcpParams, R3dTree
from bench import pcl_clouds
ctx = Context(0)
(tgt, src), _ = pcl_clouds(ctx)
for k in range(4):
    t0 = time.perf_counter(); t = R3dTree.new(ctx, tgt.points); t1 = time.perf_counter(); t.free()
    print(f"R3dTree.new {(t1-t0)*1e3:.2f} ms")
for k in range(4):
    t0 = time.perf_counter(); icp = Icp.new(ctx, IcpParams.default(), tgt); t1 = time.perf_counter(); icp.free()
    print(f"Icp.new {(t1-t0)*1e3:.2f} ms")
