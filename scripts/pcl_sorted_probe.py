"""What a leaf-ordered source cloud would buy Icp::align: the same 500k x 500k alignment with the source as given and
with the source permuted by the leaf its points fall in under the identity (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import Context, Icp, IcpParams, PointCloud, R3dTree
from bench import pcl_clouds
ctx = Context(0)
(tgt, src), _ = pcl_clouds(ctx)
tree = R3dTree.new(ctx, tgt.points)
idx, _ = tree.nearest(src.points)
split, leaves = tree.download()
slot_of = np.empty(tgt.len(), np.int64)
bits = leaves[:, 3].view(np.uint32)
valid = np.isfinite(leaves[:, 0])
slot_of[bits[valid]] = np.nonzero(valid)[0]
leaf = slot_of[idx.astype(np.int64)] // 16
order = np.argsort(leaf, kind="stable")
icp = Icp.new(ctx, IcpParams.default(), tgt)
for name, cloud in (("as given", src), ("leaf-ordered", PointCloud(src.points[order], src.normals[order]))):
    icp.align(cloud)
    ts = []
    for _ in range(5):
        T = icp.align(cloud)
        ts.append(icp.last_device_ms())
    print(f"{name}: {np.median(ts) * 1e3 / 15:.1f} us per iteration  t={T.t}")
