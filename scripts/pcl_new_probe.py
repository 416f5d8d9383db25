import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from align3d_amd import Context
ctx = Context(0, library=os.environ.get("A3D_LIBRARY"))
r, _ = bench.pcl_icp_bench(ctx)
print(json.dumps({k: r[k] for k in ("icp_new_device_ms", "new_plus_align_device_ms", "kd_build_kernel_ms", "kd_build_path", "device_ms_per_align", "icp_new_ms_incl_pcie")}))
