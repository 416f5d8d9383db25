"""benches/bench_icp.rs shape (sample1 frames 0 and 5 as clouds, 270 k points): Icp::new + align from resident clouds
(for rocprofv3 --kernel-trace --stats: which of the build's kernels a REAL depth-image cloud costs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context
import bench
ctx = Context(0, library=os.environ.get("A3D_LIBRARY"))
r, _ = bench.bench_icp_shape(ctx)
print({k: r[k] for k in ("icp_new_device_ms", "new_plus_align_device_ms", "device_ms_per_align", "icp_new_ms_incl_pcie")})
