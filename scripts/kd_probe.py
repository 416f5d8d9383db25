"""R3dTree on 500k resident points: nearest (500k queries) and the build (a3d_kdtree_new_device) (tuning aid; under
rocprofv3 --kernel-trace --stats it gives the build's per-kernel breakdown)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context
from bench import kdtree_bench
ctx = Context(0, library=os.environ.get("A3D_LIBRARY"))
r = kdtree_bench(ctx)
b = r["build"]
print(f"kdtree: {r['ms_per_500k_queries']*1e3:.1f} us per 500k queries, {r['value']:.3e} q/s, frac {r['roofline']['frac']:.3f}")
print(f"build (resident points): {b['kernel_ms']*1e3:.1f} us of launches {b['kernel_ms_stats']}, {b['device_ms']*1e3:.1f} us wall per call, "
      f"frac {b['roofline']['frac']:.3f}; from host points {r['build_ms_incl_pcie']*1e3:.1f} us")
