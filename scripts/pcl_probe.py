"""Times Icp::align on 500k x 500k resident points (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context
from bench import pcl_icp_bench
r, _ = pcl_icp_bench(Context(0))
print(f"pcl icp: {r['us_per_iteration']:.1f} us per iteration, frac {r['roofline']['frac']:.3f}, err {r['error_vs_synthetic_gt']}")
