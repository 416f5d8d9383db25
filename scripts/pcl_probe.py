import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context
from bench import pcl_icp_bench
r = pcl_icp_bench(Context(0))
print(f"pcl icp: {r['device_ms_per_align']*1e3/15:.1f} us per iteration, frac {r['roofline']['frac']:.3f}, err {r['error_vs_synthetic_gt']}")
