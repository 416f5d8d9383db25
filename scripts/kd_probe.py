"""Times R3dTree::nearest on 500k x 500k resident points (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context
from bench import kdtree_bench
ctx = Context(0)
r = kdtree_bench(ctx)
print(f"kdtree: {r['ms_per_500k_queries']*1e3:.1f} us per 500k queries, {r['value']:.3e} q/s, frac {r['roofline']['frac']:.3f}")
