"""50 frame builds on one context (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import BilateralFilter, Context, RangeImageBuilder, SyntheticDataset
ds = SyntheticDataset(7, 2)
ctx = Context(0)
b = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
for i in range(50):
    for lv in b.build_device(*ds.get(i % 2)):
        lv.free()
