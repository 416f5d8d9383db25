"""10 batched frame builds of 32 frames (argv[1]) on one context (for rocprofv3 --kernel-trace --stats); prints what a build
processed (bench.py's extra.frame_build.roofline uses the same figures)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import BilateralFilter, Context, RangeImageBuilder, synth
frames, _ = synth.frame_stream(4242, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 640, 480)
cam = synth.camera(640, 480)
ctx = Context(0, priority=int(sys.argv[2]) if len(sys.argv) > 2 else 0)  # -1: one chain (no colour fork)
b = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
for i in range(10):
    for p in b.build_many(cam, frames, synth.DEPTH_SCALE):
        for lv in p:
            lv.free()
print("last build:", ctx.last_build_stats())
