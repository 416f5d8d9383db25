import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import BilateralFilter, Context, RangeImageBuilder, synth
frames, _ = synth.frame_stream(4242, 16, 640, 480)
cam = synth.camera(640, 480)
ctx = Context(0)
b = RangeImageBuilder(ctx)
for i in range(10):
    for p in b.build_many(cam, frames, synth.DEPTH_SCALE):
        for lv in p:
            lv.free()
