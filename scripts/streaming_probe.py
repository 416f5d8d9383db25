"""bench.py's streaming loop alone, repeated: pairs/s of each repetition (how stable the pipelined build + align loop is)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from align3d_amd import Context, IcpParams, MsIcpParams
ctx = Context(0)
prm = MsIcpParams.repeat(3, IcpParams.default())
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
builders = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print([round(bench.streaming_bench(ctx, prm, 64, 640, 480, builders=builders)["pairs_per_s"]) for _ in range(reps)], flush=True)
