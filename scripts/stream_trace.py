"""A few streaming rounds (bench.py's streaming_bench) and nothing else: the workload for a rocprofv3 --kernel-trace
--memory-copy-trace timeline (scripts/busy_timeline.py reads the CSVs).   usage: stream_trace.py [rounds] [builders] [builder priority]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from align3d_amd import Context, IcpParams, MsIcpParams

ctx = Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
builders = int(sys.argv[2]) if len(sys.argv) > 2 else 1
priority = int(sys.argv[3]) if len(sys.argv) > 3 else -1
r = bench.streaming_bench(ctx, MsIcpParams.repeat(3, IcpParams.default()), 64, 640, 480, rounds=rounds, builders=builders,
                          builder_priority=priority)
print(r)
