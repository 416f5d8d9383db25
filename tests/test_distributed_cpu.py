"""The N > 1 path on CPU: two gloo ranks shard the pairs, each produces its local 4x4 poses, one
all-gather assembles them in global pair order (what bench.py does with RCCL on the GPU node)."""
import os
import socket
import subprocess
import sys

import numpy as np

from align3d_amd.distributed import owner_of, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["A3D_ROOT"])
import numpy as np, torch, torch.distributed as dist
from align3d_amd.distributed import gather_poses, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_pairs = 5 * world
lo, hi = shard_range(n_pairs, world, rank)
assert hi - lo == n_pairs // world
# stand-in for the local alignments: pose matrix of global pair j is filled with j + k/100
local = torch.tensor([[j + k / 100.0 for k in range(16)] for j in range(lo, hi)], dtype=torch.float32)
allp = gather_poses(local)
expect = torch.tensor([[j + k / 100.0 for k in range(16)] for j in range(n_pairs)], dtype=torch.float32)
assert torch.equal(allp, expect), (rank, allp)
t = torch.tensor([1.0 + rank]); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == world
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_shard_range_covers_everything_once():
    for n, w in [(512, 8), (64, 1), (10, 3), (7, 8), (0, 4)]:
        seen = []
        for r in range(w):
            lo, hi = shard_range(n, w, r)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    assert shard_range(512, 8, 3) == (192, 256)
    assert [owner_of(j, 512, 8) for j in (0, 63, 64, 511)] == [0, 0, 1, 7]


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_gather_in_global_pair_order(tmp_path, world):
    """world = 8: the rank count of BASELINE configs[4] (512 pairs over 8 GPUs), as host logic on CPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                   LOCAL_RANK=str(rank), A3D_ROOT=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out
