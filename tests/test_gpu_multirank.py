"""The N > 1 path of bench.py through REAL batches: `bench.py --gpus 2` starts two fresh rank processes itself, each
rank builds and aligns its own block of the global pair list on the GPU, one all-gather (gloo here: the box has one
GPU, both ranks use device 0) assembles the poses in global pair order.  Rank 1's block must equal, bit for bit, a
single-process batch of the same pairs (stream 1001)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_bench(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_world_size_mismatch_is_refused():
    """WORLD_SIZE set by a launcher and different from --gpus: exit non-zero before anything touches a GPU."""
    r = run_bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0"}, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
    assert r.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("shared", [False, True])
def test_two_fresh_ranks_align_their_own_blocks_and_gather(ctx, tmp_path, shared):
    P = 8
    dump = str(tmp_path / "gathered.npy")
    args = ["--gpus", "2", "--backend", "gloo", "--device", "0", "--pairs-per-gpu", str(P), "--steps", "3", "--warmup", "1",
            "--no-extras", "--cpu-pairs", "0", "--dump-gathered", dump] + (["--shared-frames"] if shared else [])
    r = run_bench(args)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_pairs"] == 2 * P
    assert out["config"]["ranks_in_collective"] == 2 and out["config"]["gathered_pairs"] == 2 * P
    assert out["config"]["gather_matches_local_poses"] is True
    assert out["config"]["distinct_frames"] is (not shared)
    assert out["roofline"]["failed_pairs"] == 0
    gathered = np.load(dump)
    assert gathered.shape == (2 * P, 16) and gathered.dtype == np.float32

    # rank 1's block = stream 1001, as one single-process batch on this process's context
    sys.path.insert(0, ROOT)
    import bench
    from align3d_amd import IcpParams, MsIcpParams, MultiscaleAlignBatch

    n_frames = P + 1 if shared else 2 * P
    pyr, _, _ = bench.build_stream_pyramids(ctx, seed=1001, n_frames=n_frames, width=640, height=480)
    pairs = [(p, p + 1) for p in range(P)] if shared else [(2 * p, 2 * p + 1) for p in range(P)]
    batch = MultiscaleAlignBatch(ctx, MsIcpParams.repeat(3, IcpParams.default()), [pyr[a] for a, _ in pairs],
                                 [pyr[b] for _, b in pairs])
    d = ctx.malloc(P * 64)
    _, status = batch.align(matrices_device=d)
    mats = ctx.to_host(d, np.zeros((P, 16), np.float32))
    ctx.free(d)
    batch.free()
    for lv in (lv for p in pyr for lv in p):
        lv.free()
    assert not status.any()
    assert np.array_equal(gathered[P:].view(np.uint32), mats.view(np.uint32)), "rank 1's block is not the single-process batch"
    # rank 0's block is another stream: not the same poses
    assert not np.array_equal(gathered[:P], gathered[P:])
    # every gathered matrix is a rigid transform
    m = gathered.reshape(-1, 4, 4)
    assert np.allclose(m[:, 3], [0, 0, 0, 1]) and np.allclose(np.linalg.det(m[:, :3, :3]), 1, atol=1e-5)


def test_launcher_fails_loudly_when_a_rank_fails(tmp_path):
    """On a machine without a GPU every rank fails at a3d_context_create: the launcher must return non-zero (and not
    hang waiting for the other ranks).  On a GPU box the ranks succeed, which the gpu test above covers."""
    from align3d_amd.multi import device_count

    if device_count() > 0:  # (the library's own count: torch.cuda.is_available() can be False on a box that has a GPU)
        pytest.skip("needs a machine without a GPU")
    r = run_bench(["--gpus", "2", "--backend", "gloo", "--device", "0", "--pairs-per-gpu", "2", "--steps", "1", "--warmup", "0",
                   "--no-extras", "--cpu-pairs", "0"], timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_c_host_gathers_poses_with_rccl_on_the_context_stream():
    """RCCL below Python: examples/rccl_gather.cpp — a plain C-ABI host that aligns its pairs, then calls ncclAllGather
    on a3d_context_stream with the device buffer a3d_multiscale_batch_align filled.  One rank here (the box has one
    GPU and RCCL refuses two ranks on one device); the N-rank form only differs in the communicator's size."""
    exe = os.path.join(ROOT, "tests", "cpp", "rccl_gather")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "rccl_gather"], stdout=subprocess.DEVNULL)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "A3D_NCCL_ID_FILE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "rccl gather OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_host_goes_through_the_n_rank_rendezvous_with_one_rank(tmp_path):
    """The same program launched the way an N-rank job launches it (RANK / WORLD_SIZE / LOCAL_RANK / A3D_NCCL_ID_FILE
    set): rank 0 writes the RCCL unique id to the file and every rank — here the only one — reads it back before
    ncclCommInitRank.  What the 8-GPU launch adds is the communicator's size, nothing else."""
    exe = os.path.join(ROOT, "tests", "cpp", "rccl_gather")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "rccl_gather"], stdout=subprocess.DEVNULL)
    id_file = tmp_path / "nccl_id"
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", A3D_NCCL_ID_FILE=str(id_file))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "rccl gather OK" in out.stdout, out.stdout + out.stderr
    assert id_file.exists() and id_file.stat().st_size == 128  # sizeof(ncclUniqueId)
    # a rank that is told WORLD_SIZE > 1 without the file refuses instead of inventing an id
    bad = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                         env={k: v for k, v in dict(env, WORLD_SIZE="2").items() if k != "A3D_NCCL_ID_FILE"})
    assert bad.returncode == 5


@pytest.mark.gpu
def test_four_fresh_ranks_keep_global_pair_order(ctx, tmp_path):
    """`bench.py --gpus 4` (gloo, all ranks on device 0: the box has one GPU and allows six GPU processes): four fresh
    rank processes, 2 pairs each.  One JSON line; ranks_in_collective == 4; the gathered buffer is in global pair order —
    rank r's block is stream 1000 + r, checked bit for bit against single-process batches of those streams."""
    P, N = 2, 4
    dump = str(tmp_path / "gathered.npy")
    r = run_bench(["--gpus", str(N), "--backend", "gloo", "--device", "0", "--pairs-per-gpu", str(P), "--steps", "2",
                   "--warmup", "1", "--no-extras", "--cpu-pairs", "2", "--cpu-orders", "3", "--dump-gathered", dump])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    # the line the driver's SCALE record is made of: inside the budget, every contract key present, value = all ranks' pairs
    import bench as bench_module
    assert len(lines[0]) < bench_module.LINE_BUDGET_BYTES
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["scaling"] == "weak" and out["unit"] == "frame-pairs/s"
    # the N > 1 line carries `roofline` AND `cpu_baseline` (a bounded sample on rank 0: chunk-order timing
    # and the parity envelope of a few of its pairs), so that a scaling record made of it does not read as unmeasured
    cpu, roof = out["cpu_baseline"], out["roofline"]
    assert cpu is not None and cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["pairs_compared"] == P
    assert cpu["pairs_over_1e-4_and_outside_the_cpu_envelope"] == 0
    for k in ("bound", "achieved", "peak", "unit", "frac"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and "failed_legs" not in out["extra"]
    assert abs(out["value"] - N * P * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"])) <= 1e-3 * out["value"]
    assert out["n_gpus"] == N and out["config"]["ranks_in_collective"] == N and out["config"]["gathered_pairs"] == N * P
    assert out["config"]["gather_matches_local_poses"] is True and out["roofline"]["failed_pairs"] == 0
    gathered = np.load(dump)
    assert gathered.shape == (N * P, 16)
    sys.path.insert(0, ROOT)
    import bench
    from align3d_amd import IcpParams, MsIcpParams, MultiscaleAlignBatch

    for rank in range(N):
        pyr, _, _ = bench.build_stream_pyramids(ctx, seed=1000 + rank, n_frames=2 * P, width=640, height=480)
        batch = MultiscaleAlignBatch(ctx, MsIcpParams.repeat(3, IcpParams.default()), [pyr[2 * p] for p in range(P)],
                                     [pyr[2 * p + 1] for p in range(P)])
        d = ctx.malloc(P * 64)
        _, status = batch.align(matrices_device=d)
        mats = ctx.to_host(d, np.zeros((P, 16), np.float32))
        ctx.free(d)
        batch.free()
        for lv in (lv for q in pyr for lv in q):
            lv.free()
        assert not status.any()
        assert np.array_equal(gathered[rank * P:(rank + 1) * P].view(np.uint32), mats.view(np.uint32)), rank


@pytest.mark.gpu
def test_two_rank_line_carries_the_copy_ceiling():
    """The default N > 1 run (no --no-extras) also measures the chip's own copy ceiling on rank 0's device (a child
    process, while the other ranks wait at the barrier): `hbm_copy_ceiling_GBs` and `frac_of_copy_ceiling` in the line."""
    r = run_bench(["--gpus", "2", "--backend", "gloo", "--device", "0", "--pairs-per-gpu", "2", "--steps", "2", "--warmup", "1",
                   "--cpu-pairs", "1", "--cpu-orders", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_in_collective"] == 2
    if not os.path.exists(os.path.join(ROOT, "scripts", "copy_ceiling")):
        pytest.skip("scripts/copy_ceiling is not built")
    assert out["roofline"]["hbm_copy_ceiling_GBs"] > 1000 and 0 < out["roofline"]["frac_of_copy_ceiling"] < 1.5
    assert out["cpu_baseline"] is not None


@pytest.mark.gpu
def test_the_drivers_launch_form_with_one_rank_goes_through_rccl():
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (one rank per GPU over RCCL).  With one GPU here: the same launcher, one rank,
    `--rehearse-collective` = the N > 1 code path (process group on backend nccl = RCCL, the all-gather on the context's
    stream through an ExternalStream, the device-count and context-device checks, the bounded cpu_baseline of a
    collective run): what an 8-GPU node adds is the communicator's size."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse-collective", "--pairs-per-gpu", "4",
           "--steps", "2", "--warmup", "1", "--no-extras", "--cpu-pairs", "1", "--cpu-orders", "2"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["ranks_in_collective"] == 1 and out["config"]["gathered_pairs"] == 4
    assert out["config"]["gather_matches_local_poses"] is True and "RCCL" in out["config"]["collective"]
    assert out["roofline"]["failed_pairs"] == 0 and out["cpu_baseline"] is not None
