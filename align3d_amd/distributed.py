"""Multi-GPU layout of the batch workload: independent frame pairs are sharded in
contiguous blocks over the ranks (one process per GPU), each rank aligns its own pairs with no
data-path communication, and ONE all-gather collects the 4x4 poses (16 f32 per pair) — RCCL over
xGMI on the GPU node (`backend="nccl"`), gloo in the CPU tests.  torch is only the transport."""


def shard_range(n_items, world_size, rank):
    """Contiguous block of items owned by `rank`: item j belongs to rank floor(j * world / n) when
    n is a multiple of world (512 pairs over 8 GPUs -> 64 each); remainders go to the first ranks."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def owner_of(item, n_items, world_size):
    for r in range(world_size):
        lo, hi = shard_range(n_items, world_size, r)
        if lo <= item < hi:
            return r
    raise IndexError(item)


def gather_poses(local_matrices, group=None):
    """local_matrices: [pairs_per_rank, 16] f32 tensor (device tensor under nccl, CPU under gloo), the
    same shape on every rank.  Returns [world * pairs_per_rank, 16], rank-major = global pair order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = torch.empty((world * local_matrices.shape[0], local_matrices.shape[1]), dtype=local_matrices.dtype,
                      device=local_matrices.device)
    dist.all_gather_into_tensor(out, local_matrices.contiguous(), group=group)
    return out
