"""Voxel-axis sharding over the GPUs of one node (one process per GPU, ``torch.distributed``).

The hot path is independent per spectrum except for autophase's GLOBAL arg-max and the single
(p0, p1, pivot) it yields (reference ``phasing.py:229, 276-290``).  So the only cross-rank traffic is
O(1) per dataset: one all_gather of (max |X|, global flat index) per rank and one broadcast of
(p0, p1) from the rank that owns the winning spectrum.  No data-path collective exists.
"""
from __future__ import annotations


def shard_bounds(n_rows: int, world: int, rank: int):
    """Contiguous block of the flattened non-FID axes owned by `rank`: rows [lo, hi)."""
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return lo, hi


def pick_winner(pairs):
    """pairs[r] = (max_abs, global_flat_index) of rank r.  np.argmax semantics over the whole
    dataset: the largest magnitude wins, ties go to the lowest global flat index.  Returns
    (owner_rank, global_flat_index, max_abs)."""
    best_r, best = 0, pairs[0]
    for r, p in enumerate(pairs):
        if p[0] > best[0] or (p[0] == best[0] and p[1] < best[1]):
            best_r, best = r, p
    return best_r, int(best[1]), float(best[0])


def exchange_argmax(max_abs: float, global_flat: int, dist=None, device="cpu", group=None):
    """All ranks learn the winner.  `dist` = an initialised torch.distributed module (or None for a
    single process).  Works on RCCL ("nccl", GPU tensors) and gloo (CPU tensors, `group` = a gloo
    process group).  For a streaming pipeline prefer a gloo group: an RCCL collective is ordered behind
    everything already queued on the GPU (other datasets' kernels), a 32-byte host exchange is not."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0, int(global_flat), float(max_abs)
    import torch

    world = dist.get_world_size()
    # the flat index needs 64 integer bits (2^19 voxels x 2^13 points x 8 GPUs > 2^32): ship it as int64
    # and the magnitude as the bit pattern of a float64
    mine = torch.tensor([torch.tensor(max_abs, dtype=torch.float64).view(torch.int64).item(), int(global_flat)],
                        dtype=torch.int64, device=device)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    pairs = []
    for a in allv:
        a = a.cpu()
        pairs.append((float(a[0:1].view(torch.float64).item()), int(a[1].item())))
    return pick_winner(pairs)


def broadcast_params(values, owner: int, dist=None, device="cpu", group=None):
    """Broadcast a short list of float64 parameters (p0, p1) from `owner` to every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    import torch

    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.broadcast(t, src=owner, group=group)
    return [float(v) for v in t.cpu()]
