"""Multi-GPU data parallelism for the update path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm).  The reference is single-device; this is
the build's own addition (§8e).

Seeds are independent units, so a replay batch is block-partitioned across ranks in seed order and
every rank keeps a replica of the adjacency and the feature table (Reddit: 0.1 GB + 0.57 GB against
288 GB of HBM) — there is NO data-path collective and no halo exchange.  The only exchange step is
the gradient all-reduce of the ~1.5 M fp32 parameters (~6 MB): one flat bucket, one collective per
step, then the identical Adam update on every rank.  Because the Philox sampler is keyed by
(seed, batch counter, layer, vertex id, slot), a vertex draws the same neighbours on whichever rank
it lands: an N-rank step is bit-identical in sampled indices to the 1-rank step on the same seeds.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block partition of n items: rank r takes [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seeds(seeds, rank=None, world=None):
    """Rank's slice of a replay batch, in the reference's seed order (so concatenated per-rank outputs
    equal the 1-GPU output)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(len(seeds), rank, world)
    return seeds[lo:hi]


class GradSynchronizer:
    """All-reduce(mean) of every parameter gradient through ONE flat bucket: one concatenation, one scale, one
    collective per step; afterwards each ``p.grad`` is a view of the reduced bucket (no copy back)."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def sync(self, weight=None):
        """grads <- sum_r w_r * grad_r (w_r = 1/world by default; pass n_local/n_global for ragged shards)."""
        if self.world == 1:
            return
        w = (1.0 / self.world) if weight is None else float(weight)
        pieces = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
        flat = torch.cat(pieces)
        flat.mul_(w)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n


def all_gather_rows(t, group=None):
    """all-gather(v) of a per-rank 1-D tensor (per-seed losses of a sharded priority forward) to every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.numel()], dtype=torch.int64, device=t.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = [torch.empty(mx, dtype=t.dtype, device=t.device) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])
