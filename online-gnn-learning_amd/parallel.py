"""Multi-GPU data parallelism for the update path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm).  The reference is single-device; this is
the build's own addition (§8e).

Seeds are independent units, so a replay batch is block-partitioned across ranks in seed order and
every rank keeps a replica of the adjacency and the feature table (Reddit: 0.1 GB + 0.57 GB against
288 GB of HBM) — sampled neighbourhoods never cross a partition, so there is no halo exchange.  The
exchange steps are (1) the gradient all-reduce of the ~1.5 M fp32 parameters (~6 MB), then the identical Adam
update on every rank, and (2) for the sharded PBR passes (train update, priority forward over the train set) ONE
all-gather of the per-seed losses per pass, so that every rank's replay-buffer replica receives every priority
(``all_gather_sharded``); the sharded evaluation pass all-reduces its C x C confusion counters.  Because the Philox sampler is keyed by
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


def is_distributed(group=None):
    """True when torch.distributed is initialised with more than one rank."""
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_seeds(seeds, rank=None, world=None):
    """Rank's slice of a replay batch, in the reference's seed order (so concatenated per-rank outputs
    equal the 1-GPU output)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(len(seeds), rank, world)
    return seeds[lo:hi]


class GradSynchronizer:
    """All-reduce(mean) of every parameter gradient through flat buckets, overlapped with the backward pass.

    The first ``sync()`` learns the order in which gradients become ready (post-accumulate hooks) and reduces ONE flat
    bucket.  From then on the parameters are split into an EARLY bucket — everything except the gradients that arrive
    last (layer 0's ``fc_pool``, whose weight gradient is the longest kernel chain of the step) — and a LATE bucket:
    the early bucket's collective is launched asynchronously from the hook of its last gradient, i.e. it runs on RCCL's
    stream under the layer-0 backward kernels; ``sync()`` then reduces the small late bucket and waits for the early
    one.  Afterwards each ``p.grad`` is a view of a reduced bucket (no copy back).  xGMI is point-to-point, so the
    bucket count stays at two: a ring all-reduce is latency-bound at these sizes (4.5 MB + 1.5 MB for the Reddit model).
    """

    def __init__(self, params, group=None, overlap=True, late_fraction=0.35):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.overlap = bool(overlap) and self.world > 1
        self.late_fraction = late_fraction
        self._order = []            # arrival order of the current backward (indices into self.params)
        self._early = None          # indices of the early bucket once learnt
        self._late = None
        self._pending = None        # (work handle, flat tensor) of the early bucket in flight
        self._arrived = 0
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._hooks = []
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- hooks ----------------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        i = self._index[id(p)]
        self._order.append(i)
        if self._early is not None and self._pending is None and i in self._early_set:
            self._arrived += 1
            if self._arrived == len(self._early):
                self._launch_early()

    def _flatten(self, idxs, w):
        pieces = [(self.params[i].grad if self.params[i].grad is not None else torch.zeros_like(self.params[i])).reshape(-1)
                  for i in idxs]
        flat = torch.cat(pieces)
        flat.mul_(w)
        return flat

    def _launch_early(self):
        flat = self._flatten(self._early, 1.0 / self.world)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending = (work, flat)

    def _scatter(self, idxs, flat):
        off = 0
        for i in idxs:
            p = self.params[i]
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n

    def _learn(self):
        """Split by arrival order: the trailing gradients (at most late_fraction of the elements) form the late bucket."""
        order = list(dict.fromkeys(self._order))
        if len(order) != len(self.params):
            return                                   # some parameter got no gradient this step: keep the single bucket
        total = sum(p.numel() for p in self.params)
        late, acc = [], 0
        for i in reversed(order):
            n = self.params[i].numel()
            if late and acc + n > self.late_fraction * total:
                break
            late.append(i); acc += n
        late = list(reversed(late))
        early = [i for i in order if i not in set(late)]
        # every rank must cut the SAME buckets (the collectives are matched by order and size): rank 0's split wins
        box = [early, late]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        early, late = box
        self._late = late
        self._early = early
        self._early_set = set(self._early)
        if not self._early:
            self._early = None

    # -- the step's exchange ---------------------------------------------------------------------------------------
    def sync(self, weight=None):
        """grads <- sum_r w_r * grad_r (w_r = 1/world by default; pass n_local/n_global for ragged shards — a custom
        weight is applied to one flat bucket at sync time, without overlap)."""
        if self.world == 1:
            return
        w = (1.0 / self.world) if weight is None else float(weight)
        if self._pending is not None and weight is None:
            work, flat_e = self._pending
            flat_l = self._flatten(self._late, w)
            dist.all_reduce(flat_l, op=dist.ReduceOp.SUM, group=self.group)
            work.wait()
            self._scatter(self._early, flat_e)
            self._scatter(self._late, flat_l)
        else:
            if self._pending is not None:            # an early bucket is in flight with the default weight: finish and undo
                work, flat_e = self._pending
                work.wait()
                raise RuntimeError("GradSynchronizer: a custom weight cannot follow an overlapped launch; "
                                   "construct with overlap=False for ragged shards")
            idxs = list(range(len(self.params)))
            flat = self._flatten(idxs, w)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            self._scatter(idxs, flat)
            if self.overlap and self._early is None and weight is None:
                try:
                    self._learn()
                except Exception:                 # the exchange stays correct with the single bucket; only the overlap is lost
                    self._early = self._late = None
                    self.overlap = False
        self._pending = None
        self._arrived = 0
        self._order = []


def all_gather_sharded(local, full_sizes, group=None):
    """The exchange step of a rank-sharded pass over batches: every rank holds, for batch b of ``full_sizes[b]`` seeds,
    the values of ITS ``shard_range`` slice (``local[b]``, 1-D); returns the full per-batch value tensors, in seed
    order, on every rank.  ONE all-gather of the concatenated slices (the slice sizes follow from ``shard_range``, so
    nothing but the values travels).  This is the collective of the sharded PBR passes: per-seed losses -> priorities."""
    rank, world = rank_world(group)
    if world == 1:
        return list(local)
    per_rank = [[shard_range(n, r, world) for n in full_sizes] for r in range(world)]
    counts = [sum(hi - lo for lo, hi in pr) for pr in per_rank]
    mine = torch.cat([x.reshape(-1) for x in local]) if local else torch.zeros(0)
    assert mine.numel() == counts[rank], "local slices do not match shard_range of the batch sizes"
    mx = max(counts)
    dev = mine.device
    staged = dist.get_backend(group) == "gloo" and mine.is_cuda          # gloo: through the host
    pad = torch.zeros(mx, dtype=mine.dtype, device="cpu" if staged else dev)
    pad[:mine.numel()] = mine.cpu() if staged else mine
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    full = []
    offs = [0] * world
    for b, n in enumerate(full_sizes):
        parts = []
        for r in range(world):
            lo, hi = per_rank[r][b]
            parts.append(out[r][offs[r]:offs[r] + hi - lo])
            offs[r] += hi - lo
        full.append(torch.cat(parts).to(dev))
    return full


def all_gather_rows(t, group=None):
    """all-gather(v) of a per-rank 1-D tensor (per-seed losses of a sharded priority forward) to every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    dev = t.device
    if dist.get_backend(group) == "gloo" and t.is_cuda:                   # gloo: through the host
        t = t.cpu()
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.numel()], dtype=torch.int64, device=t.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = [torch.empty(mx, dtype=t.dtype, device=t.device) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)]).to(dev)
