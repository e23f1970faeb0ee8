"""Multi-GPU data parallelism for the update path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm).  The reference is single-device; this is
the build's own addition (§8e).

Seeds are independent units, so a replay batch is block-partitioned across ranks in seed order and
every rank keeps a replica of the adjacency and the feature table (Reddit: 0.1 GB + 0.57 GB against
288 GB of HBM) — sampled neighbourhoods never cross a partition, so there is no halo exchange.  The
exchange steps are (1) the gradient all-reduce of the ~1.5 M fp32 parameters (~6 MB), then the identical Adam
update on every rank, and (2) for the sharded PBR passes (train update, priority forward over the train set) ONE
all-gather of the per-seed losses per pass, so that every rank's replay-buffer replica receives every priority
(``all_gather_sharded`` / ``all_gather_counts``); the sharded evaluation pass all-reduces its C x C confusion counters.
The Philox sampler is keyed by (seed, batch counter, layer, vertex id, slot) and every rank reserves the counters of ALL
batches of the one-rank pass (train batches are cut by seeds inside a batch, inference passes are cut by whole batches:
``batch_shard``), so a vertex draws the same neighbours on whichever rank it lands and the sampler state after a pass is
the one-rank state: an N-rank pass is bit-identical in sampled indices to the 1-rank pass on the same seeds.

Partitioned-feature mode (``FeaturePartition``): see its docstring.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block partition of n items: rank r takes [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_FORCE = False


def force_distributed(on=True):
    """Test switch: treat an initialised process group of ONE rank as distributed, so that every collective call, device-
    tensor requirement and stream ordering of the N-rank code runs — over RCCL when the group's backend is ``nccl`` — on a
    one-GPU box (tests/test_gpu_nccl.py).  Results are those of the one-rank path (sums over one rank)."""
    global _FORCE
    _FORCE = bool(on)


_LOCAL_ONLY = False


class local_only:
    """``with parallel.local_only():`` — inside an initialised process group, behave as ONE rank: no shard, no exchange
    (``is_distributed()`` False, ``rank_world()`` (0, 1)).  What ``bench.py --gpus N`` uses to time the one-rank step on every rank of the
    SAME invocation (the T1 of a scaling curve measured beside its TN); strategies read these at ``build_optimizer`` / per batch, so a
    strategy built and run inside the context is a plain one-rank strategy."""

    def __enter__(self):
        global _LOCAL_ONLY
        self._was, _LOCAL_ONLY = _LOCAL_ONLY, True
        return self

    def __exit__(self, *exc):
        global _LOCAL_ONLY
        _LOCAL_ONLY = self._was
        return False


def _single(world):
    """True when the N-rank code has nothing to exchange (one rank, not forced)."""
    return _LOCAL_ONLY or (world == 1 and not _FORCE)


def is_distributed(group=None):
    """True when torch.distributed is initialised with more than one rank (or with one and ``force_distributed()``)."""
    return (not _LOCAL_ONLY) and dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE)


def rank_world(group=None):
    if (not _LOCAL_ONLY) and dist.is_available() and dist.is_initialized():
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
    """All-reduce(sum of weight * grad) of every parameter gradient through flat buckets, overlapped with the backward pass.

    The first ``sync()`` learns the order in which gradients become ready (post-accumulate hooks) and reduces ONE flat
    bucket.  From then on the parameters are split into an EARLY bucket — everything except the gradients that arrive
    last (layer 0's ``fc_pool``, whose weight gradient is the longest kernel chain of the step) — and a LATE bucket:
    the early bucket's collective is launched asynchronously from the hook of its last gradient, i.e. it runs on RCCL's
    stream under the layer-0 backward kernels; ``sync()`` then reduces the small late bucket and waits for the early
    one.  Afterwards each ``p.grad`` is a view of a reduced bucket (no copy back).  xGMI is point-to-point, so the
    bucket count stays at two: a ring all-reduce is latency-bound at these sizes (4.5 MB + 1.5 MB for the Reddit model).

    The sequence of collectives is the SAME on every rank whatever happened locally (collectives are matched by order and
    size): once the split is learnt — rank 0's split, broadcast in a step every rank takes part in — every ``sync()``
    issues early then late.  A rank whose hooks did not complete the early bucket (no seeds in its shard, so no
    backward) launches it from ``sync()``.  Gradient accumulation (more than one ``backward()`` per ``sync()``) must run
    its non-final passes under ``no_sync()`` (as with DistributedDataParallel): a gradient of the early bucket that changes
    after the bucket was launched raises — detecting and repairing it collectively would cost a device->host read of a
    reduced flag every step, i.e. a pipeline drain per step on every rank.  ``weight`` scales this rank's gradients before
    the sum: 1 / world (a mean) by default, 1.0 when the loss already carries the 1 / n_global of a ragged shard (the
    strategies).  ``sync()`` never blocks the host: the collectives are enqueued, and waited for on the stream."""

    def __init__(self, params, group=None, overlap=True, late_fraction=0.35, weight=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and not _single(self.world)
        self.overlap = bool(overlap) and self.active
        self.late_fraction = late_fraction
        self.weight = (1.0 / self.world) if weight is None else float(weight)
        self._order = []            # arrival order of the current backward (indices into self.params)
        self._early = None          # indices of the early bucket once learnt
        self._late = None
        self._early_set = set()
        self._pending = None        # (work handle, flat tensor, weight used) of the early bucket in flight
        self._fired = {}            # parameter index -> hook firings since the last sync()
        self._stale = False         # an early-bucket gradient changed after the launch
        self._suspended = False
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._hooks = []
        self._buckets = {}          # name -> (flat tensor, index tuple, {param index: offset}): PERSISTENT across steps
        self._where = {}            # param index -> bucket name (where its gradient is reduced from the next step on)
        self.copied_in = 0          # instrumentation: gradient elements that had to be copied into a bucket (tests)
        self._handed = set()        # parameter indices whose slot a kernel of the CURRENT backward already writes into
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if self.active and self.params and self.params[0].is_cuda:
            from . import ops
            for i, p in enumerate(self.params):
                ops.register_grad_sink(p, self, i)

    # -- hooks ----------------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        i = self._index[id(p)]
        self._order.append(i)
        self._fired[i] = self._fired.get(i, 0) + 1
        if self._early is None or i not in self._early_set:
            return
        if self._pending is not None:
            self._stale = True                       # accumulated into after the launch: sync() raises (use no_sync())
            return
        if not self._suspended and all(self._fired.get(j, 0) >= 1 for j in self._early):
            self._launch_early(getattr(self, "_step_weight", self.weight))

    @property
    def learnt(self):
        """The early / late split is known (from then on every exchange is early bucket, then late bucket)."""
        return self._early is not None

    def begin_step(self, weight=None):
        """Start of a step whose backward pass may launch the early bucket from its hooks: they scale this rank's gradients by
        ``weight`` (default: the constructor's)."""
        self._step_weight = self.weight if weight is None else float(weight)
        self._fired, self._order = {}, []

    def no_sync(self):
        """Context manager for gradient-accumulation steps: backward passes inside it never launch the early bucket."""
        gs = self

        class _Ctx:
            def __enter__(self_inner):
                gs._suspended = True

            def __exit__(self_inner, *exc):
                gs._suspended = False
                gs._fired = {}                   # the launch condition counts the firings of the FINAL pass only
                gs._handed.clear()
                return False
        return _Ctx()

    def _bucket(self, name, idxs):
        """The persistent flat buffer of bucket ``name`` over the parameters ``idxs`` (allocated once per split)."""
        ent = self._buckets.get(name)
        if ent is None or ent[1] != tuple(idxs):
            p0 = self.params[idxs[0]]
            offs, total = {}, 0
            for i in idxs:
                offs[i] = total                                       # every slot 256-byte aligned: the kernels that write a
                total += (self.params[i].numel() + 63) // 64 * 64     # gradient into its slot use 16-byte stores
            ent = self._buckets[name] = (torch.zeros(total, dtype=p0.dtype, device=p0.device), tuple(idxs), offs)
            if name != "single":                                      # (the single bucket is filled by copies, never written in place)
                for i in idxs:
                    self._where[i] = name
        return ent

    def grad_slot(self, i):
        """A fresh view of parameter ``i``'s slot in its bucket, for a backward kernel to write the gradient into (so that
        autograd's AccumulateGrad adopts the slot as ``p.grad`` and ``_fill`` has nothing to copy); None when the slot must
        not be written: no bucket yet, a gradient already accumulated (``p.grad += new`` would alias), a collective of the
        bucket in flight — or the slot was ALREADY handed out since the last ``sync()``: inside one backward ``p.grad`` stays
        None until AccumulateGrad has every contribution, so a weight used by two products of the same graph (shared weights,
        two forwards summed into one loss) would otherwise get the same slot twice, the second kernel would overwrite the
        first and the engine would then add the slot to itself.  The second product allocates; autograd sums the two."""
        p = self.params[i]
        name = self._where.get(i)
        if (name is None or p.grad is not None or self._suspended or i in self._handed
                or (name == "early" and self._pending is not None)):
            return None
        self._handed.add(i)
        flat, _, offs = self._buckets[name]
        return flat.narrow(0, offs[i], p.numel()).view(p.shape)

    def _fill(self, name, idxs, w):
        """Bucket ``name`` <- w * the gradients of ``idxs``.  A gradient its producer already wrote into the bucket (see
        ``grad_slot``) costs nothing; the others are copied in — one multi-segment launch per 8 tensors on the GPU —,
        a missing one is a zero fill.  No allocation, no torch.cat."""
        flat, _, offs = self._bucket(name, idxs)
        todo = []
        for i in idxs:
            p = self.params[i]
            n = p.numel()
            sl = flat.narrow(0, offs[i], n)
            g = p.grad
            if g is None:
                sl.zero_()
            elif g.data_ptr() == sl.data_ptr() and g.is_contiguous():
                continue
            else:
                todo.append((g.reshape(-1) if g.is_contiguous() else g.contiguous().reshape(-1), sl, n))
                self.copied_in += n
        if todo:
            if flat.is_cuda and flat.dtype == torch.float32:
                from . import ops
                for k in range(0, len(todo), 8):
                    ops.stage_segments(todo[k:k + 8])
            else:
                for src, dst, _ in todo:
                    dst.copy_(src)
        if w != 1.0:
            flat.mul_(w)
        return flat

    def _launch_early(self, w):
        ctx = None
        if self.params and self.params[0].is_cuda:
            from . import ops
            ctx = ops.collective_section()       # a forked backward: fill + launch from the side stream (weight gradients live there)
        if ctx is None:
            flat = self._fill("early", self._early, w)
            self._early_t0 = self._t0(flat)
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            with ctx:
                flat = self._fill("early", self._early, w)
                self._early_t0 = self._t0(flat)
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending = (work, flat, w)

    def _scatter(self, name):
        """p.grad <- the parameter's slot of the (reduced) bucket: a view, no copy back."""
        flat, idxs, offs = self._buckets[name]
        for i in idxs:
            p = self.params[i]
            g = p.grad
            if not (g is not None and g.data_ptr() == flat.data_ptr() + offs[i] * flat.element_size() and g.shape == p.shape
                    and g.is_contiguous()):
                p.grad = flat.narrow(0, offs[i], p.numel()).view_as(p)     # (already so when the producer wrote into the slot)

    def _learn(self):
        """Split by arrival order: the trailing gradients (at most late_fraction of the elements) form the late bucket.
        EVERY rank calls this in the same step; rank 0's split wins (a rank without a backward has seen no arrivals)."""
        box = [None, None]
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        if dist.get_rank() == src:
            order = list(dict.fromkeys(self._order))
            if len(order) == len(self.params):       # otherwise some parameter got no gradient: keep the single bucket
                total = sum(p.numel() for p in self.params)
                late, acc = [], 0
                for i in reversed(order):
                    n = self.params[i].numel()
                    if late and acc + n > self.late_fraction * total:
                        break
                    late.append(i); acc += n
                late = list(reversed(late))
                box = [[i for i in order if i not in set(late)], late]
        dist.broadcast_object_list(box, src=src, group=self.group)
        early, late = box
        if early and late:
            self._early, self._late, self._early_set = early, late, set(early)
            self._bucket("early", early); self._bucket("late", late)      # the slots the next backward writes into
            self._buckets.pop("all", None)

    # -- measurement (bench.py --gpus N / --force-dist): device time of the step's collectives ---------------------------------------
    def enable_timing(self, on=True):
        """Record a hipEvent pair around every collective ``sync()`` issues from Python (never inside a stream capture): ``timings()``
        then gives the mean device time per kind — 'single' / 'all' / 'late': from the call to the point where the calling stream may
        continue (the EXPOSED time of that exchange); 'early': from its launch in the gradient hook to the wait in ``sync()`` (hidden
        under the backward kernels between the two + whatever is left exposed)."""
        self._timing = {} if on else None

    def _t0(self, flat):
        if getattr(self, "_timing", None) is None or not flat.is_cuda or torch.cuda.is_current_stream_capturing():
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _t1(self, kind, e0):
        if e0 is None or torch.cuda.is_current_stream_capturing():
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self._timing.setdefault(kind, []).append((e0, e1))

    def timings(self):
        """{kind: (mean milliseconds, count)} of the collectives recorded since ``enable_timing()``; synchronises the device."""
        t = getattr(self, "_timing", None)
        if not t:
            return {}
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in t.items() if v}

    # -- the step's exchange ---------------------------------------------------------------------------------------
    def sync(self, weight=None, single=False):
        """grads <- sum_r w_r * grad_r, w_r = ``weight`` (default: the constructor's).  ``single``: ONE flat bucket, one collective —
        for steps whose backward cannot overlap the exchange anyway (a replayed replica step); every rank must make the same choice."""
        if not self.active:
            return
        from . import ops
        ops.assert_no_pending_gradients()
        w = self.weight if weight is None else float(weight)
        if single:
            assert self._pending is None, "single-bucket sync after an early bucket was launched"
            idxs = list(range(len(self.params)))
            flat = self._fill("single", idxs, w)
            e0 = self._t0(flat)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            self._t1("single", e0)
            self._scatter("single")
            self._stale, self._fired, self._order = False, {}, []
            self._handed.clear()
            return
        if self._early is None:
            idxs = list(range(len(self.params)))
            flat = self._fill("all", idxs, w)
            e0 = self._t0(flat)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            self._t1("all", e0)
            self._scatter("all")
            if self.overlap and not (flat.is_cuda and torch.cuda.is_current_stream_capturing()):
                self._learn()                     # (never while a step is being recorded: it broadcasts through the host)
        else:
            if self._pending is None:                # the hooks did not complete the bucket on this rank: same collective, here
                self._launch_early(w)
            work, flat_e, w_used = self._pending
            stale = self._stale
            flat_l = self._fill("late", self._late, w)
            e0 = self._t0(flat_l)
            dist.all_reduce(flat_l, op=dist.ReduceOp.SUM, group=self.group)      # (issued even when raising below: stays matched)
            self._t1("late", e0)
            work.wait()
            self._t1("early", getattr(self, "_early_t0", None))
            self._early_t0 = None
            if stale:
                self._pending, self._stale, self._fired, self._order = None, False, {}, []
                self._handed.clear()
                raise RuntimeError("GradSynchronizer: a gradient of the early bucket changed after the bucket's all-reduce was "
                                   "launched (a second backward() before sync()); run the non-final passes of a gradient "
                                   "accumulation under `with gsync.no_sync():`")
            if w_used != w:
                flat_e.mul_(w / w_used)
            self._scatter("early")
            self._scatter("late")
        self._pending = None
        self._stale = False
        self._fired = {}
        self._order = []
        self._handed.clear()
        self._step_weight = self.weight


SHARDED_UPDATE = __import__("os").environ.get("OGL_DP_SHARDED_UPDATE", "0") == "1"


class ShardedAdam:
    """The gradient exchange and the optimiser of a data-parallel replica as reduce-scatter -> Adam on this rank's 1 / N of the flat
    parameter space -> all-gather of the updated weights (the ZeRO-1 shape), instead of all-reduce -> the identical Adam over all
    1.5 M parameters on every rank.  The lever DESIGN.md §6 names for the exposed late exchange at N = 8: a ring
    all-reduce IS a reduce-scatter followed by an all-gather, so the bytes on the links are the same; what changes is that the optimiser
    runs between the two halves on 1 / N of the elements (26 us -> ~4 us of Adam at N = 8 for the Reddit model) and that a rank keeps
    1 / N of the moments.  Off by default (``OGL_DP_SHARDED_UPDATE=1`` / ``bench.py --dp-sharded-update 1``): it costs one more
    collective launch per step, which at 6 MB over point-to-point xGMI is about what the shorter Adam saves — measured nowhere yet (no
    multi-GPU node); it exists so that the first scaling run can A/B it.

    Layout: the parameters are REBASED into one flat buffer (``p.data`` becomes a view of its 64-element-aligned slot; the total is
    padded to a multiple of 64 * world), so a rank's segment is ONE contiguous range of weights, of gradients and of moments: one Adam
    launch (ops.adam_step_multi_slabs: the arithmetic of optim.Adam, bit for bit), an in-place ``reduce_scatter_tensor`` and an
    in-place ``all_gather_into_tensor``.  Padding elements have zero gradients and stay zero.  Construct it BEFORE anything that keys
    on parameter addresses (GradSynchronizer's gradient sinks, captured graphs).  gloo has no reduce-scatter: the rehearsal backend
    all-reduces the flat gradient and reads its own segment (same sums).  ``step(weight)`` leaves ``p.grad`` as the LOCAL gradient.
    Every rank must call ``step`` every step (a rank without a backward contributes zeros)."""

    def __init__(self, params, group=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no parameters"
        self.group = group
        self.rank, self.world = rank_world(group)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        p0 = self.params[0]
        self.offs, total = [], 0
        for p in self.params:
            assert p.dtype == p0.dtype and p.device == p0.device
            self.offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        unit = 64 * self.world
        total = (total + unit - 1) // unit * unit
        self.seg = total // self.world
        self.lo, self.hi = self.rank * self.seg, (self.rank + 1) * self.seg
        self.wflat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        with torch.no_grad():
            for p, off in zip(self.params, self.offs):
                slot = self.wflat.narrow(0, off, p.numel())
                slot.copy_(p.data.reshape(-1))
                p.data = slot.view(p.shape)
        self.gflat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        self.m = torch.zeros(self.seg, dtype=p0.dtype, device=p0.device)      # the moments of this rank's segment only
        self.v = torch.zeros(self.seg, dtype=p0.dtype, device=p0.device)
        self.t = 0
        self._timing = None

    # -- measurement (bench.py --gpus N / --force-dist): device time of the two collectives, as GradSynchronizer.enable_timing ------
    def enable_timing(self, on=True):
        self._timing = {} if on else None

    def _ev(self):
        if self._timing is None or not self.gflat.is_cuda or torch.cuda.is_current_stream_capturing():
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _mark(self, kind, e0):
        e1 = self._ev()
        if e0 is not None and e1 is not None:
            self._timing.setdefault(kind, []).append((e0, e1))

    def timings(self):
        if not self._timing:
            return {}
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in self._timing.items() if v}

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _fill(self, w):
        todo = []
        for p, off in zip(self.params, self.offs):
            sl = self.gflat.narrow(0, off, p.numel())
            g = p.grad
            if g is None:
                sl.zero_()
            else:
                todo.append((g.reshape(-1) if g.is_contiguous() else g.contiguous().reshape(-1), sl, p.numel()))
        if todo and self.gflat.is_cuda and self.gflat.dtype == torch.float32:
            from . import ops
            for k in range(0, len(todo), 8):
                ops.stage_segments(todo[k:k + 8])
        else:
            for src, dst, _ in todo:
                dst.copy_(src)
        if w != 1.0:
            self.gflat.mul_(w)

    def _adam(self, pw, g):
        self.t += 1
        b1, b2 = self.betas
        if pw.is_cuda:
            from . import ops
            ops.adam_step_multi_slabs([pw], [g], [self.m], [self.v], [None], step=self.t, lr=self.lr, beta1=b1, beta2=b2, eps=self.eps)
            return
        # (the CPU rehearsal: the update of csrc/loss_optim.hip's adam_one spelled with tensor ops)
        self.m.add_(g - self.m, alpha=1.0 - b1)
        self.v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = self.v.sqrt().mul_(1.0 / (1.0 - b2 ** self.t) ** 0.5).add_(self.eps)
        pw.addcdiv_(self.m, denom, value=-(self.lr / (1.0 - b1 ** self.t)))

    @torch.no_grad()
    def step(self, weight=1.0):
        """weights <- Adam(weights, sum_r weight_r * grad_r), every rank ending with identical weights."""
        self._fill(float(weight))
        mine_g = self.gflat.narrow(0, self.lo, self.seg)
        mine_w = self.wflat.narrow(0, self.lo, self.seg)
        single = _single(self.world) or not dist.is_initialized()
        nccl = (not single) and dist.get_backend(self.group) == "nccl"
        e0 = self._ev()
        if nccl:
            dist.reduce_scatter_tensor(mine_g, self.gflat, op=dist.ReduceOp.SUM, group=self.group)     # in place: my slice of the input
        elif not single:
            if self.gflat.is_cuda:                                   # gloo with device tensors: through the host
                h = self.gflat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                mine_g.copy_(h.narrow(0, self.lo, self.seg))
            else:
                dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM, group=self.group)
        self._mark("reduce_scatter", e0)
        self._adam(mine_w, mine_g)
        e0 = self._ev()
        if nccl:
            dist.all_gather_into_tensor(self.wflat, mine_w, group=self.group)                           # in place likewise
        elif not single:
            if self.wflat.is_cuda:
                mine = mine_w.cpu()
                out = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(out, mine, group=self.group)
                for r in range(self.world):
                    if r != self.rank:
                        self.wflat.narrow(0, r * self.seg, self.seg).copy_(out[r])
            else:
                dist.all_gather([self.wflat.narrow(0, r * self.seg, self.seg) for r in range(self.world)], mine_w.clone(), group=self.group)
        self._mark("all_gather", e0)
        if self.wflat.is_cuda:
            from . import ops
            ops.invalidate_weight_images()


    # -- checkpointing: the moments live here, 1 / N per rank, not in the (never stepped) optim.Adam beside this object ---------------
    def _gather_flat(self, mine):
        """The full flat vector whose rank-r segment is rank r's ``mine`` (every rank gets it)."""
        full = torch.zeros(self.seg * self.world, dtype=mine.dtype, device=mine.device)
        single = _single(self.world) or not dist.is_initialized()
        if single:
            full.narrow(0, self.lo, self.seg).copy_(mine)
        elif dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(full, mine.contiguous(), group=self.group)
        else:
            out = [torch.empty(self.seg, dtype=mine.dtype) for _ in range(self.world)]
            dist.all_gather(out, mine.detach().cpu().contiguous(), group=self.group)
            for r in range(self.world):
                full.narrow(0, r * self.seg, self.seg).copy_(out[r])
        return full

    def state_dict(self):
        """Rank-independent optimiser state: per PARAMETER (in ``params`` order) its first and second moment, the step count and the
        hyper-parameters — the moments all-gathered from their owning ranks (a collective: every rank must call it), so a checkpoint
        written by rank 0 restores onto any world size."""
        m, v = self._gather_flat(self.m), self._gather_flat(self.v)
        state = []
        for p, off in zip(self.params, self.offs):
            n = p.numel()
            state.append(dict(exp_avg=m.narrow(0, off, n).view(p.shape).clone(), exp_avg_sq=v.narrow(0, off, n).view(p.shape).clone()))
        return dict(step=int(self.t), lr=self.lr, betas=tuple(self.betas), eps=self.eps, state=state)

    def load_state_dict(self, sd):
        """Inverse of ``state_dict`` on THIS object's layout (any world size): every rank keeps its own segment of the moments."""
        assert len(sd["state"]) == len(self.params), "the checkpoint was written for another model"
        self.t = int(sd["step"])
        self.lr, self.betas, self.eps = float(sd["lr"]), (float(sd["betas"][0]), float(sd["betas"][1])), float(sd["eps"])
        m = torch.zeros(self.seg * self.world, dtype=self.m.dtype, device=self.m.device)
        v = torch.zeros_like(m)
        for p, off, ent in zip(self.params, self.offs, sd["state"]):
            n = p.numel()
            assert tuple(ent["exp_avg"].shape) == tuple(p.shape)
            m.narrow(0, off, n).copy_(ent["exp_avg"].reshape(-1))
            v.narrow(0, off, n).copy_(ent["exp_avg_sq"].reshape(-1))
        self.m.copy_(m.narrow(0, self.lo, self.seg))
        self.v.copy_(v.narrow(0, self.lo, self.seg))


def assert_replicated(values, what="value", group=None):
    """Raise when ``values`` (array-like of integers) is not identical on every rank: the sharded passes pair losses
    computed by other ranks with THIS rank's seed list, which is only right while every rank drew the same seeds (identical
    host RNG state).  One 2-element all-reduce."""
    if not is_distributed(group):
        return
    import numpy as np
    a = np.ascontiguousarray(np.asarray(values, dtype=np.int64).reshape(-1))
    with np.errstate(over="ignore"):
        h = int(((a.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) * (np.arange(a.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum()
                & np.uint64((1 << 62) - 1)) + a.size
    dev = "cpu" if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([h, -h], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if int(t[0]) != -int(t[1]):
        raise RuntimeError("ranks disagree on %s: the replicas' host RNG streams have drifted (seed random / numpy / torch / "
                           "ogl_amd.sampling identically on every rank)" % what)


def batch_shard(n, batch, rank, world):
    """Rank's share of a pass over ``n`` seeds in batches of ``batch``: WHOLE batches of the one-rank pass, block-partitioned
    (so every local batch is a batch of the one-rank pass, with that batch's sampler counter).  Returns
    (first batch, last batch + 1, first seed position, last seed position + 1)."""
    nb = -(-n // batch) if n > 0 else 0
    b_lo, b_hi = shard_range(nb, rank, world)
    return b_lo, b_hi, min(b_lo * batch, n), min(b_hi * batch, n)


def partition_rows(n, world):
    """Rows per rank of the vertex-range partition of an n-row table: equal blocks (a multiple of 32 rows), the last
    ones short or empty — rank r owns rows [r * per, min((r + 1) * per, n))."""
    per = -(-max(int(n), 1) // world)
    return (per + 31) // 32 * 32


def all_gather_row_blocks(full, per, group=None):
    """The halo exchange of the partitioned-feature mode: ``full`` is a [world * per, D] table of which this rank has
    written row block ``rank`` (rows [rank * per, (rank + 1) * per)); afterwards every rank holds every block — global row
    order, no copy-back.  xGMI is point-to-point: one all-gather of a rank's 1 / world of the table (Reddit P0: 70 MB per
    rank to each of 7 peers over 7 links in parallel)."""
    rank, world = rank_world(group)
    if _single(world):
        return full
    assert full.shape[0] == world * per
    blocks = [full[r * per:(r + 1) * per] for r in range(world)]
    if dist.get_backend(group) == "gloo" and full.is_cuda:                # gloo: through the host
        mine = blocks[rank].cpu()
        out = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(out, mine, group=group)
        for r in range(world):
            if r != rank:
                blocks[r].copy_(out[r])
        return full
    dist.all_gather(blocks, blocks[rank], group=group)
    return full


def build_row_tables(n, widths, project, device, partition=True, padded_ld=None, group=None):
    """Per-vertex tables of a pass (one [>= n, width] fp32 matrix per entry of ``widths``), each row a function of that
    vertex's raw features only.  ``project(lo, hi, blocks)`` must fill ``blocks[k][: hi - lo]`` with the rows of
    vertices [lo, hi).  One rank or ``partition=False``: one call over [0, n).  Otherwise — the partitioned-feature mode
    — this rank projects ONLY its vertex range (``partition_rows``) in place into its block of the full tables and one
    all-gather per table exchanges the blocks (``all_gather_row_blocks``): rows outside the range are the halo."""
    rank, world = rank_world(group)
    ld = padded_ld or (lambda c: c)
    if _single(world) or not partition:
        bufs = [torch.empty((max(n, 0), ld(w)), dtype=torch.float32, device=device) for w in widths]
        views = [b[:, :w] for b, w in zip(bufs, widths)]
        if n > 0:
            project(0, n, views)
        return views
    per = partition_rows(n, world)
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    bufs = [torch.empty((world * per, ld(w)), dtype=torch.float32, device=device) for w in widths]   # whole padded rows travel
    views = [b[:, :w] for b, w in zip(bufs, widths)]
    if hi > lo:
        project(lo, hi, [v[lo:hi] for v in views])
    for b in bufs:
        all_gather_row_blocks(b, per, group)
    return views


def all_gather_counts(local, counts, group=None):
    """all-gather(v) of 1-D tensors whose per-rank lengths ``counts`` every rank already knows; returns the concatenation
    in rank order on ``local``'s device."""
    rank, world = rank_world(group)
    if _single(world):
        return local
    assert local.numel() == counts[rank], "local length does not match the agreed count"
    mx = max(max(counts), 1)
    dev = local.device
    staged = dist.get_backend(group) == "gloo" and local.is_cuda          # gloo: through the host
    pad = torch.zeros(mx, dtype=local.dtype, device="cpu" if staged else dev)
    pad[:local.numel()] = local.cpu() if staged else local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)]).to(dev)


def all_gather_sharded(local, full_sizes, group=None):
    """The exchange step of a rank-sharded pass over batches: every rank holds, for batch b of ``full_sizes[b]`` seeds,
    the values of ITS ``shard_range`` slice (``local[b]``, 1-D); returns the full per-batch value tensors, in seed
    order, on every rank.  ONE all-gather of the concatenated slices (the slice sizes follow from ``shard_range``, so
    nothing but the values travels).  This is the collective of the sharded PBR passes: per-seed losses -> priorities."""
    rank, world = rank_world(group)
    if _single(world):
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
    if not dist.is_initialized() or _single(dist.get_world_size(group)):
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
