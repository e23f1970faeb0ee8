"""Block sampler + loader with the surface the reference uses from DGL.

Replaces ``dgl.sampling.MultiLayerNeighborSampler([S, S], replace=True, return_eids=True)`` and
``dgl.sampling.NodeDataLoader(graph, seeds, sampler, batch_size, shuffle, drop_last, num_workers)``
(call sites R/train/graphsage/pytorch/model.py:44-47,128-131,174-178,224-227,280-284,312-316).

MI355X-first differences (documented in DESIGN.md):
* sampling runs on the GPU over the resident time-ordered CSR (ogl_sample_layer / ogl_build_block),
  so ``num_workers`` is accepted and ignored (HIP is not fork-safe; the forked CPU samplers are
  what this replaces);
* a loader samples ALL its batches layer by layer before yielding the first one — one batched sampler launch and
  one batched block-build sequence per layer (ogl_sample_layer_batched / ogl_build_block_batched) — so the host
  synchronises once per layer per loader (to learn the block sizes) instead of once per batch;
* randomness is a counter-based Philox stream keyed by (seed, batch counter, layer, dst id, slot):
  reproducible, independent of batch composition and of how seeds are sharded over GPUs.  One consequence differs from
  DGL: the key holds the destination's VERTEX ID, not its position in the batch, so a vertex that appears twice among the
  destinations of one layer of one batch draws the same neighbours both times (DGL draws per occurrence).  The reference's
  loaders never produce that case for the seeds (its draws are shuffle prefixes / dict keys: distinct vertices) and block
  construction de-duplicates the inner layers' destinations, so it is unreachable on this path; it is what makes a
  vertex's draws independent of which rank holds it.
"""
from __future__ import annotations

import os

import torch

from . import ops

# fused inference batches carry a byte per hidden-layer row: 1 where the next layer reads the row's fp32 values (its destinations'
# own rows) — the layer below then stores fp32 only there (OGL_FUSED_KEEP_ROWS=0: every row, as before)
FUSED_KEEP_ROWS = True

# OGL_SAMPLE_PIPELINE=1: a training loader hands out its first PIPELINE_FIRST batches as soon as they are sampled and samples the rest
# on a second stream while those train (what the reference's ``num_workers`` does with forked CPU samplers); the blocks are the
# same — every batch is sampled with its own Philox counter either way.  OFF by default — measured (round 4, Reddit rung, same box,
# 4 alternations): with the hash block build 1.007 -> 0.997 ms per step over 50-batch loaders but 1.008 -> 1.010 (outliers to 1.11) on the
# driver's 20-step command; with the direct-address build (1.3 -> 0.5 ms of sampling + build per 50 batches) 0.9850 vs 0.9855 and 0.9843
# vs 0.9857 (one outlier at 1.006): the sampling kernels take the CUs they run on away from the train step — every GEMM block owns a CU
# — so what the overlap hides it costs again.  A low-priority sampling stream starves (1.23 ms per step: the host waits for it).
PIPELINE = False
PIPELINE_FIRST = 4
_PIPE = {"stream": None}

NID = "_ID"   # same key DGL uses for block.srcdata[dgl.NID]
EID = "_EID"

_STATE = {"seed": 1, "ctr": 0}


def seed(value: int):
    """Reset the sampler stream (the analogue of ``dgl.seed``; the reference never calls it)."""
    _STATE["seed"] = int(value)
    _STATE["ctr"] = 0


def _next_ctr() -> int:
    c = _STATE["ctr"]
    _STATE["ctr"] = c + 1
    return c


def reserve_ctrs(n: int):
    """The next ``n`` batch counters.  A rank-sharded pass reserves the counters of ALL batches of the one-rank pass on
    every rank and samples its own batches with theirs: the stream state after the pass — and every batch's draws — are
    those of the one-rank run."""
    c = _STATE["ctr"]
    _STATE["ctr"] = c + int(n)
    return list(range(c, c + int(n)))


def get_state():
    return dict(_STATE)


def set_state(state):
    _STATE.update(state)


class Block:
    """Fixed-fanout bipartite block: edge j of dst d comes from src ``local_idx[d, j]``.

    Offers what the reference touches on a DGL block: ``number_of_dst_nodes()``,
    ``number_of_src_nodes()``, ``number_of_edges()``, ``in_degrees()``, ``srcdata[NID]``,
    ``dstdata[NID]``, ``edata``, ``is_block`` and ``to(device)`` (R/.../pytorch/model.py:62,99,191).
    """
    is_block = True

    def __init__(self, src_ids, dst_ids, local_idx, picks=None, dst_pos=None, dst_flag=None):
        self.src_ids = src_ids            # int64 [n_src]  (dst nodes first)
        self.dst_ids = dst_ids            # int64 [n_dst]
        self.local_idx = local_idx        # int32 [n_dst, fanout], -1 = no neighbour
        self.picks = picks                # int64 [n_dst, fanout] global ids (kept for the cached-projection path)
        # several loader batches fused into one block (sample_batches(fuse_rows=...)): the destinations are no longer the FIRST
        # sources — dst_pos[d] is destination d's own row in the source list (None: the usual h[:n_dst])
        self.dst_pos = dst_pos
        # fused batches: uint8 [n_src], 1 at the rows dst_pos lists — the only rows of the layer below whose fp32 values this block's
        # layer reads (its fc_pool reads that layer's image); `out_keep` on the block BELOW is the same array, set by GraphSAGE.forward
        self.dst_flag = dst_flag
        self.out_keep = None
        self.srcdata = {NID: src_ids}
        self.dstdata = {NID: dst_ids}
        self.edata = {}
        self._n_edges = None

    @property
    def fanout(self):
        return (self.local_idx if self.local_idx is not None else self.picks).shape[1]

    def number_of_dst_nodes(self):
        return int(self.dst_ids.numel())

    def number_of_src_nodes(self):
        if self.src_ids is None:
            raise RuntimeError("this input block was sampled with relabel_input=False: it has global picks only")
        return int(self.src_ids.numel())

    def number_of_edges(self):
        if self._n_edges is None:
            idx = self.local_idx if self.local_idx is not None else self.picks
            self._n_edges = int((idx >= 0).sum().item())
        return self._n_edges

    def in_degrees(self):
        idx = self.local_idx if self.local_idx is not None else self.picks
        return (idx >= 0).sum(dim=1)

    def to(self, device):
        dev = torch.device(device)
        if dev.type == "cuda":
            return self
        raise RuntimeError("blocks live on the GPU; the HIP path has no CPU fallback")


class MultiLayerNeighborSampler:
    """fanouts[l] neighbours per dst of block l, uniform WITH replacement (``replace`` must be True:
    every reference call site passes replace=True; fixed fanout is what makes blocks ELL-shaped)."""

    def __init__(self, fanouts, replace=True, return_eids=False):
        if not replace:
            raise NotImplementedError("only replace=True (the mode the reference uses) is implemented")
        if any(f is None for f in fanouts):
            raise NotImplementedError("full-neighbour sampling (fanout None) is outside the hot path")
        self.fanouts = [int(f) for f in fanouts]
        self.return_eids = return_eids

    def sample_batches(self, graph, seed_batches, relabel_input=True, ctrs=None, fuse_rows=0):
        """Sample every batch, output layer first.  Returns a list of (input_nodes, seeds, blocks).

        ``relabel_input=False`` (inference against a cached layer-0 projection): the input block keeps only its
        global ``picks`` — no hash relabel, no size read-back; ``input_nodes`` is then ``None``.
        ``ctrs``: the batches' Philox counters (default: the next ones of the stream, see ``reserve_ctrs``).
        ``fuse_rows`` > 0 (with ``relabel_input=False``, two layers, the seed batches consecutive slices of one tensor): consecutive
        batches are laid end to end into ONE pair of blocks while their hidden-layer rows stay below ``fuse_rows`` — every batch is
        sampled with its own counter exactly as before (same picks, same per-batch de-duplication), only the launches of the
        forward pass are shared.  The returned list then has one entry per CHUNK: (None, the chunk's seeds, blocks)."""
        g = graph.handle
        L = len(self.fanouts)
        nb = len(seed_batches)
        if ctrs is None:
            ctrs = reserve_ctrs(nb)
        assert len(ctrs) == nb
        if nb == 0:
            return []
        if relabel_input:
            job = self._job(graph, seed_batches, ctrs)
            try:
                sizes = next(job)
                while True:
                    sizes = job.send(sizes.cpu().tolist())            # the one sync of this layer
            except StopIteration as done:
                return done.value
        blocks = [[None] * L for _ in seed_batches]
        # every batch of a layer goes through ONE sampler launch and ONE block-build sequence (batched C-ABI entry points):
        # batch b's destinations are dst_base[starts[b] : starts[b] + counts[b]]
        counts = [int(t.numel()) for t in seed_batches]
        dst_base = seed_batches[0] if nb == 1 else torch.cat([t.reshape(-1) for t in seed_batches])
        starts, acc = [], 0
        for c in counts:
            starts.append(acc); acc += c
        for layer in reversed(range(L)):
            S = self.fanouts[layer]
            picks_all = ops.sample_layer_batched(g, dst_base, starts, counts, S, _STATE["seed"], ctrs, layer)
            rows, acc = [], 0
            for c in counts:
                rows.append(acc); acc += c
            dsts = [dst_base[s:s + c] for s, c in zip(starts, counts)]
            if layer == 0 and not relabel_input:
                if fuse_rows and L == 2:
                    return self._fused(seed_batches, blocks, dsts, rows, counts, picks_all, int(fuse_rows))
                for bi in range(nb):
                    blocks[bi][layer] = Block(None, dsts[bi], None, picks_all[rows[bi]:rows[bi] + counts[bi]])
                return [(None, seeds, blk) for seeds, blk in zip(seed_batches, blocks)]
            src_all, n_src, lidx_all = ops.build_block_batched_async(dst_base, starts, counts, picks_all, n_ids=g.n)
            n_next = n_src[:nb].cpu().tolist()                            # the one sync of this layer
            nstarts = []
            for bi in range(nb):
                r0, c = rows[bi], counts[bi]
                s0 = r0 * (1 + S)
                src = src_all[s0:s0 + n_next[bi]]
                blocks[bi][layer] = Block(src, dsts[bi], lidx_all[r0:r0 + c], picks_all[r0:r0 + c])
                nstarts.append(s0)
            dst_base, starts, counts = src_all, nstarts, n_next
        raise AssertionError("unreachable: a loader that relabels its input layer goes through _job")

    def _job(self, graph, seed_batches, ctrs):
        """The relabelled loader as a resumable job: launches one layer (sampler + block build, batched over every batch), YIELDS the
        device tensor of that layer's block sizes and is sent their host copy — the caller decides how to wait (``sample_batches``:
        on the spot; ``sample_batches_stream``: on a second stream, between train steps).  Returns the loader's list."""
        g = graph.handle
        L = len(self.fanouts)
        nb = len(seed_batches)
        counts = [int(t.numel()) for t in seed_batches]
        dst_base = _packed(seed_batches, counts)
        starts, acc = [], 0
        for c in counts:
            starts.append(acc); acc += c
        layers = [None] * L
        for layer in reversed(range(L)):
            S = self.fanouts[layer]
            picks_all = ops.sample_layer_batched(g, dst_base, starts, counts, S, _STATE["seed"], ctrs, layer)
            src_all, n_src, lidx_all = ops.build_block_batched_async(dst_base, starts, counts, picks_all, n_ids=g.n)
            n_next = yield n_src[:nb]
            # (only what the NEXT layer's launches need is computed here; the per-batch Block objects — four tensor views each —
            # are made when a batch is handed out: the host's share of a 50-batch loader left the GPU idle for ~0.2 ms per layer)
            rows, acc = [], 0
            for c in counts:
                rows.append(acc); acc += c
            layers[layer] = (S, dst_base, starts, counts, rows, src_all, lidx_all, picks_all, n_next)
            dst_base, starts, counts = src_all, [r * (1 + S) for r in rows], n_next
        return _LazyBatches(seed_batches, layers)

    def sample_batches_stream(self, graph, seed_batches, ctrs=None, first=None):
        """``sample_batches`` as a generator that starts yielding after the first ``first`` batches are sampled; the rest is sampled
        on a second HIP stream while the consumer trains on those (module header: PIPELINE).  Same blocks, same counters."""
        nb = len(seed_batches)
        if ctrs is None:
            ctrs = reserve_ctrs(nb)
        first = PIPELINE_FIRST if first is None else int(first)
        dev = seed_batches[0].device if nb else None
        if not PIPELINE or first <= 0 or nb < 2 * first or dev is None or dev.type != "cuda" or ops._PROFILE is not None:
            yield from self.sample_batches(graph, seed_batches, ctrs=ctrs)
            return
        head = self.sample_batches(graph, seed_batches[:first], ctrs=ctrs[:first])
        rest = _StreamJob(self._job(graph, seed_batches[first:], ctrs[first:]), dev)
        for item in head:
            yield item
            rest.poll()                      # (between two train steps: never inside a graph capture)
        yield from rest.finish()


    @staticmethod
    def _fused(seed_batches, blocks, dsts, rows, counts, picks_all, fuse_rows):
        """Chunks of consecutive batches as one (input block, output block) pair each — see ``sample_batches``.  rows / counts:
        where batch b's hidden-layer rows sit in the packed layer-0 arrays (they are consecutive); the output blocks' index
        arrays are consecutive row ranges of one packed array too (the seeds are)."""
        nb = len(seed_batches)
        out = []
        b0 = 0
        while b0 < nb:
            b1, acc = b0, 0
            while b1 < nb and b1 - b0 < 64 and (b1 == b0 or acc + counts[b1] <= fuse_rows):
                acc += counts[b1]; b1 += 1
            r0, r1 = rows[b0], rows[b1 - 1] + counts[b1 - 1]
            dev = picks_all.device
            seeds = [seed_batches[b] for b in range(b0, b1)]
            # the chunk's seeds and output-block index rows: consecutive slices -> views of their parents
            first, last = seeds[0], seeds[-1]
            base = first._base if first._base is not None else first
            s_lo = first.storage_offset() - base.storage_offset()
            n_seeds = sum(int(t.numel()) for t in seeds)
            seeds_chunk = base.reshape(-1)[s_lo:s_lo + n_seeds]
            assert last.data_ptr() + last.numel() * 8 == seeds_chunk.data_ptr() + n_seeds * 8, "seed batches must be consecutive slices"
            l_first = blocks[b0][1].local_idx
            lbase = l_first._base if l_first._base is not None else l_first
            S1 = l_first.shape[1]
            l_lo = (l_first.storage_offset() - lbase.storage_offset()) // S1
            lidx = lbase.reshape(-1, S1)[l_lo:l_lo + n_seeds]
            seg_rows, seg_offs, acc_r = [0], [], 0
            for b in range(b0, b1):
                acc_r += int(seed_batches[b].numel())
                seg_rows.append(acc_r)
                seg_offs.append(rows[b] - r0)
            flag = ops.fill_zero(torch.empty(r1 - r0, dtype=torch.uint8, device=picks_all.device)) if FUSED_KEEP_ROWS else None
            dst_pos = ops.fuse_block_segments(lidx, seg_rows, seg_offs, dst_flag=flag)  # (in place: the per-batch blocks are not handed out)
            # the hidden layer's vertex ids of the chunk, end to end (the batches' source lists sit apart in the packed array)
            ids0 = torch.empty(r1 - r0, dtype=torch.int64, device=dev)
            for k in range(b0, b1, 8):
                ops.stage_segments([(dsts[b], ids0[rows[b] - r0:rows[b] - r0 + counts[b]], counts[b]) for b in range(k, min(k + 8, b1))])
            blk0 = Block(None, ids0, None, picks_all[r0:r1])
            blk1 = Block(ids0, seeds_chunk, lidx, None, dst_pos=dst_pos, dst_flag=flag)
            out.append((None, seeds_chunk, [blk0, blk1]))
            b0 = b1
        return out


def _packed(seed_batches, counts):
    """The batches' seeds end to end: a view when they are consecutive slices of one contiguous tensor (a loader's are), else a copy."""
    if len(seed_batches) == 1:
        return seed_batches[0].reshape(-1)
    first = seed_batches[0]
    base = first._base
    if base is not None and base.is_contiguous() and first.dim() == 1:
        off, ok = first.storage_offset(), True
        for t, c in zip(seed_batches, counts):
            if t._base is not base or t.dim() != 1 or t.storage_offset() != off or (c > 1 and t.stride(0) != 1):
                ok = False
                break
            off += c
        if ok:
            lo = first.storage_offset() - base.storage_offset()
            return base.reshape(-1)[lo:lo + sum(counts)]
    return torch.cat([t.reshape(-1) for t in seed_batches])


class _LazyBatches:
    """The loader's (input_nodes, seeds, blocks) triples, each built from the layers' packed arrays when it is first asked for."""

    def __init__(self, seed_batches, layers):
        self.seeds, self.layers = seed_batches, layers
        self.made = [None] * len(seed_batches)

    def __len__(self):
        return len(self.seeds)

    def __getitem__(self, bi):
        if isinstance(bi, slice):
            return [self[i] for i in range(*bi.indices(len(self)))]
        if bi < 0:
            bi += len(self)
        if self.made[bi] is None:
            blk = []
            for S, dst_base, starts, counts, rows, src_all, lidx_all, picks_all, n_next in self.layers:
                r0, c, s0 = rows[bi], counts[bi], rows[bi] * (1 + S)
                blk.append(Block(src_all[s0:s0 + n_next[bi]], dst_base[starts[bi]:starts[bi] + c], lidx_all[r0:r0 + c], picks_all[r0:r0 + c]))
            self.made[bi] = (blk[0].src_ids, self.seeds[bi], blk)
        return self.made[bi]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _StreamJob:
    """Drives a sampler job (``MultiLayerNeighborSampler._job``) on the sampling stream: its launches are enqueued there, the block
    sizes come back through pinned memory + an event the host polls between train steps and waits for only when it has run out of
    sampled batches.  The consumer's stream waits for the job's last event before the first of its batches is handed out; the
    tensors are marked as used by that stream (the caching allocator must not recycle them for the sampling stream while queued
    train steps still read them)."""

    def __init__(self, job, device):
        if _PIPE["stream"] is None:
            _PIPE["stream"] = torch.cuda.Stream(device=device)
            _PIPE["pinned"] = torch.empty(8192, dtype=torch.int64, pin_memory=True)
        self.side = _PIPE["stream"]
        self.main = torch.cuda.current_stream(device)
        self.job, self.out, self.pending, self.stage = job, None, None, 0
        self.side.wait_stream(self.main)     # the seeds' upload, the previous loader's last reads of recycled memory
        self._advance(None)

    def _advance(self, value):
        with torch.cuda.stream(self.side):
            try:
                sizes = next(self.job) if value is None else self.job.send(value)
            except StopIteration as done:
                self.out, self.pending = done.value, None
                self.done_event = torch.cuda.Event()
                self.done_event.record(self.side)
                return
            # (a slice of ONE pinned buffer allocated with the stream: a fresh pinned allocation per stage is a hipHostMalloc whenever
            # the loader's length changes — a fraction of a millisecond, inside the first train steps of the loader)
            n, cap = sizes.numel(), _PIPE["pinned"].numel()
            if (self.stage + 1) * n <= cap:
                host = _PIPE["pinned"][self.stage * n:(self.stage + 1) * n]
            else:
                host = torch.empty(sizes.shape, dtype=sizes.dtype, pin_memory=True)
            self.stage += 1
            host.copy_(sizes, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.pending = (host, ev)

    def poll(self):
        if self.pending is not None and self.pending[1].query():
            self._advance(self.pending[0].tolist())

    def finish(self):
        while self.pending is not None:
            self.pending[1].synchronize()
            self._advance(self.pending[0].tolist())
        self.main.wait_event(self.done_event)
        for _, seeds, blocks in self.out:
            for blk in blocks:
                for t in (blk.src_ids, blk.dst_ids, blk.local_idx, blk.picks):
                    if t is not None and t.is_cuda:
                        t.record_stream(self.main)
        return self.out


class NodeDataLoader:
    """Iterable of ``(input_nodes, seeds, blocks)`` in seed order; the last partial batch is kept
    unless ``drop_last``; ``shuffle`` permutes the seeds first (torch CPU generator, as DataLoader does)."""

    def __init__(self, graph, nids, sampler, batch_size=1, shuffle=False, drop_last=False, num_workers=0,
                 relabel_input=True):
        self.relabel_input = relabel_input
        if batch_size is None or batch_size <= 0:
            raise ValueError("batch_size should be a positive integer value, but got batch_size={}".format(batch_size))
        self.graph, self.sampler = graph, sampler
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.num_workers = num_workers   # accepted for API parity; sampling is on the GPU
        nids = torch.as_tensor(nids, dtype=torch.int64).reshape(-1)
        self.nids = nids

    def __len__(self):
        n = self.nids.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        nids = self.nids
        if self.shuffle:
            nids = nids[torch.randperm(nids.numel())]
        nids = nids.to(self.graph.device, non_blocking=True).contiguous()
        n, bs = nids.numel(), self.batch_size
        stops = list(range(0, n, bs))
        batches = [nids[s:s + bs] for s in stops if not (self.drop_last and s + bs > n)]
        if self.relabel_input:
            return self.sampler.sample_batches_stream(self.graph, batches)
        return iter(self.sampler.sample_batches(self.graph, batches, relabel_input=False))
