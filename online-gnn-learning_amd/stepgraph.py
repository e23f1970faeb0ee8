"""Captured train steps: hipGraph launches per replay batch instead of ~40 launches enqueued from Python.

The update path of a batch is a fixed sequence of microsecond-to-sub-millisecond kernels (sample x2, block build x2,
forward, cross entropy, backward, Adam).  Enqueued one by one through ctypes + autograd it costs the host 0.8-0.9 ms per
step — the small rungs (pubmed / arxiv, 32 seeds) are 100 % launch- and sync-bound that way, and the Reddit rung is within
30 % of it.  The classes here record those launches ONCE (``torch.cuda.graph``: a HIP stream capture of exactly the C-ABI
launches the eager path makes, autograd backward and the optimiser step included) on static buffers and replay them.
What varies between replays lives in device memory: the seeds / block arrays (static buffers), the Philox batch counter
(``ogl_sample_layer_dev``), Adam's step count (``ogl_adam_step_multi_dev``).

* ``TrainStepGraph`` — forward + loss + backward + Adam of ONE batch on block arrays of a size BUCKET (``n0`` rounded up;
  id -1 pads).  Padding is exact, not approximate (include/ogl_hip.h, "Padding contract"): id -1 samples nothing / gathers
  the zero row, so padded rows add exact zeros to every product and gradient; only the fp32 summation ORDER of split
  reductions can differ from the unpadded eager step.
* ``SampleGraph`` (small batches) — sampling + block construction of one batch on upper-bound buffers
  (``n1 <= B (1 + S)``, ``n0 <= n1 (1 + S)``), straight into the static block arrays the train graphs read: a step is a
  host write of ``[counter | seeds]`` into mapped pinned memory (the graph's first kernel reads it), the sample graph, ONE 16-byte read-back (the two source counts: they choose
  the train graph's bucket, so that the GEMMs run at the batch's size, not at 28x the upper bound), the train graph.  The
  read-back is not a copy: the graph's last kernel stores the counts and a sequence number into pinned host memory
  (``ogl_publish_i64``) and the host polls it (52 us for sample graph + read-back, 72 with a copy node + event).
* staged (Reddit rung): the loader samples the snapshot's batches as before (one read-back per layer per LOADER, amortised
  over 50 batches); per batch ONE ``ogl_stage_segments`` launch copies the batch's block arrays into the static buffers
  of the train graph captured for its bucket (``n1`` to 256, ``n0`` to 2 048 rows: <= 3 % padded rows).

hipMemsetAsync must not appear in captured code: on ROCm 7.2 a memset node re-runs on 1/16 of its range from the second
replay on (tools/graph_probe.py); every fill on these paths is a kernel.

Which steps are replayed is the strategy's policy (``HipSupervisedGraphSage.use_graphs``): small batches always; large,
loader-fed batches only when a timed snapshot shows the host cannot keep ahead of the GPU — replayed nodes run ~1 us
further apart than eagerly queued launches, so on a fast host eager is 2-4 % faster at the Reddit rung.
Under torch.distributed a replica's step is captured up to and including backward (``apply=False``); the gradient all-reduce
and the optimiser stay eager (two collectives + one launch per step).  Not captured: dropout > 0 (its counter is host-side).
"""
from __future__ import annotations

import collections
import os
import time

import numpy as np

import torch

from . import ops, sampling
from .graphsage.sageconv import GatheredRows


def round_up(x, m):
    return (int(x) + m - 1) // m * m


N1_BUCKET = 256          # staged form
N0_BUCKET = 2048
N0_BUCKET_SMALL = 256    # sampled form (input blocks of a few hundred to a few thousand rows)
SAMPLE_FILL_BUCKET = True   # the sample graph pads src0 up to that bucket only
# A sampled step whose launches take their sizes from the DEVICE (the small first layer on the sample graph's own counts: ops._SMALL_AGNOSTIC)
# is captured ONCE on the upper-bound block and replayed right behind its sample graph — no read-back in front of the train graph (it picked
# the size bucket), no host in the device's critical path: the counts are read after both graphs are enqueued.
SIZE_AGNOSTIC = os.environ.get("OGL_SIZE_AGNOSTIC", "1") != "0"
# (Three forms of the PIPELINED step on a size-agnostic train graph — one graph per batch with the next batch's sampling on a forked
# branch; two graphs with only the train graph size-agnostic; only a snapshot's first batch as the one-graph step — and the next batch's
# sample graph enqueued BEFORE the train graph were measured slower in round 5 and removed in round 6: DESIGN.md section 8.)
_WARMED = False


class BlockBuffers:
    """The static block arrays of one batch shape: what a train graph reads, what staging / a sample graph writes."""

    def __init__(self, B, S, n1_cap, n0_cap, device):
        self.B, self.S, self.n1_cap, self.n0_cap = int(B), int(S), int(n1_cap), int(n0_cap)
        self.head = torch.zeros(1 + self.B, dtype=torch.int64, device=device)          # [Philox batch counter | seeds]
        self.src0 = torch.full((self.n0_cap,), -1, dtype=torch.int64, device=device)
        self.src1 = torch.full((self.n1_cap,), -1, dtype=torch.int64, device=device)
        self.lidx0 = torch.full((self.n1_cap, self.S), -1, dtype=torch.int32, device=device)
        self.lidx1 = torch.full((self.B, self.S), -1, dtype=torch.int32, device=device)
        self.counts = None          # sampled form: the device [n1, n0] the sample graph writes (SampleGraph sets it)

    @property
    def seeds(self):
        return self.head[1:]


LAZY_LABELS = True     # the step's label gather inside the loss launch


class TrainStepGraph:
    """One captured train step over ``buf`` restricted to (n1_pad, n0_pad) rows.
    ``loss_fn(logits, labels) -> (scalar loss to differentiate, per-seed losses or None)``."""

    def __init__(self, model, optimizer, graph, buf, n1_pad, n0_pad, loss_fn, pool=None, apply=True, loss_kind=None, dp=None, sampler=None):
        # apply=False: forward + loss + backward only (the gradients are in ``grads``; exchange and optimiser are the caller's).
        # dp = (GradSynchronizer, weight): the WHOLE step of a data-parallel replica — forward, loss, backward, the gradient exchange
        # (RCCL all-reduces recorded into the graph: the early bucket launched from the gradient hooks on the side branch, under the
        # layer-0 backward; the late one after it) and the optimiser (device-side step count) — replayed with one host call.
        self.model, self.opt, self.graph, self.buf, self.loss_fn, self.apply = model, optimizer, graph, buf, loss_fn, bool(apply)
        self.dp = dp if apply else None
        # sampler (a SampleGraph over ``buf``): its one launch is recorded at the TOP of this graph — sampling and training of a batch
        # are ONE graph launch (for a train graph that takes the block's size from the device: StepGraphCache._agnostic_graph)
        self.sampler = sampler
        # "mean" / "mean_rows": the loss is nn.CrossEntropyLoss — the model may run its last layer and the loss as one node
        # (GraphSAGE.forward_loss); None: an arbitrary loss_fn(logits, labels)
        self.loss_kind = loss_kind if hasattr(model, "forward_loss") else None
        self.n1_pad, self.n0_pad = int(n1_pad), int(n0_pad)
        assert self.n1_pad <= buf.n1_cap and self.n0_pad <= buf.n0_cap
        self.loss = self.loss_rows = None
        self.cuda_graph = torch.cuda.CUDAGraph()
        self._capture(pool)

    def _body(self, apply=None, learn=False):
        apply = self.apply if apply is None else apply
        g, b = self.graph, self.buf
        if self.sampler is not None:
            self.sampler._body()
        src0, src1, lidx0 = b.src0[:self.n0_pad], b.src1[:self.n1_pad], b.lidx0[:self.n1_pad]
        blocks = [sampling.Block(src0, src1, lidx0), sampling.Block(src1, b.seeds, b.lidx1)]
        if b.counts is not None:
            # the sampled form runs on the upper-bound block: n1_cap destination rows of which counts[0] are live — the small first layer's
            # kernels skip the padded ones (a device scalar the sample graph rewrites before every replay)
            blocks[0].n_live_dev = b.counts[:1]
            if self.n0_pad == b.n0_cap:
                blocks[0].n_src_live_dev = b.counts[1:2]      # ... and, sized for the whole source list, its live length too
        # the FULL tables: a snapshot view's row count would be frozen into the graph (ids are < n_present by construction)
        labels = ops.LazyLabels(g.target_table, b.seeds) if LAZY_LABELS else ops.gather_i64(g.target_table, b.seeds)
        self.opt.zero_grad(set_to_none=True)
        if apply and hasattr(self.opt, "prime"):
            self.opt.prime()                          # (the optimiser's per-step scalars ride in the step's first launch)
        if self.loss_kind is not None:
            loss, rows, _ = self.model.forward_loss(blocks, GatheredRows(g.feat_table, src0), labels, rows=self.loss_kind == "mean_rows",
                                                    defer_mean=True)    # (every branch below runs the backward before anything reads it)
        else:
            logits = self.model(blocks, GatheredRows(g.feat_table, src0))
            loss, rows = self.loss_fn(logits, labels)
        if apply and self.dp is not None:
            gsync, weight = self.dp
            gsync.begin_step(weight)                  # (the hooks launch the early bucket's all-reduce with this rank's weight)
            ops.backward(loss)
            gsync.sync(weight)                        # late bucket, wait for the early one, p.grad <- views of the reduced buckets
            self.opt.step()
        elif apply and hasattr(self.opt, "backward_and_step"):
            self.opt.backward_and_step(loss)          # split-K slabs summed by the optimiser launch, its early part on the side branch
        elif learn and hasattr(self.opt, "backward_learn"):
            self.opt.backward_learn(loss)             # (the warm-up pass: records the order in which the gradients arrive)
        else:
            ops.backward(loss)
            if apply:
                self.opt.step()
        self.loss, self.loss_rows = loss.detach(), (rows.detach() if rows is not None else None)

    def _capture(self, pool):
        global _WARMED
        if self.apply and hasattr(self.opt, "prepare_capture"):
            assert getattr(self.opt, "capturable", False), "a captured step needs optim.Adam(capturable=True)"
            self.opt.prepare_capture()
        # (any other optimiser is recorded as it is: its step() must not synchronise — torch.optim.SGD without momentum does not)
        ops.unit_grad(self.graph.device)
        ops.ce_counter(self.graph.device, 0)                 # (allocates the device's counter array outside the capture)
        ops._static_image(self.graph.feat_table)            # built outside the capture (a one-off 850 MB split pass)
        learn = self.apply and self.dp is None and getattr(self.opt, "needs_order", False)
        if not _WARMED or learn:
            # once per process: run the step's forward + backward for real on a side stream (autograd's device thread, lazily
            # created helpers), WITHOUT the optimiser step and with the gradients dropped — the weights do not move.  (Also once
            # per optimiser that applies its update in two parts: the pass tells it the order in which the gradients arrive, so
            # that the capture below already has the early part on its side branch.)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if self.dp is not None:                 # (a replica: the warm-up pass must not talk to the other ranks)
                    with self.dp[0].no_sync():
                        self._body(apply=False)
                else:
                    self._body(apply=False, learn=learn)
            torch.cuda.current_stream().wait_stream(side)
            self.opt.zero_grad(set_to_none=True)
            ops.invalidate_weight_images()          # no optimiser step ran: drop the weight images that forward prepared
            _WARMED = True
        ops._flush_deferred()                       # (side work an earlier eager forward left for "after the next launch")
        torch.cuda.synchronize()
        kw = {} if pool is None else dict(pool=pool)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # RCCL's watchdog thread polls its events (hipEventQuery) all the time; under the default "global" capture mode a
            # query from ANY thread while this one captures is an error that invalidates the capture and kills the watchdog
            # (found by tests/test_gpu_nccl.py).  "thread_local": only this thread's own calls are policed.
            torch.cuda.synchronize()
            kw["capture_error_mode"] = "thread_local"
        ops._SMALL_AGNOSTIC["seen"] = False
        with torch.cuda.graph(self.cuda_graph, **kw):
            self._body()
        # (True: the recorded launches take the input block's size from the device — this graph serves every batch of its shape)
        self.size_agnostic = bool(ops._SMALL_AGNOSTIC["seen"]) and self.n0_pad == self.buf.n0_cap
        self.grads = [p.grad for p in self.model.parameters()]
        for sm in (self.sampler,):
            if sm is not None:
                # The caller has prepare()d this sampler for the replay that follows (host sequence number = device's + 1).  A warm-up
                # pass above runs the sampler's launch for real: the device's number then moves without the host's — and a host that
                # is one behind would see the PREVIOUS launch's publication as this one's, i.e. overwrite the mapped seed buffer
                # before the launch that reads it has run.  Re-based on what the device holds now.
                torch.cuda.synchronize()
                sm.seq = int(sm.counts_np[2]) + 1

    def replay(self):
        """Replays the step.  ``self.loss`` / ``self.loss_rows`` / ``self.grads`` are STATIC tensors of the graph: the next replay
        of this bucket (or its re-capture after an eviction) overwrites them — a caller that keeps per-step values clones them."""
        self.cuda_graph.replay()
        if self.apply:
            # the replayed optimiser moved the weights under raw pointers: bf16x3 weight images an EAGER pass built before
            # (evaluation / priority forward between snapshots) are stale now — optim.Adam.step() drops them on the eager
            # path, a replay has to do it here (keys are (data_ptr, _version): neither changes)
            ops.invalidate_weight_images()
        return self.loss


class SampleGraph:
    """Sampling + block construction of one batch, captured on upper-bound shapes, writing ``buf``.

    The output block is sampled for the B seeds; its source list (capacity B (1 + S), -1 past the n1 found) is the input
    block's destination list AS IS — padded destinations sample nothing and keep their (all-zero) block rows, so the input
    block always has ``n1_cap`` destination rows and ``n1_cap + #new`` sources.  ``run`` returns that source count."""

    SPIN_SECONDS = 0.002     # polling budget before the synchronise fallback (a sample graph takes ~50 us)

    def __init__(self, graph, buf):
        self.graph, self.buf = graph, buf
        self.sync_fallbacks = 0
        assert buf.n1_cap == buf.B * (1 + buf.S) and buf.n0_cap == buf.n1_cap * (1 + buf.S)
        self.head_host = torch.zeros(1 + buf.B, dtype=torch.int64).pin_memory()
        self.head_np = self.head_host.numpy()                # (written through numpy: a torch tensor per step costs the host ~10 us)
        self.counts = torch.zeros(2, dtype=torch.int64, device=graph.device)
        buf.counts = self.counts
        # [n1, n0, sequence number]: pinned host memory the LAST kernel of the graph writes directly (ogl_publish_i64)
        self.counts_host = torch.zeros(3, dtype=torch.int64).pin_memory()
        self.counts_np = self.counts_host.numpy()
        self.seq_dev = torch.zeros(1, dtype=torch.int64, device=graph.device)
        self.seq = 0
        self.seed = sampling.get_state()["seed"]
        self.cuda_graph = torch.cuda.CUDAGraph()
        self._body()                                         # once for real (helpers, workspaces), then recorded
        torch.cuda.synchronize()
        self.seq = int(self.counts_np[2])
        with torch.cuda.graph(self.cuda_graph):
            self._body()

    def _body(self):
        g, b, S = self.graph, self.buf, self.buf.S
        if ops.sample_blocks_small_fits(b.B, S):
            # ONE launch for the whole phase (round 5: eleven 4-5 us graph nodes before — stage, 2 x sample, 2 x block build, publish)
            # (src0 is read up to the train graph's size bucket, round_up(n0, N0_BUCKET_SMALL): its -1 padding stops there)
            self._ws = ops.sample_blocks_small(g.handle, self.head_host, b.head, b.B, S, self.seed, b.src1, b.lidx1, b.src0, b.lidx0,
                                               self.counts, self.seq_dev, self.counts_host, ws=getattr(self, "_ws", None),
                                               src0_fill=N0_BUCKET_SMALL if SAMPLE_FILL_BUCKET else 0)
            return
        ctr = b.head[:1]
        # [counter | seeds]: read by the graph's first kernel straight from the pinned host buffer run() filled (mapped memory:
        # no copy node — a 264-byte hipMemcpyAsync is an 18 us blit kernel on the device and a runtime call on the host)
        ops.stage_segments([(self.head_host, b.head, 1 + b.B)])
        # straight into the static block arrays the train graphs read
        picks1 = ops.sample_layer_dev(g.handle, b.seeds, S, self.seed, ctr, 1)
        ops.build_block_async(b.seeds, picks1, pad_tail=True, out=(b.src1, self.counts[:1], b.lidx1))   # -1 past n1
        picks0 = ops.sample_layer_dev(g.handle, b.src1, S, self.seed, ctr, 0)
        ops.build_block_async(b.src1, picks0, pad_tail=True, out=(b.src0, self.counts[1:], b.lidx0))    # -1 past n0
        ops.publish_i64(self.counts, self.seq_dev, self.counts_host)

    def run(self, seeds_host, ctr):
        """seeds_host: int64 array-like [B] (snapshot ids); ctr: this batch's Philox counter.  Returns (n1, n0)."""
        self.launch(seeds_host, ctr)
        return self.wait()

    def prepare(self, seeds_host, ctr):
        """[counter | seeds] into the mapped host buffer the sampling launch reads — for a caller that replays a graph which CONTAINS
        that launch (TrainStepGraph(sampler=...)); ``wait()`` afterwards as after ``launch``."""
        h = self.head_np
        h[0] = int(ctr)
        h[1:] = seeds_host                                   # (not touched again until the graph's last store is seen)
        self._stream = None
        self.seq += 1

    def launch(self, seeds_host, ctr, stream=None):
        """Enqueue the sample graph (on ``stream``: the pipelined steps run it beside the previous batch's train graph)."""
        h = self.head_np
        h[0] = int(ctr)
        h[1:] = seeds_host                                   # (not touched again until the graph's last store is seen)
        if stream is None:
            self.cuda_graph.replay()
        else:
            with torch.cuda.stream(stream):
                self.cuda_graph.replay()
        self._stream = stream
        self.seq += 1

    def wait(self):
        """(n1, n0) of the launched batch, once the graph's last kernel has published them."""
        t0 = time.perf_counter()
        c, want = self.counts_np, self.seq
        spins = 0
        # The step's one read-back: 16 bytes the graph's last kernel wrote into pinned host memory (system-scope fences
        # around the sequence number); polled, no copy node, no event.  This relies on torch's pinned allocations being
        # host-coherent (fine-grained: hipHostMalloc's default, HIP_HOST_COHERENT unset or 1).  Where they are not, the
        # store only becomes visible when the stream drains — so after SPIN_SECONDS of polling the host falls back to a
        # stream synchronise (after which the data must be there) instead of burning a core until a timeout.
        while c[2] != want:
            spins += 1
            if spins & 0x3FF == 0 and time.perf_counter() - t0 > self.SPIN_SECONDS:
                (self._stream if getattr(self, "_stream", None) is not None else torch.cuda.current_stream()).synchronize()
                self.sync_fallbacks += 1
                if c[2] != want:
                    raise RuntimeError("sample graph: no block sizes from the device (sequence %d, saw %d) although its stream "
                                       "is idle" % (want, int(c[2])))
        return int(c[0]), int(c[1])


class StepGraphCache:
    """The captured graphs of one (model, optimiser): keyed by form and padded sizes, captured on first use.

    A captured step owns its intermediates (a private memory pool: ~1 GB at the Reddit rung), and a long stream walks
    through many size buckets as the graph grows, so the train graphs are kept least-recently-used up to ``MAX_GRAPHS``;
    an evicted bucket is simply captured again if it comes back."""

    MAX_GRAPHS = 24

    def __init__(self, model, optimizer, S, loss_fn, loss_kind=None):
        self.model, self.opt, self.S, self.loss_fn, self.loss_kind = model, optimizer, int(S), loss_fn, loss_kind
        self.bufs, self.samplers = {}, {}
        self.agnostic = {}            # sampler key -> True / False once known (SIZE_AGNOSTIC)
        self.graphs = collections.OrderedDict()
        self.captures = self.evictions = self.borrowed = self.deferred = 0
        self.sightings = {}

    def _train(self, graph, buf, key, n1_pad, n0_pad, apply=True, dp=None, sampler=None):
        sg = self.graphs.get(key)
        if sg is None:
            while len(self.graphs) >= self.MAX_GRAPHS:
                old_key, old = self.graphs.popitem(last=False)
                if old_key[0] == "staged":                    # its static buffers belong to that bucket alone
                    self.bufs.pop(old_key, None)
                del old
                self.evictions += 1
            sg = self.graphs[key] = TrainStepGraph(self.model, self.opt, graph, buf, n1_pad, n0_pad, self.loss_fn, apply=apply,
                                                   loss_kind=self.loss_kind, dp=dp, sampler=sampler)
            self.captures += 1
        else:
            self.graphs.move_to_end(key)
        return sg

    def sampled_step_pipelined(self, graph, seeds_host, ctr, nxt=None):
        """``sampled_step`` with the NEXT batch's sampling (``nxt = (seeds_host, ctr)`` or None) enqueued on a second stream before
        this batch's train graph is waited for: sampling does not depend on the weights, so batch i + 1 is sampled (36 us of one
        workgroup + the host's read-back and launch: ~55 us at the arxiv-like rung) while batch i trains.  Two sets of block arrays
        and two sets of train graphs (a sample graph must not write what a train graph in flight still reads): set i % 2; the side
        stream waits for the train graph that last read its set, the main stream for the sample graph that filled it.  Same blocks,
        same counters, same results as ``sampled_step``."""
        B = len(seeds_host)
        bkey = ("sampled2", id(graph), B, sampling.get_state()["seed"])
        pipe = self.samplers.get(bkey)
        if pipe is None:
            n1_cap = B * (1 + self.S)
            bufs = [BlockBuffers(B, self.S, n1_cap, n1_cap * (1 + self.S), graph.device) for _ in range(2)]
            self.bufs[bkey] = bufs
            pipe = self.samplers[bkey] = dict(smp=[SampleGraph(graph, b) for b in bufs], side=torch.cuda.Stream(device=graph.device),
                                              sampled=[None, None], trained=[None, None], cur=0, ahead=None, last=None)
        cur = pipe["cur"]
        smp = pipe["smp"][cur]
        main = torch.cuda.current_stream()
        # What the side stream waits for before it re-samples a set is recorded HERE, at the top of the call that follows the set's
        # train graph — not right behind that graph's replay: the caller reads the step's static outputs (seeds, per-seed losses) on
        # the main stream after the replay returns, and the sampling launch's first act is to overwrite the seeds.  An event
        # recorded behind the replay left those reads unordered against the re-sampling (PBR could pair batch i's losses with
        # batch i + 2's seeds once the host ran a step ahead).
        last = pipe.get("last")
        if last is not None:
            ev = pipe["trained"][last] = torch.cuda.Event()
            ev.record(main)
            pipe["last"] = None
        if pipe["ahead"] is not None and pipe["ahead"][0] == cur and pipe["ahead"][1] == int(ctr):
            n1, n0 = smp.wait()                              # launched while the previous batch trained
            main.wait_event(pipe["sampled"][cur])
        else:
            if pipe["ahead"] is not None:                    # (a prefetch nobody came for: let it finish before its set is reused)
                pipe["smp"][pipe["ahead"][0]].wait()
                main.wait_event(pipe["sampled"][pipe["ahead"][0]])
            n1, n0 = smp.run(seeds_host, ctr)
        pipe["ahead"] = None
        n0_pad = min(round_up(n0, N0_BUCKET_SMALL), smp.buf.n0_cap)
        sg = self._train(graph, smp.buf, bkey + (n0_pad, cur), smp.buf.n1_cap, n0_pad)
        other = 1 - cur
        sg.replay()
        pipe["last"] = cur                                   # (its `trained` event: at the top of the next call, behind the caller's reads)
        sg.last_sizes = (n0, n1)
        if nxt is not None and len(nxt[0]) == B:
            side = pipe["side"]
            if pipe["trained"][other] is not None:
                side.wait_event(pipe["trained"][other])      # the train graph that read set `other` has finished, and its caller's reads
            pipe["smp"][other].launch(nxt[0], nxt[1], stream=side)
            es = pipe["sampled"][other] = torch.cuda.Event()
            es.record(side)
            pipe["ahead"] = (other, int(nxt[1]))
        pipe["cur"] = other
        return sg

    def sampled_step(self, graph, seeds_host, ctr):
        """One whole step of a small batch: sample graph -> 16-byte read-back -> the train graph of the size bucket."""
        B = len(seeds_host)
        bkey = ("sampled", id(graph), B, sampling.get_state()["seed"])
        smp = self.samplers.get(bkey)
        if smp is None:
            n1_cap = B * (1 + self.S)
            buf = self.bufs[bkey] = BlockBuffers(B, self.S, n1_cap, n1_cap * (1 + self.S), graph.device)
            smp = self.samplers[bkey] = SampleGraph(graph, buf)
        if SIZE_AGNOSTIC and self.agnostic.get(bkey) is not False:
            # ONE graph per step: the sampling launch recorded at the top of a train graph sized for the upper-bound block (its first
            # layer takes the live sizes from the device) — one host call, no read-back and no graph-to-graph hand-over (8 us on this
            # part) between sampling and training
            smp.prepare(seeds_host, ctr)
            sg = self._agnostic_graph(graph, smp, bkey, bkey + ("cap", "merged"), merged=True)
            if sg is not None:
                sg.replay()
                n1, n0 = smp.wait()                          # (after the step is enqueued: the device never waits for this)
                sg.last_sizes = (n0, n1)
                return sg
            torch.cuda.synchronize()                         # (not size-agnostic: back to sample graph -> read-back -> bucket's graph)
            smp.seq = int(smp.counts_np[2])
        n1, n0 = smp.run(seeds_host, ctr)
        n0_pad = min(round_up(n0, N0_BUCKET_SMALL), smp.buf.n0_cap)
        sg = self._train(graph, smp.buf, bkey + (n0_pad,), smp.buf.n1_cap, n0_pad)
        sg.replay()
        sg.last_sizes = (n0, n1)
        return sg

    def _agnostic_graph(self, graph, smp, bkey, key, merged=False):
        """The train graph captured on the sampler's UPPER-BOUND block, if its launches take the block's size from the device (the first
        use captures it and finds out: ``TrainStepGraph.size_agnostic``); None when they do not — the caller then reads the counts back
        and replays the graph of their size bucket, as before."""
        known = self.agnostic.get(bkey)
        samp = smp if merged else None
        if known is None or key not in self.graphs:
            sg = self._train(graph, smp.buf, key, smp.buf.n1_cap, smp.buf.n0_cap, sampler=samp)
            if not sg.size_agnostic:
                self.agnostic[bkey] = False
                self.graphs.pop(key, None)                   # (it ran the general launches on 28 x the rows: never again)
                return None
            self.agnostic[bkey] = True
            return sg
        return self._train(graph, smp.buf, key, smp.buf.n1_cap, smp.buf.n0_cap, sampler=samp)

    def staged_step(self, graph, seeds, blocks, n0, n1, apply=True, defer_first=False, dp=None):
        """One step of a loader batch: stage its block arrays into the bucket's static buffers, replay its train graph.
        ``apply=False``: the graph stops after backward (the gradients are in ``sg.grads``).

        A capture costs ~5 ms — five steps' worth — so a bucket is not captured for one batch: the graph of the NEXT LARGER
        bucket (one step of n0 and / or n1: <= 3.5 % more padded rows, exact) serves the batch when it exists, and with
        ``defer_first`` a bucket met for the first time returns None (the caller runs that batch eagerly) and is captured
        when it comes back.  Common buckets are captured within the first snapshots; rare ones never stall the stream."""
        B = int(seeds.numel())
        n0_pad, n1_pad = round_up(n0, N0_BUCKET), round_up(n1, N1_BUCKET)
        # (dp = (GradSynchronizer, weight): the weight n_local / n_global is a constant of the recorded bucket scaling, and whether
        # the exchange runs in one bucket or two is frozen too: both are part of the key)
        bkey = ("staged", id(graph), B, n0_pad, n1_pad, bool(apply) if dp is None else ("dp", round(float(dp[1]), 9), dp[0].learnt))
        if bkey not in self.graphs:
            near = [k for k in self.graphs
                    if k[0] == "staged" and k[1:3] == bkey[1:3] and k[5] == bkey[5]
                    and n0_pad <= k[3] <= n0_pad + N0_BUCKET and n1_pad <= k[4] <= n1_pad + N1_BUCKET]
            if near:
                bkey = min(near, key=lambda k: (k[3] + 8 * k[4], k))
                self.borrowed += 1
            elif defer_first and self.sightings.get(bkey, 0) == 0:
                self.sightings[bkey] = 1
                if len(self.sightings) > 4096:
                    self.sightings.clear()
                self.deferred += 1
                return None
        n0_pad, n1_pad = bkey[3], bkey[4]
        buf = self.bufs.get(bkey)
        if buf is None:
            buf = self.bufs[bkey] = BlockBuffers(B, self.S, n1_pad, n0_pad, graph.device)
        sg = self._train(graph, buf, bkey, n1_pad, n0_pad, apply=apply, dp=dp)
        b0, b1 = blocks
        ops.stage_segments([(b0.src_ids, buf.src0, n0), (b1.src_ids, buf.src1, n1), (b0.local_idx, buf.lidx0, n1 * self.S),
                            (b1.local_idx, buf.lidx1, B * self.S), (seeds, buf.seeds, B)])
        sg.replay()
        return sg
