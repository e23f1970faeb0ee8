"""Captured train steps: one hipGraph launch per replay batch instead of ~40 launches enqueued from Python.

The update path of a batch is a fixed sequence of microsecond-to-sub-millisecond kernels (sample x2, block build x2,
forward, cross entropy, backward, Adam).  Enqueued one by one through ctypes + autograd it costs the host 0.8-0.9 ms per
step — the small rungs (pubmed / arxiv, 32 seeds) are 100 % launch- and sync-bound that way, and the Reddit rung is within
30 % of it.  ``TrainStepGraph`` records that sequence ONCE (``torch.cuda.graph``: a HIP stream capture of exactly the
C-ABI launches the eager path makes, autograd backward and the optimiser step included) on static buffers of padded
shapes and replays it.  What varies between replays lives in device memory: the seeds / block arrays (static input
buffers), the Philox batch counter (``ogl_sample_layer_dev``), Adam's step count (``ogl_adam_step_multi_dev``).

Two forms:

* ``sampled`` (small rungs): sampling and block construction are INSIDE the graph, shapes are the upper bounds
  ``n1 = B (1 + S)``, ``n0 = n1 (1 + S)`` — no block-size read-back, no host synchronisation at all; a step is one
  host->device copy of ``[counter | seeds]`` and one graph launch.
* ``staged`` (Reddit rung): the loader samples the snapshot's batches as before (one read-back per layer per LOADER,
  amortised over 50 batches); per batch, ONE ``ogl_stage_segments`` launch copies the batch's block arrays into the static
  buffers of the graph captured for its size bucket (``n1`` rounded up to 256, ``n0`` to 2 048 rows: <= 3 % padded rows).

Padding is exact, not approximate (include/ogl_hip.h, "Padding contract"): id -1 samples nothing / gathers the zero row,
so padded rows add exact zeros to every product and gradient; only the fp32 summation ORDER of split reductions can
differ from the unpadded eager step.

Not captured: steps under torch.distributed (the gradient all-reduce stays eager), dropout > 0 (its counter is host-side).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, sampling
from .graphsage.sageconv import GatheredRows


def round_up(x, m):
    return (int(x) + m - 1) // m * m


N1_BUCKET = 256
N0_BUCKET = 2048
_WARMED = False


class TrainStepGraph:
    """One captured train step.  ``loss_fn(logits, labels) -> (scalar loss to differentiate, per-seed losses or None)``."""

    def __init__(self, model, optimizer, graph, B, S, loss_fn, form, n1_pad=None, n0_pad=None, pool=None):
        assert form in ("sampled", "staged")
        self.model, self.opt, self.graph, self.B, self.S, self.form = model, optimizer, graph, int(B), int(S), form
        self.loss_fn = loss_fn
        dev = graph.device
        self.n1_pad = self.B * (1 + self.S) if form == "sampled" else int(n1_pad)
        self.n0_pad = self.n1_pad * (1 + self.S) if form == "sampled" else int(n0_pad)
        # static inputs
        self.head = torch.zeros(1 + self.B, dtype=torch.int64, device=dev)          # [Philox batch counter | seeds]
        self.head_host = torch.zeros(1 + self.B, dtype=torch.int64).pin_memory()
        if form == "staged":
            self.src0 = torch.full((self.n0_pad,), -1, dtype=torch.int64, device=dev)
            self.src1 = torch.full((self.n1_pad,), -1, dtype=torch.int64, device=dev)
            self.lidx0 = torch.full((self.n1_pad, self.S), -1, dtype=torch.int32, device=dev)
            self.lidx1 = torch.full((self.B, self.S), -1, dtype=torch.int32, device=dev)
        self.loss = None
        self.loss_rows = None
        self.cuda_graph = torch.cuda.CUDAGraph()
        self._capture(pool)

    # the step, written once: runs eagerly for nothing, only ever under capture
    def _body(self, apply=True):
        g, S = self.graph, self.S
        seeds = self.head[1:]
        if self.form == "sampled":
            ctr = self.head[:1]
            seed = sampling.get_state()["seed"]
            picks1 = ops.sample_layer_dev(g.handle, seeds, S, seed, ctr, 1)
            src1, _, lidx1 = ops.build_block_async(seeds, picks1, pad_tail=True)           # [B (1 + S)], -1 past n1
            picks0 = ops.sample_layer_dev(g.handle, src1, S, seed, ctr, 0)
            src0, _, lidx0 = ops.build_block_async(src1, picks0, pad_tail=True)            # [n1_pad (1 + S)], -1 past n0
        else:
            src0, src1, lidx0, lidx1 = self.src0, self.src1, self.lidx0, self.lidx1
        blocks = [sampling.Block(src0, src1, lidx0), sampling.Block(src1, seeds, lidx1)]
        # the FULL tables: a snapshot view's row count would be frozen into the graph (ids are < n_present by construction)
        labels = ops.gather_i64(g.target_table, seeds)
        self.opt.zero_grad(set_to_none=True)
        logits = self.model(blocks, GatheredRows(g.feat_table, src0))
        loss, rows = self.loss_fn(logits, labels)
        ops.backward(loss)
        if apply:
            self.opt.step()
        self.loss, self.loss_rows = loss.detach(), (rows.detach() if rows is not None else None)

    def _capture(self, pool):
        global _WARMED
        assert getattr(self.opt, "capturable", False), "a captured step needs optim.Adam(capturable=True)"
        self.opt.prepare_capture()
        ops.unit_grad(self.graph.device)
        ops._static_image(self.graph.feat_table)            # built outside the capture (a one-off 850 MB split pass)
        if not _WARMED:
            # once per process: run the step's forward + backward for real on a side stream (autograd's device thread, lazily
            # created helpers), WITHOUT the optimiser step and with the gradients dropped — the weights do not move
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._body(apply=False)
            torch.cuda.current_stream().wait_stream(side)
            self.opt.zero_grad(set_to_none=True)
            _WARMED = True
        torch.cuda.synchronize()
        kw = {} if pool is None else dict(pool=pool)
        with torch.cuda.graph(self.cuda_graph, **kw):
            self._body()
        self.grads = [p.grad for p in self.model.parameters()]

    # ---- replay --------------------------------------------------------------------------------------------------
    def run_sampled(self, seeds_host, ctr):
        """seeds_host: int64 array-like [B] (snapshot ids); ctr: this batch's Philox counter."""
        h = self.head_host
        h[0] = int(ctr)
        h[1:] = torch.as_tensor(np.asarray(seeds_host), dtype=torch.int64)
        self.head.copy_(h, non_blocking=True)
        self.cuda_graph.replay()
        return self.loss

    def run_staged(self, seeds, blocks, n0, n1):
        """seeds [B] (device), blocks = [input block, output block] of the loader; n0 / n1 their source counts."""
        assert n0 <= self.n0_pad and n1 <= self.n1_pad
        b0, b1 = blocks
        ops.stage_segments([(b0.src_ids, self.src0, n0), (b1.src_ids, self.src1, n1),
                            (b0.local_idx, self.lidx0, n1 * self.S), (b1.local_idx, self.lidx1, self.B * self.S),
                            (seeds, self.head[1:], self.B)])
        self.cuda_graph.replay()
        return self.loss


class StepGraphCache:
    """The captured steps of one (model, optimiser): keyed by form and padded sizes, captured on first use."""

    def __init__(self, model, optimizer, S, loss_fn):
        self.model, self.opt, self.S, self.loss_fn = model, optimizer, int(S), loss_fn
        self.graphs = {}
        self.captures = 0

    def sampled(self, graph, B):
        key = ("sampled", id(graph), int(B), sampling.get_state()["seed"])
        sg = self.graphs.get(key)
        if sg is None:
            sg = self.graphs[key] = TrainStepGraph(self.model, self.opt, graph, B, self.S, self.loss_fn, "sampled")
            self.captures += 1
        return sg

    def staged(self, graph, B, n0, n1):
        n0_pad, n1_pad = round_up(n0, N0_BUCKET), round_up(n1, N1_BUCKET)
        key = ("staged", id(graph), int(B), n0_pad, n1_pad)
        sg = self.graphs.get(key)
        if sg is None:
            sg = self.graphs[key] = TrainStepGraph(self.model, self.opt, graph, B, self.S, self.loss_fn, "staged", n1_pad, n0_pad)
            self.captures += 1
        return sg
