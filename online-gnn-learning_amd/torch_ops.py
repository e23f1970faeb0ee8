"""``torch.ops.ogl.*``: the C-ABI entry points (include/ogl_hip.h) registered as PyTorch custom ops.

The north star's layering is "Python host code calls PyTorch-ROCm custom ops over a thin C-ABI into hand-written HIP":
``torch.library`` schemas in the ``ogl`` namespace whose CUDA (= ROCm) kernels are the ctypes calls of ``ops.py``.  They
replace, op for op, what the reference's layer reaches through DGL / ATen at
R/train/graphsage/pytorch/graphsage_dgl.py:55-59 (``layer(block, h)`` -> sampling, ``to_block``, ``feat[input_nodes]``,
``copy_src``/``max``, ``nn.Linear``, ``CrossEntropyLoss``, ``Adam``).  Differentiable ops (``linear``, ``pool_max``,
``neighbor_reduce``, ``cross_entropy``, ``dropout``) are registered as composite ops over the package's
``autograd.Function`` nodes, so ``torch.ops.ogl.linear(...)`` back-propagates through the HIP backward kernels.  There
is no CPU kernel: a CPU tensor raises (no fallback).

The graph handle of the sampler is not a tensor; ``ogl::sample_layer`` takes the three CSR tensors + the snapshot
degrees' owner through a small registry id (``register_graph``).
"""
from __future__ import annotations

import torch

from . import ops

_LIB = None
_GRAPHS = {}


def register_graph(handle: "ops.GraphHandle") -> int:
    """Make a ``GraphHandle`` addressable from ``torch.ops.ogl.sample_layer`` (ops take tensors and scalars only)."""
    gid = id(handle)
    _GRAPHS[gid] = handle
    return gid


def unregister_graph(gid: int):
    _GRAPHS.pop(gid, None)


def _define():
    global _LIB
    if _LIB is not None:
        return _LIB
    lib = torch.library.Library("ogl", "DEF")
    D = lib.define
    D("sample_layer(int graph, Tensor dst, int fanout, int seed, int ctr, int layer) -> Tensor")
    D("build_block(Tensor dst, Tensor picks) -> (Tensor, Tensor, Tensor)")
    D("gather_rows(Tensor table, Tensor ids) -> Tensor")
    D("gather_i64(Tensor table, Tensor ids) -> Tensor")
    D("reduce_fwd(Tensor src, Tensor idx, str op, bool want_argmax=False) -> (Tensor, Tensor?)")
    D("reduce_bwd(Tensor dout, Tensor? idx32, Tensor? argmax, str op, int n_src, int fanout, Tensor? relu_out=None) -> Tensor")
    D("linear_fwd(Tensor x, Tensor w, Tensor? bias=None, Tensor? x2=None, Tensor? w2=None, bool relu=False, "
      "Tensor? x_rows=None, Tensor? x2_rows=None) -> Tensor")
    D("ce_fwd_bwd(Tensor logits, Tensor labels, float grad_scale=1.0, bool want_grad=True) -> (Tensor, Tensor?)")
    D("argmax_confusion(Tensor logits, Tensor? labels, Tensor(a!)? confusion) -> Tensor")
    D("adam_step(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, int step, float lr=1e-3, float beta1=0.9, "
      "float beta2=0.999, float eps=1e-8) -> ()")
    D("dropout_rows(Tensor x, float p, int seed, int ctr, Tensor? rows=None) -> Tensor")
    # differentiable (composite over the autograd.Function nodes of ops.py)
    D("linear(Tensor x, Tensor w, Tensor? bias=None, Tensor? x2=None, Tensor? w2=None, bool relu=False, "
      "Tensor? x_rows=None, Tensor? x2_rows=None, Tensor? bias2=None) -> Tensor")
    D("neighbor_reduce(Tensor src, Tensor idx, str op) -> Tensor")
    D("pool_max(Tensor x, Tensor w, Tensor? bias, Tensor idx, Tensor? x_rows=None) -> Tensor")
    D("cross_entropy(Tensor logits, Tensor labels, str reduction='mean') -> Tensor")
    D("dropout(Tensor x, float p, Tensor? rows=None) -> Tensor")

    def cuda(name, fn):
        lib.impl(name, fn, "CUDA")

    def composite(name, fn):
        lib.impl(name, fn, "CompositeImplicitAutograd")

    cuda("sample_layer", lambda graph, dst, fanout, seed, ctr, layer:
         ops.sample_layer(_GRAPHS[graph], dst, fanout, seed, ctr, layer))
    cuda("build_block", lambda dst, picks: _build_block(dst, picks))
    cuda("gather_rows", ops.gather_rows)
    cuda("gather_i64", ops.gather_i64)
    cuda("reduce_fwd", lambda src, idx, op, want_argmax=False: ops.reduce_fwd(src, idx, op, want_argmax))
    cuda("reduce_bwd", lambda dout, idx32, argmax, op, n_src, fanout, relu_out=None:
         ops.reduce_bwd(dout, idx32, argmax, op, n_src, fanout=fanout, relu_out=relu_out))
    cuda("linear_fwd", lambda x, w, bias=None, x2=None, w2=None, relu=False, x_rows=None, x2_rows=None:
         ops.linear_fwd(x, w, bias, x2, w2, relu, x_rows, x2_rows))
    cuda("ce_fwd_bwd", lambda logits, labels, grad_scale=1.0, want_grad=True: ops.ce_fwd_bwd(logits, labels, grad_scale, want_grad))
    cuda("argmax_confusion", lambda logits, labels, confusion: ops.argmax_confusion(logits, labels, confusion, want_pred=True))
    cuda("adam_step", lambda p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8:
         ops.adam_step(p, g, m, v, step, lr, beta1, beta2, eps))
    cuda("dropout_rows", lambda x, p, seed, ctr, rows=None: ops.dropout_rows(x, p, seed, ctr, rows))
    composite("linear", lambda x, w, bias=None, x2=None, w2=None, relu=False, x_rows=None, x2_rows=None, bias2=None:
              ops.linear(x, w, bias, x2, w2, relu, x_rows, x2_rows, bias2))
    composite("neighbor_reduce", ops.neighbor_reduce)
    composite("pool_max", lambda x, w, bias, idx, x_rows=None: ops.pool_max(x, w, bias, idx, x_rows))
    composite("cross_entropy", lambda logits, labels, reduction="mean": ops.cross_entropy(logits, labels, reduction))
    composite("dropout", lambda x, p, rows=None: ops.dropout(x, p, rows))
    _LIB = lib
    return lib


def _build_block(dst, picks):
    src_ids, n_src, local_idx = ops.build_block_async(dst, picks)
    return src_ids, n_src, local_idx


_define()
