"""Thin torch-side wrappers over the C-ABI (include/ogl_hip.h) + autograd glue.

torch is plumbing here: device memory, the current HIP stream and autograd bookkeeping.  Every
arithmetic step of the hot path is a call into libogl_hip.so; nothing in this file computes on
the CPU or through ATen kernels except trivial reshapes / scalar reads.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import REDUCE_OPS, check


def _stream():
    # torch's current HIP stream of the current device, through the raw getters: torch.cuda.current_stream() builds a
    # Stream object and re-checks lazy initialisation — ~10 us of host time per call, ~30 calls per train step
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())   # plain int: the argtypes convert it


# Optional per-call HIP-event timing (bench.py's roofline leg).  Events are recorded on the stream the
# kernels are launched on (torch's current stream), so elapsed_time brackets exactly that call's kernels.
_PROFILE = None


def profile_start():
    global _PROFILE
    _flush_deferred()
    _PROFILE = []


def profile_stop():
    """Returns [(name, meta, milliseconds)] for every C-ABI launch since profile_start()."""
    global _PROFILE
    rec, _PROFILE = _PROFILE or [], None
    torch.cuda.synchronize()
    return [(n, m, s.elapsed_time(e)) for n, m, s, e in rec]


# Side work to enqueue right AFTER the next launch (see pool_bwd_x3_plan: in a captured step the first-created child of a node keeps
# its parent's hardware queue, so the critical chain's next launch has to be created before a branch that forks off the same node).
_DEFERRED = []


def _flush_deferred():
    while _DEFERRED:
        _DEFERRED.pop(0)()


def _launch(name, fn, *args, meta=None):
    if _PROFILE is None:
        check(fn(*args), name)
        if _DEFERRED:
            _flush_deferred()
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    st = fn(*args)
    e.record()
    if meta is not None and "_x3" in name and name.startswith("ogl_linear"):
        meta = dict(meta, kernel=_lib.lib().ogl_x3_last_kernel().decode())     # which instantiation ran (bench.py: PMC staleness check)
    _PROFILE.append((name, meta, s, e))
    check(st, name)


def _ptr(t):
    return t.data_ptr() if t is not None else None      # plain int / None: the argtypes (c_void_p) convert them


GEMM_MODES = {"f32": 0, "bf16x6": 1, "auto": 2}


_MODE = {"name": "f32"}


def set_gemm_mode(mode: str):
    """'f32' = exact fp32 MFMA (default); 'bf16x6' = split-bf16 MFMA at fp32 accuracy; 'auto' = bf16x6 where it is
    faster (forward / input-gradient GEMMs), f32 for weight gradients (see include/ogl_hip.h)."""
    check(_lib.lib().ogl_set_gemm_mode(GEMM_MODES[mode]), "ogl_set_gemm_mode")
    _MODE["name"] = mode


def get_gemm_mode() -> str:
    m = _lib.lib().ogl_set_gemm_mode(-1)                  # (OGL_GEMM_QUERY)
    return [k for k, v in GEMM_MODES.items() if v == m][0]


KNOBS = {"x3_tile": 0, "x3_stagger": 1, "block_min_lds": 2, "reduce_half": 3, "seg_rows": 4}


def debug_set(knob: str, value: int) -> int:
    """Pin a kernel form (``ogl_debug_set``: diagnostics for tests and same-process A/B runs — every form returns the same bits);
    returns the previous setting."""
    prev = C.c_int(0)
    check(_lib.lib().ogl_debug_set(KNOBS[knob], int(value), C.byref(prev)), "ogl_debug_set(%s, %d)" % (knob, value))
    return prev.value


def padded_ld(cols: int) -> int:
    """Leading dimension used for matrices this package allocates (16-B rows; 128-B rows when wide)."""
    return (cols + 31) // 32 * 32 if cols > 64 else max(4, (cols + 3) // 4 * 4)


def empty_mat(rows: int, cols: int, device, zero: bool = False):
    """[rows, cols] float32 view of a [rows, padded_ld(cols)] allocation (pad columns are don't-care)."""
    ld = padded_ld(cols)
    buf = torch.empty((max(rows, 0), ld), dtype=torch.float32, device=device)
    if zero:
        fill_zero(buf)
    return buf[:, :cols]


def fill_zero(t):
    """Zeroes a contiguous tensor in place (ogl_fill_zero on the device: the step's launches are this library's; torch on the host)."""
    if not t.is_cuda:
        return t.zero_()
    assert t.is_contiguous()
    if t.numel():
        _launch("ogl_fill_zero", _lib.lib().ogl_fill_zero, _ptr(t), t.numel() * t.element_size(), _stream(), meta=dict(bytes=t.numel() * t.element_size()))
    return t


def as_mat(t: torch.Tensor) -> torch.Tensor:
    """Return a float32 2-D CUDA tensor whose rows are contiguous (stride(1)==1, stride(0)>=cols)."""
    if t.dim() != 2:
        raise ValueError("expected a 2-D tensor, got shape %s" % (tuple(t.shape),))
    if not t.is_cuda:
        raise RuntimeError("ogl_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    if t.shape[0] > 1 and (t.stride(1) != 1 or t.stride(0) < t.shape[1]):
        t = t.contiguous()
    elif t.shape[0] <= 1 and t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0) if t.stride(0) >= t.shape[1] else t.shape[1])


def _ids(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous():
        raise ValueError("ids must be a contiguous CUDA int64 tensor")
    return t


# --------------------------------------------------------------------------------------------
# graph + sampler + block
# --------------------------------------------------------------------------------------------
class GraphHandle:
    """Owns an ogl_graph_t over device CSR arrays (kept alive here)."""

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, keys: torch.Tensor | None = None):
        assert indptr.is_cuda and indptr.dtype == torch.int64 and indptr.is_contiguous()
        assert indices.is_cuda and indices.dtype == torch.int32 and indices.is_contiguous()
        if keys is not None:
            assert keys.is_cuda and keys.dtype == torch.int32 and keys.is_contiguous() and keys.numel() == indices.numel()
        self.indptr, self.indices, self.keys = indptr, indices, keys
        self.n = indptr.numel() - 1
        self.nnz = indices.numel()
        h = C.c_void_p()
        check(_lib.lib().ogl_graph_create(_ptr(indptr), _ptr(indices), _ptr(keys), self.n, self.nnz, C.byref(h)),
              "ogl_graph_create")
        self._h = h
        self.n_present, self.cut = 0, 0

    def set_snapshot(self, n_present: int, cut: int):
        check(_lib.lib().ogl_graph_set_snapshot(self._h, int(n_present), int(cut), _stream()), "ogl_graph_set_snapshot")
        self.n_present, self.cut = int(n_present), int(cut)

    def degrees(self) -> torch.Tensor:
        """Copy of the current snapshot in-degrees, int32[n] (device)."""
        out = torch.empty(self.n, dtype=torch.int32, device=self.indptr.device)
        check(_lib.lib().ogl_graph_copy_degrees(self._h, _ptr(out), _stream()), "ogl_graph_copy_degrees")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().ogl_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass


def sample_layer(g: GraphHandle, dst: torch.Tensor, fanout: int, seed: int, ctr: int, layer: int) -> torch.Tensor:
    dst = _ids(dst)
    picks = torch.empty((dst.numel(), fanout), dtype=torch.int64, device=dst.device)
    _launch("ogl_sample_layer", _lib.lib().ogl_sample_layer, g._h, _ptr(dst), dst.numel(), int(fanout), C.c_uint64(seed & (2 ** 64 - 1)),
                                      C.c_uint64(ctr & (2 ** 64 - 1)), int(layer), _ptr(picks), _stream(), meta=dict(n_dst=dst.numel(), fanout=int(fanout)))
    return picks


def sample_layer_dev(g: GraphHandle, dst: torch.Tensor, fanout: int, seed: int, ctr_dev: torch.Tensor, layer: int) -> torch.Tensor:
    """sample_layer with the Philox batch counter in device memory (``ctr_dev``: one int64 element): capturable in a hipGraph.
    A destination id of -1 (padding) gets no pick."""
    dst = _ids(dst)
    assert ctr_dev.is_cuda and ctr_dev.dtype == torch.int64 and ctr_dev.numel() >= 1
    picks = torch.empty((dst.numel(), fanout), dtype=torch.int64, device=dst.device)
    _launch("ogl_sample_layer_dev", _lib.lib().ogl_sample_layer_dev, g._h, _ptr(dst), dst.numel(), int(fanout),
            C.c_uint64(seed & (2 ** 64 - 1)), _ptr(ctr_dev), int(layer), _ptr(picks), _stream(), meta=dict(n_dst=dst.numel(), fanout=int(fanout)))
    return picks


def publish_i64(src_dev, seq_dev, dst_host_pinned):
    """Store src_dev[:] and then ++seq into pinned host memory from a kernel (dst_host_pinned: int64 [n + 1]); the host polls the
    last element.  The read-back of a captured sample graph (ogl_publish_i64)."""
    n = int(src_dev.numel())
    assert src_dev.dtype == torch.int64 and src_dev.is_cuda and seq_dev.dtype == torch.int64 and seq_dev.is_cuda
    assert dst_host_pinned.dtype == torch.int64 and dst_host_pinned.is_pinned() and dst_host_pinned.numel() >= n + 1
    _launch("ogl_publish_i64", _lib.lib().ogl_publish_i64, _ptr(src_dev), n, _ptr(seq_dev), dst_host_pinned.data_ptr(), _stream(),
            meta=dict(n=n))


SAMPLE_FUSED = True      # the sampling phase of a small batch as ONE launch


def sample_blocks_small_fits(B, fanout):
    return SAMPLE_FUSED and 0 < B <= 1023 and int(_lib.lib().ogl_sample_blocks_small_workspace_bytes(int(B), int(fanout))) > 0


def sample_blocks_small(g: GraphHandle, head_host_pinned, head_dev, B, fanout, seed, src1, lidx1, src0, lidx0, counts, seq_dev,
                        counts_host_pinned, ws=None, src0_fill=0):
    """The whole sampling phase of one small batch (``ogl_sample_blocks_small``): stage [counter | seeds] from pinned host memory,
    sample + relabel both blocks into the static arrays, publish (n1, n0, ++seq) to pinned host memory.  Returns the workspace
    (keep it alive with the captured graph)."""
    nbytes = int(_lib.lib().ogl_sample_blocks_small_workspace_bytes(int(B), int(fanout)))
    assert nbytes > 0 and head_host_pinned.is_pinned() and counts_host_pinned.is_pinned() and counts_host_pinned.numel() >= 3
    n1_cap = B * (1 + fanout)
    assert head_dev.numel() >= 1 + B and src1.numel() >= n1_cap and src0.numel() >= n1_cap * (1 + fanout)
    assert tuple(lidx1.shape) == (B, fanout) and tuple(lidx0.shape) == (n1_cap, fanout) and lidx0.is_contiguous() and lidx1.is_contiguous()
    if ws is None:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=head_dev.device)
    # (src0_fill: pad src0 with -1 up to round_up(n0, src0_fill) only — stepgraph reads it up to its train graph's size bucket)
    _launch("ogl_sample_blocks_small", _lib.lib().ogl_sample_blocks_small_fill, g._h, head_host_pinned.data_ptr(), _ptr(head_dev), int(B),
            int(fanout), C.c_uint64(seed & (2 ** 64 - 1)), _ptr(src1), _ptr(lidx1), _ptr(src0), _ptr(lidx0), _ptr(counts), _ptr(seq_dev),
            counts_host_pinned.data_ptr(), _ptr(ws), nbytes, int(src0_fill), _stream(), meta=dict(n_dst=int(B), fanout=int(fanout)))
    return ws


def stage_segments(pairs, pad=-1):
    """One launch: for every (src, dst, count) copy ``count`` leading elements of ``src`` into ``dst`` and fill the rest of
    ``dst`` with ``pad``.  int32 / int64 tensors (contiguous); the staging step of a captured train step."""
    k = len(pairs)
    srcs = (C.c_void_p * k)(*[t[0].data_ptr() if t[2] > 0 else None for t in pairs])
    dsts = (C.c_void_p * k)(*[t[1].data_ptr() for t in pairs])
    cnt = (C.c_int64 * k)(*[int(t[2]) for t in pairs])
    cap = (C.c_int64 * k)(*[t[1].numel() for t in pairs])
    el = (C.c_int * k)(*[t[1].element_size() for t in pairs])
    for src, dst, n in pairs:
        assert dst.is_contiguous() and dst.is_cuda and (n == 0 or (src.is_contiguous() and src.dtype == dst.dtype and src.numel() >= n))
    _launch("ogl_stage_segments", _lib.lib().ogl_stage_segments, k, srcs, dsts, cnt, cap, el, int(pad), _stream(), meta=dict(nseg=k))


def fuse_block_segments(local_idx, seg_rows, seg_offs, want_dst_pos=True, dst_flag=None):
    """In place: the packed block-local indices of several batches become indices into their source lists laid end to end
    (``ogl_fuse_block_segments``); returns the position of every destination's own row in the fused source list (int64 [rows])."""
    assert local_idx.dtype == torch.int32 and local_idx.is_cuda and local_idx.is_contiguous() and local_idx.dim() == 2
    k = len(seg_offs)
    assert len(seg_rows) == k + 1 and seg_rows[0] == 0 and seg_rows[-1] == local_idx.shape[0]
    dst_pos = torch.empty(local_idx.shape[0], dtype=torch.int64, device=local_idx.device) if want_dst_pos else None
    if dst_flag is not None:            # (ZEROED uint8 [rows of the fused source list]: set to 1 at every destination's own row)
        assert dst_flag.dtype == torch.uint8 and dst_flag.is_cuda and dst_flag.is_contiguous()
    _launch("ogl_fuse_block_segments", _lib.lib().ogl_fuse_block_segments, _ptr(local_idx), _ptr(dst_pos), k, _host_i64(seg_rows),
            _host_i64(seg_offs), int(local_idx.shape[1]), _ptr(dst_flag), _stream(), meta=dict(rows=int(local_idx.shape[0]), nseg=k))
    return dst_pos


# pad_tail builds go through ogl_build_block_padded, which writes the -1 tail itself (one workgroup up to 4 096 flat positions,
# the parallel phases above); tests raise / lower this to force either path
BLOCK_SMALL_MAX_P = 1 << 30


def build_block_async(dst: torch.Tensor, picks: torch.Tensor, pad_tail: bool = False, out=None):
    """Enqueue the relabelling; returns (src_ids[cap], n_src_dev[1], local_idx) without synchronising.

    ``pad_tail``: the entries of src_ids past the (device-side) source count read -1 = "no vertex" (a captured step uses the
    whole capacity as its padded source list; the eager path slices by the read-back count and never looks at them).
    ``out = (src_ids, n_src, local_idx)``: caller-owned outputs (the static buffers of a captured step)."""
    dst = _ids(dst)
    n_dst, fanout = picks.shape
    assert n_dst == dst.numel() and picks.dtype == torch.int64 and picks.is_contiguous()
    dev = dst.device
    cap = n_dst * (1 + fanout)
    small = pad_tail and 0 < cap <= BLOCK_SMALL_MAX_P and fanout > 0      # ogl_build_block_padded writes the -1 tail itself
    if out is not None:
        src_ids, n_src, local_idx = out
        assert src_ids.numel() >= cap and src_ids.dtype == torch.int64 and src_ids.is_contiguous()
        assert n_src.dtype == torch.int64 and n_src.numel() >= 1
        assert local_idx.dtype == torch.int32 and local_idx.is_contiguous() and local_idx.numel() >= n_dst * fanout
        if pad_tail and not small:
            src_ids.fill_(-1)
    else:
        if pad_tail and not small:
            src_ids = torch.full((max(cap, 1),), -1, dtype=torch.int64, device=dev)
        else:
            src_ids = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
        n_src = torch.zeros(1, dtype=torch.int64, device=dev)
        local_idx = torch.empty((n_dst, fanout), dtype=torch.int32, device=dev)
    nbytes = int(_lib.lib().ogl_block_workspace_bytes(n_dst, fanout))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    if pad_tail and 0 < cap <= BLOCK_SMALL_MAX_P and fanout > 0:
        # one launch: relabel + the -1 tail (the fill above is then redundant but harmless when out was given)
        _launch("ogl_build_block_padded", _lib.lib().ogl_build_block_padded, _ptr(dst), n_dst, _ptr(picks), int(fanout), _ptr(src_ids),
                src_ids.numel(), _ptr(n_src), _ptr(local_idx), _ptr(ws), nbytes, _stream(), meta=dict(n_dst=n_dst, fanout=int(fanout)))
        return src_ids, n_src, local_idx
    _launch("ogl_build_block", _lib.lib().ogl_build_block, _ptr(dst), n_dst, _ptr(picks), int(fanout), _ptr(src_ids), _ptr(n_src),
                                     _ptr(local_idx), _ptr(ws), nbytes, _stream(), meta=dict(n_dst=n_dst, fanout=int(fanout)))
    return src_ids, n_src, local_idx


def _host_i64(vals):
    arr = (C.c_int64 * max(len(vals), 1))(*[int(v) for v in vals])
    return arr


def sample_layer_batched(g: GraphHandle, dst_base: torch.Tensor, starts, counts, fanout: int, seed: int, ctrs, layer: int):
    """Every batch of a loader layer in one launch: batch b samples for dst_base[starts[b] : starts[b] + counts[b]] with
    Philox counter ctrs[b]; returns the packed picks [sum(counts), fanout] (bit-identical to sample_layer per batch)."""
    dst_base = _ids(dst_base)
    nb, total = len(counts), int(sum(counts))
    picks = torch.empty((total, fanout), dtype=torch.int64, device=dst_base.device)
    c_ctr = (C.c_uint64 * max(nb, 1))(*[int(c) & (2 ** 64 - 1) for c in ctrs])
    _launch("ogl_sample_layer_batched", _lib.lib().ogl_sample_layer_batched, g._h, _ptr(dst_base), _host_i64(starts), _host_i64(counts),
            nb, int(fanout), C.c_uint64(seed & (2 ** 64 - 1)), c_ctr, int(layer), _ptr(picks), _stream(),
            meta=dict(n_dst=total, fanout=int(fanout), nb=nb))
    return picks


# the batched block build on a direct-address table (one int32 pair per vertex id and batch) instead of the hash when the caller passes
# the id bound and the tables of a 64-batch chunk stay below this many bytes (OGL_BLOCK_DIRECT=0: always the hash)
BLOCK_DIRECT = os.environ.get("OGL_BLOCK_DIRECT", "1") != "0"
BLOCK_DIRECT_MAX_BYTES = 2 << 30
# ... and only when the table is not much larger than what the batches put into it: the direct build touches 8 * n_ids bytes per batch
# whatever the batch holds, the hash O(positions) — a 32-seed output block (832 positions) on a 170 k-vertex graph stays on the hash, a
# 512-seed one (9.2 M positions over 50 batches of a 233 k-vertex graph) takes the table (measured there: 1.23 -> 0.30 ms)
BLOCK_DIRECT_IDS_PER_POSITION = 16


def build_block_batched_async(dst_base: torch.Tensor, starts, counts, picks: torch.Tensor, n_ids=None):
    """Relabel every batch in one set of launches.  Returns (src_ids packed at row_off * (1 + fanout), n_src_dev [nb],
    local_idx packed like picks) without synchronising; row_off = running sum of counts.  ``n_ids``: every id is < n_ids (the
    graph's vertex count) — the build then uses a direct-address table (ogl_build_block_batched with n_ids > 0: same results)."""
    dst_base = _ids(dst_base)
    nb, total = len(counts), int(sum(counts))
    fanout = picks.shape[1]
    assert picks.shape[0] == total and picks.dtype == torch.int64 and picks.is_contiguous()
    dev = dst_base.device
    src_ids = torch.empty(max(total * (1 + fanout), 1), dtype=torch.int64, device=dev)
    # (every batch's entry is written by the build — its scan launch, or the memset of an all-empty chunk: no fill launch here)
    n_src = torch.empty(nb, dtype=torch.int64, device=dev) if nb > 0 else torch.zeros(1, dtype=torch.int64, device=dev)
    local_idx = torch.empty((total, fanout), dtype=torch.int32, device=dev)
    h_counts = _host_i64(counts)
    if (BLOCK_DIRECT and n_ids and 0 < int(n_ids) < 2 ** 31 and 8 * int(n_ids) * min(nb, 64) <= BLOCK_DIRECT_MAX_BYTES
            and int(n_ids) * nb <= BLOCK_DIRECT_IDS_PER_POSITION * total * (1 + fanout)):
        nbytes = int(_lib.lib().ogl_block_workspace_bytes_batched(h_counts, nb, int(fanout), int(n_ids)))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _launch("ogl_build_block_batched", _lib.lib().ogl_build_block_batched, _ptr(dst_base), _host_i64(starts), h_counts, nb, _ptr(picks),
                int(fanout), int(n_ids), _ptr(src_ids), _ptr(n_src), _ptr(local_idx), _ptr(ws), nbytes, _stream(),
                meta=dict(n_dst=total, fanout=int(fanout), nb=nb, direct=1))
        return src_ids, n_src, local_idx
    nbytes = int(_lib.lib().ogl_block_workspace_bytes_batched(h_counts, nb, int(fanout), 0))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    _launch("ogl_build_block_batched", _lib.lib().ogl_build_block_batched, _ptr(dst_base), _host_i64(starts), h_counts, nb, _ptr(picks),
            int(fanout), 0, _ptr(src_ids), _ptr(n_src), _ptr(local_idx), _ptr(ws), nbytes, _stream(),
            meta=dict(n_dst=total, fanout=int(fanout), nb=nb))
    return src_ids, n_src, local_idx


def build_block(dst: torch.Tensor, picks: torch.Tensor):
    src_ids, n_src, local_idx = build_block_async(dst, picks)
    n = int(n_src.item())
    return src_ids[:n], local_idx, n


# --------------------------------------------------------------------------------------------
# gather / reduce
# --------------------------------------------------------------------------------------------
def gather_rows(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    table = as_mat(table)
    ids = _ids(ids)
    d = table.shape[1]
    out = empty_mat(ids.numel(), d, table.device)
    _launch("ogl_gather_rows", _lib.lib().ogl_gather_rows, _ptr(table), _ld(table), table.shape[0], _ptr(ids), ids.numel(), d,
                                     _ptr(out), _ld(out), _stream(), meta=dict(n=ids.numel(), d=d))
    return out


def gather_i64(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    flat = table.reshape(-1)
    assert flat.dtype == torch.int64 and flat.is_cuda and flat.is_contiguous()
    ids = _ids(ids)
    out = torch.empty(ids.numel(), dtype=torch.int64, device=ids.device)
    check(_lib.lib().ogl_gather_i64(_ptr(flat), flat.numel(), _ptr(ids), ids.numel(), _ptr(out), _stream()),
          "ogl_gather_i64")
    return out


# feat_drop: counter-based (Philox) dropout stream, the analogue of the sampler's: reproducible, and the mask of an
# application is a function of (seed, application counter, row, column), so backward re-derives it instead of storing it
_DROPOUT = {"seed": 1, "ctr": 0}


def dropout_seed(value: int):
    """Reset the dropout stream (the reference never seeds torch's RNG, R/train/__main__.py:211-212)."""
    _DROPOUT["seed"], _DROPOUT["ctr"] = int(value), 0


def dropout_rows(x, p, seed, ctr, rows=None):
    """x[rows?] with Bernoulli(1 - p) dropout scaled by 1 / (1 - p) (ogl_dropout_rows)."""
    x = as_mat(x)
    M = rows.numel() if rows is not None else x.shape[0]
    N = x.shape[1]
    out = empty_mat(M, N, x.device)
    _launch("ogl_dropout_rows", _lib.lib().ogl_dropout_rows, _ptr(x), _ld(x), _ptr(_ids(rows) if rows is not None else None),
            x.shape[0], M, N, C.c_double(float(p)), C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(ctr & (2 ** 64 - 1)), _ptr(out),
            _ld(out), _stream(), meta=dict(M=M, N=N))
    return out


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rows, p, seed, ctr):
        ctx.p, ctx.seed, ctx.ctr, ctx.gathered = p, seed, ctr, rows is not None
        return dropout_rows(x, p, seed, ctr, rows)

    @staticmethod
    def backward(ctx, dy):
        if ctx.gathered:
            raise RuntimeError("gradient w.r.t. a row-gathered table is not supported (features carry no grad)")
        return dropout_rows(dy, ctx.p, ctx.seed, ctx.ctr), None, None, None, None


def dropout(x, p, rows=None):
    """nn.Dropout(p)(x[rows?]) in training mode on the HIP kernel, differentiable w.r.t. an ungathered x; takes the next
    counter of the dropout stream."""
    ctr = _DROPOUT["ctr"]
    _DROPOUT["ctr"] = ctr + 1
    if rows is not None or not x.requires_grad:
        return dropout_rows(x, p, _DROPOUT["seed"], ctr, rows)
    return _DropoutFn.apply(x, None, p, _DROPOUT["seed"], ctr)


def reduce_fwd(src: torch.Tensor, idx: torch.Tensor, op: str, want_argmax: bool = False):
    """out[d] = op_j src[idx[d, j]].  idx int32 (block-local) or int64 (global rows)."""
    src = as_mat(src)
    assert idx.is_cuda and idx.dim() == 2 and idx.is_contiguous()
    n_dst, fanout = idx.shape
    d = src.shape[1]
    out = empty_mat(n_dst, d, src.device)
    argmax = torch.empty((n_dst, d), dtype=torch.int32, device=src.device) if (want_argmax and op == "max") else None
    i32 = idx if idx.dtype == torch.int32 else None
    i64 = idx if idx.dtype == torch.int64 else None
    if i32 is None and i64 is None:
        raise ValueError("idx must be int32 or int64")
    _launch("ogl_reduce_fwd", _lib.lib().ogl_reduce_fwd, _ptr(src), _ld(src), src.shape[0], _ptr(i32), _ptr(i64), n_dst, fanout, d,
                                    REDUCE_OPS[op], _ptr(out), _ld(out), _ptr(argmax), _stream(), meta=dict(n_dst=n_dst, fanout=fanout, d=d, op=op, argmax=argmax is not None, idx_bytes=idx.element_size()))
    return out, argmax


def reduce_fwd_img(src: torch.Tensor, idx: torch.Tensor, want_argmax: bool = False, want_out: bool = True):
    """max-reduce that also returns the bf16x3 image of its output: (out, argmax, X3Image).  ``want_out=False``: the image only
    (``out`` is None: a consumer that reads nothing but the image — the cached inference layers — saves 40 % of the launch's writes)."""
    src = as_mat(src)
    assert idx.is_cuda and idx.dim() == 2 and idx.is_contiguous()
    n_dst, fanout = idx.shape
    d = src.shape[1]
    out = empty_mat(n_dst, d, src.device) if want_out else None
    argmax = torch.empty((n_dst, d), dtype=torch.int32, device=src.device) if want_argmax else None
    img = X3Image(_x3_alloc(n_dst, d, src.device), n_dst, d)
    i32 = idx if idx.dtype == torch.int32 else None
    i64 = idx if idx.dtype == torch.int64 else None
    _launch("ogl_reduce_fwd_img", _lib.lib().ogl_reduce_fwd_img, _ptr(src), _ld(src), src.shape[0], _ptr(i32), _ptr(i64), n_dst, fanout, d,
            _ptr(out), _ld(out) if out is not None else padded_ld(d), _ptr(argmax), _ptr(img.buf), _stream(),
            meta=dict(n_dst=n_dst, fanout=fanout, d=d, op="max", argmax=argmax is not None, idx_bytes=idx.element_size(), out=out is not None))
    return out, argmax, img


def reduce_fwd_mean_img(src: torch.Tensor, idx: torch.Tensor):
    """mean-reduce that also returns the bf16x3 image of its output: (out, X3Image)."""
    src = as_mat(src)
    assert idx.is_cuda and idx.dim() == 2 and idx.is_contiguous()
    n_dst, fanout = idx.shape
    d = src.shape[1]
    out = empty_mat(n_dst, d, src.device)
    img = X3Image(_x3_alloc(n_dst, d, src.device), n_dst, d)
    i32 = idx if idx.dtype == torch.int32 else None
    i64 = idx if idx.dtype == torch.int64 else None
    _launch("ogl_reduce_fwd_img", _lib.lib().ogl_reduce_fwd_mean_img, _ptr(src), _ld(src), src.shape[0], _ptr(i32), _ptr(i64), n_dst, fanout,
            d, _ptr(out), _ld(out), _ptr(img.buf), _stream(),
            meta=dict(n_dst=n_dst, fanout=fanout, d=d, op="mean", argmax=False, idx_bytes=idx.element_size(), out=True))
    return out, img


def reduce_fwd_rows_mean_img(table: torch.Tensor, rows: torch.Tensor, idx: torch.Tensor):
    """(mean_j table[rows[idx[d, j]]], its X3Image): the mean over rows of a resident table through a block's local indices — the
    gathered copy ``table[rows]`` is never made."""
    table = as_mat(table)
    assert idx.is_cuda and idx.dim() == 2 and idx.is_contiguous() and idx.dtype == torch.int32 and table.shape[0] < (1 << 31)
    rows = _ids(rows)
    n_dst, fanout = idx.shape
    d = table.shape[1]
    out = empty_mat(n_dst, d, table.device)
    img = X3Image(_x3_alloc(n_dst, d, table.device), n_dst, d)
    _launch("ogl_reduce_fwd_img", _lib.lib().ogl_reduce_fwd_rows_mean_img, _ptr(table), _ld(table), table.shape[0], _ptr(idx), _ptr(rows),
            rows.numel(), n_dst, fanout, d, _ptr(out), _ld(out), _ptr(img.buf), _stream(),
            meta=dict(n_dst=n_dst, fanout=fanout, d=d, op="mean", argmax=False, idx_bytes=4, out=True, table_rows=True))
    return out, img


def reduce_bwd(dout: torch.Tensor, idx32, argmax, op: str, n_src: int, fanout=None, relu_out=None, dsrc=None) -> torch.Tensor:
    """``dsrc``: an already ZEROED [n_src, d] matrix to scatter into (``take_zeroed``); default: allocated and cleared here."""
    dout = as_mat(dout)
    n_dst, d = dout.shape
    fanout = idx32.shape[1] if idx32 is not None else int(fanout)
    if dsrc is None:
        dsrc = empty_mat(n_src, d, dout.device, zero=True)
    assert tuple(dsrc.shape) == (n_src, d)
    if relu_out is not None:
        relu_out = as_mat(relu_out)
    _launch("ogl_reduce_bwd", _lib.lib().ogl_reduce_bwd, _ptr(dout), _ld(dout), _ptr(idx32), _ptr(argmax), _ptr(relu_out),
            _ld(relu_out) if relu_out is not None else 0, n_dst, fanout, d,
                                    REDUCE_OPS[op], n_src, _ptr(dsrc), _ld(dsrc), _stream(), meta=dict(n_dst=n_dst, fanout=fanout, d=d, op=op))
    return dsrc


# --------------------------------------------------------------------------------------------
# dense projections
# --------------------------------------------------------------------------------------------
def linear_fwd(x, w, bias=None, x2=None, w2=None, relu=False, x_rows=None, x2_rows=None, out=None, bias2=None):
    """``bias2``: the second projection's own bias (dual-input form only): (bias + bias2) is formed inside the launch.
    """
    x = as_mat(x); w = as_mat(w)
    M = x_rows.numel() if x_rows is not None else x.shape[0]
    K, N = x.shape[1], w.shape[0]
    assert w.shape[1] == K
    K2 = 0
    if x2 is not None:
        x2 = as_mat(x2); w2 = as_mat(w2)
        K2 = x2.shape[1]
        assert w2.shape == (N, K2)
        assert (x2_rows.numel() if x2_rows is not None else x2.shape[0]) == M
    if _x3_forward_ok(x, M, x2):
        # rows of a registered static table: pre-split image of the table, the weights are split (with the bias in the
        # appended slot) per call — 12 us for a 602 x 602 matrix
        img = _static_image(x)
        if img is not None:
            wimg = weight_image("wb", w, bias)
            if wimg is None:
                bvec = bias if bias is not None else torch.zeros(N, dtype=torch.float32, device=x.device)
                wimg = x3_split(w, append_vec=bvec)
            return linear_fwd_x3(img, x_rows, wimg, relu=relu, x_nrows=x.shape[0], M=M, out=out)
    y = out if out is not None else empty_mat(M, N, x.device)
    if bias2 is not None:
        assert bias is not None and x2 is not None
        _launch("ogl_linear_fwd", _lib.lib().ogl_linear_fwd_dual_bias,
                _ptr(x), _ld(x), _ptr(x_rows), x.shape[0], M, K, _ptr(w), _ld(w), N, _ptr(bias), _ptr(bias2),
                _ptr(x2), _ld(x2), _ptr(x2_rows), x2.shape[0], K2, _ptr(w2), _ld(w2), int(bool(relu)), _ptr(y), _ld(y), _stream(),
                meta=dict(M=M, K=K, N=N, K2=K2))
        return y
    _launch("ogl_linear_fwd", _lib.lib().ogl_linear_fwd, 
        _ptr(x), _ld(x), _ptr(x_rows), x.shape[0], M, K, _ptr(w), _ld(w), N, _ptr(bias),
        _ptr(x2), _ld(x2) if x2 is not None else 0, _ptr(x2_rows), x2.shape[0] if x2 is not None else 0, K2,
        _ptr(w2), _ld(w2) if w2 is not None else 0, int(bool(relu)), _ptr(y), _ld(y), _stream(), meta=dict(M=M, K=K, N=N, K2=K2))
    return y


def linear_fwd_addrows(x, w, add, add_rows=None, bias=None, relu=False, x_rows=None):
    """y = act(x[rows] @ w.T + bias + add[add_rows]): a projection whose other term comes from a per-vertex table."""
    x = as_mat(x); w = as_mat(w); add = as_mat(add)
    M = x_rows.numel() if x_rows is not None else x.shape[0]
    K, N = x.shape[1], w.shape[0]
    assert w.shape[1] == K and add.shape[1] == N and (add_rows is None or add_rows.numel() == M)
    y = empty_mat(M, N, x.device)
    _launch("ogl_linear_fwd_addrows", _lib.lib().ogl_linear_fwd_addrows, _ptr(x), _ld(x), _ptr(x_rows), x.shape[0], M, K, _ptr(w), _ld(w),
            N, _ptr(bias), _ptr(add), _ld(add), _ptr(_ids(add_rows) if add_rows is not None else None), add.shape[0], int(bool(relu)),
            _ptr(y), _ld(y), _stream(), meta=dict(M=M, K=K, N=N, K2=0))
    return y


def relu_bwd(dy, y):
    """dy (.) [y > 0] (the mask of a fused-ReLU projection; applied once, the backward GEMMs stay mask-free)."""
    dy = as_mat(dy); y = as_mat(y)
    M, N = dy.shape
    out = empty_mat(M, N, dy.device)
    _launch("ogl_relu_bwd", _lib.lib().ogl_relu_bwd, _ptr(dy), _ld(dy), _ptr(y), _ld(y), M, N, _ptr(out), _ld(out), _stream(),
            meta=dict(M=M, N=N))
    return out


def out_layer_bwd_inputs(dy, w_self, w_neigh, argmax, neigh, n_src, dp_zeroed=None, finish_loss=None):
    """(dx_self [n_dst, K], dP [n_src, K]): dy . w_self, and dy . w_neigh scattered to the max winners (csrc/out_layer.hip).
    ``dp_zeroed``: an already zeroed [n_src, K] scatter target (``take_zeroed``).  ``finish_loss = (loss_rows, mean)``: the launch
    also writes mean <- sum(loss_rows) / n (the loss of a fused forward whose mean was left to its successor)."""
    dy = as_mat(dy); w_self = as_mat(w_self); w_neigh = as_mat(w_neigh); neigh = as_mat(neigh)
    n_dst, N = dy.shape
    K = w_self.shape[1]
    dx = empty_mat(n_dst, K, dy.device)
    dp = dp_zeroed if dp_zeroed is not None else empty_mat(n_src, K, dy.device, zero=True)
    if finish_loss is not None:
        rows, mean = finish_loss
        _launch("ogl_out_layer_bwd_inputs", _lib.lib().ogl_out_layer_bwd_inputs_mean, _ptr(dy), _ld(dy), n_dst, N, K, _ptr(w_self),
                _ld(w_self), _ptr(w_neigh), _ld(w_neigh), _ptr(argmax), _ptr(neigh), _ld(neigh), n_src, _ptr(dx), _ld(dx), _ptr(dp),
                _ld(dp), _ptr(rows), rows.numel(), _ptr(mean), _stream(), meta=dict(M=n_dst, N=N, K=K))
        return dx, dp
    _launch("ogl_out_layer_bwd_inputs", _lib.lib().ogl_out_layer_bwd_inputs, _ptr(dy), _ld(dy), n_dst, N, K, _ptr(w_self), _ld(w_self),
            _ptr(w_neigh), _ld(w_neigh), _ptr(argmax), _ptr(neigh), _ld(neigh), n_src, _ptr(dx), _ld(dx), _ptr(dp), _ld(dp), _stream(),
            meta=dict(M=n_dst, N=N, K=K))
    return dx, dp


def out_layer_bwd_weights(dy, x_self, x_neigh, want_bias=True, x_self_rows=None, dws_out=None, dwn_out=None):
    """(dw_self, dw_neigh [N, K], db, db2) of a few-column combine in one launch; ``x_self_rows`` gathers x_self's rows from a table."""
    dy = as_mat(dy); x_self = as_mat(x_self); x_neigh = as_mat(x_neigh)
    M, N = dy.shape
    assert x_neigh.shape[0] == M and (x_self_rows.numel() if x_self_rows is not None else x_self.shape[0]) == M
    K = x_self.shape[1]
    dev = dy.device
    dws = dws_out if dws_out is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
    dwn = dwn_out if dwn_out is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    db2 = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    _launch("ogl_out_layer_bwd_weights", _lib.lib().ogl_out_layer_bwd_weights, _ptr(dy), _ld(dy), M, N, K, _ptr(x_self), _ld(x_self),
            _ptr(_ids(x_self_rows) if x_self_rows is not None else None), x_self.shape[0], _ptr(x_neigh), _ld(x_neigh), _ptr(dws), _ld(dws), _ptr(dwn), _ld(dwn), _ptr(db), _ptr(db2), _stream(), meta=dict(M=M, N=N, K=K))
    return dws, dwn, db, db2


OUT_LAYER_FUSED = os.environ.get("OGL_OUT_LAYER_FUSED") != "0"
# the output layer's forward tail — neighbour max, [n_dst, 2K] -> N projection, cross entropy — as ONE launch (ogl_out_layer_fwd_ce)
FUSED_OUT_FWD = True
DEFER_LOSS_MEAN = True     # the fused loss's mean is finished by the backward's first launch
OUT_FWD_ROWS = 0        # destinations per block (0: automatic)


def out_loss_fits(h, n_dst, idx, w_self, w_neigh, p_width):
    """The fused forward + loss of a few-column 'pool' layer applies (and so does its two-launch backward, ``_out_layer_fits``)."""
    N, K = w_self.shape
    return (FUSED_OUT_FWD and OUT_LAYER_FUSED and idx.dtype == torch.int32 and idx.dim() == 2 and p_width == K and h.shape[1] == K
            and 0 < n_dst <= 4096 and K >= 64 and h.shape[0] >= n_dst and w_self.is_contiguous() and w_neigh.is_contiguous()
            and w_self.data_ptr() % 16 == 0 and w_neigh.data_ptr() % 16 == 0
            and bool(_lib.lib().ogl_out_layer_fwd_ce_fits(int(n_dst), int(idx.shape[1]), int(K), int(N))))


def out_layer_fwd_ce(p, idx, h, n_dst, w_self, w_neigh, b_self, b_neigh, labels, want_grad=True, zero=None, want_mean=True):
    """(mean loss, row losses, logits, neigh, argmax, dlogits / n_dst) of the output layer from its pooled projection rows ``p`` =
    relu(fc_pool(h)) in ONE launch; ``labels``: int64 tensor or LazyLabels; ``zero``: a contiguous fp32 buffer the grid clears on
    the side (the scatter target of the layer's backward).  ``want_mean=False``: the returned mean tensor gets NaN from this launch
    (no last-block-done count, no device-scope fences) and its value from ``out_layer_bwd_inputs(finish_loss=...)``."""
    p = as_mat(p); h = as_mat(h); w_self = as_mat(w_self); w_neigh = as_mat(w_neigh)
    K, N = p.shape[1], w_self.shape[0]
    dev = p.device
    lazy = labels if isinstance(labels, LazyLabels) else None
    if lazy is None:
        labels = labels.reshape(-1)
        assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous()
    assert labels.numel() == n_dst and idx.shape[0] == n_dst and idx.is_contiguous()
    neigh = empty_mat(n_dst, K, dev)
    argmax = torch.empty((n_dst, K), dtype=torch.int32, device=dev) if want_grad else None
    logits = empty_mat(n_dst, N, dev)
    loss = torch.empty(n_dst, dtype=torch.float32, device=dev)
    mean = torch.empty((), dtype=torch.float32, device=dev)
    dl = empty_mat(n_dst, N, dev) if want_grad else None
    stream = _stream()
    table, ids = (lazy.table, lazy.ids) if lazy is not None else (labels, None)
    zn = 0
    if zero is not None:
        assert zero.is_contiguous() and zero.dtype == torch.float32 and zero.numel() % 4 == 0
        zn = zero.numel()
    _launch("ogl_out_layer_fwd_ce", _lib.lib().ogl_out_layer_fwd_ce, _ptr(p), _ld(p), p.shape[0], _ptr(idx), n_dst, int(idx.shape[1]),
            _ptr(h), _ld(h), K, _ptr(w_self), _ld(w_self), _ptr(w_neigh), _ld(w_neigh), _ptr(b_self), _ptr(b_neigh), N, _ptr(neigh),
            _ld(neigh), _ptr(argmax), _ptr(logits), _ld(logits), _ptr(table), table.numel(), _ptr(ids), C.c_float(1.0 / n_dst),
            _ptr(loss), _ptr(dl), _ld(dl) if dl is not None else 0, _ptr(mean), ce_counter(dev, stream) if want_mean else None, _ptr(zero), zn,
            OUT_FWD_ROWS, stream, meta=dict(n_dst=n_dst, fanout=int(idx.shape[1]), d=K, N=N, zero_bytes=4 * zn))
    return mean, loss, logits, neigh, argmax, dl


def _out_layer_fits(dy, h, w_self, w_neigh):
    return (OUT_LAYER_FUSED and dy.shape[1] <= 64 and 0 < dy.shape[0] <= 4096 and h.shape[1] >= 64 and _ld(dy) % 4 == 0
            and dy.data_ptr() % 16 == 0 and as_mat(w_self).data_ptr() % 16 == 0 and as_mat(w_neigh).data_ptr() % 16 == 0)


def relu_bwd_img(dy, y):
    """relu_bwd that also returns the bf16x3 image of the masked gradient: (out, X3Image)."""
    dy = as_mat(dy); y = as_mat(y)
    M, N = dy.shape
    out = empty_mat(M, N, dy.device)
    img = X3Image(_x3_alloc(M, N, dy.device), M, N)
    _launch("ogl_relu_bwd_img", _lib.lib().ogl_relu_bwd_img, _ptr(dy), _ld(dy), _ptr(y), _ld(y), M, N, _ptr(out), _ld(out),
            _ptr(img.buf), _stream(), meta=dict(M=M, N=N))
    return out, img


# The ReLU backward of a layer's output in the epilogue of the product that computes that output's gradient (ogl_linear_fwd_x3_ext's
# `mask`): removes the 16 us ogl_relu_bwd_img launch of the Reddit step and 60 MB of traffic — and measured NO gain (same box,
# alternating replayed runs: 1.056 / 1.066 ms without, 1.063 / 1.078 with): the masked epilogue costs the product 10 us on the
# critical path, while the separate pass ran beside the side stream's weight gradients.  Off (a module constant: tests/test_gpu_round3.py
# flips it to keep the masked epilogue of ogl_linear_fwd_x3_ext exercised).
FUSE_RELU_BWD = False


def linear_bwd_input(dy, w, ymask=None, dy_img=None, add_head=None, out_relu_mask=None):
    """dX = dY . W.  ``dy_img``: the bf16x3 image of dY when its producer wrote one — the product then runs on the image kernel
    against the image of W^T (a 600 x 600 transpose + split: two small launches, or one of the step's prepared weight images).
    ``add_head`` [n, K]: added to the first n rows of dX (the fc_self path of a layer's input gradient) — in the epilogue of the
    image kernel when that runs, in place otherwise.
    ``out_relu_mask`` [M, K] (image path only; ignored elsewhere): X is the output of a fused-ReLU projection whose value this is —
    dX leaves the kernel multiplied by [X > 0] with its bf16x3 image attached, and carries ``_ogl_premasked`` so that the ReLU's own
    backward recognises it (masking twice would be harmless: the mask is idempotent)."""
    dy = as_mat(dy); w = as_mat(w)
    if ymask is not None:
        dy = relu_bwd(dy, ymask)
        dy_img = None
    M, N = dy.shape
    K = w.shape[1]
    if dy_img is not None and dy_img.rows == M and dy_img.K == N:
        wt = weight_image("T", w)
        wt = wt if wt is not None else x3_split(transpose(w))
        if out_relu_mask is not None and K % 4 == 0 and _ld(as_mat(out_relu_mask)) % 4 == 0 and out_relu_mask.data_ptr() % 16 == 0:
            dx, img = linear_fwd_x3_ext(dy_img, None, wt, add=add_head, mask=out_relu_mask, want_image=True)
            attach_image(dx, img)
            dx._ogl_premasked = (out_relu_mask.data_ptr(), out_relu_mask._version)
            return dx
        if add_head is not None:
            return linear_fwd_x3_ext(dy_img, None, wt, add=add_head)
        return linear_fwd_x3(dy_img, None, wt)
    if add_head is not None:
        dx = linear_bwd_input(dy, w)
        dx[:add_head.shape[0]].add_(add_head)
        return dx
    if M >= BWD_INPUT_VIA_FWD_MIN_ROWS and get_gemm_mode() != "f32":
        # dX = dY . W as dY . (W^T)^T: with the (small) weight transposed first, both operands of the product are
        # reduction-contiguous and it runs on the forward kernel — 60 us against 75-84 us at the n1-row shapes (the
        # 13 us transpose included), bit-identical results (tools/_bwd_input_probe.py)
        return linear_fwd(dy, transpose(w), None)
    dx = empty_mat(M, K, dy.device)
    _launch("ogl_linear_bwd_input", _lib.lib().ogl_linear_bwd_input, _ptr(dy), _ld(dy), M, N, _ptr(w), _ld(w), K, _ptr(dx), _ld(dx),
            _stream(), meta=dict(M=M, K=K, N=N))
    return dx


def linear_bwd_weight(dy, x, ymask=None, x_rows=None, want_bias=True, dw_out=None):
    dy = as_mat(dy); x = as_mat(x)
    if ymask is not None:
        dy = relu_bwd(dy, ymask)
    M, N = dy.shape
    K = x.shape[1]
    dev = dy.device
    dw = dw_out if dw_out is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    nbytes = int(_lib.lib().ogl_linear_bwd_weight_workspace_bytes(M, N, K))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    _launch("ogl_linear_bwd_weight", _lib.lib().ogl_linear_bwd_weight, _ptr(dy), _ld(dy), _ptr(x), _ld(x), _ptr(x_rows), x.shape[0],
            M, N, K, _ptr(dw), _ld(dw), _ptr(db), _ptr(ws), nbytes, _stream(), meta=dict(M=M, K=K, N=N))
    return dw, db


def transpose(src, rows=None):
    """[N, M] = src[rows?].T (LDS-tiled, optional row gather): operands of the transposed weight-gradient form."""
    src = as_mat(src)
    M = rows.numel() if rows is not None else src.shape[0]
    N = src.shape[1]
    dst = empty_mat(N, M, src.device)
    _launch("ogl_transpose", _lib.lib().ogl_transpose, _ptr(src), _ld(src), _ptr(rows), src.shape[0], M, N, _ptr(dst), _ld(dst),
            _stream(), meta=dict(M=M, N=N))
    return dst


def linear_bwd_weight_t(dyT, xT, want_bias=True, dw_out=None):
    """dw [N, K] = dyT [N, M] @ xT [K, M].T, db = dyT.sum(1): the weight gradient as a reduction-contiguous product."""
    dyT = as_mat(dyT); xT = as_mat(xT)
    N, M = dyT.shape
    K = xT.shape[0]
    assert xT.shape[1] == M
    dev = dyT.device
    dw = dw_out if dw_out is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    nbytes = int(_lib.lib().ogl_linear_bwd_weight_t_workspace_bytes(M, N, K))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    _launch("ogl_linear_bwd_weight_t", _lib.lib().ogl_linear_bwd_weight_t, _ptr(dyT), _ld(dyT), _ptr(xT), _ld(xT), M, N, K, _ptr(dw),
            _ld(dw), _ptr(db), _ptr(ws), nbytes, _stream(), meta=dict(M=M, K=K, N=N))
    return dw, db


class X3Image:
    """bf16x3 image of an fp32 matrix (include/ogl_hip.h, "pre-split operands"): ``rows`` image rows over a reduction of
    length ``K``.  ``buf`` is the raw device allocation (ogl_x3_image_bytes)."""

    def __init__(self, buf, rows, K):
        self.buf, self.rows, self.K = buf, int(rows), int(K)

    @property
    def nbytes(self):
        return self.buf.numel()


def _x3_alloc(rows, K, device):
    n = int(_lib.lib().ogl_x3_image_bytes(rows, K))
    return torch.empty(max(n, 16), dtype=torch.uint8, device=device)


def x3_split(x, rows=None, append_ones=False, append_vec=None):
    """Image of x[rows] (or of x): one image row per matrix row, reduction over the columns.  ``append_ones`` /
    ``append_vec`` add the extra reduction element that folds a bias into the product (activations / weights side)."""
    x = as_mat(x)
    R = rows.numel() if rows is not None else x.shape[0]
    K = x.shape[1]
    append = 2 if append_vec is not None else (1 if append_ones else 0)
    Ki = K + (1 if append else 0)
    buf = _x3_alloc(R, Ki, x.device)
    _launch("ogl_x3_split", _lib.lib().ogl_x3_split, _ptr(x), _ld(x), _ptr(_ids(rows) if rows is not None else None), x.shape[0],
            R, K, append, _ptr(append_vec), _ptr(buf), _stream(), meta=dict(R=R, K=K))
    return X3Image(buf, R, Ki)


def x3_split_t(x, rows=None, ones_row=False, interleave=0):
    """Image of x[rows].T (reduction over the M rows); ``ones_row`` appends the all-ones image row; ``interleave`` = G
    deals the reduction index round-robin over G groups of 32 (the layout of pool_bwd_x3)."""
    x = as_mat(x)
    M = rows.numel() if rows is not None else x.shape[0]
    N = x.shape[1]
    nimg = N + (1 if ones_row else 0)
    Mi = 32 * interleave if interleave else M
    buf = _x3_alloc(nimg, Mi, x.device)
    _launch("ogl_x3_split_t", _lib.lib().ogl_x3_split_t, _ptr(x), _ld(x), _ptr(_ids(rows) if rows is not None else None), x.shape[0],
            M, N, 1 if ones_row else 0, interleave, _ptr(buf), _stream(), meta=dict(M=M, N=N))
    return X3Image(buf, nimg, Mi)


def linear_fwd_x3(x_img, x_rows, w_img, relu=False, x_nrows=None, M=None, out=None):
    """y = act(x_img[x_rows] @ w_img.T); a bias is folded into the images (x3_split append_ones / append_vec).
    ``x_nrows`` bounds the valid gather ids (default: every image row); without a gather ``M`` selects a row prefix."""
    M = x_rows.numel() if x_rows is not None else (x_img.rows if M is None else M)
    x_nrows = x_img.rows if x_nrows is None else x_nrows
    K, N = x_img.K, w_img.rows
    assert w_img.K == K, "both images must be built with the same append choice"
    y = out if out is not None else empty_mat(M, N, x_img.buf.device)
    assert y.shape[0] == M and y.shape[1] == N
    _launch("ogl_linear_fwd_x3", _lib.lib().ogl_linear_fwd_x3, _ptr(x_img.buf), x_img.rows,
            _ptr(_ids(x_rows) if x_rows is not None else None), x_nrows, M, K, _ptr(w_img.buf), N, 1 if relu else 0, _ptr(y),
            _ld(y), _stream(), meta=dict(M=M, K=K, N=N, K2=0))
    return y


# bf16x3 images that travel beside an fp32 activation from the kernel that produced it to the projection that consumes it: an
# attribute of the activation's tensor OBJECT (autograd.Function hands the same objects in and out), stamped with the tensor's
# version counter — a different tensor at a recycled address, or this one after an in-place write, never inherits an image
N1_BWD_SPLIT = True        # the scattered pool gradient dP (atomics: no producer can write its image) gets a split pass of its own
X3_N1_MIN_ROWS = (1 << 40) if os.environ.get("OGL_N1_IMAGES") == "0" else 2048      # projections with at least this many rows run on the image kernel when their operand images exist


def attach_image(t, img):
    t._ogl_image = (img, t._version, t.data_ptr())
    return t


def take_image(t, pop=True):
    ent = getattr(t, "_ogl_image", None)
    if ent is None:
        return None
    if pop:
        del t._ogl_image
    img, version, ptr = ent
    return img if (version == t._version and ptr == t.data_ptr() and img.rows == t.shape[0]) else None


def x3_split_cat(parts):
    """K-concatenated (weight) image: ``parts`` = [(matrix [N, K_p], bias-or-None), ...] with equal row counts; part p occupies
    ceil((K_p + has_bias) / 32) groups of every image row, in order.  The B operand of ``linear_fwd_x3_ext``."""
    mats = [(as_mat(m), b) for m, b in parts]
    N = mats[0][0].shape[0]
    groups = [-(-(m.shape[1] + (1 if b is not None else 0)) // 32) for m, b in mats]
    G = sum(groups)
    buf = torch.empty(max((N + 1) * G * 192, 16), dtype=torch.uint8, device=mats[0][0].device)
    off = 0
    for (m, b), gp in zip(mats, groups):
        assert m.shape[0] == N
        _launch("ogl_x3_split_into", _lib.lib().ogl_x3_split_into, _ptr(m), _ld(m), N, m.shape[1], 2 if b is not None else 0, _ptr(b),
                _ptr(buf), G * 192, off, _stream(), meta=dict(R=N, K=m.shape[1]))
        off += gp
    return X3Image(buf, N, 32 * G)


def linear_fwd_x3_ext(x_img, x_rows, w_img, x2_img=None, x2_rows=None, add=None, add_rows=None, relu=False, x_nrows=None,
                      x2_nrows=None, M=None, want_image=False, image_append_ones=False, out=None, mask=None, y_keep=None):
    """``linear_fwd_x3`` with a second A part (``w_img`` K-concatenated, x3_split_cat), a per-row addend ``add[add_rows]`` and /
    or the bf16x3 image of the output (returned as the second value when ``want_image``).  ``y_keep`` (uint8 [M], with
    ``want_image``): fp32 rows are stored only where it is non-zero — the other rows of the returned matrix are UNDEFINED."""
    M = x_rows.numel() if x_rows is not None else (x_img.rows if M is None else M)
    x_nrows = x_img.rows if x_nrows is None else x_nrows
    K1 = x_img.K
    K2 = x2_img.K if x2_img is not None else 0
    N = w_img.rows
    assert -(-w_img.K // 32) == -(-K1 // 32) + -(-K2 // 32), "the weight image must be K-concatenated over the A parts"
    dev = x_img.buf.device
    y = out if out is not None else empty_mat(M, N, dev)
    if add is not None:
        add = as_mat(add)
        assert add.shape[1] == N and (add_rows is None or add_rows.numel() == M)
    if mask is not None:
        mask = as_mat(mask)
        assert mask.shape == (M, N) and _ld(mask) % 4 == 0 and N % 4 == 0 and mask.data_ptr() % 16 == 0
    if y_keep is not None:
        assert want_image and y_keep.dtype == torch.uint8 and y_keep.is_cuda and y_keep.is_contiguous() and y_keep.numel() == M
    img = None
    if want_image:
        Ki = N + (1 if image_append_ones else 0)
        img = X3Image(_x3_alloc(M, Ki, dev), M, Ki)
    _launch("ogl_linear_fwd_x3_ext", _lib.lib().ogl_linear_fwd_x3_ext, _ptr(x_img.buf), x_img.rows,
            _ptr(_ids(x_rows) if x_rows is not None else None), x_nrows, K1,
            _ptr(x2_img.buf) if x2_img is not None else None, x2_img.rows if x2_img is not None else 0,
            _ptr(_ids(x2_rows) if x2_rows is not None else None),
            (x2_img.rows if x2_nrows is None else x2_nrows) if x2_img is not None else 0, K2, M, _ptr(w_img.buf), N,
            _ptr(add), _ld(add) if add is not None else 0, _ptr(_ids(add_rows) if add_rows is not None else None),
            add.shape[0] if add is not None else 0, 1 if relu else 0, _ptr(y), _ld(y), _ptr(img.buf) if img is not None else None,
            1 if image_append_ones else 0, _ptr(mask), _ld(mask) if mask is not None else 0, _ptr(y_keep), _stream(),
            meta=dict(M=M, K=K1, N=N, K2=K2))
    return (y, img) if want_image else y


# ---- weight images of a step -------------------------------------------------------------------------------------------
# Every image product needs the bf16x3 image of its weight matrix, rebuilt after every optimiser step.  One at a time these are
# launch-bound 6-7 us kernels (and a transpose before the input-gradient ones); a model that knows which products its step will
# run asks for all of them at once (GraphSAGE.forward -> weight_images_prepare: ONE launch).  Entries are keyed by the
# parameters' storage + version counter and dropped by the optimisers (which update through raw pointers).
_W_IMAGES = {}
PREPARE_WEIGHT_IMAGES = True
# optim.Adam (device-side step count) may ask the step's weight-image launch to compute its per-step scalars: ``req`` = (step_dev,
# scalars_dev, lr, beta1, beta2) until a launch takes it, ``served`` = (step_dev address, made while capturing?) afterwards
ADAM_PRIME_IN_SPLIT = True
_ADAM_PRIME = {"req": None, "served": None}


def adam_prime(step_dev, scalars_dev, lr, beta1, beta2):
    """The NEXT weight-image launch also prepares this optimiser's step (++step count, bias-correction scalars)."""
    _ADAM_PRIME["req"], _ADAM_PRIME["served"] = (step_dev, scalars_dev, float(lr), float(beta1), float(beta2)), None


def adam_primed(step_dev):
    """True (once) when a weight-image launch since ``adam_prime`` prepared this optimiser's step — in the mode we are in now
    (a launch recorded into a graph prepares that graph's replays, not an eager step, and vice versa)."""
    served, _ADAM_PRIME["served"] = _ADAM_PRIME["served"], None
    _ADAM_PRIME["req"] = None
    return served is not None and served == (step_dev.data_ptr(), _capturing())


class _X3SplitPart(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld", C.c_int64), ("R", C.c_int64), ("K", C.c_int32), ("transpose", C.c_int32),
                ("append", C.c_int32), ("vec1", C.c_void_p), ("vec2", C.c_void_p), ("image", C.c_void_p),
                ("image_row_bytes", C.c_int64), ("group_offset", C.c_int64)]


def _wkey(kind, *tensors):
    return (kind,) + tuple(None if t is None else (t.data_ptr(), t._version) for t in tensors)


def invalidate_weight_images():
    _W_IMAGES.clear()


def _capturing():
    return torch.cuda.is_current_stream_capturing()


def _alive(ent):
    """The tensors an entry was built from still exist.  The key is (address, version counter): a NEW tensor allocated at the
    address of a freed one (another model's parameters in the same process) carries the same key — but then the old tensor,
    which the entry remembers weakly, is gone."""
    return all(r() is not None for r in ent[2])


def weight_image(kind, *tensors):
    """The prepared image for this request, or None.  An image built while a hipGraph was being captured exists only inside
    that graph's replays (its split kernel did not run), and one built eagerly is not rebuilt by a replay: an entry serves
    only the mode it was made in."""
    ent = _W_IMAGES.get(_wkey(kind, *tensors))
    return ent[0] if (ent is not None and ent[1] == _capturing() and _alive(ent)) else None


def weight_images_prepare(requests):
    """requests: [(kind, tensors)] with kind / tensors one of
         ("wb", (w, b))                    image of w with the bias slot (b may be None: 0) — B operand against an activation image with a ones slot
         ("cat", (w, w2, b, b2))           K-concatenated [w | b + b2] [w2]                 — B operand of the two-part combine product
         ("T", (w,))                       image of w^T                                   — B operand of dX = dY . w
         ("bsum", (b, b2))                 not an image: the fp32 vector b + b2 (weight_image returns a tensor) — the summed bias of a
                                           dual projection on fp32 operands, riding in the same launch
       All of them in one launch (at most 8 parts; further requests are left to their consumers)."""
    parts, made = [], []
    cap = _capturing()
    for kind, ts in requests:
        key = _wkey(kind, *ts)
        if key in _W_IMAGES and _W_IMAGES[key][1] == cap and _alive(_W_IMAGES[key]):
            continue
        if kind == "bsum":
            if len(parts) + 1 > 8:
                break
            out = torch.empty(ts[0].numel(), dtype=torch.float32, device=ts[0].device)
            parts.append((None, ts[0].numel(), 0, 2, 0, ts[0], ts[1], out, 0, 0))
            made.append((key, out, ts))
            continue
        w = as_mat(ts[0])
        dev = w.device
        if kind == "T":
            need, R, groups = 1, w.shape[1], [-(-w.shape[0] // 32)]
        elif kind == "wb":
            need, R, groups = 1, w.shape[0], [-(-(w.shape[1] + 1) // 32)]
        else:
            w2 = as_mat(ts[1])
            need, R, groups = 2, w.shape[0], [-(-(w.shape[1] + 1) // 32), -(-w2.shape[1] // 32)]
        if len(parts) + need > 8:
            break
        G = sum(groups)
        buf = torch.empty(max((R + 1) * G * 192, 16), dtype=torch.uint8, device=dev)
        if kind == "T":
            parts.append((w, w.shape[1], w.shape[0], 1, 0, None, None, buf, G, 0))
        elif kind == "wb":
            parts.append((w, R, w.shape[1], 0, 1, ts[1], None, buf, G, 0))
        else:
            parts.append((w, R, w.shape[1], 0, 1, ts[2], ts[3], buf, G, 0))
            parts.append((w2, R, w2.shape[1], 0, 0, None, None, buf, G, groups[0]))
        made.append((key, X3Image(buf, R, 32 * G if kind == "cat" else (w.shape[0] if kind == "T" else w.shape[1] + 1)), ts))
    if not parts:
        return
    arr = (_X3SplitPart * len(parts))()
    for a, (m, R, K, tr, app, v1, v2, buf, G, off) in zip(arr, parts):
        a.src, a.ld, a.R, a.K, a.transpose, a.append = _ptr(m), (_ld(m) if m is not None else 0), R, K, tr, app
        a.vec1, a.vec2, a.image, a.image_row_bytes, a.group_offset = _ptr(v1), _ptr(v2), _ptr(buf), G * 192, off
    prime = _ADAM_PRIME.get("req")
    if prime is not None and ADAM_PRIME_IN_SPLIT:
        # the optimiser's per-step scalars ride in this launch (the step's first): its own one-thread launch at the END of the step,
        # and the gap in front of it, leave the critical path
        step_dev, scal, lr, b1, b2 = prime
        _launch("ogl_x3_split_multi", _lib.lib().ogl_x3_split_multi, C.cast(arr, C.c_void_p), len(parts), _ptr(step_dev), _ptr(scal),
                C.c_double(lr), C.c_double(b1), C.c_double(b2), _stream(), meta=dict(parts=len(parts), adam_prepare=True))
        _ADAM_PRIME["req"], _ADAM_PRIME["served"] = None, (step_dev.data_ptr(), _capturing())
    else:
        _launch("ogl_x3_split_multi", _lib.lib().ogl_x3_split_multi, C.cast(arr, C.c_void_p), len(parts), None, None, C.c_double(0.0),
                C.c_double(0.0), C.c_double(0.0), _stream(), meta=dict(parts=len(parts)))
    if len(_W_IMAGES) > 32:
        _W_IMAGES.clear()               # stale versions of re-assigned parameters: never let them pile up
    import weakref
    for key, img, ts in made:
        _W_IMAGES[key] = (img, cap, [weakref.ref(t) for t in ts if t is not None])


def _n1_images_ok(M, *widths):
    """The n1-row products of a step run on the image kernel when their operand images come for free (emitted by the kernel
    that produced the operand) and the product is tall enough to fill the chip."""
    return _MODE["name"] != "f32" and M >= X3_N1_MIN_ROWS and all(64 <= wd <= 640 for wd in widths)


def _dual_fwd_images(x, w, bias, bias2, x2, w2, relu, x_rows, x2_rows):
    """fc_self(table[dst]) + fc_neigh(neigh) (+ bias, ReLU) as ONE image product over a two-part A operand: the rows of the
    resident table's image and the image the aggregator wrote beside ``neigh`` — and the image of the result goes out beside
    it for the next layer's fc_pool.  None when the operands' images are not at hand (the caller runs the fp32-operand
    kernel)."""
    if x2 is None or x_rows is None or x2_rows is not None:
        return None
    x = as_mat(x); x2 = as_mat(x2)
    M = x_rows.numel()
    if not _n1_images_ok(M, x2.shape[1], w.shape[0]) or x2.shape[0] != M:
        return None
    x2img = take_image(x2, pop=False)
    if x2img is None or (x.data_ptr(), _ld(x), x.shape[1]) not in _X3_TABLES:
        return None
    ximg = _static_image(x)
    if ximg is None or ximg.nbytes >= (1 << 32):             # (the two-part product lives in the 32-bit-offset kernel)
        return None
    take_image(x2)
    wcat = weight_image("cat", w, w2, bias, bias2)
    if wcat is None:
        weight_images_prepare([("cat", (w, w2, bias, bias2))])
        wcat = weight_image("cat", w, w2, bias, bias2)
    y, yimg = linear_fwd_x3_ext(ximg, x_rows, wcat, x2_img=x2img, relu=relu, x_nrows=x.shape[0], want_image=True,
                                image_append_ones=True)
    return attach_image(y, yimg)


def pool_bwd_x3(dout, argmax, relu_out, idx32, n_src):
    """Image of dP^T for the relu -> max-pool backward (see include/ogl_hip.h): rows = features, reduction = the source
    rows dealt round-robin over G = ceil(n_src / 32) groups (build the other operand with x3_split_t(interleave=G))."""
    dout = as_mat(dout)
    n_dst, d = dout.shape
    assert idx32.dtype == torch.int32 and idx32.is_contiguous() and argmax.dtype == torch.int32 and argmax.is_contiguous()
    G = (n_src + 31) // 32
    buf = _x3_alloc(d, 32 * G, dout.device)
    nbytes = int(_lib.lib().ogl_pool_bwd_x3_workspace_bytes(n_dst, idx32.shape[1], d, n_src))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dout.device)
    relu_out = as_mat(relu_out) if relu_out is not None else None
    _launch("ogl_pool_bwd_x3", _lib.lib().ogl_pool_bwd_x3, _ptr(dout), _ld(dout), _ptr(argmax), _ptr(relu_out),
            _ld(relu_out) if relu_out is not None else 0, _ptr(idx32), n_dst, idx32.shape[1], d, n_src, _ptr(buf), _ptr(ws), nbytes,
            _stream(), meta=dict(n_dst=n_dst, d=d, n_src=n_src, fanout=idx32.shape[1]))
    return X3Image(buf, d, 32 * G)


class PoolPlan:
    """What ogl_pool_bwd_x3_plan left in its workspace, and the event (None: same stream) that marks the end of the plan launches;
    ``pending``: the plan's launches are still waiting for the caller's next launch to be created first (``_DEFERRED``)."""
    __slots__ = ("ws", "nbytes", "event", "shape", "pending", "groups")

    def __init__(self, ws, nbytes, event, shape):
        self.ws, self.nbytes, self.event, self.shape, self.pending, self.groups = ws, nbytes, event, shape, False, False


POOL_PLAN = os.environ.get("OGL_POOL_PLAN", "1") != "0"
def pool_bwd_x3_plan(argmax, relu_out, idx32, n_src, side=True):
    """The part of the layer-0 pool backward that needs no gradient (a destination's columns in slot order, the slot offsets, the
    per-group record counts, their scan and every (destination, slot) segment's place in the group-major record array), enqueued NOW
    — by the forward pass, on the side stream when the fork is on (beside the forward products: the backward then has the values
    pass and the streamed group pass on its critical path).  Returns the PoolPlan ``pool_bwd_x3_apply`` consumes."""
    n_dst, d = argmax.shape
    assert idx32.dtype == torch.int32 and idx32.is_contiguous() and argmax.dtype == torch.int32 and argmax.is_contiguous()
    nbytes = int(_lib.lib().ogl_pool_bwd_x3_workspace_bytes(n_dst, idx32.shape[1], d, n_src))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=argmax.device)         # (allocated on the main stream)
    relu_out = as_mat(relu_out) if relu_out is not None else None

    def launch():
        _launch("ogl_pool_bwd_x3_plan", _lib.lib().ogl_pool_bwd_x3_plan, _ptr(argmax), _ptr(relu_out),
                _ld(relu_out) if relu_out is not None else 0, _ptr(idx32), n_dst, idx32.shape[1], d, n_src, _ptr(ws), nbytes, _stream(),
                meta=dict(n_dst=n_dst, d=d, n_src=n_src, fanout=idx32.shape[1]))

    plan = PoolPlan(ws, nbytes, None, (n_dst, idx32.shape[1], d, n_src))
    return _plan_on_side(plan, launch, (ws, argmax, relu_out, idx32), side)


# (Two later starts of the layer-0 pool backward's plan were measured in round 4 and removed in round 6: parked until the next layer's
# fc_pool product was enqueued, 1.005-1.008 -> 1.013-1.017 ms per step; on a third stream of its own, +100 us: DESIGN.md section 8.)
def _plan_on_side(plan, launch, tensors, side=True):
    """Run ``launch()`` — the gradient-free half of a backward pass, enqueued by the FORWARD pass — on the side stream when the fork
    is on (else right here); ``plan.event`` then marks its end, ``plan.pending`` that its launches still wait for the caller's next
    launch to be created first (``_DEFERRED``: in a captured step the first-created child of a node keeps its parent's queue)."""
    forked = side and FORK_BACKWARD and _PROFILE is None and not _SIDE["off"] and (FORK_IN_GRAPHS or not _capturing())
    if not forked:
        launch()
        return plan
    dev = torch.cuda.current_device()
    st = _SIDE["streams"].get(dev)
    if st is None:
        st = _SIDE["streams"][dev] = torch.cuda.Stream(device=dev)
    plan.pending = True

    def make(here):
        def run():                      # its launches are created after the main stream's next one (see _DEFERRED)
            st.wait_event(here)
            with torch.cuda.stream(st):
                launch()
                plan.event = torch.cuda.Event()
                plan.event.record()
            plan.pending = False
            if not _capturing():
                # (a forward whose backward never runs frees these with nothing having waited for the side stream: the caching
                # allocator must not hand their memory to a later main-stream kernel while the plan's kernels still read it)
                for t in tensors:
                    if t is not None:
                        t.record_stream(st)
        return run

    here = torch.cuda.Event()
    here.record()                       # the plan's inputs exist from HERE on ...
    _DEFERRED.append(make(here))
    return plan


def _plan_ready(plan):
    """Make the current stream wait for a plan enqueued by ``_plan_on_side``."""
    if plan.pending:
        _flush_deferred()
    if plan.event is not None:
        torch.cuda.current_stream().wait_event(plan.event)


SEG_REDUCE_BWD = os.environ.get("OGL_SEG_REDUCE_BWD", "1") != "0"    # mean / sum backward as a planned segmented gather (no atomics)
SEG_T = os.environ.get("OGL_SEG_T", "1") != "0"      # 'meanpool' layer 0: the pooled rows' gradient as the transposed group-major image (one launch)
SEG_T_MAX_D = 640
SEG_MIN_EDGES = 4096


def seg_bwd_fits(idx, d, n_src):
    return (SEG_REDUCE_BWD and idx is not None and idx.dtype == torch.int32 and idx.dim() == 2 and idx.numel() >= SEG_MIN_EDGES
            and d % 4 == 0 and 4 <= d <= 1024 and 0 < n_src < (1 << 31))


def reduce_bwd_seg_plan(idx32, d, n_src, side=True, groups=False):
    """The gradient-free half of the segmented mean / sum backward (``ogl_reduce_bwd_seg_plan``: the block's edges sorted by source),
    enqueued NOW — by the forward pass, on the side stream when the fork is on.  Returns the plan ``reduce_bwd_seg_apply`` consumes;
    ``groups``: + the group-major copy of the lists ``reduce_bwd_seg_apply_t`` walks."""
    n_dst, fanout = idx32.shape
    assert idx32.dtype == torch.int32 and idx32.is_contiguous()
    nbytes = int(_lib.lib().ogl_reduce_bwd_seg_workspace_bytes(n_dst, fanout, d, n_src))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=idx32.device)

    def launch():
        _launch("ogl_reduce_bwd_seg_plan", _lib.lib().ogl_reduce_bwd_seg_plan, _ptr(idx32), n_dst, fanout, n_src, 1 if groups else 0, _ptr(ws),
                nbytes, _stream(), meta=dict(n_dst=n_dst, fanout=fanout, n_src=n_src, groups=bool(groups)))

    plan = PoolPlan(ws, nbytes, None, (n_dst, fanout, d, n_src))
    plan.groups = bool(groups)
    return _plan_on_side(plan, launch, (ws, idx32), side)


def reduce_bwd_seg_apply(dout, idx32, plan, op, mask=None, want_out=True, want_image=False, add=None):
    """(dsrc [n_src, d] or None, its row-major bf16x3 image or None) from a plan: dsrc[s] = (1 / fanout for 'mean') sum of dout over
    the edges into s, in edge order (no atomics, reproducible), optionally times [mask[s] > 0]; ``add`` [n_add <= n_src, d]: added onto
    the first n_add rows inside the launch (the head rows' own gradient of a SAGE layer)."""
    dout = as_mat(dout)
    n_dst, d = dout.shape
    fanout, n_src = idx32.shape[1], plan.shape[3]
    assert plan.shape == (n_dst, fanout, d, n_src) and op in ("mean", "sum") and (want_out or want_image)
    _plan_ready(plan)
    out = empty_mat(n_src, d, dout.device) if want_out else None
    img = X3Image(_x3_alloc(n_src, d, dout.device), n_src, d) if want_image else None
    if mask is not None:
        mask = as_mat(mask)
        assert tuple(mask.shape) == (n_src, d)
    if add is not None:
        add = as_mat(add)
        assert want_out and add.shape[1] == d and add.shape[0] <= n_src
    _launch("ogl_reduce_bwd_seg_apply", _lib.lib().ogl_reduce_bwd_seg_apply, _ptr(dout), _ld(dout), _ptr(idx32), n_dst, fanout, d,
            REDUCE_OPS[op], n_src, _ptr(mask), _ld(mask) if mask is not None else 0, _ptr(add), _ld(add) if add is not None else 0,
            add.shape[0] if add is not None else 0, _ptr(out), _ld(out) if out is not None else 0,
            _ptr(img.buf) if img is not None else None, _ptr(plan.ws), plan.nbytes, _stream(),
            meta=dict(n_dst=n_dst, fanout=fanout, d=d, n_src=n_src, op=op, out=out is not None, image=img is not None, mask=mask is not None))
    return out, img


def reduce_bwd_seg_apply_t(dout, idx32, plan, op, mask=None):
    """``reduce_bwd_seg_apply`` written as the TRANSPOSED group-major image (the layout of ``pool_bwd_x3``: sources dealt over
    G = ceil(n_src / 32) groups) — the dy operand of ``linear_bwd_weight_x3k(..., interleave=G)``."""
    dout = as_mat(dout)
    n_dst, d = dout.shape
    fanout, n_src = idx32.shape[1], plan.shape[3]
    assert plan.shape == (n_dst, fanout, d, n_src) and op in ("mean", "sum") and d <= SEG_T_MAX_D and getattr(plan, "groups", False)
    _plan_ready(plan)
    G = (n_src + 31) // 32
    buf = _x3_alloc(d, 32 * G, dout.device)
    if mask is not None:
        mask = as_mat(mask)
        assert tuple(mask.shape) == (n_src, d)
    _launch("ogl_reduce_bwd_seg_apply_t", _lib.lib().ogl_reduce_bwd_seg_apply_t, _ptr(dout), _ld(dout), n_dst, fanout, d,
            REDUCE_OPS[op], n_src, _ptr(mask), _ld(mask) if mask is not None else 0, _ptr(buf), _ptr(plan.ws), plan.nbytes, _stream(),
            meta=dict(n_dst=n_dst, fanout=fanout, d=d, n_src=n_src, op=op, mask=mask is not None))
    return X3Image(buf, d, 32 * G)


def pool_bwd_x3_apply(dout, idx32, plan, n_src):
    """``pool_bwd_x3`` from a plan: the gradient rows written into their planned places (one wave per destination), then one block
    per source group streaming its records into the slab and out as the image."""
    dout = as_mat(dout)
    n_dst, d = dout.shape
    assert plan.shape == (n_dst, idx32.shape[1], d, n_src) and idx32.dtype == torch.int32 and idx32.is_contiguous()
    _plan_ready(plan)
    G = (n_src + 31) // 32
    buf = _x3_alloc(d, 32 * G, dout.device)
    _launch("ogl_pool_bwd_x3_apply", _lib.lib().ogl_pool_bwd_x3_apply, _ptr(dout), _ld(dout), _ptr(idx32), n_dst, idx32.shape[1], d, n_src,
            _ptr(buf), _ptr(plan.ws), plan.nbytes, _stream(), meta=dict(n_dst=n_dst, d=d, n_src=n_src, fanout=idx32.shape[1]))
    return X3Image(buf, d, 32 * G)


def linear_bwd_weight_x3(dyT_img, xT_img, want_bias=True, dw_out=None):
    """dw [N, K], db [N] from the images of dy.T ([N rows, M]) and [x | 1].T ([K + 1 rows, M])."""
    N, M, K = dyT_img.rows, dyT_img.K, xT_img.rows - 1
    assert xT_img.K == M
    dev = dyT_img.buf.device
    dw = dw_out if dw_out is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    nbytes = int(_lib.lib().ogl_linear_bwd_weight_x3_workspace_bytes(M, N, K))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    _launch("ogl_linear_bwd_weight_x3", _lib.lib().ogl_linear_bwd_weight_x3, _ptr(dyT_img.buf), _ptr(xT_img.buf), M, N, K, _ptr(dw),
            _ld(dw), _ptr(db), _ptr(ws), nbytes, _stream(), meta=dict(M=M, K=K, N=N))
    return dw, db


# Split-K weight gradients whose reduction is left to the optimiser launch: inside ``deferred_splitk(optimizer)`` a k-major weight
# gradient that knows which PARAMETERS its results belong to (``defer_for``) runs without its reduction launch and leaves a SlabGrad
# per parameter here (keyed by the parameter's address: autograd may hand p.grad a clone of the — then unwritten — gradient tensor, the
# parameter itself stays put); ``optim.Adam`` sums the slabs inside its own launch and writes the gradient into p.grad as it goes.
# What is left when the context ends (an optimiser that does not know slabs, a gradient nobody stepped on) is reduced into p.grad then.
SLAB_ADAM = os.environ.get("OGL_SLAB_ADAM", "1") != "0"
_SLABS = {"on": False, "pending": {}}


class SlabGrad:
    """``out``: the (still unwritten) gradient tensor the product handed to autograd — where the plain reduction goes when the slabs
    cannot be left to the optimiser after all (a second product for the same parameter in one backward pass)."""
    __slots__ = ("ws", "stride", "ws_ld", "nsplit", "rows", "ncols", "col0", "out", "split", "col0b")

    def __init__(self, ws, stride, ws_ld, nsplit, rows, ncols, col0, out=None, split=0, col0b=0):
        self.ws, self.stride, self.ws_ld, self.nsplit, self.rows, self.ncols, self.col0 = ws, stride, ws_ld, nsplit, rows, ncols, col0
        # a TWO-RANGE gradient (a concat weight [rows, split + K2] behind the dual weight-gradient product): columns [0, split) at slab
        # column col0, columns [split, ncols) at slab column col0b; split == 0: one range
        self.split, self.col0b = int(split), int(col0b)
        # a WEAK reference: AccumulateGrad adopts a gradient tensor only while nobody else holds it — a strong reference here made
        # autograd CLONE every deferred gradient (eight device-to-device copies per Reddit step, +27 us: found by a same-box A/B against
        # the round-4 tree and the copy launches in the traced step)
        self.out = weakref.ref(out) if out is not None else None


def take_slabs(param):
    """The pending SlabGrad of ``param`` (removed from the table) or None."""
    return _SLABS["pending"].pop(param.data_ptr(), None) if _SLABS["pending"] else None


def _slabs_settle(params):
    """A SECOND weight-gradient product for a parameter whose slabs are still pending (shared weights, two forward passes summed into
    one loss): the pending slabs are reduced into the tensor their product returned — autograd then adds two WRITTEN gradients — and
    the caller runs its own product with the reduction launch.  True when the caller must not defer."""
    pend, hit = _SLABS["pending"], False
    for t in params:
        if t is None:
            continue
        sg = pend.pop(t.data_ptr(), None)
        if sg is not None:
            hit = True
            o = sg.out() if sg.out is not None else None
            if o is not None:
                slab_reduce(sg, o)
        if t.grad is not None:                # gradient accumulation: the optimiser launch would overwrite what is already there
            hit = True
    return hit


def slab_reduce(sg, out):
    """out[rows, ncols] <- the plain reduction of the slabs (slab order).  ``out``: contiguous, or a 2-D view whose rows are contiguous
    (a column block of a wider matrix: its row stride is passed on)."""
    if sg.split:
        # (two ranges: each into its column block of ``out``)
        assert out.dim() == 2 and tuple(out.shape) == (sg.rows, sg.ncols)
        slab_reduce(SlabGrad(sg.ws, sg.stride, sg.ws_ld, sg.nsplit, sg.rows, sg.split, sg.col0), out[:, :sg.split])
        slab_reduce(SlabGrad(sg.ws, sg.stride, sg.ws_ld, sg.nsplit, sg.rows, sg.ncols - sg.split, sg.col0b), out[:, sg.split:])
        return
    if out.dim() == 2 and not out.is_contiguous():
        assert tuple(out.shape) == (sg.rows, sg.ncols) and out.stride(1) == 1 and out.stride(0) >= sg.ncols
        ld = out.stride(0)
    else:
        assert out.is_contiguous() and out.numel() == sg.rows * sg.ncols
        ld = sg.ncols
    _launch("ogl_x3_slab_reduce", _lib.lib().ogl_x3_slab_reduce, _ptr(sg.ws), sg.stride, sg.ws_ld, sg.nsplit, sg.rows, sg.ncols, sg.col0,
            _ptr(out), ld, _stream(), meta=dict(n=sg.rows * sg.ncols, nsplit=sg.nsplit))


# Both weight gradients of a dual-input projection as ONE k-major product over a two-part B operand (ogl_linear_bwd_weight_x3k_dual_slabs).
# OGL_DUAL_DW=0: two products.
DUAL_DW = os.environ.get("OGL_DUAL_DW", "1") != "0"
# ... and for the in-repo layer's ONE concat weight: its gradient stays in the slabs as a TWO-RANGE tensor the optimiser sums
# (SlabGrad.split; ogl_adam_step_multi_slabs2).  "auto" (default): when the layer has NO input gradient to compute — the first layer of
# 'mean', whose weight gradients are all that is left of its backward: 0.4208 -> 0.3867 ms per step, same box (relu_bwd_img + one
# k-major product where relu_bwd + a transposed image + two products + two reduction launches ran); with an input gradient on the
# critical path beside it ('meanpool') the one big product is in the way: 1.2057 -> 1.2172 ms.  OGL_DUAL_DW_CAT=1 / 0: always / never.
DUAL_DW_CAT_MODE = "auto"       # (round 6, after k_seg_groups: "1" for 'meanpool' 1.0406-1.0435 against 1.0344-1.0378 ms, same box: still in the way)
DUAL_DW_CAT = DUAL_DW_CAT_MODE != "0"


def linear_bwd_weight_x3k_dual(dy_img, x_img, x_rows, x_nrows, M, K1, x2_img, K2):
    """Slabs of [dw1 | db | pad | dw2] = dy^T . [x[x_rows] | 1 | x2]: returns (ws, stride, ws_ld, nsplit, col2, N) or None when the plan
    has a single split.  ``dy_img``: row-major image of dy [M, N]; ``x_img``: row-major image of x with the ones slot (K1 + 1);
    ``x2_img``: row-major image of x2 [M, K2]."""
    N = dy_img.K
    assert dy_img.rows == M and x_img.K == K1 + 1 and x2_img.K == K2 and x2_img.rows >= M
    nbytes = int(_lib.lib().ogl_linear_bwd_weight_x3k_dual_workspace_bytes(M, N, K1, 1, K2))
    if nbytes <= 0:
        return None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy_img.buf.device)
    ns, wl, c2 = C.c_int(0), C.c_int64(0), C.c_int(0)
    rc = _lib.lib().ogl_linear_bwd_weight_x3k_dual_slabs
    _launch("ogl_linear_bwd_weight_x3k", rc, _ptr(dy_img.buf), M, N, _ptr(x_img.buf), x_img.rows,
            _ptr(_ids(x_rows) if x_rows is not None else None), x_img.rows if x_nrows is None else x_nrows, K1, 1, _ptr(x2_img.buf),
            x2_img.rows, K2, _ptr(ws), nbytes, C.byref(ns), C.byref(wl), C.byref(c2), _stream(), meta=dict(M=M, K=K1 + K2, N=N, dual=True))
    return ws, N * wl.value, wl.value, ns.value, c2.value, N


class deferred_splitk:
    """Context: backward passes inside it may leave split-K slabs for ``optimizer`` (an ``optim.Adam``; None / any other optimiser:
    nothing is deferred).  On exit every slab set still pending is reduced into its parameter's ``.grad``."""

    def __init__(self, optimizer=None):
        self.opt = optimizer
        self.on = SLAB_ADAM and optimizer is not None and getattr(optimizer, "consumes_slabs", False)

    def __enter__(self):
        self.prev = _SLABS["on"]
        _SLABS["on"] = bool(self.on)
        return self

    def __exit__(self, *exc):
        _SLABS["on"] = self.prev
        pend = _SLABS["pending"]
        if pend and self.on:
            for group in self.opt.param_groups:
                for p in group["params"]:
                    sg = pend.pop(p.data_ptr(), None)
                    if sg is not None and p.grad is not None and exc[0] is None:
                        if p.grad.is_contiguous():
                            slab_reduce(sg, p.grad)
                        else:                                      # (reduced into a contiguous temporary, then copied into place)
                            tmp = torch.empty(p.grad.shape, dtype=torch.float32, device=p.grad.device)
                            slab_reduce(sg, tmp)
                            p.grad.copy_(tmp)
            pend.clear()
        return False


def linear_bwd_weight_x3k(dyT_img, x_img, M, K, x_rows=None, x_nrows=None, interleave=0, want_bias=True, want_bias2=False, dy_rows=False,
                          dw_out=None, defer_for=None):
    """dw [N, K] (and db, db2: two copies of the bias gradient) from the image of dy.T and the ROW-MAJOR image of x (M reduction
    rows, gathered by ``x_rows``): no transposed image of x.  The bias gradient needs the ones slot in ``x_img`` (K + 1).
    ``dy_rows``: ``dyT_img`` is the row-major image of dy itself ([M, N]: what relu_bwd_img / x3_split build), read k-major too.
    ``defer_for = (w, b, b2)``: the PARAMETERS dw / db / db2 are the gradients of (b / b2 None where that gradient is not asked for):
    inside ``deferred_splitk`` the split-K reduction is left to the optimiser — the returned tensors are then UNWRITTEN until it ran."""
    has_ones = x_img.K == K + 1
    assert x_img.K in (K, K + 1) and (has_ones or not (want_bias or want_bias2))
    if dy_rows:
        assert dyT_img.rows == M and not interleave
        N, interleave = dyT_img.K, -1
    else:
        N = dyT_img.rows
        assert dyT_img.K == (32 * interleave if interleave else M), (dyT_img.K, M, interleave)
    dev = dyT_img.buf.device
    dw = dw_out if dw_out is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    db2 = torch.empty(N, dtype=torch.float32, device=dev) if want_bias2 else None
    nbytes = int(_lib.lib().ogl_linear_bwd_weight_x3k_workspace_bytes(M, interleave, N, K, 1 if has_ones else 0))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    if (defer_for is not None and _SLABS["on"] and dw_out is None and defer_for[0] is not None and tuple(defer_for[0].shape) == (N, K)
            and (db is None or defer_for[1] is not None) and (db2 is None or defer_for[2] is not None)
            # (whole PARAMETERS only: a view of one — the column slices of a concat -> Linear weight — shares its address)
            and all(t is None or (t.is_leaf and t.is_contiguous()) for t in defer_for)
            # (one product per parameter and backward pass, nothing accumulated yet: see _slabs_settle)
            and not _slabs_settle(defer_for)):
        ns, wl = C.c_int(0), C.c_int64(0)
        _launch("ogl_linear_bwd_weight_x3k", _lib.lib().ogl_linear_bwd_weight_x3k_slabs, _ptr(dyT_img.buf), interleave, _ptr(x_img.buf),
                x_img.rows, _ptr(_ids(x_rows) if x_rows is not None else None), x_img.rows if x_nrows is None else x_nrows, M, N, K,
                1 if has_ones else 0, _ptr(dw), _ld(dw), _ptr(db), _ptr(db2), _ptr(ws), nbytes, C.byref(ns), C.byref(wl), _stream(),
                meta=dict(M=M, K=K, N=N, deferred=True))
        if ns.value > 1:
            pend, stride = _SLABS["pending"], N * wl.value
            pend[defer_for[0].data_ptr()] = SlabGrad(ws, stride, wl.value, ns.value, N, K, 0, dw)
            if db is not None:
                pend[defer_for[1].data_ptr()] = SlabGrad(ws, stride, wl.value, ns.value, N, 1, K, db)
            if db2 is not None:
                pend[defer_for[2].data_ptr()] = SlabGrad(ws, stride, wl.value, ns.value, N, 1, K, db2)
        return dw, db, db2
    _launch("ogl_linear_bwd_weight_x3k", _lib.lib().ogl_linear_bwd_weight_x3k, _ptr(dyT_img.buf), interleave, _ptr(x_img.buf),
            x_img.rows, _ptr(_ids(x_rows) if x_rows is not None else None), x_img.rows if x_nrows is None else x_nrows, M, N, K,
            1 if has_ones else 0, _ptr(dw), _ld(dw), _ptr(db), _ptr(db2), _ptr(ws), nbytes, _stream(), meta=dict(M=M, K=K, N=N))
    return dw, db, db2


# Static tables (the resident feature table) are split ONCE; projections that gather their rows from such a table and
# are large enough run on the pre-split kernels.  Keyed by the allocation, so row-prefix views (ndata['feat'] of a
# snapshot) resolve to the same image.
_X3_TABLES = {}
BWD_INPUT_VIA_FWD_MIN_ROWS = 2048       # input gradients with at least this many rows go through transpose(W) + forward kernel
X3_BWW_MIN_ROWS = 2048    # weight gradients with at least this many reduction rows use the image kernels
X3_MIN_ROWS = 8192        # below this the on-the-fly kernel is as fast (one wave of tiles either way)


def register_static_table(table):
    """Declare ``table`` (a [rows, cols] matrix whose contents never change) as a projection input: its bf16x3 image
    (+ the bias slot) is built lazily on first use and reused by every forward pass."""
    table = as_mat(table)
    _X3_TABLES[(table.data_ptr(), _ld(table), table.shape[1])] = [table, None]


def _static_key(x):
    return (x.data_ptr(), _ld(x), x.shape[1]) if x.dim() == 2 else None


def _static_image(x):
    ent = _X3_TABLES.get((x.data_ptr(), _ld(x), x.shape[1])) if x.dim() == 2 else None
    if ent is None or x.shape[0] > ent[0].shape[0]:
        return None
    if ent[1] is None:
        ent[1] = x3_split(ent[0], append_ones=True)
    return ent[1]


def _x3_forward_ok(x, M, x2):
    return (_MODE["name"] != "f32" and x2 is None and M >= X3_MIN_ROWS and x.is_cuda and x.dim() == 2
            and (x.data_ptr(), _ld(x), x.shape[1]) in _X3_TABLES)


K_MAJOR_WEIGHT_GRADS = os.environ.get("OGL_BWW_KMAJOR") != "0"    # weight gradients read the activations' row-major images as they are


def _row_image_for(x, x_rows, x_img):
    """The row-major bf16x3 image the weight gradient of a projection of ``x[x_rows]`` can read: the one its forward consumed
    (``x_img``), or the resident table's."""
    if not K_MAJOR_WEIGHT_GRADS:
        return None
    img = x_img
    if img is None and x_rows is not None and _static_key(x) in _X3_TABLES:
        img = _static_image(x)
    # (the k-major product addresses its images with 32-bit offsets: a table image of 4 GB and more keeps the transposed-image form)
    return img if (img is not None and img.nbytes < (1 << 32)) else None


K_MAJOR_DY = True    # ... and dy's own row-major image when its producer wrote one


def _dy_rows_image(dy, dy_img):
    """``dy_img`` if it is the row-major image of ``dy`` and the k-major weight gradient may use it (no transposed image of dy)."""
    if K_MAJOR_WEIGHT_GRADS and K_MAJOR_DY and dy_img is not None and dy_img.rows == dy.shape[0] and dy_img.K == dy.shape[1]:
        return dy_img
    return None


BWW_DIRECT_MAX_ROWS = 1024   # below this many reduction rows: the direct k-major kernel


def weight_grad(dy, x, x_rows=None, want_bias=True, dyT=None, x_img=None, dy_img=None, dw_out=None, defer_for=None):
    """dW, db of a projection.  In the bf16x6 / auto arithmetic the product runs on the split-bf16 image kernel: x as the row-major
    image its forward already had (``x_img``, or the resident table's: read k-major, no transposed copy), dy likewise when its
    producer wrote its image (``dy_img``), else as the image of dy^T; without a row-major image of x both operands are transposed
    images; the exact-fp32 mode keeps the direct k-major kernel.  ``dyT`` lets the two weight gradients of a dual-input Linear
    share one transpose."""
    if _MODE["name"] == "f32" or dy.shape[0] < BWW_DIRECT_MAX_ROWS:
        return linear_bwd_weight(dy, x, None, x_rows, want_bias=want_bias, dw_out=dw_out)
    K = x.shape[1]
    rimg = _row_image_for(x, x_rows, x_img) if dy.shape[0] >= X3_BWW_MIN_ROWS else None
    if rimg is not None and not (rimg.K == K + 1 or (rimg.K == K and not want_bias)):
        rimg = None
    dyr = _dy_rows_image(dy, dy_img) if rimg is not None else None
    if dyr is not None:
        return linear_bwd_weight_x3k(dyr, rimg, dy.shape[0], K, x_rows=x_rows, x_nrows=x.shape[0] if x_rows is not None else None,
                                     want_bias=want_bias, dy_rows=True, dw_out=dw_out, defer_for=defer_for)[:2]
    if dyT is None:
        dyT = transposed_operand(dy)
    if isinstance(dyT, X3Image):
        if rimg is not None:
            return linear_bwd_weight_x3k(dyT, rimg, dy.shape[0], K, x_rows=x_rows, x_nrows=x.shape[0] if x_rows is not None else None,
                                         want_bias=want_bias, dw_out=dw_out, defer_for=defer_for)[:2]
    if isinstance(dyT, X3Image):
        # both operands as bf16x3 images of their transposes (one fused gather + transpose + split pass each)
        return linear_bwd_weight_x3(dyT, x3_split_t(x, x_rows, ones_row=True), want_bias=want_bias, dw_out=dw_out)
    return linear_bwd_weight_t(dyT, transpose(x, x_rows), want_bias=want_bias, dw_out=dw_out)


def transposed_operand(dy):
    """dy^T in the form the weight-gradient product of this size consumes: a bf16x3 image (>= X3_BWW_MIN_ROWS reduction
    rows) or an fp32 matrix.  Shared by the two weight gradients of a dual-input Linear."""
    return x3_split_t(dy) if dy.shape[0] >= X3_BWW_MIN_ROWS else transpose(dy)


# --------------------------------------------------------------------------------------------
# loss / optimiser
# --------------------------------------------------------------------------------------------
def ce_fwd_bwd(logits, labels, grad_scale=1.0, want_grad=True):
    logits = as_mat(logits)
    labels = labels.reshape(-1)
    assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous() and labels.numel() == logits.shape[0]
    B, Cc = logits.shape
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    dl = empty_mat(B, Cc, logits.device) if want_grad else None
    _launch("ogl_ce_fwd_bwd", _lib.lib().ogl_ce_fwd_bwd, _ptr(logits), _ld(logits), _ptr(labels), B, Cc, C.c_float(grad_scale), _ptr(loss),
                                    _ptr(dl), _ld(dl) if dl is not None else 0, _stream(), meta=dict(B=B, C=Cc))
    return loss, dl


CE_MEAN_SMALL_MAX_B = 128      # up to here the mean comes from the cross-entropy launch itself (one workgroup)


CE_SMALL_MAX_ZERO = 65536       # floats the one-workgroup loss launch clears on the side (ogl_ce_fwd_bwd_mean_gather)
SMALL_LOSS_FUSED = True    # the last small 'pool' layer + its loss + the dlogits-only gradients: ONE launch
SMALL_LOSS_ZERO_MAX = 1 << 23   # floats the fused small output layer + loss launch clears on the side (ogl_small_pool_layer_fwd_ce_bwd)


def ce_fwd_bwd_mean(logits, labels, want_grad=True):
    """(mean loss [scalar tensor], row losses, dlogits scaled by 1/B) of a small batch in ONE launch; ``labels`` may be a LazyLabels
    (gathered inside the launch), and a small pending ``request_zeroed`` buffer is cleared by the same launch."""
    logits = as_mat(logits)
    lazy = labels if isinstance(labels, LazyLabels) else None
    if lazy is None:
        labels = labels.reshape(-1)
        assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous()
    assert labels.numel() == logits.shape[0]
    B, Cc = logits.shape
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    mean = torch.empty((), dtype=torch.float32, device=logits.device)
    dl = empty_mat(B, Cc, logits.device) if want_grad else None
    zbuf, zn = None, 0
    if want_grad:
        key = (logits.device.index, _stream())
        ent = _PENDING_ZERO.get(key)
        if ent is not None and ent[0].numel() <= CE_SMALL_MAX_ZERO and ent[0].numel() % 4 == 0:
            del _PENDING_ZERO[key]
            zbuf, zn = ent[0], ent[0].numel()
            ent[3] = True
    table, ids = (lazy.table, lazy.ids) if lazy is not None else (labels, None)
    prime = _ADAM_PRIME.get("req")
    if prime is not None and ADAM_PRIME_IN_SPLIT and want_grad:
        # a step without a weight-image launch (the 32-seed rungs): the optimiser's per-step scalars ride in the loss launch instead
        step_dev, scal, lr, b1, b2 = prime
        _launch("ogl_ce_fwd_bwd_mean", _lib.lib().ogl_ce_fwd_bwd_mean_gather, _ptr(logits), _ld(logits), _ptr(table), table.numel(),
                _ptr(ids), B, Cc, C.c_float(1.0 / B), _ptr(loss), _ptr(dl), _ld(dl) if dl is not None else 0, _ptr(mean), _ptr(zbuf), zn,
                _ptr(step_dev), _ptr(scal), C.c_double(lr), C.c_double(b1), C.c_double(b2), _stream(), meta=dict(B=B, C=Cc, adam_prepare=True))
        _ADAM_PRIME["req"], _ADAM_PRIME["served"] = None, (step_dev.data_ptr(), _capturing())
        return mean, loss, dl
    _launch("ogl_ce_fwd_bwd_mean", _lib.lib().ogl_ce_fwd_bwd_mean_gather, _ptr(logits), _ld(logits), _ptr(table), table.numel(), _ptr(ids), B, Cc,
            C.c_float(1.0 / B), _ptr(loss), _ptr(dl), _ld(dl) if dl is not None else 0, _ptr(mean), _ptr(zbuf), zn, None, None,
            C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), _stream(), meta=dict(B=B, C=Cc))
    return mean, loss, dl


# The grid form (any batch size): its block counter is one zeroed word per (device, stream) — the last block resets it — and it can
# zero a buffer on the side: a layer whose backward scatters with atomics parks its (empty) scatter target here in forward
# (``request_zeroed``), the loss kernel that runs between forward and backward clears it, and the layer's backward finds it
# zeroed (``take_zeroed``) instead of launching a fill of its own.
_CE_COUNTERS = {}
_PENDING_ZERO = {}          # (device index, stream) -> the newest request not served yet (an older one is simply dropped)


def request_zeroed(rows, cols, device):
    """An EMPTY [rows, cols] matrix (padded rows) that the next cross-entropy launch on this device will zero; returns a handle
    for ``take_zeroed``."""
    buf = torch.empty((max(rows, 1), padded_ld(cols)), dtype=torch.float32, device=device)
    ent = [buf, rows, cols, False]
    _PENDING_ZERO[(buf.device.index, _stream())] = ent
    return ent


def take_zeroed(ent, rows, cols):
    """The matrix of ``request_zeroed`` as a zeroed [rows, cols] view (zeroed here when no loss launch picked it up)."""
    buf, r, c, done = ent
    assert r == rows and c == cols
    if not done:
        fill_zero(buf)
        for k, v in list(_PENDING_ZERO.items()):
            if v is ent:
                del _PENDING_ZERO[k]
    return buf[:rows, :cols]


def ce_counter(device, stream):
    """Address of the block counter of (device, stream): one word of a zeroed 64-word array per device, allocated on first use
    (stepgraph allocates it before a capture so that it never lives in a graph's private pool)."""
    ent = _CE_COUNTERS.get(device.index)
    if ent is None:
        ent = _CE_COUNTERS[device.index] = (torch.zeros(64, dtype=torch.int32, device=device), {})
    arr, slots = ent
    i = slots.get(stream)
    if i is None:
        i = slots[stream] = len(slots) % 64
    return arr.data_ptr() + 4 * i


class LazyLabels:
    """``table[ids]`` not gathered yet: the losses that can (``ce_fwd_bwd_mean_grid``) read the labels through the ids inside their own
    launch — graph.ndata['target'][seeds] (R/train/graphsage/pytorch/model.py:91,183) without a launch; every other consumer calls
    ``materialize()`` (= ``gather_i64``)."""
    __slots__ = ("table", "ids", "_out")

    def __init__(self, table, ids):
        table = table.reshape(-1)
        assert table.dtype == torch.int64 and table.is_cuda and table.is_contiguous()
        self.table, self.ids, self._out = table, _ids(ids), None

    def numel(self):
        return self.ids.numel()

    def materialize(self):
        if self._out is None:
            self._out = gather_i64(self.table, self.ids)
        return self._out


def _labels_tensor(labels):
    return labels.materialize() if isinstance(labels, LazyLabels) else labels


def ce_fwd_bwd_mean_grid(logits, labels, want_grad=True):
    """(mean loss, row losses, dlogits / B) of a batch of any size in ONE launch (ogl_ce_fwd_bwd_mean_grid; ``labels`` may be a
    LazyLabels: the gather then happens inside the launch)."""
    logits = as_mat(logits)
    lazy = labels if isinstance(labels, LazyLabels) else None
    if lazy is None:
        labels = labels.reshape(-1)
        assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous()
    assert labels.numel() == logits.shape[0]
    B, Cc = logits.shape
    dev = logits.device
    stream = _stream()
    ctr = ce_counter(dev, stream)
    loss = torch.empty(B, dtype=torch.float32, device=dev)
    mean = torch.empty((), dtype=torch.float32, device=dev)
    dl = empty_mat(B, Cc, dev) if want_grad else None
    zbuf, zn = None, 0
    if want_grad:
        ent = _PENDING_ZERO.pop((dev.index, stream), None)
        if ent is not None:
            zbuf, zn = ent[0], ent[0].numel()
            ent[3] = True
    if lazy is not None:
        _launch("ogl_ce_fwd_bwd_mean_grid", _lib.lib().ogl_ce_fwd_bwd_mean_grid, _ptr(logits), _ld(logits), _ptr(lazy.table),
                lazy.table.numel(), _ptr(lazy.ids), B, Cc, C.c_float(1.0 / B), _ptr(loss), _ptr(dl), _ld(dl) if dl is not None else 0,
                _ptr(mean), ctr, _ptr(zbuf), zn, _stream(), meta=dict(B=B, C=Cc))
        return mean, loss, dl
    _launch("ogl_ce_fwd_bwd_mean_grid", _lib.lib().ogl_ce_fwd_bwd_mean_grid, _ptr(logits), _ld(logits), _ptr(labels), 0, None, B, Cc,
            C.c_float(1.0 / B), _ptr(loss), _ptr(dl), _ld(dl) if dl is not None else 0, _ptr(mean), ctr, _ptr(zbuf), zn,
            _stream(), meta=dict(B=B, C=Cc))
    return mean, loss, dl


def argmax_confusion(logits, labels=None, confusion=None, want_pred=True):
    """pred = argmax over classes; ``confusion`` (int64 [C, C], accumulated in place) counts (true, pred) pairs."""
    logits = as_mat(logits)
    B, Cc = logits.shape
    pred = torch.empty(B, dtype=torch.int64, device=logits.device) if want_pred else None
    if labels is not None:
        labels = labels.reshape(-1)
        assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous() and labels.numel() == B
    if confusion is not None:
        assert confusion.dtype == torch.int64 and confusion.is_cuda and confusion.is_contiguous() and confusion.numel() == Cc * Cc
    _launch("ogl_argmax_confusion", _lib.lib().ogl_argmax_confusion, _ptr(logits), _ld(logits), _ptr(labels), B, Cc, _ptr(pred),
            _ptr(confusion), _stream(), meta=dict(B=B, C=Cc))
    return pred


def adam_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    invalidate_weight_images()      # parameters change under raw pointers: no version bump to key on
    for t in (p, g, m, v):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    _launch("ogl_adam_step", _lib.lib().ogl_adam_step, _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(step), C.c_double(lr),
                                   C.c_double(beta1), C.c_double(beta2), C.c_double(eps), _stream(), meta=dict(n=p.numel()))


def adam_step_multi(ps, gs, ms, vs, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """One launch for every parameter tensor (same step count for all)."""
    adam_step_multi_slabs(ps, gs, ms, vs, [None] * len(ps), step=step, lr=lr, beta1=beta1, beta2=beta2, eps=eps)


def adam_step_multi_dev(ps, gs, ms, vs, step_dev, scalars_dev, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """adam_step_multi with the step count in device memory (``step_dev`` int64[1], incremented by the call on the device;
    ``scalars_dev`` float32[2] scratch): capturable in a hipGraph."""
    adam_step_multi_slabs(ps, gs, ms, vs, [None] * len(ps), step_dev=step_dev, scalars_dev=scalars_dev, prepare=True, lr=lr, beta1=beta1,
                          beta2=beta2, eps=eps)


def adam_step_multi_slabs(ps, gs, ms, vs, slabs, step=0, step_dev=None, scalars_dev=None, prepare=True, lr=1e-3, beta1=0.9, beta2=0.999,
                          eps=1e-8):
    """Adam over several tensors in one launch; ``slabs[i]`` (a SlabGrad or None): tensor i's gradient still is a set of split-K
    slabs — summed inside the launch, written to ``gs[i]`` and applied.  ``step_dev`` None: host step count ``step``; else the
    device-side count, incremented only when ``prepare`` (a step applied in two launches prepares once)."""
    invalidate_weight_images()      # parameters change under raw pointers: no version bump to key on
    k = len(ps)
    for p, g, m, v in zip(ps, gs, ms, vs):
        for t in (p, g, m, v):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() for t in ts])
    n = (C.c_int64 * k)(*[p.numel() for p in ps])
    for sg, p in zip(slabs, ps):
        assert sg is None or sg.rows * sg.ncols == p.numel()
    ws = (C.c_void_p * k)(*[(sg.ws.data_ptr() if sg is not None else None) for sg in slabs])
    stride = (C.c_int64 * k)(*[(sg.stride if sg is not None else 0) for sg in slabs])
    i32 = lambda f: (C.c_int * k)(*[(getattr(sg, f) if sg is not None else 0) for sg in slabs])
    if _SIDE["active"]:
        _SIDE["keep"].extend(sg.ws for sg in slabs if sg is not None)       # (read on the side stream: held until the join)
    _launch("ogl_adam_step_multi_slabs", _lib.lib().ogl_adam_step_multi_slabs, k, arr(ps), arr(gs), arr(ms), arr(vs), n, ws, stride,
            i32("ws_ld"), i32("nsplit"), i32("ncols"), i32("col0"), i32("split"), i32("col0b"), int(step), _ptr(step_dev), _ptr(scalars_dev),
            1 if prepare else 0,
            C.c_double(lr), C.c_double(beta1), C.c_double(beta2), C.c_double(eps), _stream(),
            meta=dict(n=sum(p.numel() for p in ps), slab_tensors=sum(sg is not None for sg in slabs)))


# --------------------------------------------------------------------------------------------
# autograd glue
# --------------------------------------------------------------------------------------------
# Gradient sinks (data parallelism): a parameter registered here by parallel.GradSynchronizer has a slot in a persistent flat
# all-reduce bucket; the weight-gradient kernels write dW straight into that slot (``dw_out``), autograd adopts the slot as
# ``p.grad``, and the collective runs in place on the bucket: no torch.cat, no copy back, no allocation per step.
_GRAD_SINKS = {}


def register_grad_sink(param, owner, index):
    import weakref
    _GRAD_SINKS[param.data_ptr()] = (weakref.ref(owner), int(index))


def _dw_out(w, N, K):
    """The gradient slot of weight ``w`` ([N, K], contiguous) or None (then the kernel wrapper allocates)."""
    if not _GRAD_SINKS or w is None:
        return None
    ent = _GRAD_SINKS.get(w.data_ptr())
    if ent is None:
        return None
    owner = ent[0]()
    if owner is None:
        _GRAD_SINKS.pop(w.data_ptr(), None)
        return None
    t = owner.grad_slot(ent[1])
    return t if (t is not None and tuple(t.shape) == (int(N), int(K)) and not _capturing()) else None


# Forked backward: the weight gradients of the n1-row projections are leaves of the backward graph — nothing but the optimiser
# reads them — while the input-gradient chain (dh1 -> relu mask -> dneigh0 -> pool backward -> dW_pool0) is the critical path.
# ``side_section()`` runs its body on a second HIP stream that waits for everything enqueued so far; ``side_join()`` (before the
# optimiser step) makes the main stream wait for it.  In a captured step the two become parallel branches of the hipGraph.
# Tensors the side work reads were allocated on the main stream: ``side_keep`` holds them until the join, so the caching allocator
# cannot hand their memory to a later main-stream kernel while the side stream still reads it.
FORK_BACKWARD = os.environ.get("OGL_FORK_BWD", "1") != "0"
# in a captured step the fork becomes parallel graph branches (with the fork points placed where the side work's inputs are ready:
# 1.10 -> 1.05-1.07 ms per replayed Reddit step, the eager figure; a first placement that forked late was 1 % slower than serial)
FORK_IN_GRAPHS = True
_SIDE = {"streams": {}, "keep": [], "active": False, "off": 0}


def fork_point():
    """Marks the place in the main stream a later ``side_section(at=...)`` has to wait for — everything enqueued up to HERE,
    not up to where the section is opened: the section's work can then run beside what the main stream enqueues in between.
    None when the fork is off."""
    if not FORK_BACKWARD or _PROFILE is not None or _SIDE["off"] or (not FORK_IN_GRAPHS and _capturing()):
        return None
    ev = torch.cuda.Event()
    ev.record()
    return ev


class _SideSection:
    def __init__(self, at=None):
        self._at = at

    def __enter__(self):
        dev = torch.cuda.current_device()
        st = _SIDE["streams"].get(dev)
        if st is None:
            st = _SIDE["streams"][dev] = torch.cuda.Stream(device=dev)
        if self._at is not None:
            st.wait_event(self._at)
        else:
            st.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(st)
        self._ctx.__enter__()
        if not _SIDE["active"]:
            # joined when the backward pass that opened the section ends, whoever called it (loss.backward() of user code too)
            try:
                torch.autograd.Variable._execution_engine.queue_callback(side_join)
            except RuntimeError:
                pass            # (opened outside a backward pass — deferred work flushed by the optimiser's step(), which joins itself)
        _SIDE["active"] = True
        return self

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        return False


class _NoSection:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def side_section(*keep, at=None):
    """Context manager: the body's launches go to the side stream, ordered after everything the main stream had enqueued at
    ``at`` (a ``fork_point()``; default: so far).  A no-op context when the fork is off: under per-kernel profiling, data
    OGL_FORK_BWD=0.  (Data parallelism: the gradient hooks launch their bucket's collective FROM the side stream,
    ``collective_section``, so the main stream's input-gradient chain is never made to wait for a weight gradient.)"""
    if not FORK_BACKWARD or _PROFILE is not None or _SIDE["off"] or (not FORK_IN_GRAPHS and _capturing()):
        return _NoSection()
    _SIDE["keep"].extend(t for t in keep if t is not None)
    return _SideSection(at)


def collective_section():
    """Context for launching a gradient bucket's all-reduce in the middle of a forked backward: when side work is outstanding the
    launch (and the bucket fill before it) happens on the side stream, after it has also caught up with the main stream — RCCL's
    own stream then waits for the weight gradients without the main stream waiting for anything.  Otherwise a no-op context."""
    if not _SIDE["active"]:
        return _NoSection()
    return _SideSection(None)


def early_section(at=None):
    """Context for work launched from a gradient hook in the middle of a backward pass that only the END of the step needs (the
    optimiser's early part): on the side stream — behind its own queue and behind the main stream up to ``at`` (a ``fork_point()``;
    default: up to now) — when the fork is on; otherwise a no-op context (the work then sits in the main stream)."""
    if not FORK_BACKWARD or _PROFILE is not None or _SIDE["off"] or (not FORK_IN_GRAPHS and _capturing()):
        return _NoSection()
    return _SideSection(at)


def side_join():
    """The main stream waits for the side stream's work (call before anything consumes what a side section produced)."""
    if _SIDE["active"]:
        st = _SIDE["streams"].get(torch.cuda.current_device())
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)
        _SIDE["active"] = False
    _SIDE["keep"].clear()


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, x2, w2, relu, x_rows, x2_rows, bias2=None, split=None):
        # bias2: the second projection's own bias (a layer with fc_self.bias and fc_neigh.bias): summed here, and each
        # gets its gradient from its own weight-gradient product in backward (their ones columns are free) — a tracked
        # `bias + bias2` outside would hand ONE gradient tensor to two parameters, which autograd clones (a launch)
        # split: `w` is the weight of a concat -> Linear, [N, split + K2] (fc_neigh(cat(h_self, h_neigh)) of the in-repo layer,
        # aggregator_dgl.py:206): its two column blocks are the two projections.  Sliced HERE — as autograd views outside, each
        # block's gradient would pass through a SliceBackward node (zero fill + copy of the whole [N, K1 + K2] matrix, twice, + an add:
        # ATen launches on the main stream that READ the weight gradients while the forked backward is still computing them on the
        # side stream: layers.0.fc_neigh.weight.grad came out zero / garbage in its neighbour half at the Reddit shape).
        ctx.split = None if split is None else int(split)
        w_full = w
        if ctx.split is not None:
            assert w2 is None and x2 is not None and 0 < ctx.split < w.shape[1]
            w, w2 = w_full[:, :ctx.split], w_full[:, ctx.split:]
        ctx.has_bias2 = bias2 is not None
        ctx.bias_t, ctx.bias2_t = bias, bias2                    # (the parameters themselves: who a deferred gradient belongs to)
        ctx.x2_img = take_image(x2, pop=False) if x2 is not None else None     # read again by the weight gradient (k-major)
        y = _dual_fwd_images(x, w, bias, bias2, x2, w2, relu, x_rows, x2_rows)
        if y is None:
            # (bias2 comes with x2, and a dual-input product never takes linear_fwd's table-image path: the two biases are summed
            # inside the launch — ogl_linear_fwd_dual_bias — instead of by an ATen add in front of it)
            y = linear_fwd(x, w, bias, x2, w2, relu, x_rows, x2_rows, bias2=bias2)
        ctx.relu = bool(relu)
        ctx.has_bias = bias is not None
        if relu:
            y._ogl_relu_out = True               # (read by the consumer layer: its input gradient may come back pre-masked)
        if ctx.split is not None:
            ctx.save_for_backward(x, w_full, x2, None, y if relu else None, x_rows, x2_rows)
        else:
            ctx.save_for_backward(x, w, x2, w2, y if relu else None, x_rows, x2_rows)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, x2, w2, y, x_rows, x2_rows = ctx.saved_tensors
        dy = as_mat(dy)
        need = ctx.needs_input_grad
        dw_cat, w_leaf = None, w.is_leaf and (w2 is None or w2.is_leaf)
        ctx.dw_views = None
        if ctx.split is not None:
            w_full = w
            w, w2 = w_full[:, :ctx.split], w_full[:, ctx.split:]
            need = (need[0], need[1], need[2], need[3], need[1])
            if need[1]:
                # ONE gradient tensor for the concat weight: each block's product writes its columns (strided rows)
                dw_cat = torch.empty_like(w_full, memory_format=torch.contiguous_format)
                ctx.dw_views = (dw_cat[:, :ctx.split], dw_cat[:, ctx.split:])
                ctx.w_full_t, ctx.dw_cat_t = w_full, dw_cat
        dy_img = None
        if y is not None and getattr(dy, "_ogl_premasked", None) == (y.data_ptr(), y._version):
            # the product that computed dy already applied [y > 0] in its epilogue and wrote the image (linear_bwd_input)
            dy_img = take_image(dy)
            y = None
        if y is not None:
            # once; the (up to four) backward GEMMs below are mask-free.  Tall products with an input gradient to compute get
            # the image of the masked gradient from the same pass
            # (... or when both weight gradients can run as ONE k-major product over dy's row-major image: the dual product)
            cat_ok = DUAL_DW_CAT_MODE == "1" or (DUAL_DW_CAT_MODE == "auto" and not need[0] and not (x2 is not None and need[3]))
            ctx.dual_cat_ok = cat_ok
            dual_img = (DUAL_DW and x2 is not None and ctx.x2_img is not None and x2_rows is None and need[1] and dy.shape[0] >= X3_BWW_MIN_ROWS
                        and (ctx.split is None or cat_ok))
            if (need[0] or (x2 is not None and need[3]) or dual_img) and _n1_images_ok(dy.shape[0], dy.shape[1], w.shape[1]):
                dy, dy_img = relu_bwd_img(dy, y)
            else:
                dy = relu_bwd(dy, y)
            y = None
        dx = dw = db = dx2 = dw2 = db2 = None
        if need[0]:
            if x_rows is not None:
                raise RuntimeError("gradient w.r.t. a row-gathered table is not supported (features carry no grad)")
            dx = linear_bwd_input(dy, w, y, dy_img=dy_img)
        # (the weight gradients go to the side stream only when nothing but the optimiser reads them: parameters themselves — a
        # weight that is an autograd VIEW of one has a backward node behind it that reads the gradient on the main stream at once)
        forked = dy_img is not None and dy.shape[0] >= X3_BWW_MIN_ROWS and w_leaf
        at = fork_point() if forked else None                  # dy and its image are ready here
        if x2 is not None and need[3]:
            # the input gradients first: they are the critical path of the backward pass (the weight gradients below are leaves)
            if x2_rows is not None:
                raise RuntimeError("gradient w.r.t. a row-gathered table is not supported")
            dx2 = linear_bwd_input(dy, w2, y, dy_img=dy_img)
        with (side_section(dy, dy_img, x, x2, ctx.x2_img, at=at) if forked else _NoSection()):
            dw, db, dw2, db2 = _LinearFn._weight_grads(ctx, dy, dy_img, x, w, x2, w2, x_rows, x2_rows, need)
        # (no reference to the concat gradient but the one returned: AccumulateGrad adopts a gradient tensor only when nobody else
        # holds it — the two column views do, through their base — and otherwise CLONES it, on the main stream, while a forked
        # backward is still writing it on the side stream)
        ctx.dw_views = None
        ctx.w_full_t = ctx.dw_cat_t = None
        if ctx.split is not None:
            return dx, dw_cat, (db if ctx.has_bias else None), dx2, None, None, None, None, None, None
        return dx, dw, (db if ctx.has_bias else None), dx2, dw2, None, None, None, (db2 if ctx.has_bias2 else None), None

    @staticmethod
    def _weight_grads(ctx, dy, dy_img, x, w, x2, w2, x_rows, x2_rows, need):
        dw = db = dw2 = db2 = None
        dyT = None
        views = getattr(ctx, "dw_views", None)
        _dw_out_mod = globals()["_dw_out"]

        def _dw_out(t, *_shape):          # (shadows the module's: a concat weight's two blocks write into ONE gradient tensor)
            if views is not None:
                return views[0] if t is w else views[1]
            return globals()["_dw_out"](t, *_shape)
        x2_img = ctx.x2_img if (x2 is not None and x2_rows is None) else None
        tall = _MODE["name"] != "f32" and dy.shape[0] >= X3_BWW_MIN_ROWS
        rimg = _row_image_for(x, x_rows, None) if tall else None
        # every operand's row-major image at hand (dy's from the ReLU backward): the k-major product needs no transposed image
        dyr = _dy_rows_image(dy, dy_img) if (rimg is not None and x2_img is not None) else None
        if dyr is None and _MODE["name"] != "f32" and dy.shape[0] >= 1024 and x2 is not None and need[1] and need[4]:
            dyT = transposed_operand(dy)   # shared by the two weight gradients of a dual-input projection
        both = None
        if (DUAL_DW and dyr is not None and rimg is not None and rimg.K == x.shape[1] + 1 and x2_img is not None and x2_img.K == x2.shape[1]
                and x2_img.rows >= dy.shape[0] and need[1] and need[4] and rimg.buf.numel() < (1 << 32)):
            # ONE product for both weight gradients (round 5): dy^T . [x[rows] | 1 | x2] over a two-part B operand — the table's image
            # rows and the image the aggregator wrote, read where they lie
            K1, K2, N_ = x.shape[1], x2.shape[1], dy.shape[1]
            wf = getattr(ctx, "w_full_t", None)
            if (views is not None and getattr(ctx, "dual_cat_ok", False) and _SLABS["on"] and wf is not None and wf.is_leaf and wf.is_contiguous()
                    and _dw_out_mod(wf, *wf.shape) is None and (not ctx.has_bias or (ctx.bias_t is not None and ctx.bias_t.is_leaf))
                    and not _slabs_settle((wf, ctx.bias_t))):
                # the in-repo layer's ONE concat weight [N, K1 + K2]: its gradient stays in the slabs as a TWO-RANGE tensor the optimiser
                # sums (columns [0, K1) at slab column 0, [K1, K1 + K2) at col2): no reduction launch at all
                res = linear_bwd_weight_x3k_dual(dyr, rimg, x_rows, x.shape[0] if x_rows is not None else None, dy.shape[0], K1, x2_img, K2)
                if res is not None:
                    ws_, stride, wl, ns, c2, _ = res
                    pend = _SLABS["pending"]
                    dcat = ctx.dw_cat_t
                    pend[wf.data_ptr()] = SlabGrad(ws_, stride, wl, ns, N_, K1 + K2, 0, dcat, split=K1, col0b=c2)
                    db = None
                    if ctx.has_bias:
                        db = torch.empty(N_, dtype=torch.float32, device=dy.device)
                        pend[ctx.bias_t.data_ptr()] = SlabGrad(ws_, stride, wl, ns, N_, 1, K1, db)
                    return views[0], db, views[1], None
            elif views is not None:
                # (outside a slab-consuming optimiser step: the two column blocks summed out of the slabs by reduction launches)
                res = None if DUAL_DW_CAT_MODE != "1" else linear_bwd_weight_x3k_dual(dyr, rimg, x_rows, x.shape[0] if x_rows is not None else None, dy.shape[0], K1, x2_img, K2)
                if res is not None:
                    ws_, stride, wl, ns, c2, _ = res
                    slab_reduce(SlabGrad(ws_, stride, wl, ns, N_, K1, 0), views[0])
                    slab_reduce(SlabGrad(ws_, stride, wl, ns, N_, K2, c2), views[1])
                    db = None
                    if ctx.has_bias:
                        db = torch.empty(N_, dtype=torch.float32, device=dy.device)
                        slab_reduce(SlabGrad(ws_, stride, wl, ns, N_, 1, K1), db)
                    return views[0], db, views[1], None
            elif (_SLABS["on"] and ctx.has_bias and ctx.has_bias2 and globals()["_dw_out"](w, *w.shape) is None
                  and globals()["_dw_out"](w2, *w2.shape) is None
                  and all(t is not None and t.is_leaf and t.is_contiguous() for t in (w, w2, ctx.bias_t, ctx.bias2_t))
                  and tuple(w.shape) == (N_, K1) and tuple(w2.shape) == (N_, K2)
                  and not _slabs_settle((w, w2, ctx.bias_t, ctx.bias2_t))):
                res = linear_bwd_weight_x3k_dual(dyr, rimg, x_rows, x.shape[0] if x_rows is not None else None, dy.shape[0], K1, x2_img, K2)
                if res is not None:
                    ws_, stride, wl, ns, c2, _ = res
                    dev = dy.device
                    dw = torch.empty((N_, K1), dtype=torch.float32, device=dev)
                    dw2 = torch.empty((N_, K2), dtype=torch.float32, device=dev)
                    db = torch.empty(N_, dtype=torch.float32, device=dev)
                    db2 = torch.empty(N_, dtype=torch.float32, device=dev)
                    pend = _SLABS["pending"]
                    pend[w.data_ptr()] = SlabGrad(ws_, stride, wl, ns, N_, K1, 0, dw)
                    pend[w2.data_ptr()] = SlabGrad(ws_, stride, wl, ns, N_, K2, c2, dw2)
                    pend[ctx.bias_t.data_ptr()] = SlabGrad(ws_, stride, wl, ns, N_, 1, K1, db)
                    pend[ctx.bias2_t.data_ptr()] = SlabGrad(ws_, stride, wl, ns, N_, 1, K1, db2)
                    return dw, db, dw2, db2
        if ((dyr is not None or isinstance(dyT, X3Image)) and ctx.has_bias and ctx.has_bias2 and x2_img is not None
                and x2_img.K == x2.shape[1] and need[1] and need[4]):
            # the neighbour part's image has no ones slot: both copies of the bias gradient come from the first product
            if rimg is not None and rimg.K == x.shape[1] + 1:
                both = linear_bwd_weight_x3k(dyr if dyr is not None else dyT, rimg, dy.shape[0], x.shape[1], x_rows=x_rows,
                                             x_nrows=x.shape[0] if x_rows is not None else None, want_bias=True, want_bias2=True,
                                             dy_rows=dyr is not None, dw_out=_dw_out(w, *w.shape),
                                             defer_for=(w, ctx.bias_t, ctx.bias2_t))
        if (both is None and x2 is not None and x2_rows is None and need[1] and need[4] and ctx.has_bias and ctx.has_bias2
                and dy.shape[0] < 1024 and x.shape[1] == x2.shape[1] and _out_layer_fits(dy, x, w, w2)):
            # a short, narrow combine (the first layer at the 32-seed rungs): both weight gradients and both bias-gradient copies
            # from one launch instead of two latency-bound ones
            dw, dw2, db, db2 = out_layer_bwd_weights(dy, x, x2, want_bias=True, x_self_rows=x_rows, dws_out=_dw_out(w, *w.shape),
                                                     dwn_out=_dw_out(w2, *w2.shape))
            both = (dw, db, db2)
            fused_small = True
        else:
            fused_small = False
        if both is not None:
            dw, db, db2 = both
        elif need[1] or (need[2] and ctx.has_bias):
            dw, db = weight_grad(dy, x, x_rows, want_bias=ctx.has_bias, dyT=dyT, dy_img=dy_img, dw_out=_dw_out(w, *w.shape),
                                 defer_for=(w, ctx.bias_t, None))
        if x2 is not None:
            if fused_small:
                pass
            elif both is not None:
                dw2 = weight_grad(dy, x2, None, want_bias=False, dyT=dyT, x_img=x2_img, dy_img=dy_img, dw_out=_dw_out(w2, *w2.shape),
                                  defer_for=(w2, None, None))[0]
            elif need[4] or ctx.has_bias2:
                # (the pooled rows' own image, when their producer wrote one and no bias gradient has to come out of this product:
                # read k-major — no transposed images of dy and x2)
                x2i = x2_img if (x2_img is not None and not ctx.has_bias2 and x2_img.K == x2.shape[1]) else None
                dw2, db2 = weight_grad(dy, x2, x2_rows, want_bias=ctx.has_bias2, dyT=dyT, x_img=x2i, dy_img=dy_img,
                                       dw_out=_dw_out(w2, *w2.shape), defer_for=(w2, ctx.bias2_t, None))
        return dw, db, dw2, db2


def linear(x, w, bias=None, x2=None, w2=None, relu=False, x_rows=None, x2_rows=None, bias2=None):
    """y = act(x[rows] @ w.T (+ x2[rows2] @ w2.T) + bias (+ bias2)), differentiable."""
    if bias2 is not None and (bias is None or x2 is None):
        raise ValueError("bias2 belongs to the second projection of a dual-input Linear that has a first bias")
    return _LinearFn.apply(x, w, bias, x2, w2, relu, x_rows, x2_rows, bias2)


def linear_cat(x, x2, w, split, bias=None, relu=False, x_rows=None):
    """y = act(cat(x[rows], x2) @ w.T + bias) for w [N, split + K2] WITHOUT materialising the concatenation or slicing the
    parameter outside: fc_neigh(cat(h_self, h_neigh)) (R/train/graphsage/pytorch/aggregator_dgl.py:206) as one dual-input product
    whose backward returns ONE gradient tensor for w."""
    return _LinearFn.apply(x, w, bias, x2, None, relu, x_rows, None, None, int(split))


class _ReduceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx, op):
        need_grad = src.requires_grad
        if op == "mean" and _n1_images_ok(idx.shape[0], src.shape[1]):
            out, img = reduce_fwd_mean_img(src, idx)         # tall: the combine product reads the image of the pooled rows
            attach_image(out, img)
            argmax = None
        else:
            out, argmax = reduce_fwd(src, idx, op, want_argmax=need_grad)
        ctx.op, ctx.n_src, ctx.fanout = op, src.shape[0], idx.shape[1]
        ctx.seg_plan = None
        if need_grad and op in ("mean", "sum") and seg_bwd_fits(idx, src.shape[1], src.shape[0]):
            # the backward as a segmented gather over the edges sorted by source: the sort needs the indices only — planned here
            ctx.seg_plan = reduce_bwd_seg_plan(idx, src.shape[1], src.shape[0])
        ctx.save_for_backward(idx if idx.dtype == torch.int32 else None, argmax)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx32, argmax = ctx.saved_tensors
        if ctx.op == "max" and argmax is None:
            raise RuntimeError("max-reduce backward needs the argmax recorded in forward")
        if ctx.op != "max" and idx32 is None:
            raise RuntimeError("mean/sum-reduce backward needs block-local int32 indices")
        plan, ctx.seg_plan = getattr(ctx, "seg_plan", None), None
        if plan is not None:
            return reduce_bwd_seg_apply(dout, idx32, plan, ctx.op)[0], None, None
        return reduce_bwd(dout, idx32, argmax, ctx.op, ctx.n_src, fanout=ctx.fanout), None, None


class _PoolMeanFn(torch.autograd.Function):
    """relu(fc_pool(x[rows])) -> mean over the sampled neighbours, as one autograd node, for a projection input WITHOUT a gradient (the
    first layer of the in-repo 'meanpool' mode, R/train/graphsage/pytorch/aggregator_dgl.py:178-186): the pooled-row gradient dP has
    one consumer, fc_pool's weight gradient, and goes from dout straight to the row-major bf16x3 image that product reads —
    scaled, ReLU-masked, never materialised in fp32 (``reduce_bwd_seg_apply``); the edge sort it needs is planned by the forward."""

    @staticmethod
    def forward(ctx, x, w, bias, x_rows, idx):
        need = w.requires_grad or (bias is not None and bias.requires_grad)
        p = linear_fwd(x, w, bias, relu=True, x_rows=x_rows)
        if _n1_images_ok(idx.shape[0], p.shape[1]) and p.shape[1] % 4 == 0:
            out, img = reduce_fwd_mean_img(p, idx)
            attach_image(out, img)
        else:
            out, _ = reduce_fwd(p, idx, "mean")
        if _CAPTURE is not None:
            _CAPTURE.append(dict(pool_out=p))          # (test hook: the device's own ReLU decisions of the pooled projection)
        ctx.n_src, ctx.fanout, ctx.has_bias, ctx.bias_t = p.shape[0], idx.shape[1], bias is not None, bias
        ctx.seg_plan = reduce_bwd_seg_plan(idx, p.shape[1], p.shape[0], groups=SEG_T and p.shape[1] <= SEG_T_MAX_D) if need else None
        ctx.p_shape = tuple(p.shape)
        ctx.save_for_backward(x, w, x_rows, p, idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, x_rows, p, idx = ctx.saved_tensors
        plan, ctx.seg_plan = ctx.seg_plan, None
        if plan is None:                                  # (a second backward pass over a retained graph: the plan was consumed)
            plan = reduce_bwd_seg_plan(idx, ctx.p_shape[1], ctx.p_shape[0], side=False, groups=SEG_T and ctx.p_shape[1] <= SEG_T_MAX_D)
        K = x.shape[1]
        rimg = _row_image_for(x, x_rows, None) if (_MODE["name"] != "f32" and ctx.n_src >= X3_BWW_MIN_ROWS) else None
        if rimg is not None and rimg.K == K + 1:
            if getattr(plan, "groups", False) and x_rows is not None:
                # dP^T as the dealt group-major image the 'pool' mode's backward writes: the same 256 x 128 product (round 6)
                dyT = reduce_bwd_seg_apply_t(dout, idx, plan, "mean", mask=p)
                dw, db, _ = linear_bwd_weight_x3k(dyT, rimg, ctx.n_src, K, x_rows=x_rows, x_nrows=x.shape[0], interleave=(ctx.n_src + 31) // 32,
                                                  want_bias=ctx.has_bias, dw_out=_dw_out(w, *w.shape), defer_for=(w, ctx.bias_t, None))
                return None, dw, (db if ctx.has_bias else None), None, None
            # dP as the image the k-major weight gradient reads; x as the resident table's own image, gathered by the block's ids
            _, dp_img = reduce_bwd_seg_apply(dout, idx, plan, "mean", mask=p, want_out=False, want_image=True)
            dw, db, _ = linear_bwd_weight_x3k(dp_img, rimg, ctx.n_src, K, x_rows=x_rows, x_nrows=x.shape[0] if x_rows is not None else None,
                                              want_bias=ctx.has_bias, dy_rows=True, dw_out=_dw_out(w, *w.shape),
                                              defer_for=(w, ctx.bias_t, None))
        else:
            dp, _ = reduce_bwd_seg_apply(dout, idx, plan, "mean", mask=p)
            dw, db = weight_grad(dp, x, x_rows, want_bias=ctx.has_bias, dw_out=_dw_out(w, *w.shape))
        return None, dw, (db if ctx.has_bias else None), None, None


def pool_mean_fits(x, idx, pool_width, n_src):
    return (not x.requires_grad) and seg_bwd_fits(idx, pool_width, n_src)


def pool_mean(x, w, bias, idx, x_rows=None):
    """mean_j relu(fc_pool(x))[idx[:, j]] for an input without a gradient — the aggregator of the in-repo 'meanpool' first layer."""
    return _PoolMeanFn.apply(x, w, bias, x_rows, idx)


# Test hook: while a list is installed, every differentiable pool layer appends dict(argmax, neigh[, out]) — the winners and
# ReLU masks the device chose — so that a parity test can route the oracle's backward through the same winners
# (tests/test_gpu_fullsize.py); the in-repo 'meanpool' layers append dict(pool_out) (the ReLU'd pooled projection).  Not used by
# the product path.
_CAPTURE = None


def capture_pool_winners(store):
    global _CAPTURE
    _CAPTURE = store


class _PoolMaxFn(torch.autograd.Function):
    """relu(fc_pool(x[rows])) -> elementwise max over the sampled neighbours, as one autograd node.

    The projected rows P are transient: backward needs only the max output (its sign is the winner's ReLU mask,
    applied inside the scatter), the argmax and the projection input, so the [n_src, D] activation is freed right
    after the forward and neither backward GEMM reads a mask."""

    @staticmethod
    def forward(ctx, x, w, bias, x_rows, idx):
        p = linear_fwd(x, w, bias, relu=True, x_rows=x_rows)
        need = x.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad)
        ctx.n_src, ctx.fanout, ctx.has_bias = p.shape[0], idx.shape[1], bias is not None
        plan_ok = (POOL_PLAN and need and not x.requires_grad and idx.dtype == torch.int32 and _MODE["name"] != "f32"
                   and ctx.n_src >= X3_BWW_MIN_ROWS and p.shape[1] <= 640 and ctx.fanout <= 63 and idx.shape[0] * p.shape[1] < (1 << 27)
                   and (w.requires_grad or (bias is not None and bias.requires_grad)))
        if _n1_images_ok(idx.shape[0], p.shape[1]):
            # the pooled rows feed the n1-row combine product: their bf16x3 image goes out beside them
            out, argmax, img = reduce_fwd_img(p, idx, want_argmax=need)
            attach_image(out, img)
        else:
            out, argmax = reduce_fwd(p, idx, "max", want_argmax=need)
        if _CAPTURE is not None:
            _CAPTURE.append(dict(argmax=argmax, neigh=out))
        ctx.bias_t = bias
        ctx.pool_plan = None
        if plan_ok:
            # layer 0 (see backward): the gradient-free half of the pool backward starts here, beside the products that follow
            ctx.pool_plan = pool_bwd_x3_plan(argmax, out, idx, ctx.n_src)
        ctx.dp_slot = None
        if need and ctx.pool_plan is None and max(ctx.n_src, 1) * padded_ld(out.shape[1]) <= (SMALL_LOSS_ZERO_MAX if SMALL_LOSS_FUSED else CE_SMALL_MAX_ZERO):
            # a small scatter target: the loss launch that runs between this forward and its backward clears it on the side (the
            # one-workgroup loss up to CE_SMALL_MAX_ZERO floats, the fused small output layer + loss — a grid — up to SMALL_LOSS_ZERO_MAX;
            # a request nobody serves is filled by ``take_zeroed``)
            ctx.dp_slot = request_zeroed(ctx.n_src, out.shape[1], out.device)
        ctx.save_for_backward(x, w, x_rows, out, argmax, idx if idx.dtype == torch.int32 else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, x_rows, out, argmax, idx32 = ctx.saved_tensors
        need = ctx.needs_input_grad
        if (not need[0] and idx32 is not None and _MODE["name"] != "f32" and ctx.n_src >= X3_BWW_MIN_ROWS
                and out.shape[1] <= 640 and ctx.fanout <= 63 and (need[1] or (need[2] and ctx.has_bias))):
            # layer 0: the projection input carries no gradient, so dP has one consumer — the weight gradient — and goes
            # straight from (dout, argmax) to the image of its transpose
            plan = getattr(ctx, "pool_plan", None)
            ctx.pool_plan = None
            G = (ctx.n_src + 31) // 32
            rimg = _row_image_for(x, x_rows, None)
            dyT = pool_bwd_x3_apply(dout, idx32, plan, ctx.n_src) if plan is not None else pool_bwd_x3(dout, argmax, out, idx32, ctx.n_src)
            if rimg is not None and rimg.K == x.shape[1] + 1:
                # the resident table's own image, its rows gathered in the dealt order of dP^T: no X^T image
                dw, db, _ = linear_bwd_weight_x3k(dyT, rimg, ctx.n_src, x.shape[1], x_rows=x_rows, x_nrows=x.shape[0], interleave=G,
                                                  want_bias=ctx.has_bias, dw_out=_dw_out(w, *w.shape),
                                                  defer_for=(w, ctx.bias_t, None))
            else:
                dw, db = linear_bwd_weight_x3(dyT, x3_split_t(x, x_rows, ones_row=True, interleave=G), want_bias=ctx.has_bias,
                                              dw_out=_dw_out(w, *w.shape))
            return None, dw, (db if ctx.has_bias else None), None, None
        plan = getattr(ctx, "pool_plan", None)
        if plan is not None:                                       # (planned for the image path, which this call does not take)
            _plan_ready(plan)
        ctx.pool_plan = None
        slot, ctx.dp_slot = getattr(ctx, "dp_slot", None), None
        dp = reduce_bwd(dout, None, argmax, "max", ctx.n_src, fanout=ctx.fanout, relu_out=out,
                        dsrc=take_zeroed(slot, ctx.n_src, out.shape[1]) if slot is not None else None)
        dx = dw = db = None
        if need[0]:
            if x_rows is not None:
                raise RuntimeError("gradient w.r.t. a row-gathered table is not supported (features carry no grad)")
            dx = linear_bwd_input(dp, w, None)
        if need[1] or (need[2] and ctx.has_bias):
            dw, db = weight_grad(dp, x, x_rows, want_bias=ctx.has_bias, dw_out=_dw_out(w, *w.shape))
        return dx, dw, (db if ctx.has_bias else None), None, None


class _SagePoolLayerFn(torch.autograd.Function):
    """One autograd node for a whole 'pool' SAGEConv layer whose input carries a gradient (every layer but the first):
    out = act(fc_self(h[:n_dst]) + fc_neigh(max_j relu(fc_pool(h))[idx]) + b_self + b_neigh).
    h has two consumers (its first n_dst rows feed fc_self, all rows feed fc_pool); as separate nodes autograd pays a
    slice-backward (zero fill + copy of a [n_src, K] matrix) and a full-size gradient add per step.  Here the input
    gradient is the pool path's matrix with the fc_self part added in place to its first n_dst rows."""

    @staticmethod
    def forward(ctx, h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu):
        h = as_mat(h)
        ctx.h_relu_out = bool(getattr(h, "_ogl_relu_out", False))                    # h = relu(...) of the layer below
        himg = take_image(h) if _n1_images_ok(h.shape[0], h.shape[1], w_pool.shape[0]) else None
        ctx.h_img = himg if (himg is not None and himg.K == h.shape[1] + 1) else None   # read again by fc_pool's weight gradient
        if himg is not None and himg.K == h.shape[1] + 1:
            # the projection that produced h wrote its image (ones slot included): fc_pool runs on the image kernel
            wimg = weight_image("wb", w_pool, b_pool)
            if wimg is None:
                weight_images_prepare([("wb", (w_pool, b_pool))])
                wimg = weight_image("wb", w_pool, b_pool)
            p = linear_fwd_x3(himg, None, wimg, relu=True)
        else:
            p = linear_fwd(h, w_pool, b_pool, relu=True)
        need = any(t is not None and t.requires_grad for t in (h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh))
        neigh, argmax = reduce_fwd(p, idx, "max", want_argmax=need)
        bias = None
        if b_self is not None:
            bias = weight_image("bsum", b_self, b_neigh)          # from the step's one weight-image launch, when prepared
            if bias is None:
                bias = b_self + b_neigh
        out = linear_fwd(h[:n_dst], w_self, bias, x2=neigh, w2=w_neigh, relu=relu)
        if _CAPTURE is not None:
            _CAPTURE.append(dict(argmax=argmax, neigh=neigh, out=out if relu else None))
        ctx.relu, ctx.n_dst, ctx.fanout, ctx.has_bias, ctx.has_pool_bias = bool(relu), n_dst, idx.shape[1], b_self is not None, b_pool is not None
        ctx.b_pool_t = b_pool
        # the few-column output layer scatters its pooled-row gradient with atomics: park the (empty) target where the loss launch
        # that comes next will zero it
        ctx.dp_slot = None
        if need and not relu and out.shape[1] <= 64 and OUT_LAYER_FUSED and h.shape[0] >= 1024:
            ctx.dp_slot = request_zeroed(h.shape[0], h.shape[1], h.device)
        ctx.save_for_backward(h, w_pool, w_self, w_neigh, neigh, argmax, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, dy):
        h, w_pool, w_self, w_neigh, neigh, argmax, out = ctx.saved_tensors
        dy = as_mat(dy)
        if out is not None:
            dy = relu_bwd(dy, out)
        n_dst, n_src = ctx.n_dst, h.shape[0]
        h_dst = h[:n_dst]
        if _out_layer_fits(dy, h, w_self, w_neigh):
            return _out_layer_backward(ctx, dy, h, w_pool, w_self, w_neigh, neigh, argmax) + (None, None, None)
        dyT = transposed_operand(dy) if (_MODE["name"] != "f32" and dy.shape[0] >= 1024) else None
        dw_self, db = weight_grad(dy, h_dst, None, want_bias=ctx.has_bias, dyT=dyT, dw_out=_dw_out(w_self, *w_self.shape))
        # the bias gradient once more from the second product (its ones column is free): two tensors for the two biases —
        # one tensor returned for both makes autograd clone it (a launch)
        dw_neigh, db2 = weight_grad(dy, neigh, None, want_bias=ctx.has_bias, dyT=dyT, dw_out=_dw_out(w_neigh, *w_neigh.shape))
        dneigh = linear_bwd_input(dy, w_neigh, None)
        dp = reduce_bwd(dneigh, None, argmax, "max", n_src, fanout=ctx.fanout, relu_out=neigh)
        dp_img = x3_split(dp) if (N1_BWD_SPLIT and _n1_images_ok(n_src, dp.shape[1], w_pool.shape[1])) else None
        dh = linear_bwd_input(dp, w_pool, None, dy_img=dp_img)
        dw_pool, db_pool = weight_grad(dp, h, None, want_bias=ctx.has_pool_bias, x_img=ctx.h_img, dy_img=dp_img, dw_out=_dw_out(w_pool, *w_pool.shape))
        dx_self = linear_bwd_input(dy, w_self, None)
        dh[:n_dst].add_(dx_self)                                  # the fc_self path, in place on the first n_dst rows
        return (dh, dw_pool, db_pool if ctx.has_pool_bias else None, dw_self, dw_neigh, db if ctx.has_bias else None,
                db2 if ctx.has_bias else None, None, None, None)


def _out_layer_backward(ctx, dy, h, w_pool, w_self, w_neigh, neigh, argmax):
    """Backward of a 'pool' layer with few output columns (the output layer), shared by ``_SagePoolLayerFn`` and ``_SagePoolLossFn``:
    (dh, dw_pool, db_pool, dw_self, dw_neigh, db_self, db_neigh).  The combine's backward in two launches, its input gradient for
    the pooled rows scattered to the winners as it is computed; the fc_self part joins dh in the epilogue of the fc_pool input
    gradient."""
    n_dst, n_src = ctx.n_dst, h.shape[0]
    h_dst = h[:n_dst]
    # (taken once: a second backward through the same graph — retain_graph=True — must not scatter into the first one's sums)
    slot, ctx.dp_slot = getattr(ctx, "dp_slot", None), None
    tall = N1_BWD_SPLIT and _n1_images_ok(n_src, h.shape[1], w_pool.shape[1])
    # the layer's weight gradients are leaves of the backward graph: on the side stream when the layer is tall — the two
    # few-column ones right away (beside the equally small input-gradient launch), fc_pool's once dP and its image exist
    # (the critical launch FIRST: in a captured step the first-created child of a fork node stays on its parent's queue, the
    # others start ~5 us later on another one and every later cross-queue edge of their chain costs the same again —
    # measured 1.075-1.084 -> 1.056-1.059 ms per replayed Reddit step, same box, alternating runs)
    at0 = fork_point() if tall else None
    finish, ctx.loss_out = getattr(ctx, "loss_out", None), None
    dx_self, dp = out_layer_bwd_inputs(dy, w_self, w_neigh, argmax, neigh, n_src,
                                       dp_zeroed=take_zeroed(slot, n_src, h.shape[1]) if slot is not None else None, finish_loss=finish)
    finish = None
    with (side_section(dy, h, neigh, at=at0) if tall else _NoSection()):
        if finish is not None:                   # the deferred mean of the forward launch: one wave, off the critical path
            _launch("ogl_loss_mean_finish", _lib.lib().ogl_loss_mean_finish, _ptr(finish[0]), finish[0].numel(), _ptr(finish[1]), _stream(),
                    meta=dict(n=finish[0].numel()))
        dw_self, dw_neigh, db, db2 = out_layer_bwd_weights(dy, h_dst, neigh, want_bias=ctx.has_bias,
                                                           dws_out=_dw_out(w_self, *w_self.shape), dwn_out=_dw_out(w_neigh, *w_neigh.shape))
    dp_img = x3_split(dp) if tall else None
    at = fork_point() if tall else None
    dh = linear_bwd_input(dp, w_pool, None, dy_img=dp_img, add_head=dx_self,
                          out_relu_mask=h if (FUSE_RELU_BWD and ctx.h_relu_out and tall) else None)
    with (side_section(dp, dp_img, ctx.h_img, at=at) if tall else _NoSection()):
        dw_pool, db_pool = weight_grad(dp, h, None, want_bias=ctx.has_pool_bias, x_img=ctx.h_img, dy_img=dp_img, dw_out=_dw_out(w_pool, *w_pool.shape),
                                       defer_for=(w_pool, getattr(ctx, "b_pool_t", None), None))
    return (dh, dw_pool, db_pool if ctx.has_pool_bias else None, dw_self, dw_neigh, db if ctx.has_bias else None,
            db2 if ctx.has_bias else None)


class _SagePoolLossFn(torch.autograd.Function):
    """The LAST 'pool' layer of a train step together with its loss, one autograd node:
        logits = fc_self(h[:n_dst]) + fc_neigh(max_j relu(fc_pool(h))[idx]) + b_self + b_neigh;  loss = mean_d CE(logits[d], label(d)).
    Forward = the fc_pool product + ONE launch for everything after it (``out_layer_fwd_ce``: neighbour max, projection, cross
    entropy, the zero fill of the backward's scatter target); backward = ``_out_layer_backward`` from the stored dlogits.
    Returns (mean loss, per-seed losses, logits); only the mean is differentiable."""

    @staticmethod
    def forward(ctx, h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, labels, defer_mean=False):
        h = as_mat(h)
        ctx.h_relu_out = bool(getattr(h, "_ogl_relu_out", False))
        himg = take_image(h) if _n1_images_ok(h.shape[0], h.shape[1], w_pool.shape[0]) else None
        ctx.h_img = himg if (himg is not None and himg.K == h.shape[1] + 1) else None
        if ctx.h_img is not None:
            wimg = weight_image("wb", w_pool, b_pool)
            if wimg is None:
                weight_images_prepare([("wb", (w_pool, b_pool))])
                wimg = weight_image("wb", w_pool, b_pool)
            p = linear_fwd_x3(himg, None, wimg, relu=True)
        else:
            p = linear_fwd(h, w_pool, b_pool, relu=True)
        need = any(t is not None and t.requires_grad for t in (h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh))
        ent = None
        if need and h.shape[0] >= 1024:
            # the backward scatters its pooled-row gradient with float atomics into a zeroed [n_src, K] matrix: cleared by this launch
            buf = torch.empty((h.shape[0], padded_ld(h.shape[1])), dtype=torch.float32, device=h.device)
            ent = [buf, h.shape[0], h.shape[1], True]
        # The VALUE of the mean loss is written by the first launch of this node's backward (``finish_loss``): inside the forward
        # launch it costs a device-scope fence per block (58 us instead of ~20 for 512 seeds, measured).  Only a caller that owns the
        # backward may ask for that (``defer_mean``: the strategies' train steps, the captured step body); any other loss — one that is
        # logged, guarded or never backpropagated — gets its mean from the forward launch.  A deferred mean reads NaN until it exists.
        ctx.defer_mean = bool(need and DEFER_LOSS_MEAN and defer_mean)
        ctx.set_materialize_grads(False)             # (the gradients of the two non-differentiable outputs stay None: no zero fills)
        mean, rows, logits, neigh, argmax, dl = out_layer_fwd_ce(p, idx, h, n_dst, w_self, w_neigh, b_self, b_neigh, labels,
                                                                want_grad=need, zero=ent[0] if ent is not None else None,
                                                                want_mean=not ctx.defer_mean)
        ctx.loss_out = (rows, mean) if ctx.defer_mean else None
        if _CAPTURE is not None:
            _CAPTURE.append(dict(argmax=argmax, neigh=neigh, out=None))
        ctx.n_dst, ctx.fanout, ctx.has_bias, ctx.has_pool_bias = n_dst, idx.shape[1], b_self is not None, b_pool is not None
        ctx.b_pool_t = b_pool
        ctx.dp_slot = ent
        ctx.save_for_backward(h, w_pool, w_self, w_neigh, neigh, argmax, dl)
        ctx.mark_non_differentiable(rows, logits)
        return mean, rows, logits

    @staticmethod
    def backward(ctx, dloss, _drows, _dlogits):
        h, w_pool, w_self, w_neigh, neigh, argmax, dl = ctx.saved_tensors
        unit = _UNIT_GRAD.get((dloss.device.type, dloss.device.index))
        if unit is not None and dloss.data_ptr() == unit.data_ptr():
            dy = dl
        else:                                    # (a user's own root gradient: scaled into a matrix with the padded row stride)
            dy = empty_mat(dl.shape[0], dl.shape[1], dl.device)
            torch.mul(dl, dloss, out=dy)
        return _out_layer_backward(ctx, dy, h, w_pool, w_self, w_neigh, neigh, argmax) + (None, None, None, None)


def out_layer_fwd_ce_mean(h, idx, n_dst, w_cat, split, bias, labels, want_mean=True, p=None):
    """(mean loss, row losses, logits, neigh, dlogits / n_dst) of the in-repo 'mean' output layer in ONE launch
    (``ogl_out_layer_fwd_ce_mean``): neigh = mean_j P[idx[:, j]], logits = cat(h[:n_dst], neigh) . w_cat^T + bias; ``p`` = the rows the
    mean runs over (default ``h`` itself: 'mean'; relu(fc_pool(h)): 'meanpool'), same width as ``h``."""
    h = as_mat(h); w_cat = as_mat(w_cat)
    p = h if p is None else as_mat(p)
    assert p.shape == h.shape and _ld(p) % 4 == 0 and p.data_ptr() % 16 == 0
    K, N = h.shape[1], w_cat.shape[0]
    assert split == K and w_cat.shape[1] == 2 * K and w_cat.is_contiguous()
    dev = h.device
    lazy = labels if isinstance(labels, LazyLabels) else None
    if lazy is None:
        labels = labels.reshape(-1)
        assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous()
    assert labels.numel() == n_dst and idx.shape[0] == n_dst and idx.is_contiguous() and idx.dtype == torch.int32
    neigh = empty_mat(n_dst, K, dev)
    logits = empty_mat(n_dst, N, dev)
    loss = torch.empty(n_dst, dtype=torch.float32, device=dev)
    mean = torch.empty((), dtype=torch.float32, device=dev)
    dl = empty_mat(n_dst, N, dev)
    stream = _stream()
    table, ids = (lazy.table, lazy.ids) if lazy is not None else (labels, None)
    ws_ptr, wn_ptr = w_cat.data_ptr(), w_cat.data_ptr() + 4 * K
    _launch("ogl_out_layer_fwd_ce", _lib.lib().ogl_out_layer_fwd_ce_mean, _ptr(p), _ld(p), h.shape[0], _ptr(idx), n_dst, int(idx.shape[1]),
            _ptr(h), _ld(h), K, ws_ptr, 2 * K, wn_ptr, 2 * K, _ptr(bias), None, N, _ptr(neigh), _ld(neigh), _ptr(logits), _ld(logits),
            _ptr(table), table.numel(), _ptr(ids), C.c_float(1.0 / n_dst), _ptr(loss), _ptr(dl), _ld(dl), _ptr(mean),
            ce_counter(dev, stream) if want_mean else None, OUT_FWD_ROWS if OUT_FWD_ROWS in (0, 1, 2) else 0, stream,
            meta=dict(n_dst=n_dst, fanout=int(idx.shape[1]), d=K, N=N, zero_bytes=0, mean=True))
    return mean, loss, logits, neigh, dl


def out_layer_bwd_inputs_dense(dy, w_cat, K, finish_loss=None):
    """(dx_self, dneigh) [n_dst, K] = dy . w_cat[:, :K], dy . w_cat[:, K:] in one launch (``ogl_out_layer_bwd_inputs_dense``)."""
    dy = as_mat(dy); w_cat = as_mat(w_cat)
    n_dst, N = dy.shape
    dx = empty_mat(n_dst, K, dy.device)
    dn = empty_mat(n_dst, K, dy.device)
    rows, mean = finish_loss if finish_loss is not None else (None, None)
    _launch("ogl_out_layer_bwd_inputs", _lib.lib().ogl_out_layer_bwd_inputs_dense, _ptr(dy), _ld(dy), n_dst, N, K, w_cat.data_ptr(), 2 * K,
            w_cat.data_ptr() + 4 * K, 2 * K, _ptr(dx), _ld(dx), _ptr(dn), _ld(dn), _ptr(rows), rows.numel() if rows is not None else 0,
            _ptr(mean), _stream(), meta=dict(M=n_dst, N=N, K=K, dense=True))
    return dx, dn


class _SageMeanLossFn(torch.autograd.Function):
    """The LAST in-repo 'mean' layer of a train step together with its loss, one autograd node (round 5):
        logits = fc_neigh(cat(h[:n_dst], mean_j h[idx[:, j]]));  loss = mean_d CE(logits[d], label(d))
    (R/train/graphsage/pytorch/aggregator_dgl.py:156-159,199-206; pytorch/model.py:105).  Forward = ONE launch
    (``out_layer_fwd_ce_mean``); backward = the dense input gradients (one launch), the two weight-gradient blocks and the bias
    gradient (one launch, on the side branch), the mean's planned segmented backward with the head rows' gradient added in place.
    Before: mean reduce + clone + skinny product + loss + two input-gradient products + two skinny weight gradients + three ATen adds."""

    @staticmethod
    def forward(ctx, h, w_cat, bias, idx, n_dst, labels, defer_mean=False, plan=None):
        h = as_mat(h)
        K = h.shape[1]
        need = any(t is not None and t.requires_grad for t in (h, w_cat, bias))
        ctx.defer_mean = bool(need and DEFER_LOSS_MEAN and defer_mean)
        ctx.set_materialize_grads(False)
        # (``plan``: the mean's backward plan when the caller started it earlier — GraphSAGE.forward_loss does, at the top of the step:
        # its eight small launches then run beside the first layer instead of in front of this node's backward)
        if plan is not None and plan.shape != (n_dst, idx.shape[1], K, h.shape[0]):
            plan = None
        ctx.seg_plan = (plan if plan is not None else reduce_bwd_seg_plan(idx, K, h.shape[0])) if (need and h.requires_grad) else None
        mean, rows, logits, neigh, dl = out_layer_fwd_ce_mean(h, idx, n_dst, w_cat, K, bias, labels, want_mean=not ctx.defer_mean)
        ctx.loss_out = (rows, mean) if ctx.defer_mean else None
        ctx.n_dst, ctx.K, ctx.has_bias, ctx.n_src = n_dst, K, bias is not None, h.shape[0]
        ctx.save_for_backward(h, w_cat, neigh, dl, idx)
        ctx.mark_non_differentiable(rows, logits)
        return mean, rows, logits

    @staticmethod
    def backward(ctx, dloss, _drows, _dlogits):
        h, w_cat, neigh, dl, idx = ctx.saved_tensors
        unit = _UNIT_GRAD.get((dloss.device.type, dloss.device.index))
        if unit is not None and dloss.data_ptr() == unit.data_ptr():
            dy = dl
        else:
            dy = empty_mat(dl.shape[0], dl.shape[1], dl.device)
            torch.mul(dl, dloss, out=dy)
        n_dst, K = ctx.n_dst, ctx.K
        finish, ctx.loss_out = getattr(ctx, "loss_out", None), None
        need = ctx.needs_input_grad
        at0 = fork_point()
        dx_self, dneigh = out_layer_bwd_inputs_dense(dy, w_cat, K, finish_loss=finish)
        dw_cat = db = None
        if need[1] or (need[2] and ctx.has_bias):
            with (side_section(dy, h, neigh, at=at0) if at0 is not None else _NoSection()):
                dw_cat = torch.empty_like(w_cat, memory_format=torch.contiguous_format)
                _, _, db, _ = out_layer_bwd_weights(dy, h[:n_dst], neigh, want_bias=ctx.has_bias, dws_out=dw_cat[:, :K], dwn_out=dw_cat[:, K:])
        dh = None
        if need[0]:
            plan, ctx.seg_plan = getattr(ctx, "seg_plan", None), None
            if plan is None:
                plan = reduce_bwd_seg_plan(idx, K, ctx.n_src, side=False)
            dh = reduce_bwd_seg_apply(dneigh, idx, plan, "mean", add=dx_self)[0]      # (the head rows' own gradient joins inside the launch)
        return dh, dw_cat, (db if ctx.has_bias else None), None, None, None, None, None


class _SageMeanPoolLossFn(torch.autograd.Function):
    """The LAST in-repo 'meanpool' layer of a train step together with its loss (R/train/graphsage/pytorch/aggregator_dgl.py:178-186,
    199-206): p = relu(fc_pool(h)); logits = fc_neigh(cat(h[:n_dst], mean_j p[idx[:, j]])); the mean CE.  Forward = the fc_pool product +
    ONE launch; backward = ``_SageMeanLossFn``'s, with the mean's planned backward masked by [p > 0] and followed by fc_pool's two
    gradients (the head rows' gradient joins dh in the epilogue of the input-gradient product, as in ``_out_layer_backward``)."""

    @staticmethod
    def forward(ctx, h, w_pool, b_pool, w_cat, bias, idx, n_dst, labels, defer_mean=False, plan=None):
        h = as_mat(h)
        K = h.shape[1]
        if plan is not None and plan.shape != (n_dst, idx.shape[1], K, h.shape[0]):
            plan = None
        ctx.h_relu_out = bool(getattr(h, "_ogl_relu_out", False))
        himg = take_image(h) if _n1_images_ok(h.shape[0], K, w_pool.shape[0]) else None
        ctx.h_img = himg if (himg is not None and himg.K == K + 1) else None
        if ctx.h_img is not None:
            wimg = weight_image("wb", w_pool, b_pool)
            if wimg is None:
                weight_images_prepare([("wb", (w_pool, b_pool))])
                wimg = weight_image("wb", w_pool, b_pool)
            p = linear_fwd_x3(himg, None, wimg, relu=True)
        else:
            p = linear_fwd(h, w_pool, b_pool, relu=True)
        if _CAPTURE is not None:
            _CAPTURE.append(dict(pool_out=p))
        need = any(t is not None and t.requires_grad for t in (h, w_pool, b_pool, w_cat, bias))
        ctx.defer_mean = bool(need and DEFER_LOSS_MEAN and defer_mean)
        ctx.set_materialize_grads(False)
        ctx.seg_plan = (plan if plan is not None else reduce_bwd_seg_plan(idx, K, h.shape[0])) if need else None
        mean, rows, logits, neigh, dl = out_layer_fwd_ce_mean(h, idx, n_dst, w_cat, K, bias, labels, want_mean=not ctx.defer_mean, p=p)
        ctx.loss_out = (rows, mean) if ctx.defer_mean else None
        ctx.n_dst, ctx.K, ctx.has_bias, ctx.has_pool_bias, ctx.n_src = n_dst, K, bias is not None, b_pool is not None, h.shape[0]
        ctx.b_pool_t = b_pool
        ctx.save_for_backward(h, w_pool, w_cat, neigh, dl, idx, p)
        ctx.mark_non_differentiable(rows, logits)
        return mean, rows, logits

    @staticmethod
    def backward(ctx, dloss, _drows, _dlogits):
        h, w_pool, w_cat, neigh, dl, idx, p = ctx.saved_tensors
        unit = _UNIT_GRAD.get((dloss.device.type, dloss.device.index))
        if unit is not None and dloss.data_ptr() == unit.data_ptr():
            dy = dl
        else:
            dy = empty_mat(dl.shape[0], dl.shape[1], dl.device)
            torch.mul(dl, dloss, out=dy)
        n_dst, K = ctx.n_dst, ctx.K
        finish, ctx.loss_out = getattr(ctx, "loss_out", None), None
        tall = N1_BWD_SPLIT and _n1_images_ok(ctx.n_src, K, w_pool.shape[1])
        at0 = fork_point() if tall else None
        dx_self, dneigh = out_layer_bwd_inputs_dense(dy, w_cat, K, finish_loss=finish)
        with (side_section(dy, h, neigh, at=at0) if at0 is not None else _NoSection()):
            dw_cat = torch.empty_like(w_cat, memory_format=torch.contiguous_format)
            _, _, db, _ = out_layer_bwd_weights(dy, h[:n_dst], neigh, want_bias=ctx.has_bias, dws_out=dw_cat[:, :K], dwn_out=dw_cat[:, K:])
        plan, ctx.seg_plan = getattr(ctx, "seg_plan", None), None
        if plan is None:
            plan = reduce_bwd_seg_plan(idx, K, ctx.n_src, side=False)
        # dP = the mean's backward masked by [p > 0], with its image beside it for the two products that follow
        dp, dp_img = reduce_bwd_seg_apply(dneigh, idx, plan, "mean", mask=p, want_out=True, want_image=tall)
        at = fork_point() if tall else None
        dh = linear_bwd_input(dp, w_pool, None, dy_img=dp_img, add_head=dx_self,
                              out_relu_mask=h if (FUSE_RELU_BWD and ctx.h_relu_out and tall) else None)
        with (side_section(dp, dp_img, ctx.h_img, at=at) if at is not None else _NoSection()):
            dw_pool, db_pool = weight_grad(dp, h, None, want_bias=ctx.has_pool_bias, x_img=ctx.h_img, dy_img=dp_img,
                                           dw_out=_dw_out(w_pool, *w_pool.shape), defer_for=(w_pool, getattr(ctx, "b_pool_t", None), None))
        return (dh, dw_pool, db_pool if ctx.has_pool_bias else None, dw_cat, (db if ctx.has_bias else None), None, None, None, None, None)


MEAN_LOSS_FUSED = True


def sage_meanpool_layer_loss(h, w_pool, b_pool, w_cat, bias, idx, n_dst, labels, defer_mean=False, plan=None):
    """(mean CE loss, per-seed losses, logits) of the last in-repo 'meanpool' layer + nn.CrossEntropyLoss as one node, or None."""
    if (not MEAN_LOSS_FUSED or h.dim() != 2 or idx.dtype != torch.int32 or not idx.is_contiguous() or w_cat.shape[1] != 2 * h.shape[1]
            or w_pool.shape[0] != h.shape[1] or w_pool.shape[1] != h.shape[1]
            or not w_cat.is_contiguous() or not _lib.lib().ogl_out_layer_fwd_ce_fits(n_dst, idx.shape[1], h.shape[1], w_cat.shape[0])
            or n_dst > 4096 or h.shape[0] < n_dst or _ld(as_mat(h)) % 4 or as_mat(h).data_ptr() % 16 or w_cat.data_ptr() % 16
            or h.shape[1] % 4 or not seg_bwd_fits(idx, h.shape[1], h.shape[0])):
        return None
    return _SageMeanPoolLossFn.apply(h, w_pool, b_pool, w_cat, bias, idx, n_dst, labels, bool(defer_mean), plan)


def sage_mean_layer_loss(h, w_cat, bias, idx, n_dst, labels, defer_mean=False, plan=None):
    """(mean CE loss, per-seed losses, logits) of the last in-repo 'mean' layer + nn.CrossEntropyLoss as one node, or None when the fused
    form does not apply."""
    if (not MEAN_LOSS_FUSED or h.dim() != 2 or idx.dtype != torch.int32 or not idx.is_contiguous() or w_cat.shape[1] != 2 * h.shape[1]
            or not w_cat.is_contiguous() or not _lib.lib().ogl_out_layer_fwd_ce_fits(n_dst, idx.shape[1], h.shape[1], w_cat.shape[0])
            or n_dst > 4096 or h.shape[0] < n_dst or _ld(as_mat(h)) % 4 or as_mat(h).data_ptr() % 16 or w_cat.data_ptr() % 16
            or h.shape[1] % 4 or not seg_bwd_fits(idx, h.shape[1], h.shape[0])):
        return None
    return _SageMeanLossFn.apply(h, w_cat, bias, idx, n_dst, labels, bool(defer_mean), plan)


def _adopts_gradient(p):
    """AccumulateGrad will ADOPT the gradient tensor a node returns for ``p`` and nobody sees it on the way: ``p`` holds no gradient
    yet (else: ``grad += new``) and carries no gradient hook of either kind (a hook reads the tensor when autograd hands it over)."""
    return p is None or (p.grad is None and not p._backward_hooks and not getattr(p, "_post_accumulate_grad_hooks", None))


def sage_pool_layer_loss(h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, labels, defer_mean=False, h_single_use=False):
    """(mean CE loss, per-seed losses, logits) of the last 'pool' layer + nn.CrossEntropyLoss, or None when the fused form does not
    apply (the caller then runs the layer and the loss separately)."""
    if h.dim() != 2:
        return None
    if small_pool_layer_fits(h.shape[0], n_dst, idx.shape[1], h.shape[1], w_self.shape[0]):
        # (two launches for the layer, its loss and their backward; the mean's VALUE comes from the second: deferring callers only)
        if defer_mean and DEFER_LOSS_MEAN and small_pool_loss_fits(h, w_pool, w_self, w_neigh, idx, n_dst, labels):
            # (``h_single_use``: the caller made ``h`` and hands it to this layer only — GraphSAGE.forward_loss.  When it came out of the
            # small first layer, this node's whole backward moves into that layer's two launches: see ``_SmallPoolLossFn``)
            # (not under a gradient exchange that launches collectives from gradient hooks: this node's gradient tensors are filled by
            # a launch enqueued AFTER their hooks fire — a bucket made of them alone would be reduced before it is written)
            # (and only while every gradient it would hand out is ADOPTED by AccumulateGrad — ``p.grad is None`` — and nobody else can
            # see the gradient of ``h``: no retain_grad, no tensor hook.  All of it is checked again when the backward runs, which
            # takes the route only inside ``ops.backward`` and otherwise finishes the node with its own launch.)
            lazy = bool(SMALL_ROUTE and h_single_use and h.requires_grad and type(h.grad_fn) is _SmallFirstLayerFn._backward_cls
                        and n_dst * h.shape[1] <= 2048 and not _GRAD_SINKS and not h.retains_grad and not h._backward_hooks
                        and all(_adopts_gradient(p) for p in (w_pool, b_pool, w_self, w_neigh, b_self, b_neigh)))
            return _SmallPoolLossFn.apply(h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, labels, lazy)
        return None
    if (b_self is None) != (b_neigh is None) or not out_loss_fits(h, n_dst, idx, w_self, w_neigh, w_pool.shape[0]):
        return None
    return _SagePoolLossFn.apply(h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, labels, bool(defer_mean))


SMALL_LAYER = True      # small 'pool' layers run as one launch forward + one launch backward (small_layer.hip)


def small_pool_layer_fits(n_src, n_dst, fanout, hin, hout):
    return SMALL_LAYER and bool(_lib.lib().ogl_small_pool_layer_fits(int(n_src), int(n_dst), int(fanout), int(hin), int(hout)))


def small_pool_layer_fwd(h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu, want_argmax=True):
    h = as_mat(h); w_pool = as_mat(w_pool); w_self = as_mat(w_self); w_neigh = as_mat(w_neigh)
    n_src, hin = h.shape
    hout = w_self.shape[0]
    assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.shape[0] == n_dst
    neigh = empty_mat(n_dst, hin, h.device)
    argmax = torch.empty((n_dst, hin), dtype=torch.int32, device=h.device) if want_argmax else None
    y = empty_mat(n_dst, hout, h.device)
    ws = torch.empty(max(n_src * hin, 4), dtype=torch.float32, device=h.device)
    _launch("ogl_small_pool_layer_fwd", _lib.lib().ogl_small_pool_layer_fwd, _ptr(h), _ld(h), n_src, _ptr(idx), n_dst, idx.shape[1], hin,
            _ptr(w_pool), _ld(w_pool), _ptr(b_pool), _ptr(w_self), _ld(w_self), _ptr(b_self), _ptr(w_neigh), _ld(w_neigh), _ptr(b_neigh),
            hout, int(bool(relu)), _ptr(neigh), _ld(neigh), _ptr(argmax), _ptr(y), _ld(y), _ptr(ws), _stream(),
            meta=dict(n_src=n_src, n_dst=n_dst, hin=hin, hout=hout))
    return y, neigh, argmax


class _SmallPoolLayerFn(torch.autograd.Function):
    """The same layer as _SagePoolLayerFn on the one-workgroup kernels: two launches per layer per step instead of ~15."""

    @staticmethod
    def forward(ctx, h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu):
        h = as_mat(h)
        y, neigh, argmax = small_pool_layer_fwd(h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu)
        if _CAPTURE is not None:
            _CAPTURE.append(dict(argmax=argmax, neigh=neigh, out=y if relu else None))
        ctx.relu, ctx.n_dst, ctx.fanout = bool(relu), n_dst, idx.shape[1]
        ctx.flags = (b_pool is not None, b_self is not None, b_neigh is not None)
        ctx.save_for_backward(h, as_mat(w_pool), as_mat(w_self), as_mat(w_neigh), neigh, argmax, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w_pool, w_self, w_neigh, neigh, argmax, y = ctx.saved_tensors
        dy = as_mat(dy)
        n_src, hin = h.shape
        hout = w_self.shape[0]
        need = ctx.needs_input_grad
        dev = h.device
        dh = empty_mat(n_src, hin, dev) if need[0] else None
        dwp = torch.empty((hin, hin), dtype=torch.float32, device=dev)
        dws = torch.empty((hout, hin), dtype=torch.float32, device=dev)
        dwn = torch.empty((hout, hin), dtype=torch.float32, device=dev)
        has_bp, has_bs, has_bn = ctx.flags
        dbp = torch.empty(hin, dtype=torch.float32, device=dev) if has_bp else None
        dbs = torch.empty(hout, dtype=torch.float32, device=dev) if has_bs else None
        dbn = torch.empty(hout, dtype=torch.float32, device=dev) if has_bn else None
        ws = torch.empty(max(ctx.n_dst * hin, 4), dtype=torch.float32, device=dev)
        _launch("ogl_small_pool_layer_bwd", _lib.lib().ogl_small_pool_layer_bwd, _ptr(dy), _ld(dy), _ptr(y), _ld(y) if y is not None else 0,
                int(ctx.relu), _ptr(h), _ld(h), n_src, ctx.n_dst, ctx.fanout, hin, hout, _ptr(neigh), _ld(neigh), _ptr(argmax),
                _ptr(w_pool), _ld(w_pool), _ptr(w_self), _ld(w_self), _ptr(w_neigh), _ld(w_neigh), _ptr(dwp), hin, _ptr(dbp),
                _ptr(dws), hin, _ptr(dbs), _ptr(dwn), hin, _ptr(dbn), _ptr(dh), _ld(dh) if dh is not None else 0, _ptr(ws), _stream(),
                meta=dict(n_src=n_src, n_dst=ctx.n_dst, hin=hin, hout=hout))
        return dh, dwp, dbp, dws, dwn, dbs, dbn, None, None, None




def small_pool_loss_fits(h, w_pool, w_self, w_neigh, idx, n_dst, labels):
    """The last layer of a 32-seed step qualifies for ``_SmallPoolLossFn`` (ogl_small_pool_loss_fits + every weight differentiable)."""
    if not (SMALL_LOSS_FUSED and torch.is_grad_enabled() and w_pool.requires_grad and w_self.requires_grad and w_neigh.requires_grad):
        return False
    if idx.dtype != torch.int32 or not idx.is_contiguous() or idx.shape[0] != n_dst or labels.numel() != n_dst:
        return False
    return bool(_lib.lib().ogl_small_pool_loss_fits(int(h.shape[0]), int(n_dst), int(idx.shape[1]), int(h.shape[1]), int(w_self.shape[0])))


class _SmallPoolLossFn(torch.autograd.Function):
    """The LAST small 'pool' layer of a train step with nn.CrossEntropyLoss(reduction='mean'), one autograd node of TWO launches
    (R/train/graphsage/pytorch/model.py:87-107 on the 32-seed rungs), cut where the data crosses destinations: forward =
    ogl_small_pool_layer_fwd_ce_bwd — one workgroup per destination: layer, row loss, dlogits, the combine's two input gradients, and on
    the side the zero fill a first layer parked for its scatter target (``request_zeroed``, any size); backward =
    ogl_small_pool_layer_bwd_pool — the sums over destinations (dWs, dWn, bias gradients, the MEAN LOSS, the optimiser's per-step
    scalars) in one block, fc_pool through the winners in the others.  Before: 2 + 1 + 2 launches and a fill.  Same bits as those.
    Only for callers that own the backward (``defer_mean``): the loss tensor holds NaN until the backward launch has run.
    Returns (mean loss, per-seed losses, logits); only the mean is differentiable."""

    @staticmethod
    def forward(ctx, h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, labels, lazy=False):
        ctx.param_refs = [weakref.ref(p) for p in (w_pool, b_pool, w_self, w_neigh, b_self, b_neigh) if p is not None]
        h = as_mat(h); w_pool = as_mat(w_pool); w_self = as_mat(w_self); w_neigh = as_mat(w_neigh)
        n_src, hin = h.shape
        ctx.lazy = bool(lazy)
        hout = w_self.shape[0]
        dev = h.device
        lazy = labels if isinstance(labels, LazyLabels) else None
        if lazy is None:
            labels = labels.reshape(-1)
            assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous()
        table, ids = (lazy.table, lazy.ids) if lazy is not None else (labels, None)
        neigh = empty_mat(n_dst, hin, dev)
        argmax = torch.empty((n_dst, hin), dtype=torch.int32, device=dev)
        logits = empty_mat(n_dst, hout, dev)
        rows = torch.empty(n_dst, dtype=torch.float32, device=dev)
        mean = torch.empty((), dtype=torch.float32, device=dev)
        dl = empty_mat(n_dst, hout, dev)
        G = torch.empty(max(n_dst * hin, 4), dtype=torch.float32, device=dev)
        # (lazy: only the destinations' head rows of dh — the fc_self path; the consumer gathers the winners' rows itself)
        dh = (empty_mat(n_dst if ctx.lazy else n_src, hin, dev)) if h.requires_grad else None
        stream = _stream()
        zbuf, zn = None, 0
        key = (dev.index, stream)
        ent = _PENDING_ZERO.get(key)
        if ent is not None and ent[0].numel() % 4 == 0:       # (any size: fill-only blocks of the grid)
            del _PENDING_ZERO[key]
            zbuf, zn = ent[0], ent[0].numel()
            ent[3] = True
        _launch("ogl_small_pool_layer_fwd_ce_bwd", _lib.lib().ogl_small_pool_layer_fwd_ce_bwd, _ptr(h), _ld(h), n_src, _ptr(idx), n_dst,
                int(idx.shape[1]), hin, _ptr(w_pool), _ld(w_pool), _ptr(b_pool), _ptr(w_self), _ld(w_self), _ptr(b_self), _ptr(w_neigh),
                _ld(w_neigh), _ptr(b_neigh), hout, _ptr(table), table.numel(), _ptr(ids), C.c_float(1.0 / n_dst), _ptr(neigh), _ld(neigh),
                _ptr(argmax), _ptr(logits), _ld(logits), _ptr(rows), _ptr(mean), _ptr(dl), _ld(dl), _ptr(G), _ptr(dh),
                _ld(dh) if dh is not None else 0, 1 if ctx.lazy else 0, _ptr(zbuf), zn, stream,
                meta=dict(n_src=n_src, n_dst=n_dst, hin=hin, hout=hout, zero_bytes=4 * zn))
        # the optimiser's per-step scalars ride in this node's BACKWARD launch (the 32-seed steps have no weight-image launch): the
        # request is taken here so that no launch in between serves it twice; without a backward the optimiser prepares itself
        ctx.prime = None
        prime = _ADAM_PRIME.get("req")
        if prime is not None and ADAM_PRIME_IN_SPLIT:
            ctx.prime, _ADAM_PRIME["req"] = prime, None
        if _CAPTURE is not None:
            _CAPTURE.append(dict(argmax=argmax, neigh=neigh, out=None))
        ctx.n_dst, ctx.hout = n_dst, hout
        ctx.flags = (b_pool is not None, b_self is not None, b_neigh is not None)
        ctx.pre = (G, dh, rows, mean)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(h, w_pool, argmax, neigh, dl)
        ctx.mark_non_differentiable(rows, logits)
        return mean, rows, logits

    @staticmethod
    def backward(ctx, dloss, _drows, _dlogits):
        h, w_pool, argmax, neigh, dl = ctx.saved_tensors
        # (taken once and dropped: ``rows`` / ``mean`` are OUTPUTS of this node — kept on ctx past the backward they close a reference
        # cycle output -> grad_fn -> ctx -> output that keeps the step's autograd graph alive until the garbage collector runs; a
        # graph of an eager step still alive inside the capture of the next one crashed hipStreamEndCapture of a replica's step)
        if ctx.pre is None:
            raise RuntimeError("the fused small output layer + loss node was already backpropagated (its forward launch's gradients are "
                               "consumed once: no retain_graph)")
        G, dh, rows, mean = ctx.pre
        ctx.pre = None
        n_src, hin = h.shape
        hout = ctx.hout
        dev = h.device
        unit = _UNIT_GRAD.get((dloss.device.type, dloss.device.index))
        if not (unit is not None and dloss.data_ptr() == unit.data_ptr()):
            # (a root gradient of the caller's own: everything the forward launch wrote is linear in it)
            dls = empty_mat(dl.shape[0], dl.shape[1], dev)
            torch.mul(dl, dloss, out=dls)
            dl, G = dls, G * dloss
            if dh is not None:
                dhs = empty_mat(dh.shape[0], hin, dev)
                torch.mul(dh, dloss, out=dhs)
                dh = dhs
        has_bp, has_bs, has_bn = ctx.flags
        dwp = torch.empty((hin, hin), dtype=torch.float32, device=dev)
        dbp = torch.empty(hin, dtype=torch.float32, device=dev) if has_bp else None
        dws = torch.empty((hout, hin), dtype=torch.float32, device=dev)
        dwn = torch.empty((hout, hin), dtype=torch.float32, device=dev)
        dbs = torch.empty(hout, dtype=torch.float32, device=dev) if has_bs else None
        dbn = torch.empty(hout, dtype=torch.float32, device=dev) if has_bn else None
        step_dev = scal = None
        lr = b1 = b2 = 0.0
        if ctx.prime is not None:
            step_dev, scal, lr, b1, b2 = ctx.prime
            ctx.prime = None
        route_ok = False
        if ctx.lazy:
            # the route is a promise — gradient tensors a LATER node's launch fills — and is taken only where it is certain to be
            # kept: inside ``ops.backward`` (which raises if the route is still pending when the pass ends) and while AccumulateGrad
            # ADOPTS what this node returns (a parameter that already holds a gradient would get `grad += <unwritten memory>`)
            route_ok = _OWN_BWD["depth"] > 0 and all(_adopts_gradient(r()) for r in ctx.param_refs)
            if not route_ok and dh is not None:
                # finished by this node's own launch after all: the forward wrote the destinations' head rows only
                full = fill_zero(torch.empty((max(n_src, 1), padded_ld(hin)), dtype=torch.float32, device=dev))[:n_src, :hin]
                full[:dh.shape[0]].copy_(dh)
                dh = full
        if route_ok:
            # NO launch here: the gradient handed to the first layer is an EMPTY [n_src, hin] matrix that stands for the route (its
            # consumer, _SmallFirstLayerFn.backward, finds the route by the matrix' address, gathers the winners' rows inside its own
            # launch and never reads the matrix); this node's weight gradients, the mean loss and the optimiser's scalars are written by
            # that layer's record launch (ogl_record_weight_grads) — the tensors returned below are filled by it, in stream order before
            # anything reads them (the optimiser, a collective: both are enqueued after the first layer's backward)
            dh_full = empty_mat(n_src, hin, dev)
            _PENDING_ROUTES.clear()
            _PENDING_ROUTES[dh_full.data_ptr()] = dict(argmax=argmax, G=G, w_pool=w_pool, head=dh, n_head=ctx.n_dst, hin=hin, hout=hout, dl=dl,
                                                       neigh=neigh, h=h, rows=rows, mean=mean, prime=(step_dev, scal, lr, b1, b2) if step_dev is not None else None,
                                                       # (ADDRESSES, not tensors: AccumulateGrad adopts a gradient only when nobody else
                                                       # holds it and otherwise CLONES it — here: a copy of memory nothing has written yet)
                                                       out=tuple(t.data_ptr() if t is not None else None for t in (dwp, dbp, dws, dbs, dwn, dbn)),
                                                       keep=dh_full)
            return dh_full, dwp, dbp, dws, dwn, dbs, dbn, None, None, None, None
        _launch("ogl_small_pool_layer_bwd_pool", _lib.lib().ogl_small_pool_layer_bwd_pool, _ptr(h), _ld(h), ctx.n_dst, hin, hout, _ptr(argmax),
                _ptr(G), _ptr(w_pool), _ld(w_pool), _ptr(neigh), _ld(neigh), _ptr(dl), _ld(dl), _ptr(rows), _ptr(dwp), hin, _ptr(dbp),
                _ptr(dws), hin, _ptr(dbs), _ptr(dwn), hin, _ptr(dbn), _ptr(dh), _ld(dh) if dh is not None else 0, _ptr(mean), _ptr(step_dev),
                _ptr(scal), C.c_double(lr), C.c_double(b1), C.c_double(b2), _stream(),
                meta=dict(n_src=n_src, n_dst=ctx.n_dst, hin=hin, hout=hout, adam_prepare=step_dev is not None))
        if step_dev is not None:
            _ADAM_PRIME["served"] = (step_dev.data_ptr(), _capturing())
        return dh, dwp, dbp, dws, dwn, dbs, dbn, None, None, None, None


# the small last layer's whole backward inside the small first layer's two launches (its input gradient gathered by the consumer, its
# weight gradients as three more row groups of the record launch): no launch of its own, no float atomics
SMALL_ROUTE = os.environ.get("OGL_SMALL_ROUTE", "1") != "0"
_PENDING_ROUTES = {}
_SMALL_AGNOSTIC = {"seen": False}      # set by a forward whose first layer ran on the device's own source count (stepgraph reads it)
SMALL_LIVE = True      # padded rows of a captured step's upper-bound block take the kernels' early exits
SMALL_PROJ = True      # fc_pool of a small step's first layer on the small-tile fp32-MFMA kernel
SMALL_PROJ_MAX_ROWS = 4096
SMALL_FIRST_FUSED = True   # the first 'pool' layer of a 32-seed step: max + combine in one launch
SMALL_FIRST_MAX_DST = 2048
# fc_pool's weight gradient of that layer from the winners' records (ogl_small_first_layer_dw) while the rows it would gather from L2 if
# EVERY destination row were live — n_dst * F * F floats — stay below this many bytes.  (A captured 32-seed step runs on the upper-bound
# block, 832 destination rows — 1 472 at the reference's pubmed setting, fanout 45 — of which 100-300 are live: padded rows have no records
# and cost nothing, so the bound covers 1 472 x 500^2.  Measured, pubmed setting: 0.194 ms per step at 1 GiB (dense form), 0.1455 at 2 GiB.)
SMALL_FIRST_DW = True
SMALL_FIRST_DW_MAX_BYTES = (1 << 31)


def small_first_layer_fits(table, ids, idx, n_dst, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh):
    """The first 'pool' layer of a small step qualifies for ``_SmallFirstLayerFn`` (ogl_small_first_layer_fits + layouts)."""
    if not (SMALL_FIRST_FUSED and torch.is_grad_enabled() and ids is not None and table.dim() == 2 and not table.requires_grad):
        return False
    F, H = table.shape[1], w_self.shape[0]
    tm = as_mat(table)
    if (idx.dtype != torch.int32 or not idx.is_contiguous() or idx.shape[0] != n_dst or n_dst > SMALL_FIRST_MAX_DST or F % 4
            or (b_self is None) != (b_neigh is None) or tuple(w_pool.shape) != (F, F) or tuple(w_neigh.shape) != (H, F)
            or tuple(w_self.shape) != (H, F) or not (w_self.is_contiguous() and w_neigh.is_contiguous())
            or w_self.data_ptr() % 16 or w_neigh.data_ptr() % 16 or _ld(tm) % 4 or tm.data_ptr() % 16 or ids.numel() < n_dst
            or _n1_images_ok(n_dst, F, H)):
        return False
    return bool(_lib.lib().ogl_small_first_layer_fits(int(ids.numel()), int(n_dst), int(idx.shape[1]), int(F), int(H)))


class _SmallFirstLayerFn(torch.autograd.Function):
    """The FIRST 'pool' layer of a 32-seed train step (input = rows of the resident table: no input gradient) as one autograd node:
    fc_pool on the GEMM kernel, then neighbour max + combine in ONE launch (ogl_small_first_layer_fwd; before: the max aggregator and a
    skinny dual-input product, two autograd nodes); backward: the ReLU mask, dneigh = dy . Wn and — on the scatter path — the winners'
    scatter in ONE launch (ogl_small_first_layer_bwd; before: three), the combine's two weight gradients + bias gradients in one
    (ogl_out_layer_bwd_weights, as before), fc_pool's weight gradient from the winners' records in one (ogl_small_first_layer_dwpool:
    n_dst F records instead of the dense product — while n_dst F^2 floats of gathered rows stay small) or as ``_PoolMaxFn`` computes it.
    y = act(X[ids[:n_dst]] . Ws^T + max_j relu(X[ids] . Wp^T + bp)[idx] . Wn^T + bs + bn) (DGL SAGEConv 'pool')."""

    @staticmethod
    def forward(ctx, table, ids, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu, n_live=None, n_src_live=None):
        # n_live (optional device int64 scalar): the LIVE destination rows of a captured step's upper-bound block — rows behind them are
        # padding (index rows all -1, source id -1) and take the kernels' early exits
        # n_src_live (optional device int64 scalar): the LIVE source rows — the block's source list is an upper bound (a captured step
        # sized for the largest input block it can see: stepgraph's size-agnostic form), fc_pool computes the live rows only
        ctx.n_live = n_live
        x = as_mat(table)
        ids = _ids(ids)
        if (SMALL_PROJ and (ids.numel() <= SMALL_PROJ_MAX_ROWS or (n_src_live is not None and ids.numel() <= 65536))
                and w_pool.is_contiguous() and w_pool.data_ptr() % 16 == 0):
            if n_src_live is not None:
                _SMALL_AGNOSTIC["seen"] = True          # (read by stepgraph: this step's launches do not depend on the block's size)
            # the product itself on the exact-fp32 MFMA with 32 x 64 tiles (ogl_small_proj_rows): at these sizes the general kernel's
            # 64 x 64 tiles leave most of the chip idle and convert their operands on the fly (25 us for 0.38 GFLOP, pubmed-like rung)
            p = empty_mat(ids.numel(), w_pool.shape[0], x.device)
            _launch("ogl_small_proj_rows", _lib.lib().ogl_small_proj_rows, _ptr(x), _ld(x), _ptr(ids), x.shape[0], ids.numel(), x.shape[1],
                    _ptr(w_pool), _ld(as_mat(w_pool)), w_pool.shape[0], _ptr(b_pool), 1, _ptr(p), _ld(p), _ptr(n_src_live), _stream(),
                    meta=dict(M=ids.numel(), N=w_pool.shape[0], K=x.shape[1]))
        else:
            p = linear_fwd(x, w_pool, b_pool, relu=True, x_rows=ids)
        n_src, F = p.shape
        H = w_self.shape[0]
        dev = p.device
        need_pool = w_pool.requires_grad or (b_pool is not None and b_pool.requires_grad)
        neigh = empty_mat(n_dst, F, dev)
        argmax = torch.empty((n_dst, F), dtype=torch.int32, device=dev)
        y = empty_mat(n_dst, H, dev)
        _launch("ogl_small_first_layer_fwd", _lib.lib().ogl_small_first_layer_fwd, _ptr(p), _ld(p), n_src, _ptr(idx), n_dst, int(idx.shape[1]), F,
                _ptr(x), _ld(x), _ptr(ids), x.shape[0], _ptr(w_self), _ld(as_mat(w_self)), _ptr(b_self), _ptr(w_neigh), _ld(as_mat(w_neigh)),
                _ptr(b_neigh), H, int(bool(relu)), _ptr(neigh), _ld(neigh), _ptr(argmax), _ptr(y), _ld(y), _ptr(n_live), _stream(),
                meta=dict(n_src=n_src, n_dst=n_dst, fanout=int(idx.shape[1]), d=F, H=H))
        if _CAPTURE is not None:
            _CAPTURE.append(dict(argmax=argmax, neigh=neigh))
        ctx.n_src, ctx.n_dst, ctx.fanout, ctx.relu = n_src, n_dst, int(idx.shape[1]), bool(relu)
        ctx.has_bias, ctx.has_pool_bias = b_self is not None, b_pool is not None
        ctx.bias_t = b_pool
        # fc_pool's weight gradient is _PoolMaxFn's: its planned image path (the gradient-free half starts here) or the scatter path
        ctx.rec_path = bool(need_pool and SMALL_FIRST_DW and n_dst <= 2048 and 4 * n_dst * F * F <= SMALL_FIRST_DW_MAX_BYTES)
        ctx.x3_path = bool(need_pool and not ctx.rec_path and _MODE["name"] != "f32" and n_src >= X3_BWW_MIN_ROWS and F <= 640
                           and ctx.fanout <= 63)
        ctx.pool_plan = None
        if ctx.x3_path and POOL_PLAN and n_dst * F < (1 << 27):
            ctx.pool_plan = pool_bwd_x3_plan(argmax, neigh, idx, n_src)
        ctx.dp_slot = None
        if (need_pool and not ctx.x3_path and not ctx.rec_path
                and max(n_src, 1) * padded_ld(F) <= (SMALL_LOSS_ZERO_MAX if SMALL_LOSS_FUSED else CE_SMALL_MAX_ZERO)):
            ctx.dp_slot = request_zeroed(n_src, F, dev)            # (the scatter target: cleared by the loss launch on the side)
        if relu:
            y._ogl_relu_out = True
        ctx.save_for_backward(x, ids, w_pool, w_self, w_neigh, neigh, argmax, y if relu else None, idx)
        return y

    @staticmethod
    def backward(ctx, dout):
        x, ids, w_pool, w_self, w_neigh, neigh, argmax, y, idx = ctx.saved_tensors
        route = _PENDING_ROUTES.pop(dout.data_ptr(), None)      # (the small last layer's backward, handed over: see _SmallPoolLossFn)
        dout = as_mat(dout)
        n_src, n_dst = ctx.n_src, ctx.n_dst
        F, H = neigh.shape[1], w_self.shape[0]
        dev = dout.device
        need = ctx.needs_input_grad                               # (table, ids, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, ...)
        need_pool = need[2] or (need[3] and ctx.has_pool_bias)
        dy = empty_mat(n_dst, H, dev)
        dneigh = dP = None
        if need_pool and (ctx.x3_path or ctx.rec_path):
            dneigh = empty_mat(n_dst, F, dev)
        elif need_pool:
            slot, ctx.dp_slot = ctx.dp_slot, None
            dP = take_zeroed(slot, n_src, F) if slot is not None else \
                fill_zero(torch.empty((max(n_src, 1), padded_ld(F)), dtype=torch.float32, device=dev))[:n_src, :F]
        else:
            dneigh = empty_mat(n_dst, F, dev)                      # (nobody reads it: the kernel writes at least one of the two)
        _launch("ogl_small_first_layer_bwd", _lib.lib().ogl_small_first_layer_bwd, _ptr(dout), _ld(dout), _ptr(y), _ld(y) if y is not None else 0,
                int(ctx.relu), n_dst, H, F, _ptr(w_neigh), _ld(as_mat(w_neigh)), _ptr(neigh), _ld(neigh), _ptr(argmax), _ptr(dy), _ld(dy),
                _ptr(dneigh), _ld(dneigh) if dneigh is not None else 0, _ptr(dP), _ld(dP) if dP is not None else 0, n_src,
                1 if (need_pool and ctx.rec_path) else 0,
                _ptr(route["argmax"]) if route else None, _ptr(route["G"]) if route else None, route["n_head"] * route["hin"] if route else 0,
                route["hin"] if route else 0, _ptr(route["w_pool"]) if route else None, _ld(route["w_pool"]) if route else 0,
                _ptr(route["head"]) if route else None, _ld(route["head"]) if route else 0, route["n_head"] if route else 0,
                _ptr(ctx.n_live), _stream(),
                meta=dict(n_src=n_src, n_dst=n_dst, d=F, H=H, scatter=dP is not None, routed=route is not None))
        # the combine's two weight gradients (+ both bias gradients) and — on the record path — fc_pool's, ONE launch
        dws = dwn = db = db2 = dwp = dbp = None
        rec = need_pool and ctx.rec_path
        if need[4] or need[5] or rec:
            if need[4] or need[5]:
                dws = _dw_out(w_self, *w_self.shape)
                dwn = _dw_out(w_neigh, *w_neigh.shape)
                dws = dws if dws is not None else torch.empty((H, F), dtype=torch.float32, device=dev)
                dwn = dwn if dwn is not None else torch.empty((H, F), dtype=torch.float32, device=dev)
                if ctx.has_bias:
                    db = torch.empty(H, dtype=torch.float32, device=dev)
                    db2 = torch.empty(H, dtype=torch.float32, device=dev)
            if rec:
                # from the winners' records: no scatter target, no transposed operands, no plan; sums in destination order
                dwp = _dw_out(w_pool, *w_pool.shape)
                dwp = dwp if dwp is not None else torch.empty((F, F), dtype=torch.float32, device=dev)
                dbp = torch.empty(F, dtype=torch.float32, device=dev) if ctx.has_pool_bias else None
            _record_launch(rec, dneigh, argmax, dy, n_dst, F, H, x, ids, n_src, neigh, dwp, dbp, dws, db, dwn, db2, route, ctx.n_live)
            route = None
        if route is not None:                                  # (nothing of this layer's to sum: the handed-over groups alone)
            _record_launch(False, None, None, None, n_dst, F, H, x, ids, n_src, neigh, None, None, None, None, None, None, route)
        if need_pool and not rec and ctx.x3_path:
            import types
            shim = types.SimpleNamespace(saved_tensors=(x, w_pool, ids, neigh, argmax, idx), needs_input_grad=(False, need[2], need[3], False, False),
                                         n_src=n_src, fanout=ctx.fanout, has_bias=ctx.has_pool_bias, pool_plan=ctx.pool_plan, dp_slot=None,
                                         bias_t=ctx.bias_t)
            ctx.pool_plan = None
            _, dwp, dbp, _, _ = _PoolMaxFn.backward(shim, dneigh)
        elif need_pool and not rec:
            dwp, dbp = weight_grad(dP, x, ids, want_bias=ctx.has_pool_bias, dw_out=_dw_out(w_pool, *w_pool.shape))
        return (None, None, dwp, dbp if ctx.has_pool_bias else None, dws, dwn, db if ctx.has_bias else None,
                db2 if ctx.has_bias else None, None, None, None, None, None)


def _record_launch(rec, dneigh, argmax, dy, n_dst, F, H, x, ids, n_src, neigh, dwp, dbp, dws, db, dwn, db2, route, n_live=None):
    """ONE ogl_record_weight_grads launch: the small first layer's row groups (fc_pool from the winners' records when ``rec``; fc_self,
    fc_neigh) and, when the small last layer handed its backward over (``route``), that layer's three + the deferred mean loss + the
    optimiser's per-step scalars."""
    segs = []

    def seg(G, arg, ldarg, n_idx, idp, rows, n_rows, Fw, nd, n_out, dW, b1_, b2_, live=None):
        sg = _lib.RecSeg()
        sg.n_live = _ptr(live)
        sg.G, sg.ldg = _ptr(G), _ld(as_mat(G)) if G.dim() == 2 else n_out
        sg.arg, sg.ldarg, sg.n_idx = _ptr(arg), ldarg, n_idx
        sg.ids, sg.rows, sg.ldr, sg.n_rows, sg.F = _ptr(idp), _ptr(rows), _ld(rows), n_rows, Fw
        if isinstance(dW, torch.Tensor):
            sg.dW, sg.lddw, sg.db, sg.db2 = _ptr(dW), _ld(as_mat(dW)), _ptr(b1_), _ptr(b2_)
        else:                                                   # (raw addresses of contiguous [n_out, Fw] / [n_out] gradients: a route's)
            sg.dW, sg.lddw, sg.db, sg.db2 = dW, Fw, b1_, b2_
        sg.n_dst, sg.n_out = nd, n_out
        segs.append(sg)
    if rec:
        seg(dneigh, argmax, F, n_src, ids, x, x.shape[0], F, n_dst, F, dwp, dbp, None, n_live)
    if dws is not None:
        seg(dy, None, 0, 0, ids, x, x.shape[0], F, n_dst, H, dws, db, None, n_live)
    if dwn is not None:
        seg(dy, None, 0, 0, None, neigh, n_dst, F, n_dst, H, dwn, db2, None, n_live)
    rows_l = mean_l = None
    step_dev = scal = None
    lr = b1 = b2 = 0.0
    if route is not None:
        nh, hin, hout = route["n_head"], route["hin"], route["hout"]
        rdwp, rdbp, rdws, rdbs, rdwn, rdbn = route["out"]
        h1 = route["h"]
        seg(route["G"].view(nh, hin) if route["G"].numel() == nh * hin else route["G"][:nh * hin].view(nh, hin), route["argmax"], hin, h1.shape[0],
            None, h1, h1.shape[0], hin, nh, hin, rdwp, rdbp, None)
        seg(route["dl"], None, 0, 0, None, h1, h1.shape[0], hin, nh, hout, rdws, rdbs, None)
        seg(route["dl"], None, 0, 0, None, route["neigh"], nh, hin, nh, hout, rdwn, rdbn, None)
        rows_l, mean_l = route["rows"], route["mean"]
        if route["prime"] is not None:
            step_dev, scal, lr, b1, b2 = route["prime"]
    arr = (_lib.RecSeg * len(segs))(*segs)
    _launch("ogl_record_weight_grads", _lib.lib().ogl_record_weight_grads, C.addressof(arr), len(segs), _ptr(rows_l),
            rows_l.numel() if rows_l is not None else 0, _ptr(mean_l), _ptr(step_dev), _ptr(scal), C.c_double(lr), C.c_double(b1), C.c_double(b2),
            _stream(), meta=dict(n_src=n_src, n_dst=n_dst, d=F, H=H, records=bool(rec), routed=route is not None, groups=len(segs)))
    if step_dev is not None:
        _ADAM_PRIME["served"] = (step_dev.data_ptr(), _capturing())


def small_first_pool_layer(table, ids, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu, n_live=None, n_src_live=None):
    return _SmallFirstLayerFn.apply(table, ids, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu,
                                    n_live if SMALL_LIVE else None, n_src_live)


def sage_pool_layer(h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu):
    if h.dim() == 2 and small_pool_layer_fits(h.shape[0], n_dst, idx.shape[1], h.shape[1], w_self.shape[0]):
        return _SmallPoolLayerFn.apply(h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu)
    return _SagePoolLayerFn.apply(h, w_pool, b_pool, w_self, w_neigh, b_self, b_neigh, idx, n_dst, relu)


def pool_max(x, w, bias, idx, x_rows=None):
    """max_j relu(fc_pool(x))[idx[:, j]] — the aggregator of the live 'pool' layer (and in-repo 'maxpool')."""
    return _PoolMaxFn.apply(x, w, bias, x_rows, idx)


def neighbor_reduce(src, idx, op):
    """Differentiable fixed-fanout reduction (``max`` | ``mean`` | ``sum``)."""
    return _ReduceFn.apply(src, idx, op)


class _SelfNeighFn(torch.autograd.Function):
    """(src[:n_dst], reduce_j src[idx[:, j]]) as ONE autograd node: a layer input that is read twice — the destinations' own rows and
    the neighbour reduction (the in-repo 'mean' layer, R/train/graphsage/pytorch/aggregator_dgl.py:156-159,199-206) — otherwise gets
    its gradient from autograd as zeros([n_src, d]) + a copy of the head rows + an add of two [n_src, d] matrices: three ATen launches,
    21 us of the Reddit-rung 'mean' step.  Here the head rows' gradient is added into the reduction's gradient in place."""

    @staticmethod
    def forward(ctx, src, idx, n_dst, op):
        out = _ReduceFn.forward(ctx, src, idx, op)
        ctx.n_dst = int(n_dst)
        head = src[:ctx.n_dst].clone() if ctx.n_dst * src.shape[1] <= (1 << 22) else src[:ctx.n_dst].contiguous()
        return head, out

    @staticmethod
    def backward(ctx, dhead, dout):
        n_dst = ctx.n_dst
        if dout is None:
            dsrc = None
        else:
            dsrc = _ReduceFn.backward(ctx, dout)[0]
        if dhead is not None:
            if dsrc is None:
                dsrc = torch.zeros((ctx.n_src, dhead.shape[1]), dtype=dhead.dtype, device=dhead.device)
            dsrc[:n_dst].add_(dhead)
        return dsrc, None, None, None


def self_and_neighbors(src, idx, n_dst, op):
    """(src[:n_dst], neighbor_reduce(src, idx, op)) from one autograd node (see ``_SelfNeighFn``)."""
    return _SelfNeighFn.apply(src, idx, int(n_dst), op)


class _CrossEntropyRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        loss, dl = ce_fwd_bwd(logits, labels, 1.0, want_grad=logits.requires_grad)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dl,) = ctx.saved_tensors
        return dl * dloss.reshape(-1, 1), None


_UNIT_GRAD = {}


def unit_grad(device):
    """A cached scalar 1.0 on ``device``: the root gradient of ``backward(loss)``."""
    key = (device.type, device.index)
    t = _UNIT_GRAD.get(key)
    if t is None:
        t = _UNIT_GRAD[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


def backward(loss):
    """loss.backward() with the cached unit root gradient: autograd then neither fills a ones_like(loss) nor does the
    mean-reduced cross entropy multiply its stored dlogits by it (two ~5 us launches per step)."""
    # (the region in which this package owns the whole backward pass: the only place where a node may hand out a gradient tensor that
    # a LATER node's launch fills — ``_SmallPoolLossFn``'s route.  A route nobody consumed by the end of the pass is an error, never
    # silently unwritten gradients.)
    _OWN_BWD["depth"] += 1
    try:
        loss.backward(unit_grad(loss.device))
    finally:
        _OWN_BWD["depth"] -= 1
        stale = bool(_PENDING_ROUTES)
        _PENDING_ROUTES.clear()
    if stale:
        raise RuntimeError("the fused last layer handed its backward to the first layer's launches (ops.SMALL_ROUTE) and that node never "
                           "ran or received a different gradient tensor (a tensor hook on the hidden rows?): the last layer's weight "
                           "gradients were NOT written.  Set OGL_SMALL_ROUTE=0 for such a graph.")
    side_join()


_OWN_BWD = {"depth": 0}


def assert_no_pending_gradients():
    """Raises when a gradient promised by a fused node was never written (checked by the optimiser and the gradient exchange)."""
    if _PENDING_ROUTES:
        _PENDING_ROUTES.clear()
        raise RuntimeError("a gradient route of the fused last layer was never consumed: its weight gradients are unwritten memory")


class _CrossEntropyMeanFn(torch.autograd.Function):
    """reduction='mean' as one node: the kernel writes dlogits already scaled by 1/B, backward is one scalar multiply
    (none when the incoming gradient is the cached unit scalar of ``backward``)."""

    @staticmethod
    def forward(ctx, logits, labels):
        B = max(logits.shape[0], 1)
        if 0 < logits.shape[0] <= CE_MEAN_SMALL_MAX_B:
            mean, _, dl = ce_fwd_bwd_mean(logits, labels, want_grad=logits.requires_grad)
            ctx.save_for_backward(dl)
            return mean
        if logits.shape[0] > 0:
            mean, _, dl = ce_fwd_bwd_mean_grid(logits, labels, want_grad=logits.requires_grad)
            ctx.save_for_backward(dl)
            return mean
        rows, dl = ce_fwd_bwd(logits, _labels_tensor(labels), 1.0 / B, want_grad=logits.requires_grad)
        ctx.save_for_backward(dl)
        return rows.mean()

    @staticmethod
    def backward(ctx, dloss):
        (dl,) = ctx.saved_tensors
        unit = _UNIT_GRAD.get((dloss.device.type, dloss.device.index))
        if unit is not None and dloss.data_ptr() == unit.data_ptr():
            return dl, None
        return dl * dloss, None


class _CrossEntropyMeanRowsFn(torch.autograd.Function):
    """(mean loss, per-seed losses) from one launch; only the mean is differentiable.  What the PBR update needs: it trains on
    the mean of the 'none'-reduced loss and exports the rows as priorities (R/train/graphsage/pytorch/model.py:198-204)."""

    @staticmethod
    def forward(ctx, logits, labels):
        B = logits.shape[0]
        if 0 < B <= CE_MEAN_SMALL_MAX_B:
            mean, rows, dl = ce_fwd_bwd_mean(logits, labels, want_grad=logits.requires_grad)
        else:
            mean, rows, dl = ce_fwd_bwd_mean_grid(logits, labels, want_grad=logits.requires_grad)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(rows)
        ctx.set_materialize_grads(False)             # (the rows' gradient stays None: no zero fill in front of backward)
        return mean, rows

    @staticmethod
    def backward(ctx, dloss, _drows):
        (dl,) = ctx.saved_tensors
        unit = _UNIT_GRAD.get((dloss.device.type, dloss.device.index))
        if unit is not None and dloss.data_ptr() == unit.data_ptr():
            return dl, None
        return dl * dloss, None


def cross_entropy_mean_rows(logits, labels):
    """(mean, rows) of nn.CrossEntropyLoss(reduction='none') in one launch; differentiate the mean."""
    if logits.shape[0] == 0:
        rows = _CrossEntropyRowsFn.apply(logits, _labels_tensor(labels))
        return rows.mean(), rows
    return _CrossEntropyMeanRowsFn.apply(logits, labels)


def cross_entropy(logits, labels, reduction="mean"):
    """nn.CrossEntropyLoss(reduction) on the HIP kernel; 'none' returns the per-seed vector."""
    if reduction == "mean":
        return _CrossEntropyMeanFn.apply(logits, labels)
    rows = _CrossEntropyRowsFn.apply(logits, _labels_tensor(labels))
    if reduction == "none":
        return rows
    if reduction == "mean":
        return rows.mean()
    if reduction == "sum":
        return rows.sum()
    raise ValueError("unknown reduction %r" % (reduction,))
