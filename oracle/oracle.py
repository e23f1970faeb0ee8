"""CPU oracle for the streaming-GraphSAGE update path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package (``online-gnn-learning_amd``) never does; it fails loudly when
the HIP library is missing.

What this restates (reference = MassimoPerini/online-gnn-learning, ``R/`` = /root/reference):

* k-hop sampling + block construction: the reference calls the *absent, unpinned* third-party DGL
  (``dgl.sampling.MultiLayerNeighborSampler([S,S], replace=True)`` + ``NodeDataLoader``,
  R/train/graphsage/pytorch/model.py:44-47,128-131,174-178,224-227,280-284,312-316).  DGL's RNG is
  never seeded by the reference (R/train/__main__.py:211-212 seeds numpy/random only), so *no*
  implementation can be bit-exact against it.  This oracle restates the published semantics
  (uniform, with replacement, exactly S picks iff in-degree>0; dst-first / first-appearance block
  relabelling) with a counter-based Philox4x32-10 stream so that the HIP path can be bit-exact
  against *it*.  PARITY UNPINNED at the DGL boundary: see DESIGN.md.  Philox itself is pinned to
  the Random123 known-answer vectors (tests/test_oracle.py).
* The live layer (DGL ``SAGEConv(aggregator_type='pool')``, imported at
  R/train/graphsage/pytorch/graphsage_dgl.py:3; parameterisation corroborated by
  R/inference_optimized.py:136-139,256-282): ``relu(fc_pool(h))`` -> elementwise max over sampled
  in-neighbours -> ``fc_self(h_dst) + fc_neigh(neigh)``.
* The in-repo layer R/train/graphsage/pytorch/aggregator_dgl.py:128-216 (``mean``, ``meanpool``,
  ``gcn``; concat -> ``fc_neigh``).  This part IS pinned: tests/golden/sageconv_*.npz were produced
  by running the reference's own ``SAGEConv.forward`` (tests/golden/make_golden.py).
* Model stack R/train/graphsage/pytorch/graphsage_dgl.py:38-59, CE loss / Adam
  R/train/graphsage/pytorch/model.py:20-25,105-107,198-202.
* Snapshot adjacency R/train/graph/dynamic_graph_vertex.py:39-94,132-141 and
  R/train/graph/dynamic_graph_edge.py:46-82,190-218 as a time-ordered CSR + prefix-degree cut.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
# Pinned by the Random123 KAT vectors in tests/test_oracle.py.
# --------------------------------------------------------------------------------------------
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK32
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


# --------------------------------------------------------------------------------------------
# Snapshot adjacency: time-ordered CSR + prefix-degree cut.
# --------------------------------------------------------------------------------------------
def snapshot_degrees(indptr, keys, n_present, cut):
    """deg_t[v] = #{e in adj(v) : keys[e] < cut} for v < n_present, else 0.

    ``keys`` is ascending inside every adjacency list.  Vertex streams use keys = neighbour id and
    cut = n_present (induced subgraph on the first n_present arrival-ordered vertices,
    R/train/graph/dynamic_graph_vertex.py:82-85,137-140); edge streams use keys = row index of the
    time-sorted edge table and cut = t * edges_per_snapshot
    (R/train/graph/dynamic_graph_edge.py:190-218).
    """
    indptr = np.asarray(indptr, dtype=np.int64)
    keys = np.asarray(keys)
    n = len(indptr) - 1
    deg = np.zeros(n, dtype=np.int32)
    for v in range(min(n, int(n_present))):
        a, b = indptr[v], indptr[v + 1]
        deg[v] = np.searchsorted(keys[a:b], cut, side="left")
    return deg


def snapshot_degrees_fast(indptr, keys, n_present, cut):
    """Vectorised equivalent of :func:`snapshot_degrees` (counts keys<cut per row)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    keys = np.asarray(keys)
    n = len(indptr) - 1
    below = (keys < cut).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(below)])
    deg = (cs[indptr[1:]] - cs[indptr[:-1]]).astype(np.int32)
    deg[int(n_present):] = 0
    return deg


# --------------------------------------------------------------------------------------------
# Sampler: uniform with replacement, fixed fanout.
# --------------------------------------------------------------------------------------------
def sample_layer(indptr, indices, deg_t, dst, fanout, seed, ctr, layer):
    """picks[i, j] = j-th sampled in-neighbour of dst[i]  (int64, -1 when deg_t[dst[i]] == 0).

    Philox counter = (j >> 2 | layer << 16, dst_lo32, dst_hi32, ctr_lo32);
    key = (seed_lo32, seed_hi32 ^ ctr_hi32); word j & 3 of the output block is the 32-bit draw r;
    pick offset = (r * deg) >> 32 (multiply-shift range reduction).
    Semantics restated: DGL sample_neighbors(fanout=S, replace=True) — exactly S picks per dst with
    in-degree > 0, none otherwise.
    """
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices)
    dst = np.asarray(dst, dtype=np.int64)
    n = len(dst)
    picks = np.full((n, fanout), -1, dtype=np.int64)
    if n == 0 or fanout == 0:
        return picks
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    ctr = int(ctr) & 0xFFFFFFFFFFFFFFFF
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) ^ (ctr >> 32)) & 0xFFFFFFFF
    j = np.arange(fanout, dtype=np.uint64)[None, :]
    c0 = (j >> np.uint64(2)) | np.uint64((int(layer) & 0xFFFF) << 16)
    d = dst.astype(np.uint64)[:, None]
    c1 = d & _MASK32
    c2 = d >> np.uint64(32)
    c3 = np.uint64(ctr & 0xFFFFFFFF)
    r = philox4x32_10(c0 + np.zeros_like(c1), c1 + np.zeros_like(c0), c2 + np.zeros_like(c0), c3, k0, k1)
    word = (np.arange(fanout) & 3)[None, :]
    r32 = np.choose(word + np.zeros((n, 1), dtype=np.int64), r).astype(np.uint64)
    deg = np.asarray(deg_t)[dst].astype(np.uint64)[:, None]
    off = (r32 * deg) >> np.uint64(32)
    base = indptr[dst][:, None]
    has = deg[:, 0] > 0
    pos = (base + off.astype(np.int64))[has]
    picks[has] = indices[pos].astype(np.int64)
    return picks


def build_block(dst, picks):
    """Bipartite block relabelling (DGL ``to_block`` semantics restated).

    src nodes = the dst nodes in order (local id i for dst[i], duplicates keep their own row), then
    every new id in first-appearance order of the row-major picks array.  Returns
    ``(src_ids int64[n_src], local_idx int32[n_dst, fanout])`` with -1 for missing picks; a picked id
    that equals a dst id maps to that dst's *first* row.
    """
    dst = np.asarray(dst, dtype=np.int64)
    picks = np.asarray(picks, dtype=np.int64)
    n_dst = len(dst)
    flat = picks.reshape(-1)
    allv = np.concatenate([dst, flat])
    valid = allv >= 0
    pos = np.nonzero(valid)[0]
    vals = allv[pos]
    uniq, first = np.unique(vals, return_index=True)
    first_pos = pos[first]                      # position of first appearance per unique id
    is_new = first_pos >= n_dst                 # ids not among the dst nodes
    order = np.argsort(first_pos[is_new], kind="stable")
    new_ids = uniq[is_new][order]
    src_ids = np.concatenate([dst, new_ids]).astype(np.int64)
    # local index per unique id
    lidx_unique = np.empty(len(uniq), dtype=np.int64)
    lidx_unique[~is_new] = first_pos[~is_new]   # first dst row holding that id
    rank = np.empty(is_new.sum(), dtype=np.int64)
    rank[order] = np.arange(len(order))
    lidx_unique[is_new] = n_dst + rank
    local = np.full(flat.shape, -1, dtype=np.int32)
    pv = flat >= 0
    local[pv] = lidx_unique[np.searchsorted(uniq, flat[pv])].astype(np.int32)
    return src_ids, local.reshape(picks.shape)


def sample_blocks(indptr, indices, deg_t, seeds, fanouts, seed, ctr):
    """Two-(or k-)layer block list, output layer sampled first (DGL MultiLayerNeighborSampler order).

    Returns ``(input_nodes, seeds, blocks)`` where ``blocks[l] = dict(src_ids, dst_ids, local_idx)``
    and blocks[-1] is the output block (dst = seeds) — the ``(input_nodes, seeds, blocks)`` triple
    of R/train/graphsage/pytorch/model.py:52,133.
    """
    seeds = np.asarray(seeds, dtype=np.int64)
    blocks = []
    dst = seeds
    for layer in reversed(range(len(fanouts))):
        picks = sample_layer(indptr, indices, deg_t, dst, fanouts[layer], seed, ctr, layer)
        src_ids, local_idx = build_block(dst, picks)
        blocks.insert(0, dict(src_ids=src_ids, dst_ids=dst, local_idx=local_idx, picks=picks))
        dst = src_ids
    return blocks[0]["src_ids"], seeds, blocks


# --------------------------------------------------------------------------------------------
# Fixed-fanout neighbour reduction.
# --------------------------------------------------------------------------------------------
def reduce_fwd(src, local_idx, op):
    """out[d] = op over rows src[local_idx[d, :]] ; rows with local_idx[d,0] < 0 give zeros.

    ``max``: elementwise max, argmax = local src row of the first slot attaining it.
    ``mean``: float32 sum in slot order 0..S-1 then true division by S (bit-exact contract with
    the HIP kernel);  ``sum``: the same without the division (gcn mode).
    """
    src = np.asarray(src, dtype=np.float32)
    li = np.asarray(local_idx)
    n_dst, S = li.shape
    D = src.shape[1]
    out = np.zeros((n_dst, D), dtype=np.float32)
    arg = np.full((n_dst, D), -1, dtype=np.int32)
    has = li[:, 0] >= 0 if S > 0 else np.zeros(n_dst, dtype=bool)
    if not has.any():
        return out, arg
    rows = src[li[has].astype(np.int64)]          # [n, S, D]
    if op == "max":
        slot = rows.argmax(axis=1)                # first max slot
        out[has] = np.take_along_axis(rows, slot[:, None, :], axis=1)[:, 0, :]
        arg[has] = np.take_along_axis(li[has][:, :, None].astype(np.int32), slot[:, None, :], axis=1)[:, 0, :]
    else:
        acc = np.zeros((rows.shape[0], D), dtype=np.float32)
        for j in range(S):
            acc = acc + rows[:, j, :]
        out[has] = acc / np.float32(S) if op == "mean" else acc
    return out, arg


# --------------------------------------------------------------------------------------------
# Layers (torch CPU fp32).
# --------------------------------------------------------------------------------------------
def _neigh_torch(p, local_idx, op, forced=None, connect_empty=False):
    """Differentiable torch version of reduce_fwd (used for gradient parity).

    ``connect_empty``: a block WITHOUT any edge still returns a result that depends on ``p`` (with zero gradient), as DGL's
    builtin ``update_all(copy_src, max)`` of the live 'pool' layer does — ``fc_pool`` then receives a ZERO gradient tensor
    and Adam counts the step; the in-repo layer's zero-edge guard (R/.../aggregator_dgl.py:151-154) instead leaves
    ``fc_pool`` without a gradient (``None``: torch.optim.Adam skips the parameter).

    ``forced`` (max only; tests of the full-size step): dict(argmax=int [n_dst, D] local source row of the winner the
    DEVICE chose, -1 = none; mask=bool [n_dst, D] the device's ReLU mask of the pooled value).  ``p`` is then the
    PRE-activation projection and the result is ``where(mask, p[argmax[d, c], c], 0)``: the same function of p wherever
    the two evaluations agree on the winner, and on an fp32 near-tie (two candidates equal to the last bits, or a pooled
    value within rounding of 0) the gradient is routed the way the device routed it instead of the way a second fp32
    evaluation happens to break the tie.  relu is monotonic, so max_j relu(p_j) = relu(max_j p_j)."""
    li = torch.as_tensor(np.asarray(local_idx), dtype=torch.long)
    n_dst, S = li.shape
    if forced is not None:
        assert op == "max"
        am = torch.as_tensor(np.asarray(forced["argmax"]), dtype=torch.long)
        mask = torch.as_tensor(np.asarray(forced["mask"]), dtype=torch.bool) & (am >= 0)
        picked = p.gather(0, am.clamp(min=0))
        return torch.where(mask, picked, torch.zeros((), dtype=p.dtype))
    has = (li[:, 0] >= 0) if S > 0 else torch.zeros(n_dst, dtype=torch.bool)
    out = p.new_zeros((n_dst, p.shape[1]))
    if has.any():
        rows = p[li[has]]                          # [n, S, D]
        if op == "max":
            red = rows.amax(dim=1)
        elif op == "mean":
            acc = rows[:, 0, :]
            for j in range(1, S):
                acc = acc + rows[:, j, :]
            red = acc / S
        else:
            acc = rows[:, 0, :]
            for j in range(1, S):
                acc = acc + rows[:, j, :]
            red = acc
        out = out.index_put((torch.nonzero(has)[:, 0],), red)
    elif connect_empty:
        out = out + p[:0].sum() * 0
    return out


def _edge_reduce(edge_feat, local_idx, op):
    """The edge-feature half of a reduced mailbox: mean / max over a destination's S edges of ``edge_feat[d, j]`` (slot order),
    zeros for a destination without edges (the reference's mailbox reduction runs over destinations that have edges only)."""
    e = torch.as_tensor(np.asarray(edge_feat), dtype=torch.float32)
    li = np.asarray(local_idx)
    has = torch.as_tensor(li[:, 0] >= 0) if li.shape[1] > 0 else torch.zeros(len(li), dtype=torch.bool)
    out = e.new_zeros((e.shape[0], e.shape[2]))
    if has.any():
        rows = e[has]
        if op == "max":
            red = rows.amax(dim=1)
        else:
            acc = rows[:, 0, :]
            for j in range(1, rows.shape[1]):
                acc = acc + rows[:, j, :]
            red = acc / rows.shape[1]
        out = out.index_put((torch.nonzero(has)[:, 0],), red)
    return out


def _lstm_last_hidden(rows, params):
    """Final hidden state of a one-layer LSTM over the mailbox rows [n, S, D] (slot order = sequence order) from zero initial
    state: R/train/graphsage/pytorch/aggregator_dgl.py:116-126 (``nn.LSTM(D, D, batch_first=True)``; the reducer returns h_n).
    Gate order of the stacked weights: input, forget, cell, output (torch.nn.LSTM)."""
    w_ih, w_hh = params["lstm.weight_ih_l0"], params["lstm.weight_hh_l0"]
    b_ih, b_hh = params["lstm.bias_ih_l0"], params["lstm.bias_hh_l0"]
    n, S, D = rows.shape
    h = rows.new_zeros((n, D))
    c = rows.new_zeros((n, D))
    for t in range(S):
        gates = F.linear(rows[:, t, :], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
        i, f, g, o = gates.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
    return h


def sageconv_forward(mode, h_src, n_dst, local_idx, params, activation=None, forced=None, dropout=None, trace=None, edge_feat=None):
    """One SAGEConv layer on a fixed-fanout block.

    mode ``pool``      — live DGL layer (max; fc_pool in->in; fc_self + fc_neigh).
    mode ``meanpool``/``maxpool``/``mean``/``gcn``/``lstm`` — R/train/graphsage/pytorch/aggregator_dgl.py:156-206
    (fc_pool in->pool_feats; ``fc_neigh(cat(h_self, h_neigh))``; gcn: ``(sum + h_dst)/(deg+1)``).
    ``params``: dict of torch tensors named like the reference state_dict
    (fc_pool.weight, fc_pool.bias, fc_self.*, fc_neigh.*).
    ``forced`` (mode ``pool``): the device's winners / ReLU masks, see :func:`_neigh_torch`; an optional
    ``act_mask`` replaces the output ReLU by the device's mask of it (any mode).  In-repo pooling modes: an optional ``pool_mask``
    (bool [n_src, pool_feats]) replaces ``relu(fc_pool(h))`` by ``where(pool_mask, fc_pool(h), 0)`` — the same function wherever the
    two evaluations agree on the sign, and a unit within rounding of 0 is routed the way the device routed it.
    ``edge_feat`` ([n_dst, S, E], in-repo modes ``mean`` / ``meanpool`` / ``maxpool`` only): the message of edge (d, j) is
    ``cat(h_src[idx[d, j]], edge_feat[d, j])`` (R/.../aggregator_dgl.py:7-13), so the reduced vector gains E columns and
    ``fc_neigh`` takes ``cat(h_self, reduce(h), reduce(e))`` (``Linear(in_neigh + E + in, out)``, :94).
    ``trace`` (a list): appends dict(pool_mask=..., act_mask=...) — THIS evaluation's own ReLU decisions (numpy bool, None where the
    layer has none) — so that a test can count where two evaluations disagree.
    ``dropout``: dict(p, seed, ctr) — ``feat_drop`` on the layer input (R/.../graphsage_dgl.py:41 passes
    ``feat_drop=dropout``), with the counter-based mask of :func:`dropout_mask`.
    """
    if dropout is not None and dropout["p"] > 0:
        keep = torch.as_tensor(dropout_mask(h_src.shape[0], h_src.shape[1], dropout["p"], dropout["seed"], dropout["ctr"]))
        h_src = torch.where(keep, h_src / np.float32(1.0 - dropout["p"]), torch.zeros((), dtype=h_src.dtype))
    h_dst = h_src[:n_dst]
    li = np.asarray(local_idx)
    own_pool_mask = None
    if mode == "pool":
        pre = F.linear(h_src, params["fc_pool.weight"], params["fc_pool.bias"])
        if forced is not None:
            neigh = _neigh_torch(pre, li, "max", forced)
        else:
            neigh = _neigh_torch(F.relu(pre), li, "max", connect_empty=True)
        rst = F.linear(h_dst, params["fc_self.weight"], params["fc_self.bias"]) + \
            F.linear(neigh, params["fc_neigh.weight"], params["fc_neigh.bias"])
    elif mode in ("meanpool", "maxpool"):
        pre = F.linear(h_src, params["fc_pool.weight"], params["fc_pool.bias"])
        own_pool_mask = (pre.detach() > 0).numpy()
        if forced is not None and forced.get("pool_mask") is not None:
            p = torch.where(torch.as_tensor(np.asarray(forced["pool_mask"]), dtype=torch.bool), pre, torch.zeros((), dtype=pre.dtype))
        else:
            p = F.relu(pre)
        neigh = _neigh_torch(p, li, "mean" if mode == "meanpool" else "max")
        if edge_feat is not None:
            neigh = torch.cat((neigh, _edge_reduce(edge_feat, li, "mean" if mode == "meanpool" else "max")), 1)
        rst = F.linear(torch.cat((h_dst, neigh), 1), params["fc_neigh.weight"], params["fc_neigh.bias"])
    elif mode == "mean":
        neigh = _neigh_torch(h_src, li, "mean")
        if edge_feat is not None:
            neigh = torch.cat((neigh, _edge_reduce(edge_feat, li, "mean")), 1)
        rst = F.linear(torch.cat((h_dst, neigh), 1), params["fc_neigh.weight"], params["fc_neigh.bias"])
    elif mode == "lstm":
        # (aggregator_dgl.py:195-199: the LSTM reducer over each destination's mailbox; a destination without edges keeps zeros)
        lt = torch.as_tensor(li, dtype=torch.long)
        has = (lt[:, 0] >= 0) if lt.shape[1] > 0 else torch.zeros(len(lt), dtype=torch.bool)
        neigh = h_src.new_zeros((n_dst, h_src.shape[1]))
        if has.any():
            neigh = neigh.index_put((torch.nonzero(has)[:, 0],), _lstm_last_hidden(h_src[lt[has]], params))
        rst = F.linear(torch.cat((h_dst, neigh), 1), params["fc_neigh.weight"], params["fc_neigh.bias"])
    elif mode == "gcn":
        s = _neigh_torch(h_src, li, "sum")
        S = li.shape[1]
        deg = torch.as_tensor((li[:, 0] >= 0).astype(np.float32) * S if S > 0 else np.zeros(len(li), np.float32))
        neigh = (s + h_dst) / (deg.unsqueeze(-1) + 1)
        rst = F.linear(neigh, params["fc_neigh.weight"], params["fc_neigh.bias"])
    else:
        raise KeyError("Aggregator type {} not recognized.".format(mode))
    if trace is not None:
        trace.append(dict(pool_mask=own_pool_mask, act_mask=(rst.detach() > 0).numpy() if activation is not None else None))
    if activation is not None:
        if forced is not None and forced.get("act_mask") is not None:
            rst = torch.where(torch.as_tensor(np.asarray(forced["act_mask"]), dtype=torch.bool), rst,
                              torch.zeros((), dtype=rst.dtype))
        else:
            rst = activation(rst)
    return rst


def graphsage_forward(mode, x, blocks, layer_params, forced=None, dropout=None, trace=None):
    """R/train/graphsage/pytorch/graphsage_dgl.py:48-59 — relu on every layer but the last.
    ``forced`` / ``dropout``: per-layer lists (entries may be None), see :func:`sageconv_forward`; ``trace``: one entry per layer."""
    h = x
    L = len(layer_params)
    for l, (blk, prm) in enumerate(zip(blocks, layer_params)):
        h = sageconv_forward(mode, h, len(blk["dst_ids"]), blk["local_idx"], prm,
                             activation=F.relu if l < L - 1 else None,
                             forced=forced[l] if forced is not None else None,
                             dropout=dropout[l] if dropout is not None else None, trace=trace)
    return h


def dropout_mask(rows, cols, p, seed, ctr):
    """keep[i, j] of the counter-based dropout the HIP path uses for ``feat_drop`` (ogl_dropout_rows): Philox4x32-10 with
    counter (j >> 2, i_lo32, i_hi32, ctr_lo32), key (seed_lo32, seed_hi32 ^ ctr_hi32) — the sampler's keying with the
    output row in place of the vertex id —, word j & 3 is the 32-bit draw r; keep iff r >= floor(p * 2**32).
    (torch's nn.Dropout draws from an unseeded-by-the-reference global stream, R/train/__main__.py:211-212; only the
    distribution — Bernoulli(1 - p), scaled by 1 / (1 - p) — is the reference's.)"""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    ctr = int(ctr) & 0xFFFFFFFFFFFFFFFF
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) ^ (ctr >> 32)) & 0xFFFFFFFF
    j = np.arange(cols, dtype=np.uint64)[None, :]
    i = np.arange(rows, dtype=np.uint64)[:, None]
    z = np.zeros((rows, cols), dtype=np.uint64)
    r = philox4x32_10((j >> np.uint64(2)) + z, (i & _MASK32) + z, (i >> np.uint64(32)) + z, np.uint64(ctr & 0xFFFFFFFF), k0, k1)
    word = (np.arange(cols) & 3)[None, :] + np.zeros((rows, 1), dtype=np.int64)
    r32 = np.choose(word, r).astype(np.uint64)
    thr = np.uint64(min(int(float(p) * 4294967296.0), 0xFFFFFFFF))
    return r32 >= thr


def cross_entropy(logits, labels, reduction="mean"):
    """nn.CrossEntropyLoss(reduction) — R/train/graphsage/pytorch/model.py:20,105,147,198."""
    return F.cross_entropy(logits, labels.flatten(), reduction=reduction)


def init_layer_params(mode, in_feats, out_feats, pool_feats=None, gen=None):
    """Parameter shapes + init of one layer.

    ``pool``: DGL SAGEConv — fc_pool Linear(in,in), fc_self/fc_neigh Linear(in,out), xavier_uniform
    (gain sqrt 2) on all three weights, default nn.Linear bias init.
    in-repo modes: R/train/graphsage/pytorch/aggregator_dgl.py:79-114.
    """
    gain = torch.nn.init.calculate_gain("relu")
    prm = {}

    def lin(name, i, o):
        m = torch.nn.Linear(i, o)
        torch.nn.init.xavier_uniform_(m.weight, gain=gain)
        prm[name + ".weight"] = m.weight.detach().clone()
        prm[name + ".bias"] = m.bias.detach().clone()

    if mode == "pool":
        lin("fc_pool", in_feats, in_feats)
        lin("fc_self", in_feats, out_feats)
        lin("fc_neigh", in_feats, out_feats)
    elif mode in ("meanpool", "maxpool"):
        pf = pool_feats if pool_feats is not None else in_feats
        lin("fc_pool", in_feats, pf)
        lin("fc_neigh", pf + in_feats, out_feats)
    elif mode == "mean":
        lin("fc_neigh", 2 * in_feats, out_feats)
    elif mode == "gcn":
        lin("fc_neigh", in_feats, out_feats)
    elif mode == "lstm":
        m = torch.nn.LSTM(in_feats, in_feats, batch_first=True)          # (default init: aggregator_dgl.py:108-109)
        for k, v in m.state_dict().items():
            prm["lstm." + k] = v.detach().clone()
        lin("fc_neigh", 2 * in_feats, out_feats)
    else:
        raise KeyError("Aggregator type {} not recognized.".format(mode))
    return prm


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (R/train/graphsage/pytorch/model.py:24-25), in place."""
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
    return p


# --------------------------------------------------------------------------------------------
# Whole training step (the CPU baseline "port" timed by bench.py next to the GPU numbers).
# --------------------------------------------------------------------------------------------
class CpuModel:
    """2-layer GraphSAGE in torch-CPU with autograd + torch.optim.Adam(lr=1e-3)."""

    def __init__(self, mode, in_feats, n_hidden, n_classes, pool_feats=None, seed=1):
        torch.manual_seed(seed)
        self.mode = mode
        self.params = [init_layer_params(mode, in_feats, n_hidden, pool_feats),
                       init_layer_params(mode, n_hidden, n_classes, pool_feats)]
        for prm in self.params:
            for k in prm:
                prm[k].requires_grad_(True)
        self.opt = torch.optim.Adam([t for prm in self.params for t in prm.values()], lr=1e-3)

    def forward(self, x, blocks, forced=None, dropout=None, trace=None):
        return graphsage_forward(self.mode, x, blocks, self.params, forced=forced, dropout=dropout, trace=trace)

    def loss_and_grads(self, feat, labels, indptr, indices, deg_t, seeds, fanout, seed, ctr, forced=None, trace=None):
        """Loss and gradients of ONE batch WITHOUT the optimiser step (the weights stay): (loss, {"layers.i.name": grad clone})."""
        input_nodes, seeds, blocks = sample_blocks(indptr, indices, deg_t, seeds, [fanout, fanout], seed, ctr)
        x = feat[torch.as_tensor(input_nodes)]
        y = labels[torch.as_tensor(seeds)]
        self.opt.zero_grad()
        loss = cross_entropy(self.forward(x, blocks, forced=forced, trace=trace), y, "mean")
        loss.backward()
        grads = {"layers.%d.%s" % (i, k): v.grad.detach().clone() for i, prm in enumerate(self.params) for k, v in prm.items()}
        self.opt.zero_grad()
        return float(loss.detach()), grads

    def train_step(self, feat, labels, indptr, indices, deg_t, seeds, fanout, seed, ctr, forced=None):
        input_nodes, seeds, blocks = sample_blocks(indptr, indices, deg_t, seeds, [fanout, fanout], seed, ctr)
        x = feat[torch.as_tensor(input_nodes)]
        y = labels[torch.as_tensor(seeds)]
        self.opt.zero_grad()
        logits = self.forward(x, blocks, forced=forced)
        loss = cross_entropy(logits, y, "mean")
        loss.backward()
        self.opt.step()
        return float(loss.detach())

    def seed_losses(self, feat, labels, indptr, indices, deg_t, seeds, fanout, seed, ctr):
        """Per-seed CE (reduction='none') of ONE inference batch: the body of the PBR priority forward,
        R/train/graphsage/pytorch/model.py:229-248 (the reference builds and discards an autograd graph; no_grad here).
        Returns (losses float32 [B], logits [B, C])."""
        input_nodes, seeds, blocks = sample_blocks(indptr, indices, deg_t, seeds, [fanout, fanout], seed, ctr)
        with torch.no_grad():
            logits = self.forward(feat[torch.as_tensor(input_nodes)], blocks)
            rows = cross_entropy(logits, labels[torch.as_tensor(seeds)], "none")
        return rows.numpy(), logits.numpy()


# --------------------------------------------------------------------------------------------
# Vertex-stream snapshots on the host + the no-rehearsal loop (BASELINE config 1: "Pubmed pytorch CPU,
# no-rehearsal, depth=2 samples=10 batch=32 (plumbing, no GPU)").
# --------------------------------------------------------------------------------------------
class HostVertexStream:
    """Snapshots of a vertex stream as the reference forms them (R/train/graph/dynamic_graph_vertex.py:39-94,132-141):
    vertices sorted by timestamp, chunked into ``snapshots`` groups of ``N // snapshots``; snapshot t is the induced
    subgraph on the first t groups, vertex ids = positions in the time-sorted list.  Restated as ONE arrival-ordered CSR
    (in-neighbour lists sorted by arrival id) + ``n_present(t)``; ``deg(t)`` is the prefix-degree cut."""

    def __init__(self, n, src, dst, order, snapshots, feat, labels):
        order = np.asarray(order, dtype=np.int64)            # arrival position -> original id
        inv = np.empty(n, dtype=np.int64)
        inv[order] = np.arange(n)
        s = np.concatenate([inv[np.asarray(src)], inv[np.asarray(dst)]])      # both directions, arrival ids
        d = np.concatenate([inv[np.asarray(dst)], inv[np.asarray(src)]])
        o = np.lexsort((s, d))                               # rows by dst, neighbours ascending inside a row
        self.indices = s[o].astype(np.int32)
        self.indptr = np.concatenate([[0], np.cumsum(np.bincount(d, minlength=n))]).astype(np.int64)
        self.n, self.order = n, order
        self.per = int(n / snapshots)
        self.feat = torch.as_tensor(np.asarray(feat))[torch.as_tensor(order)].float().contiguous()
        self.labels = torch.as_tensor(np.asarray(labels).reshape(-1)[order]).reshape(-1, 1)
        self.t = 1

    @property
    def n_present(self):
        return min(self.t * self.per, self.n)

    def arrivals(self):
        """Snapshot ids of the vertices added by the latest snapshot."""
        return np.arange((self.t - 1) * self.per, self.n_present, dtype=np.int64)

    def degrees(self):
        return snapshot_degrees_fast(self.indptr, self.indices, self.n_present, self.n_present)

    def evolve(self):
        self.t += 1


def no_rehearsal_stream(stream, model, fanout, seeds_per_snapshot, seed, ctr0=0, after_step=None):
    """The no-rehearsal strategy's loop body over consecutive snapshots (R/train/graphsage/pytorch/model.py:300-323 +
    R/train/__main__.py:161-196): per snapshot ONE batch of the newly arrived train vertices (``seeds_per_snapshot[t]``,
    snapshot ids, in the order the loader sees them — the reference shuffles them with an unseeded RNG, so the caller
    fixes the order), sampled at ``[fanout, fanout]``, one Adam step; then ``evolve``.  Returns the per-snapshot losses.
    ``after_step(t, model)`` (tests) runs after each snapshot's step."""
    losses = []
    for t, sd in enumerate(seeds_per_snapshot):
        if len(sd) >= 2:                                     # the reference returns early below two new nodes (:308-309)
            losses.append(model.train_step(stream.feat, stream.labels, stream.indptr, stream.indices, stream.degrees(),
                                           np.asarray(sd, dtype=np.int64), fanout, seed, ctr0 + len(losses)))
            if after_step is not None:
                after_step(t, model)
        stream.evolve()
    return losses
