"""GraphSAGE layer on fixed-fanout blocks, every arithmetic step a libogl_hip kernel.

Two parameterisations, selected by ``aggregator_type`` exactly as in the reference:

* ``'pool'`` — the layer the live ``backend=pytorch`` path trains: DGL's
  ``SAGEConv(aggregator_type='pool')`` (imported at R/train/graphsage/pytorch/graphsage_dgl.py:3,
  constructed at :41-46 with ``"pool"`` from R/train/__main__.py:124-127).  ``fc_pool: in->in``,
  ReLU, elementwise MAX over the sampled in-neighbours, ``fc_self(h_dst) + fc_neigh(neigh)``.
  Parameter names/shapes corroborated by R/inference_optimized.py:136-139,260,276.
* ``'mean' | 'gcn' | 'meanpool' | 'maxpool' | 'lstm'`` — the in-repo layer
  R/train/graphsage/pytorch/aggregator_dgl.py:49-216: ``fc_pool: in->pool_feats`` (pool modes),
  reduce, ``fc_neigh(cat(h_self, h_neigh))`` (gcn: ``fc_neigh((sum + h_dst)/(deg+1))``).
  ``cat -> Linear`` runs as ONE dual-input GEMM over two column slices of ``fc_neigh.weight``.
  (``maxpool`` is implemented as the elementwise max it documents; the reference's
  ``.max(axis=1)`` at :175 returns a namedtuple and fails.)  ``'lstm'``: the library LSTM (MIOpen) over the gathered mailbox.
Unknown types raise ``KeyError`` from ``forward`` like the reference (:196-197).
"""
from __future__ import annotations

import os

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops


class GatheredRows:
    """Lazy ``table[ids]``: the row gather is fused into the consuming GEMM / reduction loaders so
    ``graph.ndata['feat'][input_nodes]`` (R/.../pytorch/model.py:54,88,182,232) is never materialised."""

    def __init__(self, table: torch.Tensor, ids, proj: torch.Tensor = None):
        # proj (optional, inference only): the (P0, S0) pair of SAGEConv.project_tables for EVERY table row, computed
        # once per weight version.  With it the first layer reduces straight from P0 through the block's global picks,
        # takes its self term from S0, and `ids` may be None (the input block then needs no relabelling at all).
        self.table, self.ids, self.proj = table, ids, proj

    @property
    def shape(self):
        return (self.ids.numel(), self.table.shape[1])

    def head(self, n):
        return GatheredRows(self.table, self.ids[:n], self.proj)

    def materialize(self):
        return ops.gather_rows(self.table, self.ids)


# inference self term S0[dst]: from this output width on it is added by the neighbour GEMM (accumulators start at the table row:
# half the reduction — Reddit H = 600: 0.20 -> 0.12 ms per 1 024-seed batch); below it the identity-weight dual product wins
# (arxiv H = 32: 25 vs 36 us — the gathered 128-byte rows cost more than 32 extra reduction steps)
ADDROWS_MIN_WIDTH = 128
# the cached layers' aggregator writes the image of the pooled rows only (OGL_IMAGE_ONLY_POOL=0: the fp32 rows too, which nobody reads)
POOL_FP32_OUT = False
# in-repo 'mean' layers behind the first: the destinations' own rows and the neighbour mean from ONE autograd node (ops._SelfNeighFn)
SELF_NEIGH_NODE = True
_EYE = {}


def _eye(n, device):
    key = (n, device.type, device.index)
    t = _EYE.get(key)
    if t is None:
        t = _EYE[key] = ops.empty_mat(n, n, device).copy_(torch.eye(n, dtype=torch.float32, device=device))
    return t


def _is_relu(fn):
    return fn is F.relu or fn is torch.relu or isinstance(fn, nn.ReLU)


class SAGEConv(nn.Module):
    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0., bias=True, norm=None, edge_feats=None,
                 activation=None, pool_feats=None):
        super().__init__()
        self._in_feats, self._out_feats, self._aggre_type = in_feats, out_feats, aggregator_type
        self.norm, self.activation = norm, activation
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_pool = self.fc_self = self.fc_neigh = None
        edge_feats = int(edge_feats or 0)
        if edge_feats and aggregator_type not in ("mean", "meanpool", "maxpool"):
            # (the reference's own gcn / lstm reducers do not survive edge features either: the gcn sum is added to h_dst of another
            # width, the LSTM's initial state has in_feats columns — aggregator_dgl.py:123-125,165-167; the live DGL layer has none)
            raise NotImplementedError("edge features: 'mean', 'meanpool' and 'maxpool' only")
        self._edge_feats = edge_feats
        self.lstm = None
        if aggregator_type == "lstm" and in_feats > 0:
            # (aggregator_dgl.py:76-77; outside the hot path — no setting file uses it: the library LSTM (MIOpen) over the gathered
            # mailbox, no kernel of this package's own)
            self.lstm = nn.LSTM(in_feats, in_feats, batch_first=True)
        in_neigh = in_feats
        if aggregator_type == "pool":
            self.fc_pool = nn.Linear(in_feats, in_feats)
            self.fc_self = nn.Linear(in_feats, out_feats, bias=bias)
            self.fc_neigh = nn.Linear(in_feats, out_feats, bias=bias)
        else:
            if pool_feats is not None and aggregator_type in ("maxpool", "meanpool"):
                in_neigh = pool_feats
            if aggregator_type in ("maxpool", "meanpool"):
                self.fc_pool = nn.Linear(in_feats, in_neigh)
            if aggregator_type == "gcn":
                self.fc_neigh = nn.Linear(in_neigh, out_feats, bias=bias)
            else:
                # fc_neigh(cat(h_self, reduce(cat(h, e)))): Linear(in_neigh + edge + in, out) (aggregator_dgl.py:94)
                self.fc_neigh = nn.Linear(in_neigh + edge_feats + in_feats, out_feats, bias=bias)
        self.in_neigh_feats = in_neigh
        self.reset_parameters()

    def reset_parameters(self):
        """xavier_uniform(gain=sqrt 2) on every weight, default Linear bias init (aggregator_dgl.py:99-114;
        DGL's SAGEConv does the same for fc_pool / fc_self / fc_neigh)."""
        gain = nn.init.calculate_gain("relu")
        for lin in (self.fc_pool, self.fc_self, self.fc_neigh):
            if lin is not None:
                nn.init.xavier_uniform_(lin.weight, gain=gain)
        if self.lstm is not None:
            self.lstm.reset_parameters()

    # ------------------------------------------------------------------------------------------
    def _project(self, lin, x, relu=False, x2=None, w2=None, bias=None, w=None):
        w = lin.weight if w is None else w
        b = lin.bias if bias is None else bias
        if isinstance(x, GatheredRows):
            return ops.linear(x.table, w, b, x2, w2, relu, x.ids, None)
        return ops.linear(x, w, b, x2, w2, relu, None, None)

    def forward(self, graph, feat):
        t = self._aggre_type
        if t not in ("pool", "mean", "gcn", "meanpool", "maxpool", "lstm"):
            raise KeyError("Aggregator type {} not recognized.".format(t))
        lazy = isinstance(feat, GatheredRows)
        if self.training and self.feat_drop.p > 0:
            # feat_drop on the layer input (graphsage_dgl.py:41): one HIP launch, fused with the row gather when the input
            # is a lazy table[ids] (the layer then runs on the materialised, dropped rows)
            if lazy and feat.proj is not None:
                raise RuntimeError("the cached-projection path is inference only (model.eval())")
            feat = ops.dropout(feat.table, self.feat_drop.p, rows=feat.ids) if lazy else ops.dropout(feat, self.feat_drop.p)
            lazy = False
        n_dst = graph.number_of_dst_nodes()
        idx = graph.local_idx
        fuse_relu = _is_relu(self.activation)
        if lazy and feat.proj is not None:
            return self._forward_cached(graph, feat, fuse_relu)
        dst_pos = getattr(graph, "dst_pos", None)
        if dst_pos is not None:
            return self._forward_fused_batches(graph, feat, idx, dst_pos, fuse_relu)
        feat_dst = feat.head(n_dst) if lazy else feat[:n_dst]

        if (t == "pool" and not lazy and self.norm is None and (self.activation is None or fuse_relu)
                and (self.fc_self.bias is None) == (self.fc_neigh.bias is None) and idx.dtype == torch.int32):
            if not torch.is_grad_enabled():
                if ops.small_pool_layer_fits(feat.shape[0], n_dst, idx.shape[1], feat.shape[1], self._out_feats):
                    return ops.small_pool_layer_fwd(feat, self.fc_pool.weight, self.fc_pool.bias, self.fc_self.weight,
                                                    self.fc_neigh.weight, self.fc_self.bias, self.fc_neigh.bias, idx, n_dst,
                                                    fuse_relu, want_argmax=False)[0]
                # inference: the same three launches without the autograd node, the summed bias from the cache
                imgs = getattr(self, "_pass_images", None)
                himg = ops.take_image(feat) if (imgs and "w_pool_b" in imgs) else None
                if himg is not None and himg.K == imgs["w_pool_b"].K:
                    p = ops.linear_fwd_x3(himg, None, imgs["w_pool_b"], relu=True)     # both images at hand: the image kernel
                else:
                    p = ops.linear_fwd(feat, self.fc_pool.weight, self.fc_pool.bias, relu=True)
                neigh, _ = ops.reduce_fwd(p, idx, "max", want_argmax=False)
                return ops.linear_fwd(feat[:n_dst], self.fc_self.weight, self._summed_bias(), x2=neigh, w2=self.fc_neigh.weight,
                                      relu=fuse_relu)
            # input with a gradient (every layer but the first): the whole layer is one autograd node
            return ops.sage_pool_layer(feat, self.fc_pool.weight, self.fc_pool.bias, self.fc_self.weight, self.fc_neigh.weight,
                                       self.fc_self.bias, self.fc_neigh.bias, idx, n_dst, fuse_relu)
        if (t == "pool" and lazy and feat.proj is None and self.norm is None and (self.activation is None or fuse_relu)
                and ops.small_first_layer_fits(feat.table, feat.ids, idx, n_dst, self.fc_pool.weight, self.fc_pool.bias, self.fc_self.weight,
                                               self.fc_neigh.weight, self.fc_self.bias, self.fc_neigh.bias)):
            # the first layer of a small (32-seed) step: everything behind the fc_pool product is one launch each way
            return ops.small_first_pool_layer(feat.table, feat.ids, self.fc_pool.weight, self.fc_pool.bias, self.fc_self.weight,
                                              self.fc_neigh.weight, self.fc_self.bias, self.fc_neigh.bias, idx, n_dst, fuse_relu,
                                              n_live=getattr(graph, "n_live_dev", None), n_src_live=getattr(graph, "n_src_live_dev", None))
        if t == "pool":
            h_neigh = self._pool_max(feat, idx)
            rst = self._linear2(feat_dst, self.fc_self.weight, h_neigh, self.fc_neigh.weight, self.fc_self.bias, fuse_relu,
                                bias2=self.fc_neigh.bias if self.fc_self.bias is not None else None)
        elif t in ("meanpool", "maxpool"):
            if t == "maxpool":
                h_neigh = self._pool_max(feat, idx)
            elif ops.pool_mean_fits(feat.table if lazy else feat, idx, self.fc_pool.weight.shape[0], feat.shape[0]) and torch.is_grad_enabled():
                # an input without a gradient (the first layer): projection + mean as one node whose backward goes from the pooled-row
                # gradient straight to the operand of fc_pool's weight gradient (a planned segmented gather, no atomics)
                h_neigh = ops.pool_mean(feat.table if lazy else feat, self.fc_pool.weight, self.fc_pool.bias, idx, feat.ids if lazy else None)
            else:
                p = self._project(self.fc_pool, feat, relu=True)
                if ops._CAPTURE is not None:
                    ops._CAPTURE.append(dict(pool_out=p.detach()))      # (test hook, see ops.capture_pool_winners)
                h_neigh = ops.neighbor_reduce(p, idx, "mean")
            rst = self._linear_cat(feat_dst, self._with_edges(graph, h_neigh, "max" if t == "maxpool" else "mean"), fuse_relu)
        elif t == "mean":
            if (lazy and idx.dtype == torch.int32 and ops._n1_images_ok(n_dst, feat.shape[1], self._out_feats)
                    and not feat.table.requires_grad and ops._static_key(feat.table) in ops._X3_TABLES):
                # the first layer over the resident table (features carry no gradient): the mean straight from the table's rows
                # through the block's source ids, its image beside it — no feat[input_nodes] copy, and the combine below takes the
                # image kernel (the table's own image rows for h_self, this image for h_neigh)
                h_neigh, img = ops.reduce_fwd_rows_mean_img(feat.table, feat.ids, idx)
                ops.attach_image(h_neigh, img)
            else:
                src = feat.materialize() if lazy else feat
                if SELF_NEIGH_NODE and not lazy and src.requires_grad and torch.is_grad_enabled():
                    # (an input with a gradient, read twice: one autograd node hands back ONE gradient — ops._SelfNeighFn)
                    feat_dst, h_neigh = ops.self_and_neighbors(src, idx, n_dst, "mean")
                else:
                    h_neigh = ops.neighbor_reduce(src, idx, "mean")
            rst = self._linear_cat(feat_dst, self._with_edges(graph, h_neigh, "mean"), fuse_relu)
        elif t == "lstm":
            # aggregator_dgl.py:116-126,195-199: h_n of nn.LSTM over each destination's mailbox (slot order), zero initial state;
            # a destination without edges keeps zeros.  Library path: ATen gather -> MIOpen LSTM (the mailbox [n_dst, S, D] is
            # materialised, as the reference's degree bucketing does)
            src = feat.materialize() if lazy else feat
            S = idx.shape[1]
            if S == 0 or self.lstm is None:
                h_neigh = src.new_zeros((n_dst, self.in_neigh_feats))
            else:
                li = idx.long()
                has = li[:, 0] >= 0
                m = src.index_select(0, li.clamp(min=0).reshape(-1)).view(n_dst, S, src.shape[1])
                _, (hn, _) = self.lstm(m)
                h_neigh = torch.where(has.unsqueeze(1), hn.squeeze(0), torch.zeros((), dtype=src.dtype, device=src.device))
            rst = self._linear_cat(feat_dst, h_neigh, fuse_relu)
        else:  # gcn
            src = feat.materialize() if lazy else feat
            s = ops.neighbor_reduce(src, idx, "sum")
            degs = graph.in_degrees().to(s.dtype)
            h_neigh = (s + src[:n_dst]) / (degs.unsqueeze(-1) + 1)
            rst = ops.linear(h_neigh, self.fc_neigh.weight, self.fc_neigh.bias, None, None, fuse_relu, None, None)

        if self.activation is not None and not fuse_relu:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst

    def preplan_loss(self, graph):
        """Called by GraphSAGE.forward_loss at the TOP of a train step for its last layer: the in-repo 'mean' / 'meanpool' layer's
        mean backward needs a plan that depends on the block's indices only — started here, its launches run beside the first layer."""
        if (ops.MEAN_LOSS_FUSED and self._aggre_type in ("mean", "meanpool") and torch.is_grad_enabled() and self._edge_feats == 0
                and getattr(graph, "dst_pos", None) is None):
            idx = graph.local_idx
            n_src = graph.number_of_src_nodes()
            if idx.dtype == torch.int32 and idx.is_contiguous() and ops.seg_bwd_fits(idx, self._in_feats, n_src):
                graph._ogl_seg_plan = ops.reduce_bwd_seg_plan(idx, self._in_feats, n_src)

    def forward_loss(self, graph, feat, labels, defer_mean=False, h_single_use=False):
        """This layer as the LAST layer of a train step, fused with nn.CrossEntropyLoss: (mean loss, per-seed losses, logits) from
        one autograd node whose forward is the fc_pool product + ONE launch (``ops.sage_pool_layer_loss``) — or None when that form
        does not apply (the caller runs ``forward`` and the loss separately; same values, same gradients).
        ``defer_mean``: see ``GraphSAGE.forward_loss`` (only a caller that runs the backward itself may ask for it)."""
        if (self._aggre_type == "mean" and not isinstance(feat, GatheredRows) and self.norm is None and self.activation is None
                and not (self.training and self.feat_drop.p > 0) and torch.is_grad_enabled() and getattr(graph, "dst_pos", None) is None
                and self._edge_feats == 0 and self._in_feats == self.in_neigh_feats):
            # the in-repo 'mean' layer as the last layer: aggregator, concat projection and loss in one launch (ops._SageMeanLossFn)
            return ops.sage_mean_layer_loss(feat, self.fc_neigh.weight, self.fc_neigh.bias, graph.local_idx, graph.number_of_dst_nodes(),
                                            labels, defer_mean=defer_mean, plan=getattr(graph, "_ogl_seg_plan", None))
        if (self._aggre_type == "meanpool" and not isinstance(feat, GatheredRows) and self.norm is None and self.activation is None
                and not (self.training and self.feat_drop.p > 0) and torch.is_grad_enabled() and getattr(graph, "dst_pos", None) is None
                and self._edge_feats == 0):
            return ops.sage_meanpool_layer_loss(feat, self.fc_pool.weight, self.fc_pool.bias, self.fc_neigh.weight, self.fc_neigh.bias,
                                                graph.local_idx, graph.number_of_dst_nodes(), labels, defer_mean=defer_mean,
                                                plan=getattr(graph, "_ogl_seg_plan", None))
        if (self._aggre_type != "pool" or isinstance(feat, GatheredRows) or self.norm is not None or self.activation is not None
                or (self.training and self.feat_drop.p > 0) or not torch.is_grad_enabled() or getattr(graph, "dst_pos", None) is not None
                or (self.fc_self.bias is None) != (self.fc_neigh.bias is None)):
            return None
        return ops.sage_pool_layer_loss(feat, self.fc_pool.weight, self.fc_pool.bias, self.fc_self.weight, self.fc_neigh.weight,
                                        self.fc_self.bias, self.fc_neigh.bias, graph.local_idx, graph.number_of_dst_nodes(), labels,
                                        defer_mean=defer_mean, h_single_use=h_single_use)

    def _forward_fused_batches(self, graph, feat, idx, dst_pos, fuse_relu):
        """Inference on several loader batches fused into one block (sampling.sample_batches(fuse_rows=...)): the same three
        launches as the per-batch inference path, with a destination's own row gathered by position instead of ``feat[:n_dst]``."""
        if self._aggre_type != "pool" or isinstance(feat, GatheredRows) or self.norm is not None or not (self.activation is None or fuse_relu):
            raise RuntimeError("fused inference batches are built for the 'pool' layers behind a cached first layer")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("fused inference batches are inference only (run them under torch.no_grad())")
        imgs = getattr(self, "_pass_images", None)
        himg = ops.take_image(feat) if (imgs and "w_pool_b" in imgs) else None
        if himg is not None and himg.K == imgs["w_pool_b"].K:
            p = ops.linear_fwd_x3(himg, None, imgs["w_pool_b"], relu=True)
        else:
            p = ops.linear_fwd(feat, self.fc_pool.weight, self.fc_pool.bias, relu=True)
        neigh, _ = ops.reduce_fwd(p, idx, "max", want_argmax=False)
        return ops.linear_fwd(feat, self.fc_self.weight, self._summed_bias(), x2=neigh, w2=self.fc_neigh.weight, relu=fuse_relu,
                              x_rows=dst_pos)

    def _forward_cached(self, graph, feat, fuse_relu):
        """Inference against the per-pass projection tables (the idea of R/inference_optimized.py:169,258 — its
        ``h0proj`` / ``neigh`` caches): the neighbour max reads ``P0[picks]`` directly and the self term is the row
        ``S0[dst]``, so a batch touches no raw feature row (what lets the tables be built per vertex range and exchanged,
        model.HipSupervisedGraphSage._projection_tables).  ``S0[dst] + fc_neigh(neigh)`` is ONE GEMM over the neighbour term
        whose accumulators start at the table row (``ogl_linear_fwd_addrows``): half the reduction of the dual-input product."""
        t = self._aggre_type
        if t not in ("pool", "maxpool", "meanpool"):
            raise KeyError("cached projections apply to the pooling aggregators only, not {}".format(t))
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("the cached-projection path is inference only (run it under torch.no_grad())")
        P0, S0 = feat.proj
        imgs = getattr(self, "_pass_images", None)
        # (whatever the number of rows: inside a pass the choice of kernels must not depend on how many batches share the
        # launch — a row's result is then the same in any chunking, which keeps rank-sharded passes bit-identical to the
        # one-rank pass; the weight images exist for the pass anyway)
        if t == "pool" and imgs and "w_neigh" in imgs and S0.shape[1] >= ADDROWS_MIN_WIDTH:
            # tall, wide layer: the aggregator writes the image of the pooled rows beside them, the neighbour projection runs on
            # the image kernel with S0[dst] added in its epilogue, and that kernel writes the image of ITS output for the next
            # layer's fc_pool — no split pass anywhere (the weight images were built once for the pass)
            _, _, nimg = ops.reduce_fwd_img(P0, graph.picks, want_out=POOL_FP32_OUT)     # (the product below reads the image only)
            keep = getattr(graph, "out_keep", None)          # (fused batches: the rows whose fp32 values the next layer reads)
            if keep is not None and (keep.numel() != graph.dst_ids.numel() or (self.activation is not None and not fuse_relu)
                                     or self.norm is not None):
                keep = None
            rst, himg = ops.linear_fwd_x3_ext(nimg, None, imgs["w_neigh"], add=S0, add_rows=graph.dst_ids, relu=fuse_relu,
                                              want_image=True, image_append_ones=True, y_keep=keep)
            ops.attach_image(rst, himg)
            if self.activation is not None and not fuse_relu:
                rst = self.activation(rst)
            if self.norm is not None:
                rst = self.norm(rst)
            return rst
        h_neigh, _ = ops.reduce_fwd(P0, graph.picks, "mean" if t == "meanpool" else "max")
        w_neigh = self.fc_neigh.weight if t == "pool" else self.fc_neigh.weight[:, self._in_feats:]
        if S0.shape[1] >= ADDROWS_MIN_WIDTH:
            rst = ops.linear_fwd_addrows(h_neigh, w_neigh, S0, add_rows=graph.dst_ids, relu=fuse_relu)
        else:       # narrow layers: the dual-input product with an identity first weight (exact: x * 1 and zeros) is faster
            rst = ops.linear_fwd(S0, _eye(S0.shape[1], S0.device), None, x2=h_neigh, w2=w_neigh, relu=fuse_relu, x_rows=graph.dst_ids)
        if self.activation is not None and not fuse_relu:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst

    def _summed_bias(self):
        """b_self + b_neigh for the inference paths.  Inside an inference pass (``GraphSAGE.inference_pass()``: weights are
        fixed for its duration) the sum is computed once and reused by every batch; outside one it is recomputed per call.
        (Round 1 keyed a cache on the tensors' version counters — but the optimiser updates parameters through raw pointers
        (ogl_adam_step*), which never bumps them: after the first evaluation every later one added a STALE bias sum.)"""
        bs, bn = self.fc_self.bias, self.fc_neigh.bias
        if bs is None:
            return None
        if getattr(self, "_bias_sum", None) is not None:
            return self._bias_sum
        with torch.no_grad():
            return bs + bn

    def project_tables(self, table, rows=None, out=None):
        """The per-weight-version tables consumed by _forward_cached, for ``table[rows]`` (default: every row):
        ``P0 = relu(fc_pool(x))`` and ``S0`` = the self term of the combine with every bias of it folded in
        (``fc_self(x) + b_self + b_neigh`` for 'pool', ``x . W[:, :in]^T + b`` of the concat -> fc_neigh modes).
        ``out = (P0_rows, S0_rows)`` writes into caller-owned row blocks (a rank's slice of the exchanged tables)."""
        if self.fc_pool is None:
            raise KeyError("aggregator {} has no pooling projection".format(self._aggre_type))
        if self._aggre_type == "pool":
            w_self, b_self = self.fc_self.weight, self._summed_bias()
        else:
            w_self, b_self = self.fc_neigh.weight[:, :self._in_feats], self.fc_neigh.bias
        p0 = ops.linear_fwd(table, self.fc_pool.weight, self.fc_pool.bias, relu=True, x_rows=rows, out=out[0] if out else None)
        s0 = ops.linear_fwd(table, w_self, b_self, x_rows=rows, out=out[1] if out else None)
        return p0, s0

    def _pool_max(self, feat, idx):
        if isinstance(feat, GatheredRows):
            return ops.pool_max(feat.table, self.fc_pool.weight, self.fc_pool.bias, idx, feat.ids)
        return ops.pool_max(feat, self.fc_pool.weight, self.fc_pool.bias, idx, None)

    def _with_edges(self, graph, h_neigh, op):
        """``cat(h_neigh, reduce_j edge_feat[d, j])`` when the layer was built with edge features (the message of an edge is
        ``cat(src h, edge feat)``, aggregator_dgl.py:7-13: the reduction of the concatenation is the concatenation of the
        reductions), else ``h_neigh``.  ``graph.edata['feat']``: [n_dst, S, E] or [n_dst * S, E], one row per (destination, slot).
        The edge half is reduced by the package's aggregator over an identity index (a destination without edges: zeros)."""
        if not self._edge_feats:
            return h_neigh
        e = graph.edata.get("feat") if hasattr(graph, "edata") else None
        if e is None:
            raise KeyError("this layer was built with edge_feats=%d: the block needs edata['feat']" % self._edge_feats)
        idx = graph.local_idx
        n_dst, S = idx.shape
        e = ops.as_mat(e.reshape(n_dst * S, -1))
        assert e.shape[1] == self._edge_feats
        eidx = torch.arange(n_dst * S, dtype=torch.int32, device=idx.device).view(n_dst, S)
        eidx = torch.where(idx >= 0, eidx, torch.full_like(eidx, -1)).contiguous()    # (an edge exists where its slot holds a source)
        if e.requires_grad:
            # learned edge features: the differentiable aggregator (the reference's edge features flow through autograd with the message)
            e_red = ops.neighbor_reduce(e, eidx, op)
        else:
            e_red, _ = ops.reduce_fwd(e, eidx, op)
        if ops.take_image(h_neigh, pop=False) is not None:
            ops.take_image(h_neigh)               # (the pooled rows' image does not cover the appended columns)
        return torch.cat((h_neigh, e_red[:, :self._edge_feats]), 1)

    def _linear_cat(self, x1, x2, relu):
        """fc_neigh(cat(h_self, h_neigh)) (aggregator_dgl.py:206): one dual-input product over the two column blocks of the weight,
        one gradient tensor for it (``ops.linear_cat``)."""
        if isinstance(x1, GatheredRows):
            return ops.linear_cat(x1.table, x2, self.fc_neigh.weight, self._in_feats, self.fc_neigh.bias, relu, x1.ids)
        return ops.linear_cat(x1, x2, self.fc_neigh.weight, self._in_feats, self.fc_neigh.bias, relu)

    @staticmethod
    def _linear2(x1, w1, x2, w2, bias, relu, bias2=None):
        if isinstance(x1, GatheredRows):
            return ops.linear(x1.table, w1, bias, x2, w2, relu, x1.ids, None, bias2)
        return ops.linear(x1, w1, bias, x2, w2, relu, None, None, bias2)
