"""GraphSAGE model: same constructor and ``forward(blocks, x)`` as
R/train/graphsage/pytorch/graphsage_dgl.py:5-59; state_dict keys
``layers.{i}.{fc_pool,fc_self,fc_neigh}.{weight,bias}`` (what R/export_model.py:107 saves and
R/inference_optimized.py:136-139 reads)."""
import contextlib

import torch
import torch.nn as nn

from .. import ops
from .sageconv import SAGEConv


INREPO_T_IMAGES = True   # in-repo pooling layers: W^T images prepared with the step's others


class GraphSAGE(nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, dropout, aggregator_type, edge_feats=None,
                 pool_feats=None):
        super().__init__()
        # the live reference layer (DGL SAGEConv) takes neither edge_feats nor pool_feats (graphsage_dgl.py:41
        # comments them out); the in-repo layer does, so they are forwarded for its modes only
        extra = {}
        if aggregator_type != "pool":
            extra = dict(edge_feats=edge_feats, pool_feats=pool_feats)
        self.layers = nn.ModuleList()
        self.layers.append(SAGEConv(in_feats, n_hidden, aggregator_type, feat_drop=dropout, activation=activation, **extra))
        for _ in range(n_layers - 1):
            self.layers.append(SAGEConv(n_hidden, n_hidden, aggregator_type, feat_drop=dropout, activation=activation, **extra))
        self.layers.append(SAGEConv(n_hidden, n_classes, aggregator_type, feat_drop=dropout, activation=None, **extra))

    @contextlib.contextmanager
    def inference_pass(self):
        """A pass over many batches with FIXED weights (evaluation, the PBR priority forward): per-layer constants derived
        from the parameters (the summed combine bias) are computed once here instead of once per batch, and dropped at the
        end — never kept across a weight update."""
        try:
            with torch.no_grad():
                for li, layer in enumerate(self.layers):
                    if getattr(layer, "fc_self", None) is not None and layer.fc_self.bias is not None and layer.fc_neigh.bias is not None:
                        layer._bias_sum = layer.fc_self.bias + layer.fc_neigh.bias
                    # weight images of the tall projections of the pass (wide 'pool' layers, split-bf16 arithmetic): fc_neigh of
                    # the cached first layer (its self term comes from the table S0), [fc_pool | b] of the layers after it
                    if layer._aggre_type == "pool" and layer._in_feats >= 128 and ops.get_gemm_mode() != "f32":
                        if li == 0:
                            layer._pass_images = dict(w_neigh=ops.x3_split_cat([(layer.fc_neigh.weight, None)]))
                        elif layer.fc_pool.bias is not None:
                            layer._pass_images = dict(w_pool_b=ops.x3_split(layer.fc_pool.weight, append_vec=layer.fc_pool.bias))
            yield self
        finally:
            for layer in self.layers:
                layer._bias_sum = None
                layer._pass_images = None

    def _image_plan(self):
        """Per layer, once: the parameters and widths the step's weight-image requests are made of (attribute lookups through
        nn.Module.__getattr__ are a measurable part of a 0.7 ms host step; ``_apply`` — .cuda(), .float() ... — drops the plan)."""
        plan = self.__dict__.get("_img_plan")
        if plan is None:
            plan = []
            for layer in self.layers:
                if layer._aggre_type == "pool" and layer.fc_pool is not None and layer.fc_self is not None:
                    wp, bp = layer.fc_pool.weight, layer.fc_pool.bias
                    ws, wn, bs = layer.fc_self.weight, layer.fc_neigh.weight, layer.fc_self.bias
                    bn = layer.fc_neigh.bias if bs is not None else None
                    plan.append((layer.feat_drop, ("wb", (wp, bp)), ("cat", (ws, wn, bs, bn)), ("T", (wn,)), ("T", (wp,)),
                                 wp.shape[0], wp.shape[1], ws.shape[0], ("bsum", (bs, bn)) if bn is not None else None))
                else:
                    plan.append(None)
            self.__dict__["_img_plan"] = plan
        return plan

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_img_plan", None)
        return super()._apply(fn, *args, **kwargs)

    def _prepare_step_images(self, blocks, x):
        """The weight images the tall products of this train step will ask for (ops.weight_images_prepare: one launch instead
        of one split — and for the input gradients a transpose — per product).  Which products run on images is decided by
        the same size rules the layers apply; a request nobody consumes costs a few microseconds, a missing one is built by
        its consumer."""
        from .sageconv import GatheredRows
        req = []
        if isinstance(x, GatheredRows) and x.ids is None:
            return                                             # (un-relabelled batches: the layers take, or in training mode refuse, them)
        n_src = x.shape[0]
        training = self.training
        for li, (ent, block) in enumerate(zip(self._image_plan(), blocks)):
            n_dst = block.number_of_dst_nodes()
            if ent is not None and not (training and ent[0].p > 0):
                _, r_wb, r_cat, r_tn, r_tp, p_out, p_in, s_out, r_bsum = ent
                if li == 0 and isinstance(x, GatheredRows) and x.proj is None:
                    if getattr(block, "n_src_live_dev", None) is not None:
                        pass        # (an upper-bound block of a captured small step: its first layer runs on the small kernels, no images)
                    elif ops._x3_forward_ok(ops.as_mat(x.table), n_src, None):
                        req.append(r_wb)
                    if (getattr(block, "n_src_live_dev", None) is None and ops._n1_images_ok(n_dst, p_out, s_out)
                            and ops._static_key(x.table) in ops._X3_TABLES):
                        req.append(r_cat)
                        req.append(r_tn)
                elif li > 0 and ops._n1_images_ok(n_src, p_in, p_out):
                    req.append(r_wb)
                    req.append(r_tp)
                    if r_bsum is not None and s_out <= 64:      # the few-column output layer: fp32 operands + a summed bias
                        req.append(r_bsum)
            n_src = n_dst
        # the in-repo pooling layers behind the first ('meanpool' / 'maxpool'): the images of W_pool^T and of the neighbour block of
        # fc_neigh's weight, transposed — what their two input-gradient products read (a transpose + a split launch each, on the critical
        # path of the backward pass, when nobody prepared them: 4 launches, 20 us of the Reddit-rung 'meanpool' step)
        n_src = x.shape[0]
        for li, (layer, block) in enumerate(zip(self.layers, blocks)):
            n_dst = block.number_of_dst_nodes()
            if (INREPO_T_IMAGES and getattr(layer, "_aggre_type", None) in ("meanpool", "maxpool") and not (training and layer.feat_drop.p > 0)
                    and getattr(layer, "_edge_feats", 0) == 0):
                K = layer._in_feats
                if li > 0 and ops._n1_images_ok(n_src, K, layer.fc_pool.weight.shape[0]):
                    req.append(("T", (layer.fc_pool.weight,)))
                if ops._n1_images_ok(n_dst, layer.fc_neigh.weight.shape[0], K) and len(req) < 7:
                    # (the neighbour block as ONE view object kept on the layer: a prepared image lives as long as the tensor it was
                    # built from, and the consumer's own slice has the same address and version)
                    wv = layer.__dict__.get("_wn_view")
                    wfull = layer.fc_neigh.weight
                    if wv is None or wv.data_ptr() != wfull.data_ptr() + 4 * K or wv.shape[0] != wfull.shape[0]:
                        # (detached: a slice taken under grad mode carries an autograd node — kept across steps it reached into a
                        # later step's capture and hipStreamEndCapture crashed; the detached tensor shares address and version counter)
                        wv = layer.__dict__["_wn_view"] = wfull.detach()[:, K:]
                    req.append(("T", (wv,)))
            n_src = n_dst
        if req:
            ops.weight_images_prepare(req)

    def forward_loss(self, blocks, x, labels, rows=False, defer_mean=False):
        """``CrossEntropyLoss(self(blocks, x), labels)`` for a train step — the per-batch body R/train/graphsage/pytorch/model.py:87-105
        (``reduction='mean'``) and :193-200 (``'none'`` + ``.mean()``: ``rows=True`` also returns the per-seed losses) — with the last
        layer and the loss as ONE autograd node when that layer is a tall few-column 'pool' layer (``SAGEConv.forward_loss``).
        ``labels``: int64 tensor or ``ops.LazyLabels``.  Returns (loss, per-seed losses or None, logits).
        ``defer_mean=True`` is a contract for callers that run ``backward()`` on the returned loss THEMSELVES before anything reads its
        value (the strategies' train steps, the captured step body): the fused node then leaves the VALUE of the mean to the first
        launch of its backward (a device-scope fence per block inside the forward launch costs 33 us at 512 seeds) — until that
        backward has run the loss tensor holds NaN.  The default writes the value in the forward launch: a loss that is only
        logged, guarded or evaluated is always valid."""
        if torch.is_grad_enabled() and ops.get_gemm_mode() != "f32" and ops.PREPARE_WEIGHT_IMAGES:
            self._prepare_step_images(blocks, x)
        h = x
        if len(self.layers) == len(blocks) and len(blocks) > 1 and hasattr(self.layers[-1], "preplan_loss"):
            self.layers[-1].preplan_loss(blocks[-1])
        for layer, block in zip(self.layers[:-1], blocks[:-1]):
            h = layer(block, h)
        # (``h`` was made here and goes to the last layer only: that layer may hand its whole backward to the node that made ``h``)
        out = self.layers[-1].forward_loss(blocks[-1], h, labels, defer_mean=defer_mean, h_single_use=len(blocks) > 1) \
            if len(self.layers) == len(blocks) else None
        if len(blocks) > 0 and hasattr(blocks[-1], "_ogl_seg_plan"):
            del blocks[-1]._ogl_seg_plan
        if out is not None:
            return out[0], (out[1] if rows else None), out[2]
        logits = self.layers[-1](blocks[-1], h)
        if rows:
            loss, r = ops.cross_entropy_mean_rows(logits, labels)
            return loss, r, logits
        return ops.cross_entropy(logits, labels, "mean"), None, logits

    def forward(self, blocks, x):
        if torch.is_grad_enabled() and ops.get_gemm_mode() != "f32" and ops.PREPARE_WEIGHT_IMAGES:
            self._prepare_step_images(blocks, x)
        h = x
        if not torch.is_grad_enabled():
            # inference on fused batches: a layer's fp32 output is read by the next layer only at that block's destination rows (its
            # fc_pool reads the image) — tell the layer below which rows those are
            for i, (below, above) in enumerate(zip(blocks[:-1], blocks[1:])):
                imgs = getattr(self.layers[i + 1], "_pass_images", None) if i + 1 < len(self.layers) else None
                # (only when the layer above is sure to read the IMAGE of this output: its fc_pool weight image of the pass, with the
                # reduction length the emitted image has)
                if (getattr(above, "dst_flag", None) is not None and imgs and "w_pool_b" in imgs
                        and imgs["w_pool_b"].K == getattr(self.layers[i], "_out_feats", -2) + 1):
                    below.out_keep = above.dst_flag
        for layer, block in zip(self.layers, blocks):
            h = layer(block, h)
        return h
