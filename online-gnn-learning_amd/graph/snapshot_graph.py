"""Device-resident snapshot graph: ONE time-ordered CSR of the final graph + a prefix-degree cut.

Stands in for the DGLGraph the reference re-induces (vertex streams,
R/train/graph/dynamic_graph_vertex.py:85,132-141) or grows in place (edge streams,
R/train/graph/dynamic_graph_edge.py:190-218) every snapshot.  ``set_snapshot`` is O(N) on the
device (one ogl_graph_set_snapshot launch) instead of O(V_t + E_t) host work, and ``ndata`` keeps
the reference's ``graph.ndata['feat'] / ['target']`` contract as views of resident tables.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def build_time_ordered_csr(n, src, dst, key):
    """CSR of in-neighbours: row v lists ``src`` of every edge (src -> v), sorted by ``key``.

    Returns (indptr int64[n+1], indices int32[nnz], keys int32[nnz])."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    key = np.asarray(key, dtype=np.int64)
    if len(src) and (src.max() >= n or dst.max() >= n or src.min() < 0 or dst.min() < 0):
        raise ValueError("edge endpoint outside [0, n)")
    if len(src) and key.max() >= (1 << 31):
        raise ValueError("keys must fit in 31 bits")
    # one stable sort on the composite (dst, key); ties are identical entries.  torch sorts with all host
    # cores, or on the GPU when one is present (host-side preprocessing, once per dataset).
    comp = torch.from_numpy((dst << 31) | key)
    if torch.cuda.is_available():
        order = torch.argsort(comp.cuda(), stable=True).cpu().numpy()
    else:
        order = torch.argsort(comp, stable=True).numpy()
    indices = src[order].astype(np.int32)
    keys = key[order].astype(np.int32)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(dst, minlength=n), out=indptr[1:])
    return indptr, indices, keys


class SnapshotGraph:
    """What the hot path needs of a graph: the sampler handle, ``ndata`` and the node count."""

    def __init__(self, indptr, indices, keys, feat, target, device="cuda"):
        self.device = torch.device(device)
        n = len(indptr) - 1
        self.n_total = n
        self._indptr = torch.as_tensor(np.ascontiguousarray(indptr), dtype=torch.int64).to(self.device)
        self._indices = torch.as_tensor(np.ascontiguousarray(indices), dtype=torch.int32).to(self.device)
        same = keys is None or keys is indices
        self._keys = None if same else torch.as_tensor(np.ascontiguousarray(keys), dtype=torch.int32).to(self.device)
        self.handle = ops.GraphHandle(self._indptr, self._indices, self._keys)
        feat = torch.as_tensor(feat)
        if feat.dtype == torch.float64:                     # utils.to_nn_lib: float64 -> float32 (R/train/utils.py:62-64)
            feat = feat.float()
        assert feat.shape[0] == n
        self.feat_table = ops.empty_mat(n, feat.shape[1], self.device)   # rows padded to 128 B multiples
        self.feat_table.copy_(feat.to(self.device))
        ops.register_static_table(self.feat_table)          # layer-0 projections read its pre-split image (built lazily)
        target = torch.as_tensor(np.asarray(target)).reshape(n, -1)[:, :1].to(torch.int64)
        self.target_table = target.to(self.device).contiguous()
        self.edata = {}
        self.n_present = 0
        self.cut = 0
        self.ndata = {"feat": self.feat_table[:0], "target": self.target_table[:0]}

    @classmethod
    def shared(cls, other):
        """A second snapshot view of the SAME stream: the CSR arrays, the feature table (and with it its bf16x3 image,
        ``ops.register_static_table``) and the label table are ``other``'s — static data, never written after the upload — and only
        the sampler handle (its per-snapshot degree array) is this view's own.  What the look-ahead test stream of a dataset is to
        its train stream (R/train/dataset_utils/pubmed.py:85-86 builds the graph twice): one resident copy of 0.57 + 0.85 GB at the
        Reddit size instead of two."""
        g = cls.__new__(cls)
        g.device, g.n_total = other.device, other.n_total
        g._indptr, g._indices, g._keys = other._indptr, other._indices, other._keys
        g.handle = ops.GraphHandle(g._indptr, g._indices, g._keys)
        g.feat_table, g.target_table = other.feat_table, other.target_table
        g.edata = {}
        g.n_present = g.cut = 0
        g.ndata = {"feat": g.feat_table[:0], "target": g.target_table[:0]}
        return g

    def set_snapshot(self, n_present, cut):
        self.handle.set_snapshot(n_present, cut)
        self.n_present, self.cut = int(n_present), int(cut)
        self.ndata = {"feat": self.feat_table[:self.n_present], "target": self.target_table[:self.n_present]}

    def number_of_nodes(self):
        return self.n_present

    def __len__(self):
        return self.n_present

    def nodes(self):
        return torch.arange(self.n_present, device=self.device)

    def in_degrees(self):
        return self.handle.degrees()[:self.n_present]

    def number_of_edges(self):
        return int(self.handle.degrees().sum().item())
