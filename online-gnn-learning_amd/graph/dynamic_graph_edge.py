"""Edge-stream dynamic graph (timestamps on edges).

Same surface as R/train/graph/dynamic_graph_edge.py:10-265.  The reference appends
``edges_per_snapshot`` rows of a time-sorted (src, dst) table per snapshot (both directions) with
``add_nodes``/``add_edges``; vertex ids are already relabelled by first appearance
(R/train/dataset_utils/reddit.py:87-123) so new nodes are contiguous and the id maps are the
identity.  Here both directions of EVERY row go into one CSR whose adjacency lists are sorted by
row index; snapshot t keeps the entries with row < t * edges_per_snapshot.
"""
from __future__ import annotations

import numpy as np

from .dynamic_graph import DynamicGraph
from .snapshot_graph import SnapshotGraph, build_time_ordered_csr


class Wrap:
    """Identity id map (R/train/graph/dynamic_graph_edge.py:263-265)."""

    def __getitem__(self, item):
        return item


class DynamicGraphEdge(DynamicGraph):
    def __init__(self, snapshots, labelled_vertices, search_depth=1, device="cuda"):
        super().__init__(None, snapshots, labelled_vertices, search_depth)
        self.device = device
        self.snapshot_edges = None
        self.new_vertices = set()
        self.evolving_vertices = set()
        self.current_subgraph = None
        self.edge_feats = None

    def build(self, vertex_feats, targets, cuda=True, edge_timestamps=None, ensure_labelled=None, restrict=None,
              edge_feats=None):
        if edge_timestamps is None:
            raise NotImplementedError
        if edge_feats is not None:
            raise NotImplementedError("edge features are outside the hot path (settings use edge_feats=0)")
        self.vertex_feats, self.targets = vertex_feats, targets
        src = np.asarray(edge_timestamps["src"].values if hasattr(edge_timestamps["src"], "values") else edge_timestamps["src"])
        dst = np.asarray(edge_timestamps["dst"].values if hasattr(edge_timestamps["dst"], "values") else edge_timestamps["dst"])
        n_rows_all = len(src)
        if restrict is not None and restrict < n_rows_all:
            src, dst = src[:restrict], dst[:restrict]
        self.src, self.dst = src.astype(np.int64), dst.astype(np.int64)
        self.edges_per_snapshot = int(n_rows_all / self.snapshots)
        n = int(max(self.src.max(), self.dst.max())) + 1
        # vertices must appear in id order (relabelled by first appearance): n_present(t) = running max + 1
        rmax = np.maximum.accumulate(np.maximum(self.src, self.dst))
        first_seen = np.full(n, len(self.src), dtype=np.int64)
        np.minimum.at(first_seen, self.src, np.arange(len(self.src)))
        np.minimum.at(first_seen, self.dst, np.arange(len(self.dst)))
        if not (np.diff(first_seen) >= 0).all():
            raise ValueError("edge-stream vertex ids must be relabelled by first appearance "
                             "(as R/train/dataset_utils/reddit.py:87-123 does)")
        self._rmax = rmax
        rows = np.arange(len(self.src), dtype=np.int64)
        u = np.concatenate([self.src, self.dst])
        v = np.concatenate([self.dst, self.src])
        k = np.concatenate([rows, rows])
        indptr, indices, keys = build_time_ordered_csr(n, u, v, k)
        feat = np.asarray(self.vertex_feats)[:n]
        target = np.asarray(self.targets).reshape(len(self.targets), -1)[:n]
        self.current_subgraph = SnapshotGraph(indptr, indices, keys, feat, target, device=self.device)
        self.evolution_index = 1
        self.subgraph_to_original_map = Wrap()
        self.original_to_subgraph_map = self.subgraph_to_original_map
        self._n_present_prev = 0
        self._apply()

    def twin(self):
        """A second stream over the same edge table, back at its first snapshot and advancing on its own (the look-ahead test
        stream): the host-side edge arrays are shared (read-only after ``build``), the device tables too (``SnapshotGraph.shared``)."""
        import copy
        t = copy.copy(self)
        t.current_subgraph = SnapshotGraph.shared(self.current_subgraph)
        t.evolution_index = 1
        t.subgraph_to_original_map = Wrap()
        t.original_to_subgraph_map = t.subgraph_to_original_map
        t.new_vertices, t.evolving_vertices = set(), set()
        t._n_present_prev = 0
        t._apply()
        return t

    def _n_present_at(self, cut):
        cut = min(int(cut), len(self.src))
        return int(self._rmax[cut - 1]) + 1 if cut > 0 else 0

    def _apply(self):
        cut = min(self.evolution_index * self.edges_per_snapshot, len(self.src))
        n_present = self._n_present_at(cut)
        self.current_subgraph.set_snapshot(n_present, cut)
        self.new_vertices = set(range(self._n_present_prev, n_present))
        self._n_present_prev = n_present

    def get_added_vertices(self, delta=None):
        if delta is None:
            vertices = self.new_vertices
        else:
            lo = max(0, (self.evolution_index - delta) * self.edges_per_snapshot)
            hi = self.evolution_index * self.edges_per_snapshot
            vertices = np.unique(np.concatenate([self.src[lo:hi], self.dst[lo:hi]]))
        return vertices, [v in self.labelled_vertices for v in vertices]

    def get_graph(self):
        return self.current_subgraph

    def __len__(self):
        return self.snapshots

    def evolve(self):
        self.evolution_index += 1
        self._apply()

    def get_original_to_subgraph_map(self):
        return self.original_to_subgraph_map

    def get_subgraph_to_original_map(self):
        return self.subgraph_to_original_map

    def get_vertices_changed(self):
        lo = (self.evolution_index - 1) * self.edges_per_snapshot
        hi = self.evolution_index * self.edges_per_snapshot
        return set(np.unique(np.concatenate([self.src[lo:hi], self.dst[lo:hi]]))), self.search_depth
