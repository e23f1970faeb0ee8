"""Vertex-stream dynamic graph (timestamps on vertices).

Same surface as R/train/graph/dynamic_graph_vertex.py:11-169.  The reference keeps the full DGL
graph and calls ``graph.subgraph(evolving_vertices)`` each snapshot; snapshot ids are positions in
the time-sorted vertex list.  Here the vertices are relabelled ONCE into arrival order, the CSR is
built once with each adjacency list sorted by (arrival-ordered) neighbour id, and ``evolve`` only
moves the cut: snapshot t = induced subgraph on ids < n_present.
"""
from __future__ import annotations

import numpy as np

from .dynamic_graph import DynamicGraph
from .snapshot_graph import SnapshotGraph, build_time_ordered_csr


class FullGraphData:
    """Host description of the complete graph handed to DynamicGraphVertex (the role the full
    DGLGraph plays in R/train/dataset_utils/pubmed.py:85-109): directed edge list (both directions
    present for undirected data), float features [N, F], integer targets [N] or [N, 1]."""

    def __init__(self, n, src, dst, feat, target):
        self.n, self.src, self.dst, self.feat, self.target = int(n), np.asarray(src), np.asarray(dst), feat, target

    def __len__(self):
        return self.n


class DynamicGraphVertex(DynamicGraph):
    def __init__(self, graph, snapshots, labelled_vertices, search_depth=2, device="cuda"):
        super().__init__(graph, snapshots, labelled_vertices, search_depth)
        self.device = device
        self.evolving_vertices = None
        self.vertex_per_snapshot = int(len(self.graph) / self.snapshots)

    def build(self, vertex_timestamps=None, ensure_labelled=None):
        if vertex_timestamps is None:
            raise NotImplementedError
        self._generate_snapshot(vertex_timestamps, ensure_labelled)

    def _generate_snapshot(self, vertex_timestamps, ensure_labelled=None):
        # stable sort by timestamp, ties in dict order (list.sort on (vertex, ts) pairs in the reference)
        items = list(vertex_timestamps.items())
        ts = np.array([t for _, t in items])
        vertices = [items[i][0] for i in np.argsort(ts, kind="stable")]
        n = len(self.graph)
        if ensure_labelled is None:
            step = self.vertex_per_snapshot
            self.snapshot_vertices = [vertices[i:i + step] for i in range(0, n, step)]
        else:
            assert 0 <= ensure_labelled <= 1
            quota = int(self.vertex_per_snapshot * ensure_labelled)
            self.snapshot_vertices, count = [[]], 0
            for v in vertices:
                count += v in self.labelled_vertices
                self.snapshot_vertices[-1].append(v)
                if count == quota:
                    count = 0
                    self.snapshot_vertices.append([])
            if not self.snapshot_vertices[-1]:
                self.snapshot_vertices.pop()
        order = np.array([v for snap in self.snapshot_vertices for v in snap], dtype=np.int64)
        self._order = order                                  # snapshot id -> original id
        inv = np.full(n, -1, dtype=np.int64)
        inv[order] = np.arange(len(order))
        self._inv = inv                                      # original id -> snapshot id
        gd = self.graph
        keep = (inv[gd.src] >= 0) & (inv[gd.dst] >= 0)
        s, d = inv[gd.src[keep]], inv[gd.dst[keep]]
        indptr, indices, _ = build_time_ordered_csr(len(order), s, d, s)     # key = neighbour id
        feat = np.asarray(gd.feat)[order]
        target = np.asarray(gd.target).reshape(n, -1)[order]
        self.sub_g = SnapshotGraph(indptr, indices, None, feat, target, device=self.device)
        self._cum = np.cumsum([len(s_) for s_ in self.snapshot_vertices])
        self.evolving_vertices = list(self.snapshot_vertices[0])
        self.evolution_index = 1
        self._apply()

    def twin(self):
        """A second stream over the same data, back at its first snapshot and advancing on its own (the look-ahead test stream):
        host-side snapshot lists and id maps are shared (read-only after ``build``), the device tables too (``SnapshotGraph.shared``)."""
        import copy
        t = copy.copy(self)
        t.sub_g = SnapshotGraph.shared(self.sub_g)
        t.evolving_vertices = list(self.snapshot_vertices[0])
        t.evolution_index = 1
        t._apply()
        return t

    def _apply(self):
        n_present = int(self._cum[self.evolution_index - 1])
        self.sub_g.set_snapshot(n_present, n_present)
        self.subgraph_to_original_map = self._order[:n_present]
        self.original_to_subgraph_map = self._inv        # entries of absent vertices are never queried

    def get_added_vertices(self, delta=None):
        delta = 1 if delta is None else delta
        v_set = set()
        for i in range(delta):
            v_set.update(self.snapshot_vertices[self.evolution_index - i - 1])
        vertices = list(v_set)
        return vertices, [v in self.labelled_vertices for v in vertices]

    def get_graph(self):
        return self.sub_g

    def __len__(self):
        return len(self.snapshot_vertices)

    def evolve(self):
        self.evolving_vertices += self.snapshot_vertices[self.evolution_index]
        self.evolution_index += 1
        self._apply()

    def get_original_to_subgraph_map(self):
        return self.original_to_subgraph_map

    def get_subgraph_to_original_map(self):
        return self.subgraph_to_original_map

    def get_vertices_changed(self):
        return set(self.snapshot_vertices[self.evolution_index - 1]), self.search_depth
