"""Shared parsing for the reference's dataset directories.

Vertex streams (R/train/dataset_utils/pubmed.py:70-124, arxiv.py:18-54, bitcoin.py:78-113):
  ``<feats>.npy`` float64/float32 [N, F]; ``targets.npy`` [N] or [N, 1] (-1 = unlabelled);
  ``graph.adjlist`` networkx adjacency list, undirected, integer node ids = feature rows;
  ``<timestamps>.json`` {vertex id: timestamp}.
Edge streams (R/train/dataset_utils/reddit.py:144-177):
  ``feat_data.npy``, ``targets.npy``, ``edges_dataframe.csv`` — integer columns including ``src``, ``dst``,
  rows in time order, vertex ids relabelled by first appearance.
"""
from __future__ import annotations

import json
import os

import numpy as np


def _need(path, files):
    missing = [f for f in files if not os.path.isfile(os.path.join(path, f))]
    if missing:
        raise FileNotFoundError("dataset files missing under %s: %s (the reference would download them; this build has "
                                "no network access)" % (path, ", ".join(missing)))


def read_adjlist(path):
    """Parse a networkx adjlist into the directed edge list ``dgl.from_networkx`` would hold: every undirected
    edge once in each direction, self loops once.  Returns (src, dst) int64 arrays."""
    us, vs = [], []
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].split()
            if not line:
                continue
            u = int(line[0])
            for tok in line[1:]:
                us.append(u); vs.append(int(tok))
    u = np.asarray(us, dtype=np.int64); v = np.asarray(vs, dtype=np.int64)
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    und = np.unique(np.stack([lo, hi], 1), axis=0) if len(u) else np.zeros((0, 2), dtype=np.int64)
    loops = und[:, 0] == und[:, 1]
    a, b = und[~loops, 0], und[~loops, 1]
    src = np.concatenate([a, b, und[loops, 0]])
    dst = np.concatenate([b, a, und[loops, 1]])
    return src, dst


def read_timestamps(path):
    with open(path) as f:
        return {int(k): v for k, v in json.load(f).items()}


def read_edge_table(path):
    """``src`` / ``dst`` integer columns of a csv with a header row (pandas.read_csv(..., dtype=int) in the reference)."""
    with open(path) as f:
        header = [h.strip().strip('"') for h in f.readline().rstrip("\n").split(",")]
    cols = {h: i for i, h in enumerate(header)}
    if "src" not in cols or "dst" not in cols:
        raise ValueError("%s needs 'src' and 'dst' columns, found %s" % (path, header))
    data = np.loadtxt(path, delimiter=",", skiprows=1, usecols=(cols["src"], cols["dst"]), dtype=np.int64, ndmin=2)
    return {"src": data[:, 0], "dst": data[:, 1]}


def _targets(path):
    t = np.load(os.path.join(path, "targets.npy"))
    return t.astype(np.int64).reshape(len(t), -1)[:, 0]


def load_vertex_stream(path, feat_file, ts_file, snapshots=100, cuda=True, copy_to_gpu=True, ensure_labelled=None):
    from ..graph.dynamic_graph_vertex import DynamicGraphVertex, FullGraphData
    if not cuda:
        raise RuntimeError("the hip backend keeps the dataset on the GPU: call load(..., cuda=True)")
    _need(path, [feat_file, "targets.npy", "graph.adjlist", ts_file])
    feat = np.load(os.path.join(path, feat_file))
    targets = _targets(path)
    src, dst = read_adjlist(os.path.join(path, "graph.adjlist"))
    timestamps = read_timestamps(os.path.join(path, ts_file))
    labelled = set(np.argwhere(targets != -1)[:, 0].tolist())
    n_classes = len(np.unique(targets))
    # the training stream and the look-ahead test stream advance independently over the SAME static data: one CSR, one feature
    # table (+ its bf16x3 image) and one label table in HBM, two snapshot views
    gd = FullGraphData(len(feat), src, dst, feat, targets)
    g = DynamicGraphVertex(gd, snapshots, labelled)
    g.build(vertex_timestamps=timestamps, ensure_labelled=ensure_labelled)
    return feat.shape[1], targets.reshape(-1, 1), g, n_classes, g.twin()


def load_edge_stream(path, snapshots=100, cuda=True, copy_to_gpu=True, restrict=None):
    from ..graph.dynamic_graph_edge import DynamicGraphEdge
    if not cuda:
        raise RuntimeError("the hip backend keeps the dataset on the GPU: call load(..., cuda=True)")
    _need(path, ["feat_data.npy", "targets.npy", "edges_dataframe.csv"])
    feat = np.load(os.path.join(path, "feat_data.npy"))
    targets = _targets(path)
    table = read_edge_table(os.path.join(path, "edges_dataframe.csv"))
    labelled = set(np.argwhere(targets != -1)[:, 0].tolist())
    n_classes = len(np.unique(targets))
    g = DynamicGraphEdge(snapshots, labelled)
    g.build(feat, targets, True, edge_timestamps=table, restrict=restrict)
    return feat.shape[1], targets.reshape(-1, 1), g, n_classes, g.twin()          # (one resident copy, two snapshot views)
