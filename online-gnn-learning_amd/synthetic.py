"""Seeded synthetic stand-ins for the reference's datasets (none is available offline).

Shapes, seeds and generators follow SURVEY.md §8(d) / BASELINE.md: Chung-Lu power-law degrees
(gamma ~ 2.1, max expected degree capped at N/10), symmetrised edges, fp32 N(0,1) features, uniform
int64 labels, every vertex labelled.  ``pubmed`` / ``arxiv`` are vertex streams
(R/train/dataset_utils/pubmed.py:70-124, arxiv.py:18-54); ``reddit`` is an edge stream with ids
relabelled by first appearance (R/train/dataset_utils/reddit.py:87-123,144-177).
"""
from __future__ import annotations

import numpy as np
import torch

SEEDS = dict(graph=1234, feat=1235, labels=1236, split=1237)

SPECS = {
    #            N        F    C   undirected edges  snapshots  stream
    "pubmed": dict(n=19717, f=500, c=3, e=44338, snapshots=400, stream="vertex"),
    "arxiv": dict(n=169343, f=128, c=40, e=1166243, snapshots=3500, stream="vertex"),
    "reddit": dict(n=232965, f=602, c=41, e=11606919, snapshots=5000, stream="edge"),
    # Elliptic bitcoin transactions (R/train/dataset_utils/bitcoin.py:55,78-113: 165 features, classes = unique targets incl. the
    # unlabelled marker, vertex stream by time step; N / E are the public dataset's; R/settings/elliptic.json: 1000 snapshots)
    "bitcoin": dict(n=203769, f=165, c=3, e=234355, snapshots=1000, stream="vertex"),
    "toy": dict(n=600, f=20, c=4, e=3000, snapshots=20, stream="vertex"),
    "toy_edge": dict(n=600, f=20, c=4, e=3000, snapshots=20, stream="edge"),
}


def chung_lu_edges(n, e, rng, gamma=2.1):
    """e undirected endpoint pairs with P(endpoint = v) proportional to a power-law weight."""
    w = (np.arange(1, n + 1, dtype=np.float64)) ** (-1.0 / (gamma - 1.0))
    w = np.minimum(w / w.sum() * 2 * e, n / 10.0)
    rng.shuffle(w)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    tc = torch.from_numpy(cdf)
    u = torch.searchsorted(tc, torch.from_numpy(rng.random(e)), right=True).numpy()
    v = torch.searchsorted(tc, torch.from_numpy(rng.random(e)), right=True).numpy()
    np.minimum(u, n - 1, out=u)
    np.minimum(v, n - 1, out=v)
    return u, v


def make_arrays(name, scale=1.0):
    """Host arrays of one dataset: dict(n, f, c, snapshots, stream, src, dst, feat, labels[, order])."""
    spec = dict(SPECS[name])
    n, e = max(8, int(spec["n"] * scale)), max(8, int(spec["e"] * scale))
    rng = np.random.default_rng(SEEDS["graph"])
    u, v = chung_lu_edges(n, e, rng)
    gen = torch.Generator().manual_seed(SEEDS["feat"])
    if spec["stream"] == "edge":
        # rows are already in time order; relabel vertices by first appearance, drop the never-seen ones
        flat = np.stack([u, v], 1).reshape(-1)
        first = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
        first[flat[::-1]] = np.arange(len(flat), dtype=np.int64)[::-1]      # earliest position wins
        seen = np.nonzero(first != np.iinfo(np.int64).max)[0]
        newid = np.full(n, -1, dtype=np.int64)
        newid[seen[np.argsort(first[seen], kind="stable")]] = np.arange(len(seen))
        u, v, n = newid[u], newid[v], len(seen)
    feat = torch.randn((n, spec["f"]), generator=gen, dtype=torch.float32)
    labels = np.random.default_rng(SEEDS["labels"]).integers(0, spec["c"], size=n).astype(np.int64)
    out = dict(name=name, n=n, f=spec["f"], c=spec["c"], snapshots=spec["snapshots"], stream=spec["stream"],
               src=u, dst=v, feat=feat, labels=labels)
    if spec["stream"] == "vertex":
        out["order"] = np.random.default_rng(SEEDS["graph"] + 1).permutation(n)   # arrival order of original ids
    return out


def load(name, snapshots=None, device="cuda", scale=1.0):
    """``load(path, snapshots, cuda, copy_to_gpu)`` analogue of the reference's dataset modules: returns
    ``(feat_size, targets, dynamic_graph, n_classes, dynamic_graph_test)``."""
    from .graph.dynamic_graph_edge import DynamicGraphEdge
    from .graph.dynamic_graph_vertex import DynamicGraphVertex, FullGraphData
    a = make_arrays(name, scale)
    snapshots = snapshots or a["snapshots"]
    labelled = set(range(a["n"]))
    # the train graph and the look-ahead test graph: two snapshot views of ONE resident copy of the static data (twin())
    if a["stream"] == "vertex":
        gd = FullGraphData(a["n"], np.concatenate([a["src"], a["dst"]]), np.concatenate([a["dst"], a["src"]]),
                           a["feat"], a["labels"])
        g = DynamicGraphVertex(gd, snapshots, labelled, device=device)
        ts = {int(v): int(t) for t, v in enumerate(a["order"])}
        g.build(vertex_timestamps=ts)
    else:
        g = DynamicGraphEdge(snapshots, labelled, device=device)
        g.build(a["feat"], a["labels"], True, edge_timestamps={"src": a["src"], "dst": a["dst"]})
    return a["f"], a["labels"].reshape(-1, 1), g, a["c"], g.twin()
