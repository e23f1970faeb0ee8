#!/usr/bin/env python
"""The reference's snapshot loop (R/train/__main__.py:97-196) on the HIP backend, over a seeded synthetic stream
(or an on-disk dataset directory in the reference's formats): RBR, PBR and no-rehearsal every snapshot, the offline
model every ``train_offline`` snapshots, evaluation every ``eval`` snapshots, then ``evolve``.

  python examples/run_stream.py reddit --start 2500 --steps 10 --out /tmp/res.csv
  python examples/run_stream.py pubmed --path datasets/pubmed --steps 50

Prints per-strategy ``delay`` (the reference's metric: wall seconds of _run_custom_train per snapshot,
R/train/graphsage/model.py:110-117), the PBR priority-forward time and the evolve time.
"""
import argparse
import gc
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ogl_amd  # noqa: E402,F401
from ogl_amd import dataset_utils, ops, sampling, synthetic  # noqa: E402
from ogl_amd.graph import TrainTestGraph  # noqa: E402
from ogl_amd.prioritized_replay import LossPriority  # noqa: E402
from ogl_amd.utils import Lib_supported, init  # noqa: E402

# per-dataset defaults of R/settings/*.json, with BASELINE.json's samples=25
SETTINGS = {
    "pubmed": dict(embedding_size=32, latent_dim=32, samples=25, batch_size=32, batch_timestep=2, eval=4, snapshots=400, delta=14,
                   batch_full=1024, epochs_offline=8, train_offline=133, priority_forward=1),
    "arxiv": dict(embedding_size=32, latent_dim=32, samples=25, batch_size=32, batch_timestep=1, eval=7, snapshots=3500, delta=45,
                  batch_full=1024, epochs_offline=1, train_offline=580, priority_forward=1),
    "reddit": dict(embedding_size=600, latent_dim=600, samples=25, batch_size=512, batch_timestep=50, eval=8, snapshots=5000, delta=4,
                   batch_full=900, epochs_offline=32, train_offline=600, priority_forward=2),
    "toy": dict(embedding_size=16, latent_dim=16, samples=5, batch_size=16, batch_timestep=2, eval=2, snapshots=20, delta=2,
                batch_full=64, epochs_offline=1, train_offline=5, priority_forward=1),
}
START_PRIOR_ALPHA, END_PRIOR_ALPHA, SCALE = 4, 50, 1     # R/train/__main__.py:10-12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset", choices=sorted(SETTINGS))
    ap.add_argument("--path", default=None, help="dataset directory in the reference's file formats (default: synthetic)")
    ap.add_argument("--start", type=int, default=0, help="fast-forward this many snapshots before the loop (no training)")
    ap.add_argument("--steps", type=int, default=10, help="snapshots to run")
    ap.add_argument("--out", default="/tmp/ogl_results.csv")
    ap.add_argument("--gemm", default="auto", choices=["f32", "bf16x6", "auto"])
    ap.add_argument("--no-offline", action="store_true", help="skip the offline (full retrain) strategy")
    args = ap.parse_args()
    cfg = SETTINGS[args.dataset]
    np.random.seed(1); random.seed(1); torch.manual_seed(1); sampling.seed(1)     # R/train/__main__.py:211-212 (+ sampler)
    ops.set_gemm_mode(args.gemm)

    GraphSAGE, Random, Prioritized, NoReh, Full, act = init(Lib_supported.HIP, True, 0)
    t0 = time.time()
    if args.path:
        load = getattr(dataset_utils, {"pubmed": "pubmed", "arxiv": "arxiv", "reddit": "reddit"}[args.dataset]).load
        feat_size, labels, graph, n_classes, graph_test = load(args.path, snapshots=cfg["snapshots"], cuda=True, copy_to_gpu=True)
    else:
        feat_size, labels, graph, n_classes, graph_test = synthetic.load(args.dataset, snapshots=cfg["snapshots"])
    print("loaded %s in %.1f s: F=%d C=%d snapshots=%d" % (args.dataset, time.time() - t0, feat_size, n_classes, len(graph)), flush=True)
    for _ in range(cfg["delta"]):
        graph_test.evolve()
    for _ in range(args.start):                           # the device CSR makes fast-forwarding O(1) per snapshot
        graph.evolve(); graph_test.evolve()
    gu = TrainTestGraph(graph, split=0.15, start_prior_alpha=START_PRIOR_ALPHA, end_prior_alpha=END_PRIOR_ALPHA, scale=SCALE,
                        max_priority=10)
    if args.start:
        # everything that arrived during the fast-forward is labelled history: enrol it like the reference would have
        seen = [v for v in range(graph.get_graph().n_present)]
        seen = list(graph.get_subgraph_to_original_map()[seen]) if not hasattr(graph.get_subgraph_to_original_map(), "__call__") else seen
        gu._admit([int(v) for v in seen if v in graph.labelled_vertices])

    def mk():
        return GraphSAGE(feat_size, cfg["embedding_size"], n_classes, 1, act, 0, "pool", edge_feats=0, pool_feats=cfg["latent_dim"]).cuda()

    kw = dict(cuda=True, batch_full=cfg["batch_full"], n_workers=0)
    strategies = [Random(mk(), cfg["batch_timestep"], cfg["batch_size"], labels, cfg["samples"], **kw),
                  Prioritized(mk(), cfg["batch_timestep"], cfg["batch_size"], labels, cfg["samples"], LossPriority(),
                              full_pass=cfg["priority_forward"], **kw),
                  NoReh(mk(), cfg["batch_timestep"], cfg["batch_size"], labels, cfg["samples"], **kw)]
    full = None if args.no_offline else Full(mk(), cfg["epochs_offline"], cfg["batch_size"], labels, cfg["samples"], **kw)
    for s in strategies + ([full] if full else []):
        s.build_optimizer()

    delays = {s.get_model(): [] for s in strategies}
    prio_forward, evolve_t, snap_t = [], [], []
    # the loop keeps the reference's per-snapshot gc.collect(); the stream's long-lived id lists (millions of ints) are frozen out of
    # it once, so that a collection looks at the snapshot's garbage only (0.1 s -> 1 ms per Reddit-like snapshot)
    gc.collect()
    gc.freeze()
    for step in range(args.steps):
        t_snap = time.time()
        for s in strategies:
            if s.get_model() == "prioritized":             # time the priority forward (inside choose_vertices) separately
                torch.cuda.synchronize(); t = time.time()
                batch_nodes = s.choose_vertices(gu)
                torch.cuda.synchronize(); prio_forward.append(time.time() - t)
                s.choose_vertices = lambda _gu, _b=batch_nodes: _b
                s.train_timestep(gu)
                del s.choose_vertices
            else:
                s.train_timestep(gu)
            delays[s.get_model()].append(s.delay)
        if full is not None and step % cfg["train_offline"] == 0:
            full.train_timestep(gu)
        if step % cfg["eval"] == 0:
            for s in strategies + ([full] if full else []):
                s.evaluate(gu, args.out)
                s.evaluate_next_snapshots(graph_test, cfg["delta"], args.out)
        if step + args.start + cfg["delta"] + 1 < len(gu):
            torch.cuda.synchronize(); t = time.time()
            gu.evolve(); graph_test.evolve()
            torch.cuda.synchronize(); evolve_t.append(time.time() - t)
            gc.collect()
        torch.cuda.synchronize()
        snap_t.append(time.time() - t_snap)
        print("snapshot %d: n_present=%d |train|=%d  %.3f s" % (args.start + step, graph.get_graph().n_present, len(gu.get_train_set()),
                                                                 snap_t[-1]), flush=True)
    seeds_per_snapshot = cfg["batch_timestep"] * cfg["batch_size"]
    for name, d in delays.items():
        d = np.array(d[1:] or d)
        print("%-12s delay mean %.4f s  (%.0f trained vertices/s)" % (name, d.mean(), seeds_per_snapshot / d.mean() if name != "no_rehersal" else 0))
    if prio_forward:
        print("priority forward mean %.4f s per snapshot (|train| = %d)" % (np.mean(prio_forward[1:] or prio_forward), len(gu.get_train_set())))
    if evolve_t:
        print("evolve (both streams + split + buffer) mean %.4f s" % np.mean(evolve_t))
    print("whole snapshot mean %.3f s; results appended to %s" % (np.mean(snap_t[1:] or snap_t), args.out))


if __name__ == "__main__":
    main()
