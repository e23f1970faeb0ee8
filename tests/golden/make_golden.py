"""Generate golden vectors by RUNNING the importable parts of the reference in this container.

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (not present on the GPU box); only the resulting small .npz/.json data
files are committed.  Nothing from the reference's sources is copied: the script imports
``graphsage.pytorch.aggregator_dgl.SAGEConv`` (R/train/graphsage/pytorch/aggregator_dgl.py:16-216),
``prioritized_replay.segment_tree.SumSegmentTree``, ``prioritized_replay.replay_buffer.
PrioritizedReplayBuffer`` and ``prioritized_replay.generate_priority.{LossPriority, TrendPriority, HybridPriority}`` and
records their inputs/outputs.

The reference layer expects a DGL block.  DGL is absent, so ``FakeBlock`` below (our code) offers
the handful of attributes the layer touches: ``local_scope``, ``is_block``,
``number_of_dst_nodes``, ``number_of_edges``, ``in_degrees``, ``srcdata/dstdata/edata`` and a
degree-bucketed ``update_all(message_fn, reduce_fn)``.
"""
import contextlib
import json
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference/train"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from graphsage.pytorch.aggregator_dgl import SAGEConv  # noqa: E402  (reference code, imported not copied)
from prioritized_replay.segment_tree import SumSegmentTree  # noqa: E402
from prioritized_replay.replay_buffer import PrioritizedReplayBuffer  # noqa: E402
from prioritized_replay.generate_priority import LossPriority  # noqa: E402


class FakeBlock:
    """Fixed-fanout bipartite block: edge (src=local_idx[d, j]) -> dst d, slot order = mailbox order."""
    is_block = True

    def __init__(self, n_src, local_idx):
        self.n_src = n_src
        self.local_idx = np.asarray(local_idx)
        self.srcdata, self.dstdata, self.edata = {}, {}, {}

    @contextlib.contextmanager
    def local_scope(self):
        yield

    def number_of_dst_nodes(self):
        return self.local_idx.shape[0]

    def number_of_src_nodes(self):
        return self.n_src

    def number_of_edges(self):
        return int((self.local_idx >= 0).sum())

    def in_degrees(self):
        return torch.as_tensor((self.local_idx >= 0).sum(axis=1))

    def update_all(self, message_fn, reduce_fn):
        li = self.local_idx
        has = li[:, 0] >= 0
        out = None
        if has.any():
            idx = torch.as_tensor(li[has].astype(np.int64))            # [n, S]
            n, S = idx.shape
            data = {}
            if "feat" in self.edata:                                   # edge features [n_dst, S, E]: the mailbox keeps slot order
                data["feat"] = self.edata["feat"][torch.as_tensor(has)].reshape(n * S, -1)
            edges = types.SimpleNamespace(src={"h": self.srcdata["h"][idx.reshape(-1)]}, data=data)
            m = message_fn(edges)["m"].reshape(n, S, -1)
            nodes = types.SimpleNamespace(mailbox={"m": m})
            red = reduce_fn(nodes)
            for k, v in red.items():
                full = v.new_zeros((li.shape[0], v.shape[1]))
                full = full.index_put((torch.nonzero(torch.as_tensor(has))[:, 0],), v)
                self.dstdata[k] = full
        else:
            pass  # the layer pre-fills dstdata['neigh'] with zeros when there are no edges


def make_block(rng, n_dst, n_src, fanout, frac_isolated):
    li = rng.integers(0, n_src, size=(n_dst, fanout)).astype(np.int32)
    iso = rng.random(n_dst) < frac_isolated
    li[iso] = -1
    return li


def sageconv_case(tag, mode, n_dst, n_src, fanout, fin, fout, pool, frac_iso, seed, edge=0):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    layer = SAGEConv(fin, fout, mode, feat_drop=0.0, activation=torch.nn.functional.relu,
                     edge_feats=edge, pool_feats=pool)
    li = make_block(rng, n_dst, n_src, fanout, frac_iso)
    x = torch.randn(n_src, fin, requires_grad=True)
    blk = FakeBlock(n_src, li)
    e = None
    if edge:
        # edge features of every (destination, slot) edge: message = cat(src h, edge feat) (aggregator_dgl.py:7-13)
        e = torch.randn(n_dst, fanout, edge)
        blk.edata["feat"] = e
    y = layer(blk, x)
    gy = torch.randn_like(y)
    y.backward(gy)
    out = dict(mode=mode, local_idx=li, x=x.detach().numpy(), y=y.detach().numpy(), gy=gy.numpy(),
               gx=x.grad.numpy(), pool_feats=-1 if pool is None else pool)
    if e is not None:
        out["edge"] = e.numpy()
    for k, v in layer.state_dict().items():
        out["param." + k] = v.numpy()
    for k, p in layer.named_parameters():
        out["grad." + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "sageconv_%s.npz" % tag), **out)
    print("wrote sageconv_%s.npz" % tag, {k: getattr(v, "shape", v) for k, v in out.items()})


def sageconv_fullsize_case(mode):
    """The BASELINE shape: 512 destinations x 25 neighbours over 602 features, pool_feats 600, 600 outputs,
    through the reference's own SAGEConv (R/train/graphsage/pytorch/aggregator_dgl.py:128-216).  Inputs and weights come from
    ``fullsize_inputs.make`` (a PCG64 stream the tests regenerate); committed: the reference's outputs, digested (see there)."""
    import fullsize_inputs as FI
    inp = FI.make(mode)
    layer = SAGEConv(FI.FIN, FI.FOUT, mode, feat_drop=0.0, activation=torch.nn.functional.relu, edge_feats=0,
                     pool_feats=FI.POOL if mode == "meanpool" else None)
    layer.load_state_dict({k: torch.tensor(v) for k, v in inp["params"].items()})
    x = torch.tensor(inp["x"], requires_grad=True)
    blk = FakeBlock(FI.N_SRC, inp["local_idx"])
    y = layer(blk, x)
    y.backward(torch.tensor(inp["gy"]))
    proj, norms, rows = FI.digest(x.grad.numpy(), inp)
    out = dict(mode=mode, y=y.detach().numpy(), gx_proj=proj.astype(np.float32), gx_norms=norms.astype(np.float32),
               gx_rows=rows.astype(np.float32))
    for k, p in layer.named_parameters():
        g = p.grad.numpy()
        out["grad." + k] = g[::FI.GRAD_ROW_STRIDE] if g.ndim == 2 else g
    np.savez_compressed(os.path.join(OUT, "fullsize_%s.npz" % mode), **out)
    print("wrote fullsize_%s.npz" % mode, {k: getattr(v, "shape", v) for k, v in out.items()})


def replay_cases():
    res = {}
    # segment tree
    t = SumSegmentTree(8)
    vals = [0.5, 1.0, 0.25, 2.0, 0.0, 3.5, 0.125, 0.75]
    for i, v in enumerate(vals):
        t[i] = v
    res["tree"] = dict(capacity=8, values=vals,
                       sum_0_3=t.sum(0, 3), sum_all=t.sum(), sum_2_7=t.sum(2, 7),
                       prefix_queries=[0.0, 0.49, 0.5, 1.6, 1.75, 3.74, 3.75, 7.0, 8.0],
                       prefix_idx=[t.find_prefixsum_idx(q) for q in [0.0, 0.49, 0.5, 1.6, 1.75, 3.74, 3.75, 7.0, 8.0]])
    # replay buffer: add_all / update_priorities / dump / stratified sampling under random.seed
    buf = PrioritizedReplayBuffer(64, 4, max_priority=10, min_priority=1e-7)
    first = {10: 2.0, 11: 0.5, 12: 9.0, 13: 1e-9, 14: 3.0}
    buf.add_all(first)
    d1 = buf.dump_priorities(list(first))
    second = {20 + i: float(v) for i, v in enumerate(np.random.default_rng(7).uniform(0.01, 12.0, 40))}
    buf.add_all(second)
    d2 = buf.dump_priorities(list(first) + list(second))
    upd = {11: 4.0, 25: 0.001, 40: 20.0}
    buf.update_priorities(upd)
    d3 = buf.dump_priorities(list(first) + list(second))
    random.seed(1)
    s1 = sorted(int(i) for i in buf._sample_proportional(8))
    random.seed(2)
    s2 = sorted(int(i) for i in buf._sample_proportional(16))
    random.seed(3)
    s_all = sorted(int(i) for i in buf._sample_proportional(100))
    res["buffer"] = dict(alpha=4, max_priority=10, min_priority=1e-7, size=64,
                         first={str(k): v for k, v in first.items()}, dump_after_first=d1,
                         second={str(k): v for k, v in second.items()}, dump_after_second=d2,
                         update={str(k): v for k, v in upd.items()}, dump_after_update=d3,
                         sample8_seed1=s1, sample16_seed2=s2, sample100_seed3=s_all,
                         storage=[int(x) for x in buf._storage],
                         min_val=buf.get_min_priority(), max_val=buf.get_max_priority())
    lp = LossPriority()
    losses = np.array([0.3, 1.2, 0.0], dtype=np.float32)
    res["loss_priority"] = dict(nodes=[4, 9, 2], losses=losses.tolist(),
                                out=np.asarray(lp.get_priorities([4, 9, 2], losses)).tolist())
    with open(os.path.join(OUT, "replay.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote replay.json")


def priority_cases():
    """TrendPriority / HybridPriority (R/train/prioritized_replay/generate_priority.py:11-58) over a sequence of batches with
    recurring vertices: per-batch outputs and the final state.  The reference spells float64 ``np.float`` (generate_priority.py:13),
    an alias numpy >= 1.24 no longer has: it is restored here, in the generating environment, so that the reference's own code
    runs unchanged."""
    if not hasattr(np, "float"):
        np.float = float
    from prioritized_replay.generate_priority import HybridPriority, TrendPriority
    rng = np.random.default_rng(5)
    n_vertices = 40
    batches = []
    for step in range(6):
        ids = rng.choice(n_vertices, size=int(rng.integers(3, 12)), replace=False)
        losses = rng.uniform(0.0, 4.0, len(ids)).astype(np.float32)          # float32 losses, as they come off the device
        batches.append((ids, losses))
    res = {}
    for name, obj in (("trend_priority", TrendPriority(n_vertices, alpha=0.85)),
                      ("hybrid_priority", HybridPriority(n_vertices, alpha=0.7, loss_contrib=0.4))):
        outs = [np.asarray(obj.get_priorities(ids, losses), dtype=np.float64).tolist() for ids, losses in batches]
        tp = obj.trend_p if name == "hybrid_priority" else obj
        res[name] = dict(n_vertices=n_vertices, alpha=float(tp.alpha), loss_contrib=float(getattr(obj, "loss_contrib", -1.0)),
                         batches=[dict(ids=[int(i) for i in ids], losses=[float(x) for x in losses]) for ids, losses in batches],
                         outputs=outs, final_values=tp.values.tolist(), final_prev_loss=tp.prev_loss.tolist(),
                         final_init=[bool(x) for x in tp.init], final_avg=float(tp.avg), final_n_items=int(tp.n_items))
    path = os.path.join(OUT, "replay.json")
    with open(path) as f:
        allres = json.load(f)
    allres.update(res)
    with open(path, "w") as f:
        json.dump(allres, f, indent=1)
    print("added trend_priority / hybrid_priority to replay.json")


def edge_cases():
    """Round 5: the in-repo layer WITH edge features (edge_feats > 0: fc_neigh takes cat(h_self, reduce(cat(h_src, e))))."""
    sageconv_case("edge_mean_toy", "mean", 4, 9, 3, 6, 5, None, 0.25, 31, edge=2)
    sageconv_case("edge_meanpool_toy", "meanpool", 4, 9, 3, 6, 5, 7, 0.25, 32, edge=3)
    sageconv_case("edge_mean_mid", "mean", 96, 700, 25, 50, 32, None, 0.1, 33, edge=5)
    sageconv_case("edge_meanpool_mid", "meanpool", 96, 700, 25, 50, 32, 40, 0.1, 34, edge=8)


if __name__ == "__main__":
    if "--edge-only" in sys.argv:
        edge_cases()
        sys.exit(0)
    edge_cases()
    #            tag            mode      n_dst n_src fanout fin fout pool  iso   seed
    sageconv_case("mean_toy", "mean", 4, 9, 3, 6, 5, None, 0.25, 11)
    sageconv_case("meanpool_toy", "meanpool", 4, 9, 3, 6, 5, 7, 0.25, 12)
    sageconv_case("gcn_toy", "gcn", 4, 9, 3, 6, 5, None, 0.25, 13)
    sageconv_case("mean_mid", "mean", 96, 700, 25, 50, 32, None, 0.1, 21)
    sageconv_case("meanpool_mid", "meanpool", 96, 700, 25, 50, 32, 40, 0.1, 22)
    sageconv_case("gcn_mid", "gcn", 96, 700, 25, 50, 32, None, 0.1, 23)
    sageconv_case("lstm_toy", "lstm", 4, 9, 3, 6, 5, None, 0.25, 14)
    sageconv_case("lstm_mid", "lstm", 96, 700, 25, 50, 32, None, 0.1, 24)
    sys.path.insert(0, OUT)
    sageconv_fullsize_case("mean")
    sageconv_fullsize_case("meanpool")
    replay_cases()
    priority_cases()
