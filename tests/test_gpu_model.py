"""End-to-end parity of the HIP path (through the C-ABI) against the oracle and against the
reference's own SAGEConv outputs (tests/golden/sageconv_*.npz).  Run with -m gpu.

Tolerances: block indices bit-exact; embeddings/logits rtol 1e-4 / atol 1e-5 (fp32 MFMA vs CPU fp32);
losses rtol 1e-4; gradients rtol 1e-3 / atol 1e-5 (atomics reorder float sums); weights after Adam atol 1e-5.
"""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import ogl_amd
    from ogl_amd import graphsage, ops, optim, sampling, synthetic, utils  # noqa: F401
    assert torch.cuda.is_available()
    return ogl_amd


def cuda(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    return (t.to(dtype) if dtype is not None else t).cuda()


def toy_graph(pkg, name="toy", evolve=8):
    from ogl_amd import synthetic
    feat_size, targets, dyn, n_classes, dyn_test = synthetic.load(name, device="cuda")
    for _ in range(evolve):
        dyn.evolve()
    return feat_size, targets, dyn, n_classes


def host_csr(g):
    h = g.handle
    keys = (h.keys if h.keys is not None else h.indices).cpu().numpy()
    return h.indptr.cpu().numpy(), h.indices.cpu().numpy(), keys


@pytest.mark.parametrize("name", ["toy", "toy_edge"])
def test_loader_matches_oracle(pkg, name):
    from ogl_amd import sampling
    _, _, dyn, _ = toy_graph(pkg, name)
    g = dyn.get_graph()
    indptr, indices, keys = host_csr(g)
    deg = O.snapshot_degrees_fast(indptr, keys, g.n_present, g.cut)
    assert np.array_equal(g.handle.degrees().cpu().numpy(), deg)
    assert g.ndata["feat"].shape[0] == g.n_present == g.number_of_nodes()
    seeds = np.random.default_rng(0).permutation(g.n_present)[:70].astype(np.int64)
    sampling.seed(11)
    loader = sampling.NodeDataLoader(g, torch.as_tensor(seeds), sampling.MultiLayerNeighborSampler([5, 5]), batch_size=32)
    batches = list(loader)
    assert [len(b[1]) for b in batches] == [32, 32, 6]
    for ctr, (input_nodes, sd, blocks) in enumerate(batches):
        want_in, want_seeds, want_blocks = O.sample_blocks(indptr, indices, deg, sd.cpu().numpy(), [5, 5], 11, ctr)
        assert np.array_equal(input_nodes.cpu().numpy(), want_in)
        for b, wb in zip(blocks, want_blocks):
            assert np.array_equal(b.local_idx.cpu().numpy(), wb["local_idx"])
            assert np.array_equal(b.srcdata[sampling.NID].cpu().numpy(), wb["src_ids"])
            assert np.array_equal(b.dstdata[sampling.NID].cpu().numpy(), wb["dst_ids"])
            assert b.number_of_dst_nodes() == len(wb["dst_ids"]) and b.number_of_src_nodes() == len(wb["src_ids"])
            assert b.number_of_edges() == int((wb["local_idx"] >= 0).sum())
        assert blocks[0].to(torch.device("cuda")) is blocks[0]
    # every pick is a snapshot in-neighbour; exactly S picks iff in-degree > 0
    picks = batches[0][2][1].picks.cpu().numpy()
    for i, d in enumerate(batches[0][1].cpu().numpy()):
        nb = indices[indptr[d]:indptr[d] + deg[d]]
        assert (picks[i] == -1).all() if deg[d] == 0 else np.isin(picks[i], nb).all()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sageconv_*.npz"))))
def test_sageconv_matches_reference_golden(pkg, path):
    """The HIP layer reproduces the reference's own SAGEConv.forward / backward (in-repo modes)."""
    from ogl_amd.graphsage import SAGEConv
    from ogl_amd.sampling import Block
    g = np.load(path)
    mode = str(g["mode"])
    li = g["local_idx"]
    n_dst, n_src = li.shape[0], g["x"].shape[0]
    fin, fout = g["x"].shape[1], g["y"].shape[1]
    pool = int(g["pool_feats"])
    edge = g["edge"] if "edge" in g.files else None               # (round 5: the layer with edge features)
    layer = SAGEConv(fin, fout, mode, activation=F.relu, pool_feats=None if pool < 0 else pool,
                     edge_feats=0 if edge is None else edge.shape[2]).cuda()
    sd = {k[len("param."):]: torch.tensor(g[k]) for k in g.files if k.startswith("param.")}
    layer.load_state_dict(sd)
    blk = Block(torch.arange(n_src).cuda(), torch.arange(n_dst).cuda(), cuda(li))
    if edge is not None:
        blk.edata["feat"] = cuda(edge)
    x = cuda(g["x"]).requires_grad_(True)
    y = layer(blk, x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    # relu mask from the kernel's own output keeps the comparison well-posed at y ~ 0
    y.backward(cuda(g["gy"]))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx"], rtol=1e-3, atol=1e-5)
    for k, p in layer.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g["grad." + k], rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("gemm", ["f32", "auto"])
@pytest.mark.parametrize("mode", ["mean", "meanpool"])
def test_sageconv_matches_reference_golden_at_the_baseline_shape(pkg, mode, gemm):
    """SURVEY.md §8(c) G1 at the BASELINE shape: the HIP layer against the reference's own SAGEConv on the 512 x 25 x 602 block
    (pool_feats 600, 600 outputs).  Inputs / weights: the PCG64 stream of tests/golden/fullsize_inputs.py, regenerated here; the
    reference's outputs: tests/golden/fullsize_<mode>.npz.  §8(c) tolerances: rtol 1e-4 / atol 1e-5 on the embeddings, gradients
    1e-3 (atomics reorder sums) with the absolute part scaled to each tensor's size."""
    import sys
    from ogl_amd import ops
    from ogl_amd.graphsage import SAGEConv
    from ogl_amd.sampling import Block
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import fullsize_inputs as FI
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_%s.npz" % mode))
    inp = FI.make(mode)
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(gemm)
    try:
        layer = SAGEConv(FI.FIN, FI.FOUT, mode, activation=F.relu, pool_feats=FI.POOL if mode == "meanpool" else None).cuda()
        layer.load_state_dict({k: torch.tensor(v) for k, v in inp["params"].items()})
        blk = Block(torch.arange(FI.N_SRC).cuda(), torch.arange(FI.N_DST).cuda(), cuda(inp["local_idx"]))
        x = cuda(inp["x"]).requires_grad_(True)
        y = layer(blk, x)
        np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)
        y.backward(cuda(inp["gy"]))
        proj, norms, rows = FI.digest(x.grad.cpu().numpy(), inp)
        np.testing.assert_allclose(proj, g["gx_proj"], rtol=1e-3, atol=1e-5 * float(np.abs(g["gx_proj"]).max()))
        np.testing.assert_allclose(norms, g["gx_norms"], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(rows, g["gx_rows"], rtol=1e-3, atol=1e-5)
        for k, p in layer.named_parameters():
            got = p.grad.cpu().numpy()
            got = got[::FI.GRAD_ROW_STRIDE] if got.ndim == 2 else got
            np.testing.assert_allclose(got, g["grad." + k], rtol=1e-3, atol=2e-5 * float(np.abs(g["grad." + k]).max()))
    finally:
        ops.set_gemm_mode(prev)


def _copy_params(model, layer_params):
    with torch.no_grad():
        for l, prm in zip(model.layers, layer_params):
            for k, v in prm.items():
                mod, attr = k.split(".")
                getattr(getattr(l, mod), attr).copy_(v)


@pytest.mark.parametrize("mode,pool", [("pool", None), ("meanpool", 12), ("mean", None), ("gcn", None), ("maxpool", 9)])
@pytest.mark.parametrize("fuse", [True, False])
def test_two_layer_model_step_matches_oracle(pkg, mode, pool, fuse):
    from ogl_amd import ops, optim, sampling
    from ogl_amd.graphsage import GatheredRows, GraphSAGE
    feat_size, _, dyn, n_classes = toy_graph(pkg)
    g = dyn.get_graph()
    torch.manual_seed(5)
    cpu = O.CpuModel(mode, feat_size, 16, n_classes, pool_feats=pool, seed=5)
    model = GraphSAGE(feat_size, 16, n_classes, 1, F.relu, 0, mode, edge_feats=0, pool_feats=pool).cuda()
    assert sorted(model.state_dict()) == sorted("layers.%d.%s" % (i, k) for i in range(2) for k in cpu.params[i])
    _copy_params(model, [{k: v.detach() for k, v in p.items()} for p in cpu.params])
    opt = optim.Adam(model.parameters(), lr=1e-3)
    indptr, indices, keys = host_csr(g)
    deg = O.snapshot_degrees_fast(indptr, keys, g.n_present, g.cut)
    feat_cpu = g.ndata["feat"].cpu().contiguous()
    lab_cpu = g.ndata["target"].cpu()
    sampling.seed(3)
    seeds = torch.as_tensor(np.random.default_rng(1).permutation(g.n_present)[:96].astype(np.int64))
    loader = sampling.NodeDataLoader(g, seeds, sampling.MultiLayerNeighborSampler([7, 7]), batch_size=48)
    for ctr, (input_nodes, sd, blocks) in enumerate(loader):
        x = GatheredRows(g.ndata["feat"], input_nodes) if fuse else ops.gather_rows(g.ndata["feat"], input_nodes)
        labels = ops.gather_i64(g.ndata["target"], sd)
        opt.zero_grad()
        logits = model(blocks, x)
        loss_rows = ops.cross_entropy(logits, labels, "none")
        loss = loss_rows.mean()
        loss.backward()
        # oracle: same seeds, same Philox stream
        in_ref, _, blocks_ref = O.sample_blocks(indptr, indices, deg, sd.cpu().numpy(), [7, 7], 3, ctr)
        assert np.array_equal(in_ref, input_nodes.cpu().numpy())
        cpu.opt.zero_grad()
        logits_ref = cpu.forward(feat_cpu[torch.as_tensor(in_ref)], blocks_ref)
        rows_ref = O.cross_entropy(logits_ref, lab_cpu[sd.cpu()], "none")
        rows_ref.mean().backward()
        np.testing.assert_allclose(logits.detach().cpu().numpy(), logits_ref.detach().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(loss_rows.detach().cpu().numpy(), rows_ref.detach().numpy(), rtol=1e-4, atol=1e-6)
        for l, prm in zip(model.layers, cpu.params):
            for k, v in prm.items():
                mod, attr = k.split(".")
                got = getattr(getattr(l, mod), attr).grad.cpu().numpy()
                np.testing.assert_allclose(got, v.grad.numpy(), rtol=1e-3, atol=1e-5, err_msg="%s %s" % (mode, k))
        opt.step()
        cpu.opt.step()
    for l, prm in zip(model.layers, cpu.params):
        for k, v in prm.items():
            mod, attr = k.split(".")
            np.testing.assert_allclose(getattr(getattr(l, mod), attr).detach().cpu().numpy(), v.detach().numpy(),
                                       rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode,pool", [("pool", None), ("maxpool", 9), ("meanpool", 12)])
def test_cached_projection_inference_matches_per_batch(pkg, mode, pool):
    """Priority-forward fast path: logits from the cached relu(fc_pool_0(X)) table + unrelabelled input block equal
    the per-batch path on the same Philox stream (same picks), within GEMM tolerance."""
    from ogl_amd import sampling
    from ogl_amd.graphsage import GatheredRows, GraphSAGE
    feat_size, _, dyn, n_classes = toy_graph(pkg)
    g = dyn.get_graph()
    torch.manual_seed(2)
    model = GraphSAGE(feat_size, 16, n_classes, 1, F.relu, 0, mode, edge_feats=0, pool_feats=pool).cuda().eval()
    seeds = torch.as_tensor(np.random.default_rng(4).permutation(g.n_present)[:150].astype(np.int64))
    smp = sampling.MultiLayerNeighborSampler([6, 6])
    with torch.no_grad():
        sampling.seed(21)
        ref = [model(b, GatheredRows(g.ndata["feat"], i)) for i, _, b in sampling.NodeDataLoader(g, seeds, smp, batch_size=64)]
        sampling.seed(21)
        proj = model.layers[0].project_tables(g.ndata["feat"])
        got = []
        for i, sd, b in sampling.NodeDataLoader(g, seeds, smp, batch_size=64, relabel_input=False):
            assert i is None and b[0].local_idx is None and b[0].picks.shape == (b[1].number_of_src_nodes(), 6)
            got.append(model(b, GatheredRows(g.ndata["feat"], None, proj)))
    assert len(ref) == len(got) == 3
    for a, b in zip(ref, got):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=1e-5)
    # training mode refuses the cache
    model.train()
    with pytest.raises(RuntimeError):
        model(b if False else sampling.NodeDataLoader(g, seeds[:8], smp, batch_size=8, relabel_input=False).__iter__().__next__()[2],
              GatheredRows(g.ndata["feat"], None, proj))


def test_model_step_bf16x6_mode(pkg):
    """The whole train step under the split-bf16 GEMM mode meets the same tolerances against the oracle."""
    from ogl_amd import ops
    ops.set_gemm_mode("bf16x6")
    try:
        test_two_layer_model_step_matches_oracle(pkg, "pool", None, True)
        test_two_layer_model_step_matches_oracle(pkg, "meanpool", 12, True)
    finally:
        ops.set_gemm_mode("f32")


def test_unknown_aggregator_raises_keyerror(pkg):
    from ogl_amd.graphsage import SAGEConv
    from ogl_amd.sampling import Block
    layer = SAGEConv(4, 3, "bogus").cuda() if False else None
    with pytest.raises(KeyError):
        l2 = SAGEConv.__new__(SAGEConv)
        torch.nn.Module.__init__(l2)
        l2._aggre_type = "bogus"; l2.feat_drop = torch.nn.Dropout(0.0)
        l2(Block(torch.arange(3).cuda(), torch.arange(1).cuda(), torch.zeros((1, 2), dtype=torch.int32).cuda()),
           torch.zeros(3, 4).cuda())
    assert layer is None


def test_strategies_end_to_end(pkg, tmp_path):
    """utils.init 6-tuple -> the reference driver's loop body (R/train/__main__.py:99-196) on a toy stream."""
    import random
    from ogl_amd import sampling, synthetic
    from ogl_amd.graph import TrainTestGraph
    from ogl_amd.prioritized_replay import LossPriority
    from ogl_amd.utils import Lib_supported, init
    np.random.seed(1); random.seed(1); torch.manual_seed(1); sampling.seed(1)
    GraphSAGE, Random, Prioritized, NoReh, Full, act = init(Lib_supported.HIP, True, 0)
    feat_size, labels, graph, n_classes, graph_test = synthetic.load("toy", device="cuda")
    delta = 2
    for _ in range(delta):
        graph_test.evolve()
    gu = TrainTestGraph(graph, split=0.15, start_prior_alpha=4, end_prior_alpha=50, scale=1, max_priority=10)
    mk = lambda: GraphSAGE(feat_size, 8, n_classes, 1, act, 0, "pool", edge_feats=0, pool_feats=8).cuda()
    rnd = Random(mk(), 2, 8, labels, 5, cuda=True, batch_full=64); rnd.build_optimizer()
    pri = Prioritized(mk(), 2, 8, labels, 5, LossPriority(), cuda=True, full_pass=1, batch_full=64); pri.build_optimizer()
    nor = NoReh(mk(), 2, 8, labels, 5, cuda=True, batch_full=64); nor.build_optimizer()
    ful = Full(mk(), 1, 8, labels, 5, cuda=True, batch_full=64); ful.build_optimizer()
    out = str(tmp_path / "res.csv")
    w0 = rnd.graphsage_model.layers[0].fc_pool.weight.detach().clone()
    for t in range(6):
        rnd.train_timestep(gu); pri.train_timestep(gu); nor.train_timestep(gu)
        if t % 3 == 0:
            ful.train_timestep(gu)
        if t % 2 == 0:
            for s in (rnd, pri, nor, ful):
                s.evaluate(gu, out)
                s.evaluate_next_snapshots(graph_test, delta, out)
        if t + delta + 1 < len(gu):
            gu.evolve(); graph_test.evolve()
    assert rnd.delay > 0 and not torch.equal(w0, rnd.graphsage_model.layers[0].fc_pool.weight)
    rows = open(out).read().strip().split("\n")
    assert len(rows) == 3 * 4 * 2
    names = {r.split(";")[0] for r in rows}
    assert names == {"random", "prioritized", "no_rehersal", "offline"}
    full_rows = [r for r in rows if r.split(";")[1] != ""]
    assert full_rows and all(0.0 <= float(r.split(";")[1]) <= 1.0 for r in full_rows)
    # priorities were written by the priority forward: every train vertex has a finite tree weight
    pr = gu.dump_priorities(gu.get_train_set())
    assert len(pr) == len(gu.get_train_set()) and np.isfinite(pr).all()
    assert [s.get_model() for s in (rnd, pri, nor, ful)] == ["random", "prioritized", "no_rehersal", "offline"]


def test_state_dict_roundtrip_names(pkg, tmp_path):
    from ogl_amd.graphsage import GraphSAGE
    m = GraphSAGE(20, 8, 4, 1, F.relu, 0, "pool").cuda()
    keys = sorted(m.state_dict())
    assert keys == sorted("layers.%d.%s.%s" % (i, fc, p) for i in range(2) for fc in ("fc_pool", "fc_self", "fc_neigh")
                          for p in ("weight", "bias"))
    assert m.layers[0].fc_pool.weight.shape == (20, 20) and m.layers[1].fc_neigh.weight.shape == (4, 8)
    torch.save(m.state_dict(), tmp_path / "gnn.pt")
    m2 = GraphSAGE(20, 8, 4, 1, F.relu, 0, "pool").cuda()
    m2.load_state_dict(torch.load(tmp_path / "gnn.pt"))
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_on_disk_loaders_roundtrip(pkg, tmp_path):
    """Write a toy dataset in the reference's file formats, load it through dataset_utils, and check the streams
    equal the ones built directly from the same arrays."""
    import json
    from ogl_amd import dataset_utils, synthetic
    a = synthetic.make_arrays("toy")
    d = tmp_path / "pubmed"; d.mkdir()
    np.save(d / "feat_data.npy", a["feat"].numpy().astype(np.float64))
    lab = a["labels"].copy(); lab[::10] = -1
    np.save(d / "targets.npy", lab.astype(np.float64))
    adj = {}
    for u, v in zip(a["src"].tolist(), a["dst"].tolist()):
        adj.setdefault(u, []).append(v)
    (d / "graph.adjlist").write_text("\n".join("%d %s" % (u, " ".join(map(str, vs))) for u, vs in adj.items()) + "\n")
    ts = {int(v): float(t) for t, v in enumerate(a["order"])}
    (d / "postponed_timestamp.json").write_text(json.dumps(ts))
    feat_size, targets, dyn, n_classes, dyn_test = dataset_utils.pubmed.load(str(d), snapshots=20)
    assert feat_size == 20 and targets.shape == (600, 1) and n_classes == len(np.unique(lab))
    g = dyn.get_graph()
    assert g.n_present == 30 and len(dyn) == 20
    assert 0 not in dyn.labelled_vertices and 1 in dyn.labelled_vertices
    # features arrive in arrival order, float64 -> float32
    order = np.array(a["order"][:30])
    np.testing.assert_array_equal(g.ndata["feat"].cpu().numpy(), a["feat"].numpy()[order])
    assert np.array_equal(dyn.get_subgraph_to_original_map(), order)
    dyn.evolve(); dyn_test.evolve(); dyn_test.evolve()
    assert dyn.get_graph().n_present == 60 and dyn_test.get_graph().n_present == 90
    # the two streams are snapshot views of ONE resident copy of the static data (feature / label tables, CSR), each with its own
    # per-snapshot degrees
    gt = dyn_test.get_graph()
    assert gt is not dyn.get_graph() and gt.feat_table.data_ptr() == dyn.get_graph().feat_table.data_ptr()
    assert gt.target_table.data_ptr() == dyn.get_graph().target_table.data_ptr() and gt.handle._h.value != dyn.get_graph().handle._h.value
    d60, d90 = dyn.get_graph().handle.degrees().cpu().numpy(), gt.handle.degrees().cpu().numpy()
    assert (d60[60:] == 0).all() and (d90[:60] >= d60[:60]).all() and d90.sum() > d60.sum()
    # edge stream
    e = synthetic.make_arrays("toy_edge")
    d2 = tmp_path / "reddit"; d2.mkdir()
    np.save(d2 / "feat_data.npy", e["feat"].numpy().astype(np.float64)); np.save(d2 / "targets.npy", e["labels"])
    rows = ["idx,src,dst"] + ["%d,%d,%d" % (i, s, t) for i, (s, t) in enumerate(zip(e["src"], e["dst"]))]
    (d2 / "edges_dataframe.csv").write_text("\n".join(rows) + "\n")
    fs, tg, dyn_e, nc, _ = dataset_utils.reddit.load(str(d2), snapshots=20)
    ge = dyn_e.get_graph()
    assert fs == 20 and ge.cut == len(e["src"]) // 20 and nc == 4
    with pytest.raises(FileNotFoundError):
        dataset_utils.arxiv.load(str(d), snapshots=5)
