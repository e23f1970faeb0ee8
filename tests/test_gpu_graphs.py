"""Captured train steps (stepgraph.py) against the eager launches of the same strategy: same seeds, same Philox counters,
same initial weights -> the same losses and updates up to fp32 summation order (padded rows add exact zeros; split
reductions may cut the padded length elsewhere).  The oracle parity of the captured path itself is in test_gpu_rungs.py
(graphs=True).  Run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _twins(feat_size, H, n_classes, cls, *args, **kw):
    from ogl_amd.graphsage import GraphSAGE
    out = []
    for graphs in (False, True):
        torch.manual_seed(9)
        model = GraphSAGE(feat_size, H, n_classes, 1, F.relu, 0, "pool", edge_feats=0, pool_feats=H).cuda()
        st = cls(model, *args, **kw)
        st.use_graphs = graphs
        st.build_optimizer()
        out.append(st)
    return out


def _weights_close(a, b, frac=1e-3):
    """Adam turns a gradient within summation noise of zero into a step anywhere in [-lr, lr] (see test_gpu_rungs.py):
    per entry 2e-5, at most ``frac`` of the entries outside, none by more than 2 lr per step."""
    bad = total = 0
    for x, y in zip(a.parameters(), b.parameters()):
        d = (x.detach() - y.detach()).abs()
        bad += int((d > 2e-5).sum()); total += d.numel()
    assert bad <= frac * total, (bad, total)


@pytest.mark.parametrize("name,B,S,H", [("toy", 16, 5, 16), ("pubmed", 32, 25, 32)])
def test_sampled_graph_steps_equal_eager(name, B, S, H):
    import ogl_amd  # noqa: F401
    from ogl_amd import sampling, synthetic
    from ogl_amd.graphsage.model import PrioritizedHipSupervisedGraphSage, RandomHipSupervisedGraphSage
    from ogl_amd.prioritized_replay import LossPriority
    feat_size, labels, dyn, n_classes, _ = synthetic.load(name, snapshots=4, device="cuda")
    dyn.evolve(); dyn.evolve()
    g = dyn.get_graph()
    rng = np.random.default_rng(0)
    for cls, extra in ((RandomHipSupervisedGraphSage, ()), (PrioritizedHipSupervisedGraphSage, (LossPriority(),))):
        eager, graph = _twins(feat_size, H, n_classes, cls, 3, B, labels, S, *extra, cuda=True, batch_full=256)
        # (snapshot 2 is run on a smaller cut, see below: its seeds must be present there)
        seeds = [rng.choice(g.n_present - (7 if k == 2 else 0), 3 * B + (5 if k == 1 else 0), replace=False).astype(np.int64)
                 for k in range(3)]
        out = {}
        for st in (eager, graph):
            rec = []
            st.step_hook = lambda info, rec=rec: rec.append((info["form"], float(info["loss"])))
            rows = []
            sampling.seed(4)
            for k, sd in enumerate(seeds):
                if k == 2:
                    dyn_n = g.n_present                      # the snapshot moves on under the SAME captured graph
                    g.set_snapshot(max(dyn_n - 7, 1), max(dyn_n - 7, 1))
                st._train_batches(g, sd, B, on_rows=(lambda s_, r_: rows.append(r_.cpu())) if extra else None)
                if k == 2:
                    g.set_snapshot(dyn_n, dyn_n)
            out[st.use_graphs] = (rec, rows, sampling.get_state()["ctr"])
        (rec_e, rows_e, ctr_e), (rec_g, rows_g, ctr_g) = out[False], out[True]
        assert ctr_e == ctr_g == 3 + 4 + 3
        assert [f for f, _ in rec_e] == ["eager"] * 10
        # three full batches per snapshot replay the captured step; the ragged batch of snapshot 1 runs eagerly
        assert [f for f, _ in rec_g] == ["sampled"] * 3 + ["sampled"] * 3 + ["eager"] + ["sampled"] * 3
        np.testing.assert_allclose([l for _, l in rec_g], [l for _, l in rec_e], rtol=2e-4)
        if extra:
            assert len(rows_e) == len(rows_g) == 10
            for a, b in zip(rows_e, rows_g):
                np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=2e-4, atol=1e-6)
        _weights_close(eager.graphsage_model, graph.graphsage_model, frac=5e-3)
        assert 1 <= graph._step_graphs().captures <= 6          # train graphs: one per input-size bucket met


def test_pipelined_steps_hand_out_their_own_seeds():
    """The pipelined captured step (batch i + 1 sampled on a second stream while batch i trains, two sets of block arrays): what
    ``on_rows`` receives for batch i — seeds AND per-seed losses, read from the step's static buffers without a host sync — are batch
    i's, also when the host runs far ahead of the device (the re-sampling of a set is ordered behind the caller's reads of it)."""
    import ogl_amd  # noqa: F401
    from ogl_amd import sampling, synthetic
    from ogl_amd.graphsage import model as M
    from ogl_amd.prioritized_replay import LossPriority
    assert M.SAMPLE_PIPELINE
    B, S, nb = 32, 25, 12
    feat_size, labels, dyn, n_classes, _ = synthetic.load("pubmed", snapshots=4, device="cuda")
    dyn.evolve(); dyn.evolve()
    g = dyn.get_graph()
    eager, graph = _twins(feat_size, 32, n_classes, M.PrioritizedHipSupervisedGraphSage, nb, B, labels, S, LossPriority(), cuda=True,
                          batch_full=256)
    seeds = np.random.default_rng(3).choice(g.n_present, nb * B, replace=False).astype(np.int64)
    got = {}
    for st in (eager, graph):
        for rep in range(2):                                  # (second pass: every graph is captured, the host only enqueues)
            rows = []
            sampling.seed(11)
            st._train_batches(g, seeds, B, on_rows=lambda s_, r_: rows.append((s_, r_)))
        torch.cuda.synchronize()
        got[st.use_graphs] = [(a.cpu().numpy(), b.cpu().numpy()) for a, b in rows]
    assert len(got[True]) == len(got[False]) == nb
    for i, ((sd_g, _), (sd_e, _)) in enumerate(zip(got[True], got[False])):
        np.testing.assert_array_equal(sd_g, seeds[i * B:(i + 1) * B], err_msg="batch %d" % i)
        np.testing.assert_array_equal(sd_e, seeds[i * B:(i + 1) * B])


def test_staged_graph_steps_equal_eager_reddit_size():
    """The Reddit rung: loader-sampled batches staged into per-bucket captured steps; 4 steps of 512 seeds."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops, sampling, stepgraph, synthetic
    from ogl_amd.graphsage.model import RandomHipSupervisedGraphSage
    # as in a fresh process: the first capture is preceded by a real forward + backward WITHOUT an optimiser step — whatever that
    # pass caches (weight images) must not leak into the capture
    stepgraph._WARMED = False
    feat_size, labels, dyn, n_classes, _ = synthetic.load("reddit", snapshots=2, device="cuda")
    dyn.evolve()
    g = dyn.get_graph()
    ops.set_gemm_mode("auto")
    try:
        eager, graph = _twins(feat_size, 600, n_classes, RandomHipSupervisedGraphSage, 4, 512, labels, 25, cuda=True, batch_full=1024)
        seeds = np.random.default_rng(1).choice(g.n_present, 4 * 512, replace=False).astype(np.int64)
        losses = {}
        for st in (eager, graph):
            rec = []
            st.step_hook = lambda info, rec=rec: rec.append((info["form"], float(info["loss"]), info["n0"], info["n1"]))
            sampling.seed(8)
            st._train_batches(g, seeds, 512)
            losses[st.use_graphs] = rec
        assert [r[0] for r in losses[True]] == ["staged"] * 4 and [r[0] for r in losses[False]] == ["eager"] * 4
        assert [r[2:] for r in losses[True]] == [r[2:] for r in losses[False]]           # the same sampled blocks
        np.testing.assert_allclose([r[1] for r in losses[True]], [r[1] for r in losses[False]], rtol=2e-4)
        # four Adam steps: ~0.35 % of the 1.5 M entries per step have a gradient inside the summation noise of the two orders
        _weights_close(eager.graphsage_model, graph.graphsage_model, frac=2.5e-2)
        assert 1 <= graph._step_graphs().captures <= 4
    finally:
        ops.set_gemm_mode("f32")


def test_padded_small_block_build_bit_exact():
    """ogl_build_block_padded (one workgroup) == the oracle's relabelling, with -1 padded destinations and the -1 tail."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    from oracle import oracle as O
    rng = np.random.default_rng(2)
    for n_dst, fan, n_valid, hi in ((32, 25, 32, 5000), (832, 25, 120, 300), (1500, 25, 1500, 100000), (7, 3, 5, 9)):
        dst = np.full(n_dst, -1, dtype=np.int64)
        dst[:n_valid] = rng.choice(hi, n_valid, replace=False)
        picks = rng.integers(0, hi, size=(n_dst, fan)).astype(np.int64)
        picks[rng.random(n_dst) < 0.2] = -1
        picks[n_valid:] = -1
        # the padded entry point (one workgroup up to 4 096 flat positions, the parallel phases with the fill folded into the
        # table reset above) ...
        src, n_src, lidx = ops.build_block_async(torch.as_tensor(dst).cuda(), torch.as_tensor(picks).cuda(), pad_tail=True)
        # ... against the plain multi-launch build + a separate fill
        ops.BLOCK_SMALL_MAX_P, keep = 0, ops.BLOCK_SMALL_MAX_P
        try:
            src2, n_src2, lidx2 = ops.build_block_async(torch.as_tensor(dst).cuda(), torch.as_tensor(picks).cuda(), pad_tail=True)
        finally:
            ops.BLOCK_SMALL_MAX_P = keep
        assert torch.equal(src, src2) and torch.equal(lidx, lidx2) and torch.equal(n_src, n_src2)
        n = int(n_src.item())
        # oracle on the valid destinations only; padded destinations keep a row each (id -1), so new sources start at n_dst
        want_src, want_l = O.build_block(dst[:n_valid], picks[:n_valid])
        new = want_src[n_valid:]
        assert n == n_dst + len(new)
        got = src.cpu().numpy()
        assert np.array_equal(got[:n_valid], dst[:n_valid]) and (got[n_valid:n_dst] == -1).all()
        assert np.array_equal(got[n_dst:n], new) and (got[n:] == -1).all()
        gl = lidx.cpu().numpy()
        shifted = np.where(want_l >= n_valid, want_l + (n_dst - n_valid), want_l)
        assert np.array_equal(gl[:n_valid], shifted) and (gl[n_valid:] == -1).all()


def test_graph_cache_is_bounded():
    """Train graphs are kept least-recently-used: a stream that walks through many size buckets does not accumulate pools."""
    import ogl_amd  # noqa: F401
    from ogl_amd import sampling, synthetic
    from ogl_amd.graphsage.model import RandomHipSupervisedGraphSage
    feat_size, labels, dyn, n_classes, _ = synthetic.load("toy", snapshots=4, device="cuda")
    dyn.evolve(); dyn.evolve()
    g = dyn.get_graph()
    (_, st) = _twins(feat_size, 8, n_classes, RandomHipSupervisedGraphSage, 1, 16, labels, 5, cuda=True, batch_full=64)
    cache = st._step_graphs()
    cache.MAX_GRAPHS = 2
    import ogl_amd.stepgraph as sgm
    keep, keep_agn = sgm.N0_BUCKET_SMALL, sgm.SIZE_AGNOSTIC
    sgm.N0_BUCKET_SMALL = 4                                   # tiny buckets: nearly every batch is a new one
    sgm.SIZE_AGNOSTIC = False                                 # (the bucketed form: a size-agnostic step is ONE graph whatever the batch)
    try:
        rng = np.random.default_rng(0)
        sampling.seed(1)
        losses = []
        st.step_hook = lambda info: losses.append(float(info["loss"]))
        for _ in range(12):
            st._train_batches(g, rng.choice(g.n_present, 16, replace=False).astype(np.int64), 16)
        assert len(cache.graphs) <= 2 and cache.evictions >= 1 and cache.captures >= 3
        assert np.isfinite(losses).all() and len(losses) == 12
    finally:
        sgm.N0_BUCKET_SMALL, sgm.SIZE_AGNOSTIC = keep, keep_agn
