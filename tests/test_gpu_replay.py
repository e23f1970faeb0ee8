"""Device-side prioritised replay structure (prioritized_replay/device_buffer.py, csrc/replay.hip) against the
reference's own outputs (tests/golden/replay.json, produced by running R/train/prioritized_replay/*) and against the
golden-pinned host class on random workloads.  Run with -m gpu.

Tolerances: dumped priorities rtol 1e-12 (fp64 on the device; its log / pow are not glibc's bit for bit) — the fixture is
also checked at the 1e-6 an fp32 structure would be held to; sampled index SETS exact under the same ``random.seed``;
running extrema exact."""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "replay.json")))


def _dev(*a, **k):
    import ogl_amd  # noqa: F401
    from ogl_amd.prioritized_replay.device_buffer import DevicePrioritizedReplayBuffer
    return DevicePrioritizedReplayBuffer(*a, device="cuda", **k)


def test_device_buffer_matches_reference_golden():
    g = GOLD["buffer"]
    buf = _dev(g["size"], g["alpha"], max_priority=g["max_priority"], min_priority=g["min_priority"])
    first = {int(k): v for k, v in g["first"].items()}
    second = {int(k): v for k, v in g["second"].items()}
    buf.add_all(first)
    np.testing.assert_allclose(buf.dump_priorities(list(first)), g["dump_after_first"], rtol=1e-12, atol=0)
    buf.add_all(second)
    keys = list(first) + list(second)
    np.testing.assert_allclose(buf.dump_priorities(keys), g["dump_after_second"], rtol=1e-12, atol=0)
    buf.update_priorities({int(k): v for k, v in g["update"].items()})
    got = buf.dump_priorities(keys)
    np.testing.assert_allclose(got, g["dump_after_update"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(np.float32(got), np.float32(g["dump_after_update"]), rtol=1e-6, atol=0)
    assert buf.get_min_priority() == g["min_val"] and buf.get_max_priority() == g["max_val"]
    assert [int(x) for x in buf._storage] == g["storage"]
    buf.check_errors()
    for seed, n, key in ((1, 8, "sample8_seed1"), (2, 16, "sample16_seed2"), (3, 100, "sample100_seed3")):
        random.seed(seed)
        assert sorted(int(i) for i in buf._sample_proportional(n)) == g[key]
    random.seed(4)
    got = buf.sample(8)
    assert len(got) == 8 and set(got) <= set(keys)


def test_device_tree_walks_match_reference_tree_golden():
    """The golden sum tree (capacity 8): p_total excludes the last item; a walk lands on the golden leaf for every query that
    is not exactly on a leaf boundary (masses reach the kernel as u * p_total)."""
    import ogl_amd  # noqa: F401
    from ogl_amd import _lib
    g = GOLD["tree"]
    cap = g["capacity"]
    node = torch.zeros(2 * cap, dtype=torch.float64, device="cuda")
    node[cap:] = torch.tensor(g["values"], dtype=torch.float64)
    _lib.check(_lib.lib().ogl_replay_rebuild(node.data_ptr(), cap, None))
    torch.cuda.synchronize()
    assert float(node[1]) == g["sum_all"]
    ptot_want = g["sum_all"] - g["values"][-1]
    qs = [(q, i) for q, i in zip(g["prefix_queries"], g["prefix_idx"]) if q not in (0.5, 1.75, 3.75) and q < ptot_want]
    u = torch.tensor([q / ptot_want for q, _ in qs], dtype=torch.float64, device="cuda")
    out = torch.empty(1 + len(qs), dtype=torch.int64, device="cuda")
    ptot = torch.empty(1, dtype=torch.float64, device="cuda")
    u0 = torch.zeros(1, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().ogl_replay_sample(node.data_ptr(), cap, cap, 1, u0.data_ptr(), u.data_ptr(), len(qs), out.data_ptr(),
                                            ptot.data_ptr(), None))
    assert float(ptot) == ptot_want
    assert out[1:].cpu().tolist() == [i for _, i in qs] and int(out[0]) == 0


@pytest.mark.parametrize("alpha", [4, 7.3])
def test_device_buffer_equals_host_class_on_random_workloads(alpha):
    import ogl_amd  # noqa: F401
    from ogl_amd.prioritized_replay.replay_buffer import PrioritizedReplayBuffer
    rng = np.random.default_rng(0)
    host = PrioritizedReplayBuffer(10 ** 7, alpha, max_priority=10, min_priority=1e-7)
    dev = _dev(10 ** 7, alpha, max_priority=10, min_priority=1e-7, key_space=64)
    keys = rng.permutation(50000)[:3000].astype(np.int64)                  # more than the initial 1 024 leaves: the tree grows
    # admission of the first arrivals (nothing scored yet), then scored updates, then more arrivals at lo + 0.95 (hi - lo)
    host.add_all_arrays(keys[:700], np.full(700, 2.0)); dev.admit(keys[:700], 2.0)
    losses = (rng.standard_exponential(500) * 3).astype(np.float32); losses[:5] = [0.0, 1e-12, 50.0, 10.0, 1e-7]
    host.update_arrays(keys[100:600], losses.astype(np.float64))
    dev.update_device(torch.as_tensor(keys[100:600]).cuda(), torch.as_tensor(losses).cuda())   # float32 device losses, as PBR
    hi, lo = host.get_max_priority(), host.get_min_priority()
    host.add_all_arrays(keys[700:], np.full(2300, lo + (hi - lo) * 0.95)); dev.admit(keys[700:], 2.0)
    l2 = rng.standard_exponential(2000).astype(np.float64)
    host.update_arrays(keys[500:2500], l2); dev.update_arrays(keys[500:2500], l2)
    dev.check_errors()
    np.testing.assert_allclose(dev.dump_priorities(list(keys)), host.dump_priorities(list(keys)), rtol=1e-12, atol=1e-300)
    assert (dev.get_max_priority(), dev.get_min_priority()) == (host.get_max_priority(), host.get_min_priority())
    st = dev.state.cpu().tolist()
    assert st[0] == host._max_priority and st[1] == host._min_priority
    assert dev._storage == host._storage and len(dev) == len(host) == 3000
    # the same draws under the same Python stream — stratified, with re-draws / top-up when the batch nears the buffer size
    for seed, n in ((1, 32), (2, 512), (3, 2900), (4, 2999), (5, 3000), (6, 5000)):
        random.seed(seed); a = host._sample_proportional(n)
        sa = random.getstate()
        random.seed(seed); b = dev._sample_proportional(n)
        assert sorted(a) == sorted(b) and random.getstate() == sa, (seed, n)   # same set AND the same stream position after
    random.seed(9); ha = host.sample(64)
    random.seed(9); da = dev.sample(64)
    assert ha == da                                                           # same ids in the same (set-iteration) order
    # to_host(): the equivalent reference-semantics object
    h2 = dev.to_host()
    np.testing.assert_allclose(h2.dump_priorities(list(keys)), host.dump_priorities(list(keys)), rtol=1e-12, atol=1e-300)
    # a key that is not in the buffer raises the flag (the host class raises KeyError)
    dev.update_device(torch.tensor([49999 if 49999 not in set(keys.tolist()) else 49998]).cuda(), torch.ones(1, device="cuda"))
    with pytest.raises(AssertionError):
        dev.check_errors()


def test_pbr_passes_feed_the_device_buffer_without_host_copies():
    """The PBR train update and the priority forward write their per-seed losses into the HBM buffer with no float tensor
    crossing to the host; the resulting buffer equals the host-buffer twin's."""
    import torch.nn.functional as F
    import ogl_amd  # noqa: F401
    from ogl_amd import sampling, synthetic
    from ogl_amd.graph import TrainTestGraph
    from ogl_amd.graphsage import GraphSAGE
    from ogl_amd.graphsage.model import PrioritizedHipSupervisedGraphSage
    from ogl_amd.prioritized_replay import LossPriority
    outs = {}
    for device_replay in (True, False):
        np.random.seed(1); random.seed(1); torch.manual_seed(1); sampling.seed(1)
        feat_size, labels, dyn, n_classes, _ = synthetic.load("toy", device="cuda")
        for _ in range(5):
            dyn.evolve()
        gu = TrainTestGraph(dyn, split=0.15, start_prior_alpha=4, end_prior_alpha=50, scale=1, max_priority=10,
                            device_replay=device_replay)
        assert gu.device_replay == device_replay
        model = GraphSAGE(feat_size, 8, n_classes, 1, F.relu, 0, "pool").cuda()
        pri = PrioritizedHipSupervisedGraphSage(model, 2, 8, labels, 5, LossPriority(), full_pass=1, cuda=True, batch_full=64)
        pri.use_graphs = False
        pri.build_optimizer()
        moved = []
        orig_cpu = torch.Tensor.cpu
        torch.Tensor.cpu = lambda t, *a, **k: (moved.append(t.dtype), orig_cpu(t, *a, **k))[1]
        try:
            for _ in range(3):
                pri.train_timestep(gu)
                gu.evolve()
        finally:
            torch.Tensor.cpu = orig_cpu
        if device_replay:
            assert not [d for d in moved if d.is_floating_point], moved       # block sizes (int64) only
            gu.priority_replay_buffer.check_errors()
        else:
            assert [d for d in moved if d.is_floating_point]                 # the host buffer needs the losses on the host
        outs[device_replay] = np.asarray(gu.dump_priorities(gu.get_train_set()))
    # two independent training runs: the backward's float atomics sum in a run-dependent order, so weights — and losses — agree
    # to ~1e-7 relative, not bit for bit; the buffer arithmetic itself is held to 1e-12 in the tests above
    np.testing.assert_allclose(outs[True], outs[False], rtol=1e-4, atol=1e-12)


@pytest.mark.parametrize("name", ["trend_priority", "hybrid_priority"])
def test_device_trend_priorities_match_reference_golden(name):
    """TrendPriority / HybridPriority with their state in HBM (ogl_priority_trend) against the reference's own outputs
    (tests/golden/replay.json, generated by running R/train/prioritized_replay/generate_priority.py:11-58)."""
    import json
    import os
    from ogl_amd.prioritized_replay import HybridPriority, TrendPriority
    from ogl_amd.prioritized_replay.priorities import DeviceTrend
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "replay.json")))[name]
    host = TrendPriority(g["n_vertices"], g["alpha"]) if name == "trend_priority" else HybridPriority(g["n_vertices"], g["alpha"], g["loss_contrib"])
    dev = DeviceTrend(host, "cuda")
    for b, want in zip(g["batches"], g["outputs"]):
        ids = torch.as_tensor(b["ids"], dtype=torch.int64).cuda()
        losses = torch.as_tensor(b["losses"], dtype=torch.float32).cuda()        # float32, as the CE kernel writes them
        got = dev.get_priorities_device(ids, losses)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-12, atol=1e-14)
    dev.check()
    back = dev.to_host()
    tp = back.trend_p if name == "hybrid_priority" else back
    np.testing.assert_allclose(tp.values, g["final_values"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(tp.prev_loss, g["final_prev_loss"], rtol=0, atol=0)
    assert [bool(x) for x in tp.init] == g["final_init"] and tp.n_items == g["final_n_items"]
    assert abs(tp.avg - g["final_avg"]) <= 1e-12 * max(1.0, abs(g["final_avg"]))
    # a large batch (many strided trips per thread) against the host class
    rng = np.random.default_rng(2)
    n = 50000
    h2 = TrendPriority(n, 0.85) if name == "trend_priority" else HybridPriority(n, 0.85, 0.5)
    d2 = DeviceTrend(h2, "cuda")
    for _ in range(3):
        ids = rng.choice(n, 30000, replace=False)
        ls = rng.uniform(0, 5, len(ids)).astype(np.float32)
        want = h2.get_priorities(ids, ls)
        got = d2.get_priorities_device(torch.as_tensor(ids).cuda(), torch.as_tensor(ls).cuda())
        np.testing.assert_allclose(got.cpu().numpy(), np.asarray(want, dtype=np.float64), rtol=1e-11, atol=1e-13)


def test_pbr_strategy_keeps_trend_priorities_on_the_device():
    """A PBR strategy built with HybridPriority: the priority forward's losses become priorities and reach the HBM buffer without
    a host copy, and equal the host pipeline's (host class fed with the same losses)."""
    import ogl_amd  # noqa: F401
    import torch.nn.functional as F
    from ogl_amd import sampling, synthetic
    from ogl_amd.graph import TrainTestGraph
    from ogl_amd.graphsage import GraphSAGE
    from ogl_amd.graphsage.model import PrioritizedHipSupervisedGraphSage
    from ogl_amd.prioritized_replay import HybridPriority
    feat_size, labels, dyn, n_classes, _ = synthetic.load("toy", device="cuda")
    gu = TrainTestGraph(dyn, split=0.15, start_prior_alpha=4, end_prior_alpha=50, scale=1, max_priority=10)
    for _ in range(5):
        gu.evolve()
    torch.manual_seed(0)
    model = GraphSAGE(feat_size, 16, n_classes, 1, F.relu, 0, "pool").cuda()
    n_vertices = len(labels)
    strat = PrioritizedHipSupervisedGraphSage(model, 2, 8, labels, 5, HybridPriority(n_vertices, 0.85, 0.5), full_pass=1, cuda=True, batch_full=64)
    strat.build_optimizer()
    twin = HybridPriority(n_vertices, 0.85, 0.5)
    seen = []
    inner = gu.update_priorities_device
    gu.update_priorities_device = lambda ids, pr: (seen.append((np.asarray(ids).copy(), pr)), inner(ids, pr))
    losses_seen = []
    orig = strat._priorities_device
    strat._priorities_device = lambda ids, ls: (losses_seen.append(ls.detach().cpu().numpy().copy()), orig(ids, ls))[1]
    sampling.seed(3)
    for _ in range(2):
        strat.recompute_priorities(gu, gu.get_train_set())
    assert len(seen) == 2 and all(pr.is_cuda and pr.dtype == torch.float64 for _, pr in seen)
    for (ids, pr), ls in zip(seen, losses_seen):
        want = twin.get_priorities(ids, ls)
        np.testing.assert_allclose(pr.cpu().numpy(), np.asarray(want, dtype=np.float64), rtol=1e-11, atol=1e-13)
    # the host object the driver handed in is not left stale (round-3 advisor finding): read through the strategy it is brought
    # up to date from the device state — and equals the twin that lived on the host all along
    host = strat.priority_strategy
    assert host is strat._priority_strategy and not strat._device_trend.dirty
    np.testing.assert_allclose(host.trend_p.values, twin.trend_p.values, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(host.trend_p.prev_loss, twin.trend_p.prev_loss, rtol=0, atol=0)
    assert host.trend_p.n_items == twin.trend_p.n_items and np.array_equal(host.trend_p.init, twin.trend_p.init)
    # the kernel's preconditions are checked where the ids come from (the host): out of range / duplicates never reach the tree
    ls = torch.zeros(2, device="cuda")
    with pytest.raises(IndexError):
        strat._priorities_device(np.array([0, n_vertices]), ls)
    with pytest.raises(ValueError):
        strat._priorities_device(np.array([3, 3]), ls)
