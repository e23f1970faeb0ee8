"""Host-side logic that needs no GPU: replay buffer vs the reference-generated golden vectors
(tests/golden/replay.json), sum tree, time-ordered CSR construction, synthetic generators,
TrainTestGraph behaviour over a stub dynamic graph."""
import json
import os
import random

import numpy as np
import pytest

import ogl_amd  # noqa: F401
from ogl_amd.graph.snapshot_graph import build_time_ordered_csr
from ogl_amd.prioritized_replay import LossPriority, PrioritizedReplayBuffer, SumTree, TrendPriority
from oracle import oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "replay.json")))


def test_sum_tree_matches_reference_golden():
    g = GOLD["tree"]
    t = SumTree(g["capacity"])
    for i, v in enumerate(g["values"]):
        t[i] = v
    assert t.sum(0, 3) == g["sum_0_3"] and t.sum() == g["sum_all"] and t.sum(2, 7) == g["sum_2_7"]
    assert [t.find_prefixsum_idx(q) for q in g["prefix_queries"]] == g["prefix_idx"]


def test_sum_tree_growth_is_invisible():
    rng = np.random.default_rng(0)
    vals = rng.random(3000)
    small, big = SumTree(4), SumTree(1 << 14)
    small.set_many(np.arange(3000), vals)
    big.set_many(np.arange(3000), vals)
    assert small.sum(0, 2999) == big.sum(0, 2999)
    for q in rng.random(200) * vals.sum():
        assert small.find_prefixsum_idx(q) == big.find_prefixsum_idx(q)


def test_replay_buffer_matches_reference_golden():
    g = GOLD["buffer"]
    buf = PrioritizedReplayBuffer(g["size"], g["alpha"], max_priority=g["max_priority"], min_priority=g["min_priority"])
    first = {int(k): v for k, v in g["first"].items()}
    second = {int(k): v for k, v in g["second"].items()}
    buf.add_all(first)
    np.testing.assert_allclose(buf.dump_priorities(list(first)), g["dump_after_first"], rtol=1e-13, atol=0)
    buf.add_all(second)
    keys = list(first) + list(second)
    np.testing.assert_allclose(buf.dump_priorities(keys), g["dump_after_second"], rtol=1e-13, atol=0)
    buf.update_priorities({int(k): v for k, v in g["update"].items()})
    np.testing.assert_allclose(buf.dump_priorities(keys), g["dump_after_update"], rtol=1e-13, atol=0)
    assert buf.get_min_priority() == g["min_val"] and buf.get_max_priority() == g["max_val"]
    assert [int(x) for x in buf._storage] == g["storage"]
    random.seed(1)
    assert sorted(int(i) for i in buf._sample_proportional(8)) == g["sample8_seed1"]
    random.seed(2)
    assert sorted(int(i) for i in buf._sample_proportional(16)) == g["sample16_seed2"]
    random.seed(3)
    assert sorted(int(i) for i in buf._sample_proportional(100)) == g["sample100_seed3"]
    random.seed(4)
    got = buf.sample(8)
    assert len(got) == 8 and set(got) <= set(keys)


def test_priority_strategies():
    g = GOLD["loss_priority"]
    out = LossPriority().get_priorities(g["nodes"], np.array(g["losses"], dtype=np.float32))
    np.testing.assert_array_equal(np.asarray(out, dtype=np.float32), np.array(g["out"], dtype=np.float32))
    tp = TrendPriority(10)
    p1 = tp.get_priorities([1, 2], np.array([1.0, 2.0]))
    p2 = tp.get_priorities([1, 2], np.array([3.0, 1.0]))
    assert p1.shape == (2,) and p2[0] > p2[1] >= 0


@pytest.mark.parametrize("name", ["trend_priority", "hybrid_priority"])
def test_trend_and_hybrid_priorities_match_reference_golden(name):
    """The host classes against outputs of the reference's own TrendPriority / HybridPriority
    (R/train/prioritized_replay/generate_priority.py:11-58, run by tests/golden/make_golden.py): every batch's priorities and
    the final per-vertex state."""
    from ogl_amd.prioritized_replay import HybridPriority
    g = GOLD[name]
    obj = TrendPriority(g["n_vertices"], g["alpha"]) if name == "trend_priority" else HybridPriority(g["n_vertices"], g["alpha"], g["loss_contrib"])
    for b, want in zip(g["batches"], g["outputs"]):
        got = obj.get_priorities(np.asarray(b["ids"]), np.asarray(b["losses"], dtype=np.float32))
        np.testing.assert_allclose(np.asarray(got, dtype=np.float64), want, rtol=1e-13, atol=1e-15)
    tp = obj.trend_p if name == "hybrid_priority" else obj
    np.testing.assert_allclose(tp.values, g["final_values"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(tp.prev_loss, g["final_prev_loss"], rtol=0, atol=0)
    assert [bool(x) for x in tp.init] == g["final_init"] and tp.n_items == g["final_n_items"]
    assert abs(tp.avg - g["final_avg"]) <= 1e-13 * max(1.0, abs(g["final_avg"]))


def test_time_ordered_csr_vertex_and_edge():
    rng = np.random.default_rng(1)
    n, e = 50, 400
    s, d = rng.integers(0, n, e), rng.integers(0, n, e)
    ip, ix, ky = build_time_ordered_csr(n, s, d, s)
    assert ip[-1] == e and (ix == ky).all()
    for v in range(n):
        row = ix[ip[v]:ip[v + 1]]
        assert (np.diff(row) >= 0).all() and sorted(row.tolist()) == sorted(s[d == v].tolist())
    rows = np.arange(e)
    ip, ix, ky = build_time_ordered_csr(n, np.concatenate([s, d]), np.concatenate([d, s]), np.concatenate([rows, rows]))
    for v in range(n):
        assert (np.diff(ky[ip[v]:ip[v + 1]]) >= 0).all()
    # snapshot degrees at a cut = number of incident rows before the cut (self loops count twice)
    cut = 123
    deg = O.snapshot_degrees_fast(ip, ky, n, cut)
    want = np.bincount(np.concatenate([s[:cut], d[:cut]]), minlength=n)
    assert (deg == want).all()
    with pytest.raises(ValueError):
        build_time_ordered_csr(3, [0, 5], [1, 1], [0, 1])


def test_synthetic_generators_are_seeded_and_shaped():
    from ogl_amd import synthetic
    a = synthetic.make_arrays("toy")
    b = synthetic.make_arrays("toy")
    assert a["n"] == 600 and a["feat"].shape == (600, 20) and a["labels"].max() < 4
    assert (a["src"] == b["src"]).all() and (a["feat"] == b["feat"]).all() and (a["order"] == b["order"]).all()
    e = synthetic.make_arrays("toy_edge")
    flat = np.stack([e["src"], e["dst"]], 1).reshape(-1)
    seen = np.concatenate([[-1], np.maximum.accumulate(flat)[:-1]])
    assert (flat <= seen + 1).all(), "edge-stream ids must be relabelled by first appearance"
    assert e["n"] == flat.max() + 1
    spec = synthetic.SPECS["reddit"]
    assert (spec["n"], spec["f"], spec["c"]) == (232965, 602, 41)


class _StubDynamicGraph:
    """10 labelled vertices per snapshot; enough surface for TrainTestGraph."""

    def __init__(self, snapshots=6, per=10):
        self.snapshots, self.per, self.evolution_index = snapshots, per, 1

    def __len__(self):
        return self.snapshots

    def get_graph(self):
        return None

    def get_added_vertices(self, delta=None):
        lo = (self.evolution_index - 1) * self.per
        v = list(range(lo, lo + self.per))
        return v, [True] * len(v)

    def evolve(self):
        self.evolution_index += 1

    def get_original_to_subgraph_map(self):
        return np.arange(1000)

    def get_subgraph_to_original_map(self):
        return np.arange(1000)


def test_train_test_graph_behaviour():
    from ogl_amd.graph.train_test_graph import TrainTestGraph
    np.random.seed(1); random.seed(1)
    g = TrainTestGraph(_StubDynamicGraph(), split=0.15, start_prior_alpha=4, end_prior_alpha=50, scale=1, max_priority=10)
    assert len(g.get_train_set()) == 8 and len(g.get_test_set()) == 2            # 85/15 of 10
    assert set(g.get_train_set()) | set(g.get_test_set()) == set(range(10))
    assert g.priority_replay_buffer.get_max_priority() == 2                       # start_priority while empty
    g.evolve()
    assert len(g.get_train_set()) == 16 and g.prior_alpha == pytest.approx(4 + (46 / 6) * 1)
    assert len(g.get_new_train_nodes()) == 8 and len(g.get_new_train_nodes(3)) == 3
    # the train set as an incrementally kept array (arrival order): the same members as the reference's list(set), every one once;
    # the list itself is rebuilt on first use after an admission and is then ONE object until the next admission
    arr = g.get_train_array()
    assert arr.dtype == np.int64 and len(arr) == 16 and set(arr.tolist()) == set(g.get_train_set()) and not arr.flags.writeable
    assert set(arr[8:].tolist()) == set(g.get_new_train_nodes())                  # (the latest arrivals at its end)
    assert g.get_train_set() is g.get_train_set() and g.get_train_set() == list(g.train_set)
    # partial Fisher-Yates == full shuffle in distribution; exact_shuffle reproduces the reference's draws bit for bit
    g.exact_shuffle = True
    random.seed(5); ref_list = list(g.train_set_list); random.shuffle(ref_list)
    random.seed(5); assert g.draw_random_train_nodes(4) == ref_list[:4]
    g.exact_shuffle = False
    counts = {}
    random.seed(6)
    for _ in range(4000):
        for v in g.draw_random_train_nodes(3):
            counts[v] = counts.get(v, 0) + 1
    assert set(counts) == set(g.get_train_set()) and max(counts.values()) < 1.25 * min(counts.values())
    # uniform shuffle prefix whenever n <= |train| (reference quirk kept), buffer sampling otherwise
    draw = g.draw_priority_train_nodes(5)
    assert len(draw) == 5 and set(draw) <= set(g.get_train_set())
    assert set(g.draw_random_train_nodes(100)) == set(g.get_train_set())
    # partial update keeps the buffer, a full-size update re-creates it
    before = g.priority_replay_buffer
    g.update_priorities({g.get_train_set()[0]: 3.0})
    assert g.priority_replay_buffer is before
    g.update_priorities({v: 1.0 + i for i, v in enumerate(g.get_train_set())})
    assert g.priority_replay_buffer is not before and len(g.priority_replay_buffer) == 16
    with pytest.raises(AssertionError):
        g.update_priorities({v: 1.0 for v in range(100)})


def test_loader_batching_rules():
    import torch
    from ogl_amd import sampling
    ld = sampling.NodeDataLoader(None, torch.arange(103), None, batch_size=103 // 2)
    assert len(ld) == 3                                   # len//bpt batches + one remainder batch (drop_last=False)
    assert len(sampling.NodeDataLoader(None, torch.arange(103), None, batch_size=51, drop_last=True)) == 2
    with pytest.raises(ValueError):
        sampling.NodeDataLoader(None, torch.arange(3), None, batch_size=3 // 8)
    with pytest.raises(NotImplementedError):
        sampling.MultiLayerNeighborSampler([5, 5], replace=False)
    sampling.seed(7)
    assert sampling.get_state() == {"seed": 7, "ctr": 0}


def test_dataset_file_parsers(tmp_path):
    """File formats of R/train/dataset_utils/*: adjlist, timestamp json, edge csv (parsing needs no GPU)."""
    from ogl_amd.dataset_utils import common_utils as cu
    (tmp_path / "graph.adjlist").write_text("# comment\n0 1 2\n1 2\n2\n3 3\n4 0 # trailing\n")
    src, dst = cu.read_adjlist(str(tmp_path / "graph.adjlist"))
    pairs = sorted(zip(src.tolist(), dst.tolist()))
    assert pairs == sorted([(0, 1), (1, 0), (0, 2), (2, 0), (1, 2), (2, 1), (0, 4), (4, 0), (3, 3)])
    (tmp_path / "vertex_timestamp.json").write_text(json.dumps({"0": 5.0, "3": 1.5}))
    assert cu.read_timestamps(str(tmp_path / "vertex_timestamp.json")) == {0: 5.0, 3: 1.5}
    (tmp_path / "edges_dataframe.csv").write_text(",src,dst,time\n0,0,1,10\n1,1,2,11\n2,0,2,12\n")
    t = cu.read_edge_table(str(tmp_path / "edges_dataframe.csv"))
    assert t["src"].tolist() == [0, 1, 0] and t["dst"].tolist() == [1, 2, 2]
    with pytest.raises(FileNotFoundError, match="targets.npy"):
        cu._need(str(tmp_path), ["graph.adjlist", "targets.npy"])
    from ogl_amd.dataset_utils import pubmed, arxiv, reddit
    assert pubmed.FILES[3] == "postponed_timestamp.json" and arxiv.FILES[0] == "feats.npy" and "edges_dataframe.csv" in reddit.FILES


def test_macro_f1_from_confusion_matches_sklearn():
    """CSV row contents of R/train/graphsage/model.py:84-91 from a full C x C count matrix."""
    from sklearn.metrics import confusion_matrix, f1_score
    from ogl_amd.graphsage.model import macro_f1_from_confusion
    rng = np.random.default_rng(0)
    for C in (3, 7, 41):
        y = rng.integers(0, C, 500); p = rng.integers(0, max(1, C - 2), 500)
        if C > 3:
            y[y == 1] = 0; p[p == 1] = 0                       # a class that never occurs is dropped, as sklearn does
        cm = np.zeros((C, C), dtype=np.int64); np.add.at(cm, (y, p), 1)
        f1, flat = macro_f1_from_confusion(cm)
        assert abs(f1 - f1_score(y, p, average="macro")) < 1e-12
        assert flat == [int(v) for row in confusion_matrix(y, p) for v in row]
    assert macro_f1_from_confusion(np.zeros((4, 4)))[0] == 0.0


def _tree_state(buf):
    n = len(buf)
    return (buf._it_sum.node[buf._it_sum.capacity:buf._it_sum.capacity + n].copy(), buf._it_sum.node[1],
            buf._max_priority, buf._min_priority, buf.max_val, buf.min_val)


@pytest.mark.parametrize("helper", [True, False])
def test_replay_array_entry_points_equal_dict_api(helper, monkeypatch):
    """add_all_arrays / update_arrays (the C loop of libogl_host.so, or its Python fallback) leave EXACTLY the state the
    dict API leaves: leaves, root, running extrema — over growing trees, changing extrema, alpha up to 50."""
    from ogl_amd.prioritized_replay import replay_buffer as rb
    monkeypatch.setattr(rb, "_HOST", None if helper else False)
    if helper and rb.host_helper() is None:
        pytest.skip("libogl_host.so not built")
    rng = np.random.default_rng(7)
    for alpha in (0.6, 4.0, 50.0):
        a = PrioritizedReplayBuffer(10_000_000, alpha, 10, 1e-7)
        b = PrioritizedReplayBuffer(10_000_000, alpha, 10, 1e-7)
        keys = rng.permutation(40_000)[:5_000].astype(np.int64)
        pr = np.exp(rng.uniform(-20, 4, keys.size))                      # far outside the clip range on both sides
        a.add_all(dict(zip(keys.tolist(), pr.tolist())))
        b.add_all_arrays(keys, pr)
        for step in range(12):
            sel = rng.choice(keys, 300, replace=False)
            newp = np.exp(rng.uniform(-25 + step, 6 - step * 0.3, 300)).astype(np.float32)   # float32 losses, as from the GPU
            a.update_priorities(dict(zip(sel.tolist(), newp.tolist())))
            b.update_arrays(sel, newp)
            more = np.arange(50_000 + step * 900, 50_000 + (step + 1) * 900, dtype=np.int64)   # forces tree growth
            a.add_all(dict.fromkeys(more.tolist(), 3.5))
            b.add_all_arrays(more, np.full(more.size, 3.5))
        sa, sb = _tree_state(a), _tree_state(b)
        assert np.array_equal(sa[0], sb[0]) and sa[1:] == sb[1:]
        assert a.dump_priorities(keys[:100].tolist()) == b.dump_priorities(keys[:100].tolist())
        random.seed(5); ia = sorted(a._sample_proportional(200))
        random.seed(5); ib = sorted(b._sample_proportional(200))
        assert ia == ib
    with pytest.raises(KeyError):
        b.update_arrays(np.array([10 ** 7], dtype=np.int64), np.array([1.0]))


def test_whole_slab_dealing_is_a_bijection():
    """The tile walk of the k-major weight gradients (k_gemm_x3p `decode`, X3Args.xcd_slabs): XCD x owns the
    logical blocks [x T / 8 .. ) of T = tiles x nsplit, takes floor(nsplit / 8) WHOLE slabs and a share of the left-over slabs tile by
    tile.  Restated here in Python: every (slab, tile) pair is produced exactly once, and a whole slab's tiles stay on one XCD."""
    def decode(logical, tiles, nsplit):
        T = tiles * nsplit
        q = nsplit >> 3
        # the XCD whose run holds `logical`
        for xcd in range(8):
            begin = xcd * (T >> 3) + min(xcd, T & 7)
            length = (T >> 3) + (1 if xcd < (T & 7) else 0)
            if begin <= logical < begin + length:
                break
        local, whole = logical - begin, q * tiles
        if local < whole:
            return xcd, xcd * q + local // tiles, local % tiles
        e = (begin - xcd * whole) + (local - whole)
        return xcd, 8 * q + e // tiles, e % tiles

    for tiles in (1, 3, 15, 25, 40):
        for nsplit in (8, 9, 10, 16, 17, 23, 64):
            seen, home = set(), {}
            for l in range(tiles * nsplit):
                xcd, slab, tile = decode(l, tiles, nsplit)
                assert 0 <= slab < nsplit and 0 <= tile < tiles and (slab, tile) not in seen
                seen.add((slab, tile))
                if slab < 8 * (nsplit >> 3):
                    assert home.setdefault(slab, xcd) == xcd              # a whole slab never straddles two XCDs
            assert len(seen) == tiles * nsplit


def test_small_draws_from_a_long_train_set_do_not_rebuild_the_list():
    """A draw of n with 8 n < |train set| (32 of 1.4e5 at the arxiv-like rung, 512 of 2e5 at the Reddit one) indexes the arrival-order array
    with positions from a numpy Generator seeded once from Python's `random` stream: distinct members of the train set, uniform, the same
    batches under the same seed — and the list form (rebuilt after every admission: 0.7 ms at 136 k vertices, twice per PBR snapshot before
    round 6) is never materialised.  exact_shuffle keeps the reference's list and RNG consumption."""
    from ogl_amd.graph.train_test_graph import TrainTestGraph

    def make():
        np.random.seed(2); random.seed(2)
        return TrainTestGraph(_StubDynamicGraph(snapshots=6, per=400), split=0.15, start_prior_alpha=4, end_prior_alpha=50, scale=1, max_priority=10)
    g = make()
    g.evolve()
    assert len(g.train_set) == 680 and g._train_n == 680 and g._train_list is None
    random.seed(11)
    a = [g.draw_random_train_nodes(16) for _ in range(3)] + [g.draw_priority_train_nodes(16)]
    assert g._train_list is None                                       # (no list(train_set) behind any of the four draws)
    for d in a:
        assert len(d) == 16 and len(set(d)) == 16 and set(d) <= g.train_set and all(isinstance(v, int) for v in d)
    g2 = make()
    g2.evolve()
    random.seed(11)
    assert [g2.draw_random_train_nodes(16) for _ in range(3)] + [g2.draw_priority_train_nodes(16)] == a    # identically seeded replicas
    counts = np.zeros(800, np.int64)
    for _ in range(3000):
        counts[g.draw_random_train_nodes(16)] += 1
    seen = counts[sorted(g.train_set)]
    assert counts.sum() == seen.sum() and seen.min() > 0.5 * seen.mean() and seen.max() < 1.6 * seen.mean()      # 70.6 expected per vertex
    g.evolve()                                                          # an admission: the array grows, still no list
    assert g._train_n == len(g.train_set) == 1020 and g._train_list is None
    assert set(g.draw_random_train_nodes(16)) <= g.train_set and g._train_list is None
    g.exact_shuffle = True                                              # the reference's own draw: shuffles the list form
    random.seed(5); ref_list = list(g.train_set); random.shuffle(ref_list)
    random.seed(5); assert g.draw_random_train_nodes(16) == ref_list[:16]
