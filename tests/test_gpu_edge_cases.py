"""Edge cases through the C-ABI: empty and single-element inputs, fanout 1, all-isolated batches, duplicate seeds,
ragged last batches, out-of-range ids (must not fault), workspace / argument errors.  Run with -m gpu."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import ogl_amd  # noqa: F401
    from ogl_amd import ops as _ops
    return _ops


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    return (t.to(dtype) if dtype is not None else t).cuda()


def test_empty_inputs(ops):
    g = ops.GraphHandle(dev(np.zeros(6, np.int64)), dev(np.zeros(0, np.int32)))     # 5 vertices, no edges
    g.set_snapshot(5, 5)
    assert g.degrees().cpu().tolist() == [0] * 5
    assert ops.sample_layer(g, torch.zeros(0, dtype=torch.int64).cuda(), 25, 1, 0, 0).shape == (0, 25)
    picks = ops.sample_layer(g, dev(np.arange(5)), 3, 1, 0, 0)
    assert (picks == -1).all()
    src, li, n = ops.build_block(dev(np.arange(5)), picks)
    assert n == 5 and (li == -1).all() and src.cpu().tolist() == [0, 1, 2, 3, 4]
    _, _, n0 = ops.build_block(torch.zeros(0, dtype=torch.int64).cuda(), torch.zeros((0, 4), dtype=torch.int64).cuda())
    assert n0 == 0
    x = ops.empty_mat(7, 6, "cuda").normal_()
    out, _ = ops.reduce_fwd(x, li, "max")                       # every dst isolated -> zeros
    assert out.shape == (5, 6) and (out == 0).all()
    assert ops.reduce_fwd(x, torch.zeros((0, 3), dtype=torch.int32).cuda(), "mean")[0].shape == (0, 6)
    assert ops.gather_rows(x, torch.zeros(0, dtype=torch.int64).cuda()).shape == (0, 6)
    w = torch.randn(4, 6).cuda()
    assert ops.linear_fwd(x[:0], w, None).shape == (0, 4)
    dw, db = ops.linear_bwd_weight(x[:0, :4], x[:0], None, None)   # no rows: gradients are exact zeros
    assert (dw == 0).all() and (db == 0).all() and dw.shape == (4, 6)
    loss, dl = ops.ce_fwd_bwd(x[:0, :3], torch.zeros(0, dtype=torch.int64).cuda())
    assert loss.numel() == 0


def test_single_seed_fanout_one_and_duplicates(ops):
    rng = np.random.default_rng(0)
    n = 40
    deg = rng.integers(1, 5, n)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.integers(0, n, d)) for d in deg]).astype(np.int32)
    g = ops.GraphHandle(dev(indptr), dev(indices)); g.set_snapshot(n, n)
    d_t = O.snapshot_degrees_fast(indptr, indices, n, n)
    for dst in (np.array([7]), np.array([3, 3, 9, 3])):                 # B = 1; duplicate seeds keep their own rows
        picks = ops.sample_layer(g, dev(dst), 1, 5, 2, 1)
        want = O.sample_layer(indptr, indices, d_t, dst, 1, 5, 2, 1)
        assert np.array_equal(picks.cpu().numpy(), want)
        src, li, nsrc = ops.build_block(dev(dst), picks)
        s_ref, li_ref = O.build_block(dst, want)
        assert np.array_equal(src.cpu().numpy(), s_ref) and np.array_equal(li.cpu().numpy(), li_ref)
    # duplicate seeds draw identical neighbours (the stream is keyed by vertex id, not by position)
    p = ops.sample_layer(g, dev(np.array([3, 3])), 8, 5, 2, 1).cpu().numpy()
    assert np.array_equal(p[0], p[1])


def test_out_of_range_ids_do_not_fault(ops):
    indptr = np.array([0, 2, 3, 3], np.int64); indices = np.array([1, 2, 0], np.int32)
    g = ops.GraphHandle(dev(indptr), dev(indices)); g.set_snapshot(3, 3)
    picks = ops.sample_layer(g, dev(np.array([0, 99, -5, 2], np.int64)), 4, 1, 0, 0).cpu().numpy()
    assert (picks[1] == -1).all() and (picks[2] == -1).all() and (picks[3] == -1).all() and (picks[0] >= 0).all()
    tab = ops.empty_mat(10, 8, "cuda").normal_()
    got = ops.gather_rows(tab, dev(np.array([0, 10, -1, 9], np.int64)))
    assert torch.equal(got[0], tab[0]) and torch.equal(got[3], tab[9]) and (got[1] == 0).all() and (got[2] == 0).all()
    li = dev(np.array([[1, 50, 2], [-1, -1, -1]], np.int32))
    out, arg = ops.reduce_fwd(tab, li, "max", want_argmax=True)
    want = torch.maximum(tab[1], tab[2])
    assert torch.equal(out[0], want) and (out[1] == 0).all() and int(arg.max()) < 10
    lab = ops.gather_i64(dev(np.arange(10).reshape(10, 1)), dev(np.array([3, 77], np.int64))).cpu().tolist()
    assert lab == [3, -1]
    # GEMM with gathered rows outside the table: treated as zero rows
    w = torch.randn(5, 8).cuda()
    y = ops.linear_fwd(tab, w, None, x_rows=dev(np.array([2, 10_000, 4], np.int64)))
    assert torch.allclose(y[0], tab[2] @ w.T, atol=1e-5) and (y[1] == 0).all()


def test_status_codes(ops):
    from ogl_amd import _lib
    h = _lib.lib()
    dst = dev(np.arange(4, dtype=np.int64)); picks = torch.zeros((4, 3), dtype=torch.int64).cuda()
    src = torch.empty(16, dtype=torch.int64).cuda(); n = torch.zeros(1, dtype=torch.int64).cuda()
    li = torch.empty((4, 3), dtype=torch.int32).cuda(); ws = torch.empty(64, dtype=torch.uint8).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    st = h.ogl_build_block(p(dst), 4, p(picks), 3, p(src), p(n), p(li), p(ws), 64, None)
    assert st == -4 and b"OGL_EWORKSPACE" in h.ogl_status_string(st)
    assert h.ogl_build_block(p(dst), 4, p(picks), 3, p(src), p(n), p(li), None, 0, None) == -4
    x = torch.zeros(4, 8).cuda()
    assert h.ogl_reduce_fwd(p(x), 8, 4, p(li), p(picks), 4, 3, 8, 1, p(x), 8, None, None) == -1       # both index kinds
    assert h.ogl_reduce_fwd(p(x), 8, 4, p(li), None, 4, 3, 8, 7, p(x), 8, None, None) == -1           # unknown op
    assert h.ogl_linear_fwd(p(x), 4, None, 0, 4, 8, p(x), 8, 4, None, None, 0, None, 0, 0, None, 0, 0, p(x), 8, None) == -1  # ldx < K
    with pytest.raises(_lib.OglError, match="OGL_EINVAL"):
        _lib.check(-1, "probe")
    assert h.ogl_set_gemm_mode(9) == -1 and h.ogl_set_gemm_mode(-1) in (0, 1, 2)        # (-1 = OGL_GEMM_QUERY)


def test_ragged_batches_and_tiny_training_steps(ops):
    """Loader keeps the ragged last batch; a 1-seed batch trains; an all-isolated batch gives finite loss and grads."""
    from ogl_amd import optim, sampling, synthetic
    from ogl_amd.graphsage import GatheredRows, GraphSAGE
    fs, _, dyn, nc, _ = synthetic.load("toy", device="cuda")
    g = dyn.get_graph()                                       # first snapshot: 30 vertices, many isolated
    model = GraphSAGE(fs, 8, nc, 1, F.relu, 0, "pool").cuda()
    opt = optim.Adam(model.parameters())
    deg = g.in_degrees().cpu().numpy()
    iso = np.nonzero(deg == 0)[0]
    seeds = np.concatenate([iso[:3], np.arange(g.n_present)])[:23].astype(np.int64)
    sizes = []
    for input_nodes, sd, blocks in sampling.NodeDataLoader(g, torch.as_tensor(seeds), sampling.MultiLayerNeighborSampler([4, 4]),
                                                           batch_size=11):
        sizes.append(sd.numel())
        opt.zero_grad()
        loss = ops.cross_entropy(model(blocks, GatheredRows(g.ndata["feat"], input_nodes)), ops.gather_i64(g.ndata["target"], sd))
        loss.backward(); opt.step()
        assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in model.parameters())
    assert sizes == [11, 11, 1]
    if len(iso) >= 2:                                         # a batch whose seeds have no neighbours at all
        (inp, sd, blocks), = list(sampling.NodeDataLoader(g, torch.as_tensor(iso[:2].astype(np.int64)),
                                                          sampling.MultiLayerNeighborSampler([4, 4]), batch_size=2))
        assert inp.numel() == 2 and blocks[0].number_of_edges() == 0 and blocks[1].number_of_edges() == 0
        out = model(blocks, GatheredRows(g.ndata["feat"], inp))
        # neighbour part is exactly the bias path: fc_self(h) + fc_neigh(0)
        assert torch.isfinite(out).all()
