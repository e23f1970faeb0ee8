"""HIP-vs-oracle parity through the C-ABI (run with -m gpu on the MI355X box).

Tolerances: integer work (picks, block indices, argmax) bit-exact; max reduce
bit-exact; mean reduce bit-exact (same slot-order summation); fp32-MFMA GEMM outputs
rtol 1e-4 / atol 1e-5 vs torch-CPU fp32; losses rtol 1e-4.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

GEMM_RTOL, GEMM_ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def ops():
    import ogl_amd  # noqa: F401
    from ogl_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


@pytest.fixture(params=["f32", "bf16x6", "auto"])
def gemm_mode(request, ops):
    """Both GEMM arithmetics must meet the SAME tolerances: exact fp32 MFMA and the split-bf16 (x6) MFMA path."""
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode("f32")


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rand_csr(rng, n, max_deg, hubs=0):
    deg = rng.integers(0, max_deg + 1, n)
    for h in range(hubs):
        deg[rng.integers(0, n)] = n // 2
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.integers(0, n, d)) for d in deg] + [np.zeros(0, np.int64)]).astype(np.int32)
    return indptr, indices


@pytest.mark.parametrize("n,max_deg,hubs,n_present,fanout,n_dst", [
    (50, 5, 0, 30, 3, 7), (2000, 12, 3, 1500, 25, 512), (2000, 12, 3, 2000, 10, 1), (300, 4, 0, 300, 64, 33),
    (300, 4, 0, 1, 25, 5), (20000, 30, 5, 17000, 25, 4096),
])
def test_sampler_bit_exact(ops, n, max_deg, hubs, n_present, fanout, n_dst):
    rng = np.random.default_rng(n + fanout)
    indptr, indices = rand_csr(rng, n, max_deg, hubs)
    g = ops.GraphHandle(dev(indptr), dev(indices))
    g.set_snapshot(n_present, n_present)
    deg_ref = O.snapshot_degrees_fast(indptr, indices, n_present, n_present)
    assert np.array_equal(g.degrees().cpu().numpy(), deg_ref)
    dst = rng.integers(0, n_present, n_dst).astype(np.int64)
    for layer, ctr, seed in [(0, 0, 1), (1, 5, 1), (1, 2 ** 33 + 7, 2 ** 40 + 3)]:
        got = ops.sample_layer(g, dev(dst), fanout, seed, ctr, layer).cpu().numpy()
        want = O.sample_layer(indptr, indices, deg_ref, dst, fanout, seed, ctr, layer)
        assert np.array_equal(got, want)


def test_sampler_edge_stream_keys(ops):
    """Edge streams: adjacency sorted by edge-table row, cut = t * edges_per_snapshot."""
    rng = np.random.default_rng(5)
    n, E = 400, 3000
    src = rng.integers(0, n, E); dst = rng.integers(0, n, E)
    rows = np.concatenate([np.arange(E), np.arange(E)])
    u = np.concatenate([src, dst]); v = np.concatenate([dst, src])     # in-neighbours of v: u
    order = np.lexsort((rows, v))
    indices = u[order].astype(np.int32); keys = rows[order].astype(np.int32)
    indptr = np.concatenate([[0], np.cumsum(np.bincount(v, minlength=n))]).astype(np.int64)
    g = ops.GraphHandle(dev(indptr), dev(indices), dev(keys))
    for cut in (0, 1, 777, E):
        g.set_snapshot(n, cut)
        want = O.snapshot_degrees_fast(indptr, keys, n, cut)
        assert np.array_equal(g.degrees().cpu().numpy(), want)
        d = rng.integers(0, n, 64).astype(np.int64)
        got = ops.sample_layer(g, dev(d), 25, 9, 3, 1).cpu().numpy()
        assert np.array_equal(got, O.sample_layer(indptr, indices, want, d, 25, 9, 3, 1))


@pytest.mark.parametrize("n_dst,fanout,universe,iso", [
    (1, 1, 5, 0.0), (7, 3, 20, 0.3), (512, 25, 5000, 0.1), (1000, 25, 300, 0.0), (4096, 25, 200000, 0.05),
    (33, 64, 1000, 0.5), (5, 4, 10, 1.0),
])
def test_block_bit_exact(ops, n_dst, fanout, universe, iso):
    rng = np.random.default_rng(n_dst * 31 + fanout)
    dst = rng.permutation(max(universe, n_dst))[:n_dst].astype(np.int64)
    picks = rng.integers(0, universe, size=(n_dst, fanout)).astype(np.int64)
    picks[rng.random(n_dst) < iso] = -1
    src_ref, li_ref = O.build_block(dst, picks)
    src, li, n = ops.build_block(dev(dst), dev(picks))
    assert n == len(src_ref)
    assert np.array_equal(src.cpu().numpy(), src_ref)
    assert np.array_equal(li.cpu().numpy(), li_ref)


def test_block_duplicate_dst(ops):
    dst = np.array([4, 9, 4, 2], dtype=np.int64)
    picks = np.array([[9, 7], [4, 4], [-1, -1], [7, 8]], dtype=np.int64)
    src_ref, li_ref = O.build_block(dst, picks)
    src, li, n = ops.build_block(dev(dst), dev(picks))
    assert np.array_equal(src.cpu().numpy(), src_ref) and np.array_equal(li.cpu().numpy(), li_ref)


@pytest.mark.parametrize("n_src,n_dst,fanout,d", [
    (9, 4, 3, 6), (700, 96, 25, 50), (5000, 512, 25, 602), (3000, 300, 25, 600), (2000, 100, 10, 128),
    (500, 64, 70, 33), (4000, 128, 25, 1000), (100, 10, 5, 1100),
])
@pytest.mark.parametrize("op", ["max", "mean", "sum"])
def test_reduce_fwd_bit_exact(ops, n_src, n_dst, fanout, d, op):
    rng = np.random.default_rng(n_src + d)
    src = rng.standard_normal((n_src, d)).astype(np.float32)
    li = rng.integers(0, n_src, size=(n_dst, fanout)).astype(np.int32)
    li[rng.random(n_dst) < 0.1] = -1
    want, arg = O.reduce_fwd(src, li, op)
    # 16-B-aligned padded rows (vector path) and tight rows (generic path when d % 4 != 0)
    padded = ops.empty_mat(n_src, d, "cuda").copy_(torch.as_tensor(src))
    for srct in (dev(src), padded):
        got, garg = ops.reduce_fwd(srct, dev(li), op, want_argmax=True)
        assert np.array_equal(got.cpu().numpy(), want), (op, srct.stride())
        if op == "max":
            assert np.array_equal(garg.cpu().numpy(), arg)
    # int64 global-row indexing gives the same values
    got64, _ = ops.reduce_fwd(dev(src), dev(li.astype(np.int64)), op)
    assert np.array_equal(got64.cpu().numpy(), want)


@pytest.mark.parametrize("n_dst,d", [(7054, 602), (5200, 600), (10300, 602), (512, 600), (2000, 260)])
def test_reduce_fwd_column_slices_bit_exact(ops, n_dst, d):
    """BASELINE-size destinations: the kernel cuts them into 1-4 column slices (one wave each) by batch size; values and
    argmax stay bit-exact whatever the cut."""
    rng = np.random.default_rng(n_dst + d)
    n_src, fanout = 20000, 25
    src = rng.standard_normal((n_src, d)).astype(np.float32)
    li = rng.integers(0, n_src, size=(n_dst, fanout)).astype(np.int32)
    li[rng.random(n_dst) < 0.05] = -1
    padded = ops.empty_mat(n_src, d, "cuda").copy_(torch.as_tensor(src))
    for op in ("max", "mean"):
        want, arg = O.reduce_fwd(src, li, op)
        got, garg = ops.reduce_fwd(padded, dev(li), op, want_argmax=(op == "max"))
        assert np.array_equal(got.cpu().numpy(), want), op
        if op == "max":
            assert np.array_equal(garg.cpu().numpy(), arg)


@pytest.mark.parametrize("op", ["max", "mean", "sum"])
def test_reduce_bwd(ops, op):
    rng = np.random.default_rng(8)
    n_src, n_dst, fanout, d = 900, 200, 25, 77
    src = torch.tensor(rng.standard_normal((n_src, d)).astype(np.float32), requires_grad=True)
    li = rng.integers(0, n_src, size=(n_dst, fanout)).astype(np.int32)
    li[::7] = -1
    gy = torch.tensor(rng.standard_normal((n_dst, d)).astype(np.float32))
    O._neigh_torch(src, li, op).backward(gy)
    s = src.detach().cuda().requires_grad_(True)
    ops.neighbor_reduce(s, dev(li), op).backward(gy.cuda())
    np.testing.assert_allclose(s.grad.cpu().numpy(), src.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_pool_max_fused_backward(ops, gemm_mode):
    """relu(fc_pool) -> max as one node: ReLU mask applied inside the scatter (no mask pass in the GEMMs)."""
    torch.manual_seed(1)
    rng = np.random.default_rng(1)
    n_src, n_dst, S, D = 700, 150, 25, 70
    x = torch.randn(n_src, D); w = torch.randn(D, D) / D ** 0.5; b = torch.randn(D) * 0.1
    li = rng.integers(0, n_src, size=(n_dst, S)).astype(np.int32); li[::9] = -1
    gy = torch.randn(n_dst, D)
    xc = x.cuda().requires_grad_(True); wc = w.cuda().requires_grad_(True); bc = b.cuda().requires_grad_(True)
    out = ops.pool_max(xc, wc, bc, dev(li))
    out.backward(gy.cuda())
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = O._neigh_torch(torch.relu(xr @ wr.T + br), li, "max")
    ref.backward(gy)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)
    np.testing.assert_allclose(wc.grad.cpu().numpy(), wr.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(bc.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(xc.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-4)


def test_gather_rows(ops):
    rng = np.random.default_rng(2)
    for d in (602, 128, 7):
        tab = rng.standard_normal((1000, d)).astype(np.float32)
        ids = rng.integers(0, 1000, 333).astype(np.int64)
        t = ops.empty_mat(1000, d, "cuda"); t.copy_(torch.as_tensor(tab))
        assert np.array_equal(ops.gather_rows(t, dev(ids)).cpu().numpy(), tab[ids])
        assert np.array_equal(ops.gather_rows(dev(tab), dev(ids)).cpu().numpy(), tab[ids])
    lab = rng.integers(0, 41, (1000, 1)).astype(np.int64)
    assert np.array_equal(ops.gather_i64(dev(lab), dev(ids)).cpu().numpy(), lab[ids, 0])


GEMM_SHAPES = [(1, 1, 1), (5, 7, 3), (130, 33, 17), (257, 602, 602), (1000, 600, 41), (300, 128, 32), (64, 500, 32),
               (513, 602, 600), (2048, 32, 3)]


@pytest.mark.parametrize("M,K,N", GEMM_SHAPES)
@pytest.mark.parametrize("relu", [False, True])
def test_linear_fwd(ops, gemm_mode, M, K, N, relu):
    torch.manual_seed(M + K + N)
    x = torch.randn(M, K); w = torch.randn(N, K) / K ** 0.5; b = torch.randn(N)
    want = torch.nn.functional.linear(x, w, b)
    if relu:
        want = want.relu()
    for xm in (x.cuda(), ops.empty_mat(M, K, "cuda").copy_(x)):
        got = ops.linear_fwd(xm, w.cuda(), b.cuda(), relu=relu)
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)


def test_linear_fwd_dual_and_rows(ops, gemm_mode):
    torch.manual_seed(3)
    T, M, K, K2, N = 900, 300, 602, 602, 600
    tab = torch.randn(T, K); rows = torch.randint(0, T, (M,))
    x2 = torch.randn(M, K2); w = torch.randn(N, K) / 25; w2 = torch.randn(N, K2) / 25; b = torch.randn(N)
    want = (tab[rows] @ w.T + x2 @ w2.T + b).relu()
    tabm = ops.empty_mat(T, K, "cuda").copy_(tab)
    got = ops.linear_fwd(tabm, w.cuda(), b.cuda(), x2=x2.cuda(), w2=w2.cuda(), relu=True, x_rows=rows.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=GEMM_RTOL, atol=2e-5)
    # concat -> Linear expressed as two column slices of one weight (in-repo layer, aggregator_dgl.py:206)
    W = torch.randn(N, K + K2) / 30
    Wc = W.cuda()
    got = ops.linear_fwd(tab[rows].cuda(), Wc[:, :K], b.cuda(), x2=x2.cuda(), w2=Wc[:, K:])
    want = torch.cat((tab[rows], x2), 1) @ W.T + b
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=GEMM_RTOL, atol=2e-5)


@pytest.mark.parametrize("M,K,K2,N,relu", [(512, 600, 600, 41, False), (33, 300, 7, 3, True), (4096, 258, 0, 64, True),
                                             (100, 1001, 0, 33, False), (512, 602, 13, 41, False), (64, 1500, 900, 41, True),
                                             (2000, 8, 600, 17, False)])
def test_linear_fwd_skinny(ops, M, K, K2, N, relu):
    """Few output tiles x long K: the in-block split-K kernel (final [B, 2H] -> C projection)."""
    torch.manual_seed(M + K)
    x = torch.randn(M, K); w = torch.randn(N, K) / K ** 0.5; b = torch.randn(N)
    want = x @ w.T + b
    kw = {}
    if K2:
        x2 = torch.randn(M, K2); w2 = torch.randn(N, K2) / K2 ** 0.5
        want = want + x2 @ w2.T
        kw = dict(x2=x2.cuda(), w2=w2.cuda())
    if relu:
        want = want.relu()
    got = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), relu=relu, **kw)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=GEMM_RTOL, atol=2e-5)


@pytest.mark.parametrize("M,K,N", GEMM_SHAPES + [(20000, 602, 602), (5000, 128, 128)])
@pytest.mark.parametrize("relu", [False, True])
def test_linear_bwd(ops, gemm_mode, M, K, N, relu):
    torch.manual_seed(M * 3 + K + N)
    x = torch.randn(M, K)
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N)
    gy = torch.randn(M, N)
    xc = x.cuda().requires_grad_(True); wc = w.cuda().requires_grad_(True); bc = b.cuda().requires_grad_(True)
    yc = ops.linear(xc, wc, bc, relu=relu)
    yc.backward(gy.cuda())
    # the ReLU mask is taken from the kernel's own forward output: pre-activations within rounding of 0
    # may legitimately fall on either side between two fp32 summation orders
    g = gy * (yc.detach().cpu() > 0) if relu else gy
    want_x, want_w, want_b = g @ w, g.T @ x, g.sum(0)
    scale = max(1.0, float(M) ** 0.5)
    np.testing.assert_allclose(xc.grad.cpu().numpy(), want_x.numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)
    np.testing.assert_allclose(wc.grad.cpu().numpy(), want_w.numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale)
    np.testing.assert_allclose(bc.grad.cpu().numpy(), want_b.numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale)


@pytest.mark.parametrize("T,M,K,N", [(5000, 3000, 602, 602), (900, 700, 300, 41), (600, 512, 601, 3), (100, 33, 64, 64)])
def test_linear_bwd_weight_rows(ops, gemm_mode, T, M, K, N):
    """(the last three: the one-launch few-column kernel of the output layer, gathered rows)"""
    torch.manual_seed(4)
    tab = torch.randn(T, K); rows = torch.randint(0, T, (M,)); dy = torch.randn(M, N)
    want_w = dy.T @ tab[rows]; want_b = dy.sum(0)
    tabm = ops.empty_mat(T, K, "cuda").copy_(tab)
    dw, db = ops.linear_bwd_weight(dy.cuda(), tabm, x_rows=rows.cuda())
    np.testing.assert_allclose(dw.cpu().numpy(), want_w.numpy(), rtol=GEMM_RTOL, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), want_b.numpy(), rtol=GEMM_RTOL, atol=1e-3)


@pytest.mark.parametrize("B,C", [(1, 3), (32, 3), (512, 41), (1024, 40), (7, 200)])
def test_cross_entropy(ops, B, C):
    torch.manual_seed(B + C)
    logits = (torch.randn(B, C) * 3).requires_grad_(True)
    labels = torch.randint(0, C, (B, 1))
    for red in ("none", "mean"):
        logits.grad = None
        want = torch.nn.functional.cross_entropy(logits, labels.flatten(), reduction=red)
        (want.sum() if red == "none" else want).backward()
        lc = logits.detach().cuda().requires_grad_(True)
        got = ops.cross_entropy(lc, labels.cuda(), red)
        (got.sum() if red == "none" else got).backward()
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(lc.grad.cpu().numpy(), logits.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_adam(ops):
    torch.manual_seed(0)
    n = 100003
    p = torch.randn(n); m = torch.zeros(n); v = torch.zeros(n)
    pc, mc, vc = p.cuda(), m.cuda(), v.cuda()
    for step in range(1, 5):
        g = torch.randn(n)
        O.adam_step(p, g, m, v, step)
        ops.adam_step(pc, g.cuda(), mc, vc, step)
    np.testing.assert_allclose(pc.cpu().numpy(), p.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(vc.cpu().numpy(), v.numpy(), rtol=1e-5, atol=1e-9)


def test_adam_multi(ops):
    torch.manual_seed(1)
    shapes = [(602, 602), (602,), (600, 602), (41, 600), (41,), (1,)]
    ps = [torch.randn(*s_) for s_ in shapes]; ms = [torch.zeros_like(p) for p in ps]; vs = [torch.zeros_like(p) for p in ps]
    pc = [p.cuda() for p in ps]; mc = [m.cuda() for m in ms]; vc = [v.cuda() for v in vs]
    for step in range(1, 4):
        gs = [torch.randn_like(p) for p in ps]
        for p, g, m, v in zip(ps, gs, ms, vs):
            O.adam_step(p, g, m, v, step)
        ops.adam_step_multi(pc, [g.cuda() for g in gs], mc, vc, step)
    for a, b in zip(pc, ps):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-5, atol=1e-7)


def test_x6_split_is_exact_and_accurate(ops):
    """x6 vs exact-fp32 MFMA on an ill-scaled problem (wide dynamic range, K = 4096): both within fp32-GEMM error of fp64."""
    torch.manual_seed(9)
    M, K, N = 257, 4096, 130
    x = torch.randn(M, K) * torch.logspace(-3, 3, K)[None, :]
    w = torch.randn(N, K) * torch.logspace(2, -2, K)[None, :]
    ref = (x.double() @ w.double().T)
    scale = (x.double().abs() @ w.double().abs().T)              # sum |a||b|: the natural error unit
    outs = {}
    for mode in ("f32", "bf16x6"):
        ops.set_gemm_mode(mode)
        outs[mode] = ops.linear_fwd(x.cuda(), w.cuda()).cpu().double()
    ops.set_gemm_mode("f32")
    e32 = ((outs["f32"] - ref).abs() / scale).max().item()
    e6 = ((outs["bf16x6"] - ref).abs() / scale).max().item()
    assert e32 < 1e-6 and e6 < 1e-6, (e32, e6)                     # fp32 chain: ~K * 2^-24 worst case, typically 1e-7
    assert e6 < 4 * max(e32, 6e-8), (e32, e6)


@pytest.mark.parametrize("B,C", [(1, 3), (513, 41), (1000, 200), (64, 64)])
def test_argmax_confusion(ops, B, C):
    rng = np.random.default_rng(B + C)
    logits = rng.standard_normal((B, C)).astype(np.float32)
    logits[::7, :] = np.round(logits[::7, :])                     # ties: the first maximum wins, like numpy
    labels = rng.integers(0, C, B).astype(np.int64)
    cm = torch.zeros(C * C, dtype=torch.int64).cuda()
    lt = ops.empty_mat(B, C, "cuda").copy_(torch.as_tensor(logits))
    pred = ops.argmax_confusion(lt, dev(labels), cm)
    want = logits.argmax(axis=1)
    assert np.array_equal(pred.cpu().numpy(), want)
    ref = np.zeros((C, C), dtype=np.int64); np.add.at(ref, (labels, want), 1)
    assert np.array_equal(cm.cpu().numpy().reshape(C, C), ref)
    ops.argmax_confusion(lt, dev(labels), cm, want_pred=False)       # accumulates
    assert np.array_equal(cm.cpu().numpy().reshape(C, C), 2 * ref)


@pytest.mark.parametrize("M,N", [(1, 1), (70, 33), (5000, 602), (64, 64), (129, 600)])
def test_transpose(ops, M, N):
    rng = np.random.default_rng(M + N)
    a = rng.standard_normal((M, N)).astype(np.float32)
    got = ops.transpose(ops.empty_mat(M, N, "cuda").copy_(torch.as_tensor(a)))
    assert got.shape == (N, M) and np.array_equal(got.cpu().numpy(), a.T)
    assert np.array_equal(ops.transpose(dev(a)).cpu().numpy(), a.T)
    rows = rng.integers(0, M, 77).astype(np.int64)
    assert np.array_equal(ops.transpose(dev(a), dev(rows)).cpu().numpy(), a[rows].T)


@pytest.mark.parametrize("M,K,N", [(5, 7, 3), (1500, 602, 600), (20000, 602, 602), (4096, 128, 32), (2000, 600, 41)])
def test_weight_grad_transposed_form(ops, gemm_mode, M, K, N):
    """dW, db from transposed operands (reduction-contiguous product + ones row) == dy.T @ x, dy.sum(0)."""
    torch.manual_seed(M + K)
    T = M + 50
    tab = torch.randn(T, K); rows = torch.randint(0, T, (M,)); dy = torch.randn(M, N)
    tabc = ops.empty_mat(T, K, "cuda").copy_(tab)
    dyT = ops.transpose(dy.cuda()); xT = ops.transpose(tabc, rows.cuda())
    dw, db = ops.linear_bwd_weight_t(dyT, xT)
    scale = max(1.0, float(M) ** 0.5)
    np.testing.assert_allclose(dw.cpu().numpy(), (dy.T @ tab[rows]).numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale)
    np.testing.assert_allclose(db.cpu().numpy(), dy.sum(0).numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale)
    dw2, db2 = ops.weight_grad(dy.cuda(), tabc, rows.cuda())          # whichever form the mode selects
    np.testing.assert_allclose(dw2.cpu().numpy(), (dy.T @ tab[rows]).numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale)


def test_batched_sampler_and_block_build_bit_exact(ops):
    """ogl_sample_layer_batched / ogl_build_block_batched == the per-batch entry points, batch by batch (ragged sizes,
    an empty batch, more batches than one descriptor chunk holds)."""
    rng = np.random.default_rng(5)
    n = 4000
    deg = rng.integers(0, 12, n); deg[rng.random(n) < 0.1] = 0
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.integers(0, n, d)) for d in deg] + [np.zeros(0, np.int64)]).astype(np.int32)
    g = ops.GraphHandle(dev(indptr), dev(indices)); g.set_snapshot(n, n)
    sizes = [37, 0, 512, 1, 100] + [int(x) for x in rng.integers(1, 60, 70)]            # 75 batches > 64
    base = dev(rng.integers(0, n, sum(sizes) + 50).astype(np.int64))
    starts, acc = [], 7
    for c in sizes:
        starts.append(acc); acc += c
    ctrs = list(range(100, 100 + len(sizes)))
    S = 6
    picks = ops.sample_layer_batched(g, base, starts, sizes, S, 9, ctrs, 1)
    src_all, n_src, lidx_all = ops.build_block_batched_async(base, starts, sizes, picks)
    n_src = n_src.cpu().tolist()
    r = 0
    for b, c in enumerate(sizes):
        dst = base[starts[b]:starts[b] + c]
        want_p = ops.sample_layer(g, dst, S, 9, ctrs[b], 1)
        assert torch.equal(picks[r:r + c], want_p)
        if c:
            s_ref, li_ref, n_ref = ops.build_block(dst, want_p)
            assert n_src[b] == n_ref
            assert torch.equal(src_all[r * (1 + S): r * (1 + S) + n_ref], s_ref)
            assert torch.equal(lidx_all[r:r + c], li_ref)
        else:
            assert n_src[b] == 0
        r += c
