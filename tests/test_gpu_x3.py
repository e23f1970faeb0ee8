"""Pre-split ("bf16x3 image") projections through the C-ABI: image format bit-exact against a numpy model of the
split, forward GEMM bit-identical to the on-the-fly x6 kernel (same products, same order) and within the GEMM
tolerance of fp64, gather semantics, weight gradient + bias gradient.  Run with -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GEMM_RTOL, GEMM_ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def ops():
    import ogl_amd  # noqa: F401
    from ogl_amd import ops as _ops
    return _ops


def bf16_split3(x):
    """numpy/torch-CPU model of split3 (csrc/x6_arith.h): three round-to-nearest-even bf16 terms, residuals in fp32."""
    x = torch.as_tensor(x, dtype=torch.float32)
    h = x.to(torch.bfloat16)
    r = x - h.float()
    m = r.to(torch.bfloat16)
    s = r - m.float()
    l = s.to(torch.bfloat16)
    return h, m, l


def image_model(x):
    """[R, K] fp32 -> uint8 image [(R + 1), ceil(K/32), 3, 32] bf16 as raw bytes (include/ogl_hip.h)."""
    R, K = x.shape
    G = (K + 31) // 32
    pad = torch.zeros((R + 1, G * 32), dtype=torch.float32)
    pad[:R, :K] = x
    planes = torch.stack(bf16_split3(pad), 0)                     # [3, R+1, G*32] bf16
    img = planes.view(3, R + 1, G, 32).permute(1, 2, 0, 3).contiguous()
    return img.view(torch.int16).numpy().view(np.uint8).reshape(-1)


def image_model_t(x):
    """Transposed images are GROUP-MAJOR: [ceil(K/32)][R + 1][3][32] (include/ogl_hip.h)."""
    R, K = x.shape
    G = (K + 31) // 32
    return image_model(x).reshape(R + 1, G, 192).transpose(1, 0, 2).copy().reshape(-1)


@pytest.mark.parametrize("R,K", [(1, 1), (7, 31), (33, 32), (50, 33), (129, 602), (5, 1204)])
def test_image_bit_exact(ops, R, K):
    torch.manual_seed(R * 1000 + K)
    x = torch.randn(R, K) * torch.logspace(-3, 3, K)[None, :]
    xm = ops.empty_mat(R, K, "cuda"); xm.copy_(x)
    img = ops.x3_split(xm)
    assert img.rows == R and img.K == K and img.nbytes == (R + 1) * ((K + 31) // 32) * 192
    assert np.array_equal(img.buf.cpu().numpy(), image_model(x))
    # the split is exact: the three terms sum back to the fp32 value
    h, m, l = bf16_split3(x)
    assert torch.equal((h.float() + m.float()) + l.float(), x)
    # appended reduction element: ones on the activations side (zero row included), a vector on the weights side
    vec = torch.randn(R)
    ones = torch.cat([x, torch.ones(R, 1)], 1)
    want1 = image_model(ones).reshape(R + 1, -1).copy()
    want1[R] = image_model(torch.cat([torch.zeros(1, K), torch.ones(1, 1)], 1)).reshape(2, -1)[0]
    assert np.array_equal(ops.x3_split(xm, append_ones=True).buf.cpu().numpy(), want1.reshape(-1))
    img2 = ops.x3_split(xm, append_vec=vec.cuda())
    assert img2.K == K + 1
    assert np.array_equal(img2.buf.cpu().numpy(), image_model(torch.cat([x, vec[:, None]], 1)))


@pytest.mark.parametrize("M,N,ones", [(1, 1, False), (40, 7, True), (64, 64, True), (65, 130, False), (1000, 41, True), (2500, 602, True)])
def test_transposed_image_bit_exact(ops, M, N, ones):
    torch.manual_seed(M + N)
    T = M + 13
    tab = torch.randn(T, N)
    rows = torch.randint(0, T, (M,))
    tm = ops.empty_mat(T, N, "cuda"); tm.copy_(tab)
    img = ops.x3_split_t(tm, rows.cuda(), ones_row=ones)
    want = tab[rows].T.contiguous()
    if ones:
        want = torch.cat([want, torch.ones(1, M)], 0)
    assert img.rows == N + int(ones) and img.K == M
    assert np.array_equal(img.buf.cpu().numpy(), image_model_t(want))
    img2 = ops.x3_split_t(tm[:M], None, ones_row=ones)           # ungathered
    want2 = tab[:M].T.contiguous()
    if ones:
        want2 = torch.cat([want2, torch.ones(1, M)], 0)
    assert np.array_equal(img2.buf.cpu().numpy(), image_model_t(want2))


@pytest.mark.parametrize("M,K,N,relu", [(1, 1, 1, False), (37, 33, 5, True), (255, 602, 41, False), (257, 64, 129, True),
                                         (1000, 602, 602, True), (3000, 1204, 256, False), (513, 31, 600, True),
                                         (700, 32, 130, False), (2600, 608, 602, True),
                                         (7199, 602, 600, True), (6700, 96, 640, False),    # 192 x 128 tiles
                                         (9000, 602, 600, True),                             # 192 x 128 tiles
                                         (53001, 70, 600, True), (140001, 40, 160, False)])  # tall, ragged columns
def test_forward_matches_on_the_fly_x6_and_fp64(ops, M, K, N, relu):
    torch.manual_seed(M * 7 + K + N)
    T = M + 50
    tab = torch.randn(T, K); w = torch.randn(N, K) / np.sqrt(K); b = torch.randn(N)
    rows = torch.randint(0, T, (M,))
    tm = ops.empty_mat(T, K, "cuda"); tm.copy_(tab)
    wc, bc, rc = w.cuda(), b.cuda(), rows.cuda()
    # bias folded into the images (one extra reduction element: ones on the x side, the bias on the w side)
    got = ops.linear_fwd_x3(ops.x3_split(tm, append_ones=True), rc, ops.x3_split(wc, append_vec=bc), relu=relu)
    want = tab[rows].double() @ w.double().T + b.double()
    if relu:
        want = want.clamp_min(0)
    np.testing.assert_allclose(got.cpu().numpy(), want.float().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)
    # without a bias the six product terms are those of the on-the-fly x6 kernel; the image kernel sums each term over
    # 32 reduction elements per MFMA (16x16x32) where that one sums over 16 (32x32x16): fp32 summation-order noise only
    got0 = ops.linear_fwd_x3(ops.x3_split(tm), rc, ops.x3_split(wc), relu=relu)
    old = ops.get_gemm_mode()
    try:
        ops.set_gemm_mode("bf16x6")
        ref = ops.linear_fwd(tm, wc, None, relu=relu, x_rows=rc)
    finally:
        ops.set_gemm_mode(old)
    np.testing.assert_allclose(got0.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    # ungathered operand
    got2 = ops.linear_fwd_x3(ops.x3_split(tm[:M]), None, ops.x3_split(wc), relu=False)
    np.testing.assert_allclose(got2.cpu().numpy(), (tab[:M].double() @ w.double().T).float().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)


def test_forward_many_tiles_per_block(ops):
    """More row tiles than CUs: every persistent block walks several tiles (cross-tile prefetch, stores left in flight)."""
    torch.manual_seed(11)
    M, K, N = 256 * 300 + 17, 70, 200
    x = torch.randn(M, K); w = torch.randn(N, K); b = torch.randn(N)
    xm = ops.empty_mat(M, K, "cuda"); xm.copy_(x)
    got = ops.linear_fwd_x3(ops.x3_split(xm, append_ones=True), None, ops.x3_split(w.cuda(), append_vec=b.cuda()), relu=True)
    want = (x.double() @ w.double().T + b.double()).clamp_min(0).float()
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)
    y2 = ops.linear_fwd_x3(ops.x3_split(xm, append_ones=True), None, ops.x3_split(w.cuda(), append_vec=b.cuda()), relu=True)
    assert torch.equal(got, y2)


def test_forward_out_of_range_rows_are_zero_rows(ops):
    torch.manual_seed(3)
    tab = torch.randn(20, 40); w = torch.randn(9, 40); b = torch.randn(9)
    tm = ops.empty_mat(20, 40, "cuda"); tm.copy_(tab)
    rows = torch.tensor([3, 20, -1, 19, 10**9, 2**32 + 3], dtype=torch.int64).cuda()
    y = ops.linear_fwd_x3(ops.x3_split(tm, append_ones=True), rows, ops.x3_split(w.cuda(), append_vec=b.cuda())).cpu()
    np.testing.assert_allclose(y[0].numpy(), (tab[3] @ w.T + b).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y[3].numpy(), (tab[19] @ w.T + b).numpy(), rtol=1e-4, atol=1e-5)
    for i in (1, 2, 4, 5):
        assert torch.equal(y[i], b)                               # zero row . w + bias: the bias exactly
    y0 = ops.linear_fwd_x3(ops.x3_split(tm), rows, ops.x3_split(w.cuda())).cpu()
    assert (y0[[1, 2, 4, 5]] == 0).all()
    # a row-prefix bound below the image's row count: ids in [x_nrows, image rows) read the zero row, not table rows
    y1 = ops.linear_fwd_x3(ops.x3_split(tm), rows, ops.x3_split(w.cuda()), x_nrows=19).cpu()
    assert (y1[3] == 0).all() and torch.equal(y1[0], y0[0])
    # ungathered row prefix
    y2 = ops.linear_fwd_x3(ops.x3_split(tm), None, ops.x3_split(w.cuda()), M=7).cpu()
    np.testing.assert_allclose(y2.numpy(), (tab[:7] @ w.T).numpy(), rtol=1e-4, atol=1e-5)


def test_registered_table_dispatch(ops):
    """ops.linear_fwd / weight_grad route large products over a registered static table to the image kernels."""
    torch.manual_seed(4)
    T, K, N, M = 30000, 70, 50, 20000
    tab = ops.empty_mat(T, K, "cuda"); tab.normal_()
    w = torch.randn(N, K).cuda(); b = torch.randn(N).cuda()
    rows = torch.randint(0, T - 100, (M,)).cuda()
    old = ops.get_gemm_mode()
    try:
        ops.set_gemm_mode("auto")
        ref = ops.linear_fwd(tab, w, b, relu=True, x_rows=rows)          # not registered: on-the-fly kernel
        ops.register_static_table(tab)
        ops.profile_start()
        got = ops.linear_fwd(tab[:T - 100], w, b, relu=True, x_rows=rows)   # a row-prefix view resolves to the same image
        full = ops.linear_fwd(tab[:T - 100], w, b, relu=True)
        dy = ops.empty_mat(M, N, "cuda"); dy.normal_()
        dw, db = ops.weight_grad(dy, tab, rows)
        names = [r[0] for r in ops.profile_stop()]
    finally:
        ops.set_gemm_mode(old)
        ops._X3_TABLES.clear()
    # (the weight gradient reads the table's own row-major image: the k-major product, no transposed image of the rows)
    assert names.count("ogl_linear_fwd_x3") == 2 and "ogl_linear_bwd_weight_x3k" in names and "ogl_linear_fwd" not in names
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    want_full = (tab[:T - 100].double() @ w.double().T + b.double()).clamp_min(0).float()
    np.testing.assert_allclose(full.cpu().numpy(), want_full.cpu().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)
    want_w = (dy.double().T @ tab[rows].double()).float()
    np.testing.assert_allclose(dw.cpu().numpy(), want_w.cpu().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * 150)
    np.testing.assert_allclose(db.cpu().numpy(), dy.double().sum(0).float().cpu().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * 150)


@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (100, 33, 7), (1024, 602, 602), (7000, 602, 600), (20000, 256, 41), (333, 1204, 130)])
def test_weight_gradient(ops, M, K, N):
    torch.manual_seed(M + K * 3 + N)
    T = M + 9
    tab = torch.randn(T, K); dy = torch.randn(M, N)
    rows = torch.randint(0, T, (M,))
    tm = ops.empty_mat(T, K, "cuda"); tm.copy_(tab)
    dym = ops.empty_mat(M, N, "cuda"); dym.copy_(dy)
    dw, db = ops.linear_bwd_weight_x3(ops.x3_split_t(dym), ops.x3_split_t(tm, rows.cuda(), ones_row=True))
    scale = max(1.0, np.sqrt(M))
    want_w = (dy.double().T @ tab[rows].double()).float()
    np.testing.assert_allclose(dw.cpu().numpy(), want_w.numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale)
    np.testing.assert_allclose(db.cpu().numpy(), dy.double().sum(0).float().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale)
    dw2, db2 = ops.linear_bwd_weight_x3(ops.x3_split_t(dym), ops.x3_split_t(tm, rows.cuda(), ones_row=True), want_bias=False)
    assert db2 is None and torch.equal(dw, dw2)                   # deterministic (fixed split-K order)


def test_x3_accuracy_ill_scaled(ops):
    """< 1e-6 * sum|a||b| of fp64 on an ill-scaled K = 4096 product (the bound the on-the-fly x6 path is held to)."""
    torch.manual_seed(5)
    M, K, N = 300, 4096, 200
    a = torch.randn(M, K) * torch.logspace(-4, 4, K)[None, :]
    b = torch.randn(N, K) * torch.logspace(4, -4, K)[None, :]
    am = ops.empty_mat(M, K, "cuda"); am.copy_(a)
    bm = ops.empty_mat(N, K, "cuda"); bm.copy_(b)
    got = ops.linear_fwd_x3(ops.x3_split(am), None, ops.x3_split(bm)).cpu().double()
    want = a.double() @ b.double().T
    bound = a.double().abs() @ b.double().abs().T
    assert float(((got - want).abs() / bound).max()) < 1e-6


def test_status_codes(ops):
    import ctypes as C
    from ogl_amd import _lib
    h = _lib.lib()
    assert h.ogl_x3_image_bytes(1, 602) == 2 * 19 * 192 and h.ogl_x3_image_bytes(10, 33) == 11 * 2 * 192
    assert h.ogl_x3_image_bytes(-1, 8) == -1
    x = torch.zeros(4, 8).cuda(); img = torch.zeros(4096, dtype=torch.uint8).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    assert h.ogl_x3_split(p(x), 4, None, 0, 4, 8, 0, None, p(img), None) == -1   # ld < K
    assert h.ogl_x3_split(p(x), 8, None, 0, 4, 8, 0, None, None, None) == -1     # no image
    assert h.ogl_x3_split(p(x), 8, None, 0, 4, 8, 2, None, p(img), None) == -1   # append = vector without a vector
    assert h.ogl_linear_fwd_x3(p(img), 4, None, 4, 9, 8, p(img), 4, 0, p(x), 8, None) == -1   # M > image rows, no gather
    assert h.ogl_linear_fwd_x3(p(img), 4, p(img), 5, 4, 8, p(img), 4, 0, p(x), 8, None) == -1   # id bound beyond the image
    assert h.ogl_linear_bwd_weight_x3(p(img), p(img), 64, 4, 4, p(x), 8, None, None, 0, None) in (0, -4)


def image_decode_t(buf, rows, K):
    """uint8 group-major (transposed) image -> [rows, ceil(K/32)*32] fp32 (the three planes summed; exact)."""
    G = (K + 31) // 32
    raw = torch.as_tensor(buf.cpu().numpy().view(np.int16).reshape(G, rows + 1, 3, 32).copy()).view(torch.bfloat16).float()
    val = (raw[:, :, 0] + raw[:, :, 1]) + raw[:, :, 2]                 # [G, rows + 1, 32]
    assert (val[:, rows] == 0).all()                                   # the zero row of every group
    return val[:, :rows].permute(1, 0, 2).reshape(rows, G * 32)


@pytest.mark.parametrize("M,N,G", [(40, 7, 2), (64, 64, 2), (1000, 41, 32), (2500, 130, 79), (33, 5, 4)])
def test_transposed_image_interleaved(ops, M, N, G):
    """interleave = G: image reduction index m holds source row (m % 32) * G + m // 32 (zeros past M)."""
    torch.manual_seed(M + N + G)
    tab = torch.randn(M + 5, N)
    rows = torch.randint(0, M + 5, (M,))
    tm = ops.empty_mat(M + 5, N, "cuda"); tm.copy_(tab)
    img = ops.x3_split_t(tm, rows.cuda(), ones_row=True, interleave=G)
    assert img.rows == N + 1 and img.K == 32 * G
    m = np.arange(32 * G)
    s_of_m = (m % 32) * G + m // 32
    want = torch.zeros(N + 1, 32 * G)
    ok = s_of_m < M
    want[:N, ok] = tab[rows[s_of_m[ok]]].T
    want[N, :] = 1.0
    assert np.array_equal(img.buf.cpu().numpy(), image_model_t(want))


@pytest.mark.parametrize("n_dst,S,D,n_src,relu", [(1, 1, 1, 1, True), (50, 4, 33, 70, True), (300, 25, 602, 2000, True),
                                                   (2500, 10, 130, 900, False), (700, 25, 64, 40, True), (4100, 25, 40, 5000, True),
                                                   (7060, 25, 602, 62495, True), (3000, 63, 640, 40000, True)])   # the Reddit step's shape (1 953 groups: the scan's carry); the limits
def test_pool_backward_image(ops, n_dst, S, D, n_src, relu):
    """Fused relu->max backward == fp32 scatter of the winners' gradients, as the dealt transposed image."""
    torch.manual_seed(n_dst + D)
    rng = np.random.default_rng(n_dst * 3 + D)
    idx = rng.integers(0, n_src, (n_dst, S)).astype(np.int32)
    idx[rng.random((n_dst, S)) < 0.05] = -1                           # missing neighbours
    p = torch.randn(n_src, D).clamp_min(0)                            # ReLU'd projections: many exact zeros
    pm = ops.empty_mat(n_src, D, "cuda"); pm.copy_(p)
    out, argmax = ops.reduce_fwd(pm, torch.as_tensor(idx).cuda(), "max", want_argmax=True)
    dout = torch.randn(n_dst, D)
    dm = ops.empty_mat(n_dst, D, "cuda"); dm.copy_(dout)
    img = ops.pool_bwd_x3(dm, argmax, out if relu else None, torch.as_tensor(idx).cuda(), n_src)
    G = (n_src + 31) // 32
    assert img.rows == D and img.K == 32 * G
    a = argmax.cpu().numpy(); o = out.cpu().numpy(); g = dout.numpy()
    dP = np.zeros((n_src, D), np.float64)
    cnt = np.zeros((n_src, D), np.int64)
    cols = np.arange(D)
    for d in range(n_dst):
        m = a[d] >= 0
        if relu:
            m &= o[d] > 0
        dP[a[d, m], cols[m]] += g[d, m]
        cnt[a[d, m], cols[m]] += 1
    got = image_decode_t(img.buf, D, 32 * G).numpy()                     # [D, 32 G], dealt
    mm = np.arange(32 * G)
    s_of_m = (mm % 32) * G + mm // 32
    ok = s_of_m < n_src
    assert (got[:, ~ok] == 0).all()
    got_s = np.zeros((n_src, D), np.float32)
    got_s[s_of_m[ok]] = got[:, ok].T
    # cells with at most two contributions are order-independent: exactly the fp32 sum
    two = cnt <= 2
    assert np.array_equal(got_s[two], dP.astype(np.float32)[two]) or np.allclose(got_s[two], dP[two], rtol=0, atol=0)
    np.testing.assert_allclose(got_s, dP, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(dP).max()))
    # the unfused path agrees
    dP2 = ops.reduce_bwd(dm, None, argmax, "max", n_src, fanout=S, relu_out=out if relu else None)
    np.testing.assert_allclose(dP2.cpu().numpy(), dP, rtol=1e-4, atol=1e-4)
    # plan (no gradient: could have run in the forward pass: column order, group totals -> scan -> every segment's place) + apply (the
    # values pass + the streamed group pass) build the same image: identical where a cell has at most two contributions, fp32-order
    # noise elsewhere; on the side stream or on the caller's
    for side in (False, True):
        plan = ops.pool_bwd_x3_plan(argmax, out if relu else None, torch.as_tensor(idx).cuda(), n_src, side=side)
        assert plan.pending == (side and ops.FORK_BACKWARD)              # a side plan is enqueued behind the caller's next launch
        dm2 = dm.clone()                                                  # (the plan never saw this tensor)
        img2 = ops.pool_bwd_x3_apply(dm2, torch.as_tensor(idx).cuda(), plan, n_src)
        got2 = image_decode_t(img2.buf, D, 32 * G).numpy()
        got2_s = np.zeros((n_src, D), np.float32)
        got2_s[s_of_m[ok]] = got2[:, ok].T
        assert (got2[:, ~ok] == 0).all()
        assert np.array_equal(got2_s[two], got_s[two])
        np.testing.assert_allclose(got2_s, dP, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(dP).max()))


def test_pool_max_autograd_uses_fused_backward(ops):
    """pool_max over a registered table: backward runs pool_bwd_x3 + the image weight gradient and matches the unfused path."""
    torch.manual_seed(8)
    T, K, H, n_src, n_dst, S = 40000, 50, 48, 20000, 1500, 25
    tab = ops.empty_mat(T, K, "cuda"); tab.normal_()
    rows = torch.randperm(T)[:n_src].cuda()
    idx = torch.randint(0, n_src, (n_dst, S), dtype=torch.int32).cuda()
    w = (torch.randn(H, K) / 7).cuda(); b = torch.randn(H).cuda()
    gout = torch.randn(n_dst, H).cuda()
    res = {}
    old = ops.get_gemm_mode()
    try:
        ops.set_gemm_mode("auto")
        for tag in ("unfused", "fused"):
            if tag == "fused":
                ops.register_static_table(tab)
            wv, bv = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            ops.profile_start()
            out = ops.pool_max(tab, wv, bv, idx, rows)
            out.backward(gout)
            names = [r[0] for r in ops.profile_stop()]
            res[tag] = (out.detach(), wv.grad, bv.grad, names)
    finally:
        ops.set_gemm_mode(old)
        ops._X3_TABLES.clear()
    names = res["fused"][3]
    assert "ogl_reduce_bwd" not in names
    if ops.POOL_PLAN:                      # the gradient-free half planned by the forward pass; the backward consumes the plan
        consumer = "ogl_pool_bwd_x3_apply"
        assert names.index("ogl_pool_bwd_x3_plan") < names.index(consumer) and "ogl_pool_bwd_x3" not in names
    else:
        assert "ogl_pool_bwd_x3" in names
    np.testing.assert_allclose(res["fused"][0].cpu().numpy(), res["unfused"][0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    scale = float(res["unfused"][1].abs().max())
    np.testing.assert_allclose(res["fused"][1].cpu().numpy(), res["unfused"][1].cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(res["fused"][2].cpu().numpy(), res["unfused"][2].cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)


def test_pool_backward_limits(ops):
    """Largest shapes the fused pool backward accepts (fanout 63, 640 columns) and its argument errors."""
    import ctypes as C
    from ogl_amd import _lib
    torch.manual_seed(2)
    n_dst, S, D, n_src = 90, 63, 640, 3000
    idx = torch.randint(0, n_src, (n_dst, S), dtype=torch.int32).cuda()
    pm = ops.empty_mat(n_src, D, "cuda"); pm.normal_().clamp_min_(0)
    out, argmax = ops.reduce_fwd(pm, idx, "max", want_argmax=True)
    dm = ops.empty_mat(n_dst, D, "cuda"); dm.normal_()
    img = ops.pool_bwd_x3(dm, argmax, out, idx, n_src)
    G = (n_src + 31) // 32
    got = image_decode_t(img.buf, D, 32 * G)
    dP = ops.reduce_bwd(dm, None, argmax, "max", n_src, fanout=S, relu_out=out).cpu()
    mm = np.arange(32 * G); s_of_m = (mm % 32) * G + mm // 32; ok = s_of_m < n_src
    np.testing.assert_allclose(got[:, ok].T.numpy(), dP[s_of_m[ok]].numpy(), rtol=1e-5, atol=1e-5)
    h = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    ws = torch.empty(1 << 20, dtype=torch.uint8).cuda()
    args = lambda fan, d: (p(dm), 640, p(argmax), p(out), 640, p(idx), n_dst, fan, d, n_src, p(img.buf), p(ws), ws.numel(), None)
    assert h.ogl_pool_bwd_x3(*args(64, 640)) == -1                 # fanout > 63
    assert h.ogl_pool_bwd_x3(*args(63, 641)) == -1                 # more columns than a bucket wave holds
    assert h.ogl_pool_bwd_x3(p(dm), 640, p(argmax), p(out), 640, p(idx), n_dst, S, D, n_src, p(img.buf), p(ws), 64, None) == -4
    nb = int(h.ogl_pool_bwd_x3_workspace_bytes(n_dst, S, D, n_src))
    assert h.ogl_pool_bwd_x3_plan(p(argmax), p(out), 640, p(idx), n_dst, 64, D, n_src, p(ws), nb, None) == -1
    assert h.ogl_pool_bwd_x3_plan(p(argmax), p(out), D - 1, p(idx), n_dst, S, D, n_src, p(ws), nb, None) == -1
    assert h.ogl_pool_bwd_x3_plan(p(argmax), p(out), 640, p(idx), n_dst, S, D, n_src, p(ws), 64, None) == -4
    assert h.ogl_pool_bwd_x3_plan(None, p(out), 640, p(idx), n_dst, S, D, n_src, p(ws), nb, None) == -1
    assert h.ogl_pool_bwd_x3_apply(p(dm), 640, p(idx), n_dst, S, 641, n_src, p(img.buf), p(ws), nb, None) == -1
    assert h.ogl_pool_bwd_x3_apply(p(dm), 640, p(idx), n_dst, S, D, n_src, None, p(ws), nb, None) == -1
    assert h.ogl_pool_bwd_x3_apply(p(dm), 640, p(idx), n_dst, S, D, n_src, p(img.buf), p(ws), 64, None) == -4
    assert h.ogl_pool_bwd_x3_apply(None, 640, p(idx), n_dst, S, D, n_src, p(img.buf), p(ws), nb, None) == -1


@pytest.mark.parametrize("M,K,N", [(7060, 602, 600), (70000, 100, 602), (300, 33, 161), (1, 1, 1), (5000, 64, 321), (62495, 602, 602)])
def test_every_tile_shape_computes_the_same_bits(ops, M, K, N):
    """The three tiles of k_gemm_x3p (256 / 128 / 192 rows x 128 columns) differ in what a block fetches per step, not in the MFMA
    sequence behind an output element: pinned one after the other (ogl_debug_set: OGL_KNOB_X3_TILE) they return the same bits — plain products and
    EXT products (addend, second A part, output image)."""
    from ogl_amd import _lib
    torch.manual_seed(M + N)
    T = M + 10
    tm = ops.empty_mat(T, K, "cuda").copy_(torch.randn(T, K, device="cuda"))
    rows = torch.randint(0, T, (M,), device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    xi, wi = ops.x3_split(tm, append_ones=True), ops.x3_split(w, append_vec=b)
    x2 = ops.empty_mat(M, 40, "cuda").copy_(torch.randn(M, 40, device="cuda"))
    w2 = torch.randn(N, 40, device="cuda")
    wcat = ops.x3_split_cat([(w, b), (w2, None)])
    S0 = ops.empty_mat(T, N, "cuda").copy_(torch.randn(T, N, device="cuda"))
    outs = []
    try:
        for cfg in (0, 1, 2):
            ops.debug_set("x3_tile", cfg)
            y = ops.linear_fwd_x3(xi, rows, wi, relu=True, x_nrows=T)
            y2, img = ops.linear_fwd_x3_ext(xi, rows, wcat, x2_img=ops.x3_split(x2), add=S0, add_rows=rows, relu=True, x_nrows=T,
                                            want_image=True, image_append_ones=True)
            outs.append((y.clone(), y2.clone(), img.buf.clone()))
    finally:
        ops.debug_set("x3_tile", -1)
    assert _lib.lib().ogl_debug_set(0, 3, None) != 0 and _lib.lib().ogl_debug_set(0, -2, None) != 0 and _lib.lib().ogl_debug_set(9, 0, None) != 0
    want = (tm[rows].double() @ w.double().T + b.double()).clamp_min(0).float()
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), want.cpu().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)
    for cfg in (1, 2):
        for k in range(3):
            assert torch.equal(outs[0][k], outs[cfg][k]), (cfg, k)
    # and the automatic choice is one of them
    assert torch.equal(ops.linear_fwd_x3(xi, rows, wi, relu=True, x_nrows=T), outs[0][0])


def test_forward_degenerate_shapes(ops):
    torch.manual_seed(6)
    for M, K, N in [(70000, 1, 1), (300, 5, 1), (1, 700, 3)]:
        x = torch.randn(M, K); w = torch.randn(N, K); b = torch.randn(N)
        xm = ops.empty_mat(M, K, "cuda"); xm.copy_(x)
        y = ops.linear_fwd_x3(ops.x3_split(xm, append_ones=True), None, ops.x3_split(w.cuda(), append_vec=b.cuda()))
        np.testing.assert_allclose(y.cpu().numpy(), (x.double() @ w.double().T + b.double()).float().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL)


def test_random_shape_sweep(ops):
    """Seeded sweep over ragged shapes: image GEMMs (forward with folded bias, weight gradient) and the fused pool
    backward against fp64 / the unfused path.  Sizes straddle every tile, group and chunk boundary."""
    rng = np.random.default_rng(2026)
    for _ in range(24):
        M = int(rng.choice([1, 31, 32, 33, 127, 128, 129, 255, 256, 257, 511, 1000, 4097]))
        K = int(rng.choice([1, 7, 31, 32, 33, 63, 64, 65, 200, 602]))
        N = int(rng.choice([1, 3, 31, 32, 33, 127, 128, 129, 300]))
        x = torch.tensor(rng.standard_normal((M, K)), dtype=torch.float32)
        w = torch.tensor(rng.standard_normal((N, K)) / np.sqrt(K), dtype=torch.float32)
        b = torch.tensor(rng.standard_normal(N), dtype=torch.float32)
        xm = ops.empty_mat(M, K, "cuda"); xm.copy_(x)
        y = ops.linear_fwd_x3(ops.x3_split(xm, append_ones=True), None, ops.x3_split(w.cuda(), append_vec=b.cuda()), relu=bool(M & 1))
        want = x.double() @ w.double().T + b.double()
        if M & 1:
            want = want.clamp_min(0)
        np.testing.assert_allclose(y.cpu().numpy(), want.float().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * 2, err_msg=str((M, K, N)))
        dy = torch.tensor(rng.standard_normal((M, N)), dtype=torch.float32)
        dym = ops.empty_mat(M, N, "cuda"); dym.copy_(dy)
        G = (M + 31) // 32 if (K & 1) else 0
        dw, db = ops.linear_bwd_weight_x3(ops.x3_split_t(dym, interleave=G), ops.x3_split_t(xm, None, ones_row=True, interleave=G))
        scale = max(1.0, np.sqrt(M))
        np.testing.assert_allclose(dw.cpu().numpy(), (dy.double().T @ x.double()).float().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale,
                                   err_msg=str((M, K, N)))
        np.testing.assert_allclose(db.cpu().numpy(), dy.double().sum(0).float().numpy(), rtol=GEMM_RTOL, atol=GEMM_ATOL * scale)
    for _ in range(12):
        n_dst = int(rng.choice([1, 5, 64, 333, 2049])); S = int(rng.choice([1, 2, 16, 17, 25, 40]))
        D = int(rng.choice([1, 63, 64, 65, 130, 602])); n_src = int(rng.choice([1, 31, 32, 33, 500, 5000]))
        idx = torch.tensor(rng.integers(-1, n_src, (n_dst, S)), dtype=torch.int32).cuda()
        pm = ops.empty_mat(n_src, D, "cuda"); pm.copy_(torch.tensor(rng.standard_normal((n_src, D)), dtype=torch.float32).clamp_min(0))
        out, argmax = ops.reduce_fwd(pm, idx, "max", want_argmax=True)
        dm = ops.empty_mat(n_dst, D, "cuda"); dm.copy_(torch.tensor(rng.standard_normal((n_dst, D)), dtype=torch.float32))
        img = ops.pool_bwd_x3(dm, argmax, out, idx, n_src)
        G = (n_src + 31) // 32
        got = image_decode_t(img.buf, D, 32 * G)
        dP = ops.reduce_bwd(dm, None, argmax, "max", n_src, fanout=S, relu_out=out).cpu()
        mm = np.arange(32 * G); s_of_m = (mm % 32) * G + mm // 32; ok = s_of_m < n_src
        assert (got[:, ~ok] == 0).all()
        np.testing.assert_allclose(got[:, ok].T.numpy(), dP[s_of_m[ok]].numpy(), rtol=1e-4, atol=1e-4, err_msg=str((n_dst, S, D, n_src)))


def test_debug_stamps_report_a_plausible_clock(ops):
    """ogl_x3_debug_stamps: per block {s_memtime, s_memrealtime} at entry and exit of the image GEMM (diagnostics only)."""
    from ogl_amd import _lib
    torch.manual_seed(5)
    M, K, N = 20000, 602, 602
    xm = ops.empty_mat(M, K, "cuda"); xm.normal_()
    xi, wi = ops.x3_split(xm), ops.x3_split(torch.randn(N, K, device="cuda"))
    ref = ops.linear_fwd_x3(xi, None, wi)
    stamps = torch.zeros(1024 + 256 * 8 * 4, dtype=torch.int64, device="cuda")
    assert _lib.lib().ogl_x3_debug_stamps(stamps.data_ptr(), 0) == 0
    try:
        got = ops.linear_fwd_x3(xi, None, wi)
        torch.cuda.synchronize()
    finally:
        _lib.lib().ogl_x3_debug_stamps(None, 0)
    assert torch.equal(got, ref)                                   # the stamps do not touch the result
    st = stamps.cpu()[:1024].view(256, 4).double()
    st = st[st[:, 3] > st[:, 1]]
    assert st.shape[0] >= 64                                       # every launched block stamped entry and exit
    ghz = (st[:, 2] - st[:, 0]) / (st[:, 3] - st[:, 1]) * 0.1     # s_memrealtime ticks at 100 MHz
    assert 0.5 < ghz.median().item() < 3.0
    after = stamps.clone()
    ops.linear_fwd_x3(xi, None, wi); torch.cuda.synchronize()
    assert torch.equal(stamps, after)                              # switched off again



def test_self_fetching_kernels_parity():
    """The image GEMM has two forms: producer / consumer (k_gemm_x3p, the default while both images fit 32-bit offsets)
    and self-fetching (k_gemm_x3: images of 4 GB and more).  OGL_X3_PC=0 forces the second form; the forward / weight-
    gradient / dispatch tests of this file must pass on it too (the switch is read once per process: child interpreter)."""
    import os, subprocess, sys
    env = dict(os.environ, OGL_X3_PC="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x",
                        "-k", "forward or weight or random or dispatch or limits", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------------------------
# k_gemm_x3p<..., EXT>: a second A part (K-concatenated product), a per-row addend, the output's own image
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K1,K2,N", [(7060, 602, 602, 600), (300, 40, 70, 33), (14000, 602, 0, 600), (2049, 600, 0, 600)])
def test_x3_ext_two_part_addend_and_output_image(M, K1, K2, N):
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(M + K2)
    dev = "cuda"
    T = M + 500
    table = ops.empty_mat(T, K1, dev).copy_(torch.randn(T, K1, device=dev))
    rows = torch.randint(0, T, (M,), device=dev)
    rows[::97] = -1                                                   # ids outside the table read the zero row
    w1 = torch.randn(N, K1, device=dev) / K1 ** 0.5
    b = torch.randn(N, device=dev)
    t_img = ops.x3_split(table, append_ones=True)
    parts = [(w1, b)]
    x2 = x2_img = None
    if K2:
        x2 = ops.empty_mat(M, K2, dev).copy_(torch.randn(M, K2, device=dev))
        w2 = torch.randn(N, K2, device=dev) / K2 ** 0.5
        parts.append((w2, None))
        x2_img = ops.x3_split(x2)
    w_img = ops.x3_split_cat(parts)
    S0 = ops.empty_mat(T, N, dev).copy_(torch.randn(T, N, device=dev))
    add_rows = torch.randint(0, T, (M,), device=dev)
    xg = torch.where((rows >= 0).unsqueeze(1), table[rows.clamp(min=0)], torch.zeros((), device=dev))
    base = xg.double() @ w1.double().T + b.double() * (rows >= 0).double().unsqueeze(1) * 0 + b.double()   # the ones slot of the zero row is 1 too
    if K2:
        base = base + x2.double() @ w2.double().T
    for use_add, relu in ((False, True), (True, False), (True, True)):
        y, img = ops.linear_fwd_x3_ext(t_img, rows, w_img, x2_img=x2_img, add=S0 if use_add else None,
                                       add_rows=add_rows if use_add else None, relu=relu, x_nrows=T, want_image=True,
                                       image_append_ones=True)
        want = base + (S0[add_rows].double() if use_add else 0)
        if relu:
            want = want.clamp(min=0)
        np.testing.assert_allclose(y.cpu().numpy(), want.float().cpu().numpy(), rtol=1e-4, atol=2e-5)
        # the emitted image is, byte for byte, what a split pass over the fp32 output would have built
        ref_img = ops.x3_split(y, append_ones=True)
        assert img.rows == ref_img.rows == M and img.K == ref_img.K == N + 1
        assert torch.equal(img.buf[:ref_img.buf.numel()], ref_img.buf)
    # without the extensions the EXT entry point equals the plain one bit for bit
    if not K2:
        y0 = ops.linear_fwd_x3(t_img, rows, ops.x3_split(w1, append_vec=b), relu=True, x_nrows=T)
        y1 = ops.linear_fwd_x3_ext(t_img, rows, w_img, relu=True, x_nrows=T, add=torch.zeros_like(S0), add_rows=add_rows)
        assert torch.equal(y0, y1)


@pytest.mark.parametrize("n_src,n_dst,S,d,dtype", [(62000, 7060, 25, 602, torch.int32), (20000, 1500, 25, 128, torch.int64),
                                                   (3000, 512, 25, 600, torch.int32), (500, 64, 7, 36, torch.int32)])
def test_aggregator_emits_the_image_of_its_output(n_src, n_dst, S, d, dtype):
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(n_dst)
    src = ops.empty_mat(n_src, d, "cuda").copy_(torch.randn(n_src, d, device="cuda"))
    idx = torch.randint(0, n_src, (n_dst, S), device="cuda").to(dtype)
    idx[::13] = -1
    out0, arg0 = ops.reduce_fwd(src, idx, "max", want_argmax=True)
    out1, arg1, img = ops.reduce_fwd_img(src, idx, want_argmax=True)
    assert torch.equal(out0, out1) and torch.equal(arg0, arg1)
    ref = ops.x3_split(out0)
    assert img.rows == ref.rows and img.K == ref.K and torch.equal(img.buf[:ref.buf.numel()], ref.buf)
    out2, _, img2 = ops.reduce_fwd_img(src, idx, want_argmax=False)
    assert torch.equal(out2, out0) and torch.equal(img2.buf[:ref.buf.numel()], ref.buf)


@pytest.mark.parametrize("M,N", [(7060, 600), (2500, 602), (1, 1), (300, 33), (4096, 64)])
def test_relu_backward_emits_the_image_of_its_output(M, N):
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(M + N)
    dy = ops.empty_mat(M, N, "cuda").copy_(torch.randn(M, N, device="cuda"))
    y = ops.empty_mat(M, N, "cuda").copy_(torch.randn(M, N, device="cuda").clamp(min=0))
    want = ops.relu_bwd(dy, y)
    assert torch.equal(want, torch.where(y > 0, dy, torch.zeros((), device="cuda")))
    got, img = ops.relu_bwd_img(dy, y)
    assert torch.equal(got, want)
    ref = ops.x3_split(want)
    assert img.rows == ref.rows and img.K == ref.K and torch.equal(img.buf[:ref.buf.numel()], ref.buf)
    # the input gradient on the image kernel = the fp32-operand kernel's, to fp32-GEMM accuracy
    w = torch.randn(N, 70, device="cuda") / N ** 0.5
    ops.set_gemm_mode("auto")
    try:
        dx = ops.linear_bwd_input(want, w, dy_img=img)
    finally:
        ops.set_gemm_mode("f32")
    np.testing.assert_allclose(dx.cpu().numpy(), (want.double() @ w.double()).float().cpu().numpy(), rtol=1e-4, atol=2e-5)


def test_training_layers_run_their_tall_products_on_images():
    """The wiring: in the split-bf16 modes a Reddit-shaped two-layer step takes the two-part image product for the layer-0
    combine, the image kernel for fc_pool of layer 1 and for both n1-row input gradients — and produces the same loss and
    gradients as the fp32-operand kernels (fp32-GEMM accuracy either way)."""
    import torch.nn.functional as F
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    from ogl_amd.graphsage import GatheredRows, GraphSAGE
    torch.manual_seed(5)
    dev = "cuda"
    n_tab, Fi, H, C, n0, n1, B, S = 30000, 602, 600, 41, 20000, 3000, 512, 25
    table = ops.empty_mat(n_tab, Fi, dev).copy_(torch.randn(n_tab, Fi, device=dev))
    ops.register_static_table(table)
    model = GraphSAGE(Fi, H, C, 1, F.relu, 0, "pool").to(dev)

    class Blk:
        def __init__(self, n_src, n_dst):
            self.local_idx = torch.randint(0, n_src, (n_dst, S), device=dev, dtype=torch.int32)
            self._n = n_dst
        def number_of_dst_nodes(self):
            return self._n
    blocks = [Blk(n0, n1), Blk(n1, B)]
    ids = torch.randperm(n_tab, device=dev)[:n0]
    labels = torch.randint(0, C, (B,), device=dev)

    def step(mode):
        ops.set_gemm_mode(mode)
        for p in model.parameters():
            p.grad = None
        ops.profile_start()
        logits = model(blocks, GatheredRows(table, ids))
        loss = ops.cross_entropy(logits, labels, "mean")
        loss.backward()
        names = [n for n, _, _ in ops.profile_stop()]
        return float(loss.detach()), [p.grad.clone() for p in model.parameters()], names
    floor = ops.X3_N1_MIN_ROWS
    try:
        l1, g1, on_images = step("auto")
        ops.X3_N1_MIN_ROWS = 1 << 40          # same arithmetic (split-bf16 x6), operands split on the fly by k_gemm
        l0, g0, seen = step("auto")
    finally:
        ops.X3_N1_MIN_ROWS = floor
        ops.set_gemm_mode("f32")
    assert on_images.count("ogl_linear_fwd_x3_ext") == 1 and "ogl_reduce_fwd_img" in on_images and "ogl_relu_bwd_img" in on_images
    assert on_images.count("ogl_linear_fwd_x3") >= 4      # fc_pool of both layers + the two n1-row input gradients
    # every weight image of the step from ONE launch, no transpose launches for the input gradients
    assert on_images.count("ogl_x3_split_multi") == 1 and "ogl_x3_split_into" not in on_images and "ogl_transpose" not in on_images
    assert "ogl_linear_fwd_x3_ext" not in seen
    assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))
    errs = [float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12) for a, b in zip(g1, g0)]
    assert max(errs) <= 1e-3, errs


def test_weight_images_in_one_launch_equal_the_single_splits():
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(11)
    dev = "cuda"
    w = torch.randn(600, 602, device=dev); w2 = torch.randn(600, 600, device=dev)
    b = torch.randn(600, device=dev); b2 = torch.randn(600, device=dev)
    wp = torch.randn(33, 70, device=dev); bp = torch.randn(33, device=dev)
    wn = torch.randn(41, 129, device=dev)
    ops.invalidate_weight_images()
    ops.profile_start()
    ops.weight_images_prepare([("wb", (wp, bp)), ("cat", (w, w2, b, b2)), ("T", (w2,)), ("wb", (wn, None)), ("T", (wp,)),
                               ("cat", (wp, wp, None, None))])
    assert [n for n, _, _ in ops.profile_stop()] == ["ogl_x3_split_multi"]
    same = lambda a, r: a.rows == r.rows and a.K == r.K and torch.equal(a.buf[:r.buf.numel()], r.buf)
    assert same(ops.weight_image("wb", wp, bp), ops.x3_split(wp, append_vec=bp))
    assert same(ops.weight_image("cat", w, w2, b, b2), ops.x3_split_cat([(w, b + b2), (w2, None)]))
    assert same(ops.weight_image("T", w2), ops.x3_split(ops.transpose(w2)))
    assert same(ops.weight_image("wb", wn, None), ops.x3_split(wn, append_vec=torch.zeros(41, device=dev)))
    assert same(ops.weight_image("T", wp), ops.x3_split(ops.transpose(wp)))
    assert same(ops.weight_image("cat", wp, wp, None, None), ops.x3_split_cat([(wp, torch.zeros(33, device=dev)), (wp, None)]))
    # keyed by storage + version: an in-place update or an optimiser step drops the entry
    wp.add_(1.0)
    assert ops.weight_image("wb", wp, bp) is None
    assert ops.weight_image("T", w2) is not None
    m, v = torch.zeros_like(w2), torch.zeros_like(w2)
    ops.adam_step(w2, torch.ones_like(w2), m, v, 1)
    assert ops.weight_image("T", w2) is None


@pytest.mark.parametrize("M,N,K,gather,interleave,ones", [(7060, 600, 602, True, False, True), (7060, 600, 600, False, False, False),
                                                          (20000, 600, 602, True, True, True), (2500, 41, 130, True, True, True),
                                                          (2049, 33, 70, False, False, True), (5000, 200, 31, True, False, True)])
def test_weight_gradient_reads_the_row_major_image(M, N, K, gather, interleave, ones):
    """dw = dy^T . x[rows] with x as its ROW-MAJOR image (k_gemm_x3p<BK>: k-major B operand through ds_read_b64_tr_b16) against
    the transposed-image product: the same MFMAs on the same operands in the same order -> equal to the last bit wherever the
    split-K plans agree, and always to fp32-GEMM accuracy against fp64."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(M + K)
    dev = "cuda"
    T = M + 700
    table = ops.empty_mat(T, K, dev).copy_(torch.randn(T, K, device=dev))
    rows = None
    if gather:
        rows = torch.randint(0, T, (M,), device=dev)
        rows[::53] = -1
    dy = ops.empty_mat(M, N, dev).copy_(torch.randn(M, N, device=dev))
    G = (M + 31) // 32 if interleave else 0
    dyT = ops.x3_split_t(dy, interleave=G)
    x_img = ops.x3_split(table if gather else table[:M], append_ones=ones)
    dw, db, db2 = ops.linear_bwd_weight_x3k(dyT, x_img, M, K, x_rows=rows, x_nrows=T if gather else None, interleave=G,
                                            want_bias=ones, want_bias2=ones)
    xg = table[:M] if rows is None else torch.where((rows >= 0).unsqueeze(1), table[rows.clamp(min=0)], torch.zeros((), device=dev))
    want = dy.double().T @ xg.double()
    scale = float(want.abs().max())
    assert float((dw.double() - want).abs().max()) <= 2e-5 * scale
    if ones:
        wantb = dy.double().sum(0)
        assert float((db.double() - wantb).abs().max()) <= 2e-5 * float(wantb.abs().max())
        assert torch.equal(db, db2)
    ref_dw, ref_db = ops.linear_bwd_weight_x3(dyT, ops.x3_split_t(table, rows, ones_row=True, interleave=G) if gather
                                              else ops.x3_split_t(table[:M], None, ones_row=True, interleave=G), want_bias=True)
    assert float((dw - ref_dw).abs().max()) <= 1e-5 * scale
    if not interleave:
        # dy too as its own row-major image (what the ReLU backward / a split pass writes), read k-major: no transposed image at all
        dw2, db3, _ = ops.linear_bwd_weight_x3k(ops.x3_split(dy), x_img, M, K, x_rows=rows, x_nrows=T if gather else None,
                                                want_bias=ones, dy_rows=True)
        assert float((dw2 - dw).abs().max()) <= 1e-5 * scale
        if ones:
            assert float((db3 - db).abs().max()) <= 1e-5 * float(wantb.abs().max())


def test_train_step_needs_no_transposed_activation_images():
    """With the k-major weight-gradient product the Reddit-shaped step builds NO transposed image by a pass of its own (the fused
    pool backward writes dP0^T directly): x^T of the table rows, neigh^T, h1^T and the gradients' dy^T / dP1^T are gone."""
    import torch.nn.functional as F
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    from ogl_amd.graphsage import GatheredRows, GraphSAGE
    torch.manual_seed(6)
    dev = "cuda"
    n_tab, Fi, H, C, n0, n1, B, S = 30000, 602, 600, 41, 20000, 3000, 512, 25
    table = ops.empty_mat(n_tab, Fi, dev).copy_(torch.randn(n_tab, Fi, device=dev))
    ops.register_static_table(table)
    model = GraphSAGE(Fi, H, C, 1, F.relu, 0, "pool").to(dev)

    class Blk:
        def __init__(self, n_src, n_dst):
            self.local_idx = torch.randint(0, n_src, (n_dst, S), device=dev, dtype=torch.int32)
            self._n = n_dst
        def number_of_dst_nodes(self):
            return self._n
    blocks = [Blk(n0, n1), Blk(n1, B)]
    ids = torch.randperm(n_tab, device=dev)[:n0]
    labels = torch.randint(0, C, (B,), device=dev)

    def step():
        for p in model.parameters():
            p.grad = None
        ops.profile_start()
        loss = ops.cross_entropy(model(blocks, GatheredRows(table, ids)), labels, "mean")
        loss.backward()
        names = [n for n, _, _ in ops.profile_stop()]
        return float(loss.detach()), [p.grad.clone() for p in model.parameters()], names
    ops.set_gemm_mode("auto")
    flag = ops.K_MAJOR_WEIGHT_GRADS
    try:
        l1, g1, names = step()
        ops.K_MAJOR_WEIGHT_GRADS = False
        l0, g0, names0 = step()
    finally:
        ops.K_MAJOR_WEIGHT_GRADS = flag
        ops.set_gemm_mode("f32")
    assert "ogl_x3_split_t" not in names and names.count("ogl_linear_bwd_weight_x3k") == 4 and "ogl_linear_bwd_weight_x3" not in names
    assert names0.count("ogl_x3_split_t") == 6 and "ogl_linear_bwd_weight_x3k" not in names0
    assert l1 == l0
    errs = [float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12) for a, b in zip(g1, g0)]
    assert max(errs) <= 1e-5, errs


def test_activation_images_never_outlive_their_tensor_or_its_contents():
    """An image travels as an attribute of the activation's tensor object, stamped with its version: a new tensor at a recycled
    address or the same tensor after an in-place write has none."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    t = ops.empty_mat(300, 64, "cuda").copy_(torch.randn(300, 64, device="cuda"))
    img = ops.x3_split(t)
    ops.attach_image(t, img)
    assert ops.take_image(t, pop=False) is img
    ptr = t.data_ptr()
    u = t
    u.add_(1.0)                                     # contents changed: the image is stale
    assert ops.take_image(t, pop=False) is None
    ops.attach_image(t, img)
    del t, u
    v = ops.empty_mat(300, 64, "cuda")               # very likely the same address (caching allocator)
    assert ops.take_image(v) is None or v.data_ptr() != ptr
    ops.attach_image(v, img)
    assert ops.take_image(v) is img and ops.take_image(v) is None      # popped


def test_status_codes_of_the_round2_entry_points(ops):
    """Wrong shapes are refused (OGL_EINVAL = -1) before anything is launched."""
    import ctypes as C
    from ogl_amd import _lib
    h = _lib.lib()
    x = torch.zeros(64, 64).cuda(); img = torch.zeros(1 << 16, dtype=torch.uint8).cuda()
    i32 = torch.zeros(64, 64, dtype=torch.int32).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    f = C.c_float
    # k-major weight gradient: a bias gradient needs the ones slot; interleave must cover M; gather bound within the image
    assert h.ogl_linear_bwd_weight_x3k(p(img), 0, p(img), 64, None, 64, 64, 8, 8, 0, p(x), 64, p(x), None, None, 0, None) == -1
    assert h.ogl_linear_bwd_weight_x3k(p(img), 1, p(img), 64, None, 64, 64, 8, 8, 1, p(x), 64, None, None, None, 0, None) == -1
    assert h.ogl_linear_bwd_weight_x3k(p(img), 0, p(img), 64, p(img), 65, 64, 8, 8, 1, p(x), 64, None, None, None, 0, None) == -1
    assert h.ogl_linear_bwd_weight_x3k(p(img), -2, p(img), 64, None, 64, 64, 8, 8, 1, p(x), 64, None, None, None, 0, None) == -1
    # split-multi: more than 8 parts, a bias vector without the slot
    assert h.ogl_x3_split_multi(p(img), 9, None, None, 0.0, 0.0, 0.0, None) == -1
    part = ops._X3SplitPart()
    part.src, part.ld, part.R, part.K, part.transpose, part.append = x.data_ptr(), 64, 64, 64, 0, 0
    part.vec1, part.image, part.image_row_bytes, part.group_offset = x.data_ptr(), img.data_ptr(), 2 * 192, 0
    assert h.ogl_x3_split_multi(C.byref(part), 1, None, None, 0.0, 0.0, 0.0, None) == -1
    assert h.ogl_x3_split_multi(C.byref(part), 1, p(x), None, 0.0, 0.0, 0.0, None) == -1         # the optimiser's scalars: both addresses or neither
    # ReLU backward + image: no image / bad strides
    assert h.ogl_relu_bwd_img(p(x), 64, p(x), 64, 64, 64, p(x), 64, None, None) == -1
    assert h.ogl_relu_bwd_img(p(x), 8, p(x), 64, 64, 64, p(x), 64, p(img), None) == -1
    # output-layer backward: at most 64 columns, at most 4096 rows for the weights kernel
    assert h.ogl_out_layer_bwd_inputs(p(x), 64, 64, 65, 64, p(x), 64, p(x), 64, p(i32), p(x), 64, 64, p(x), 64, p(x), 64, None) == -1
    assert h.ogl_out_layer_bwd_weights(p(x), 64, 5000, 8, 64, p(x), 64, None, 0, p(x), 64, p(x), 64, p(x), 64, None, None, None) == -1
    assert h.ogl_out_layer_bwd_weights(p(x), 64, 64, 8, 64, p(x), 64, p(img), 0, p(x), 64, p(x), 64, p(x), 64, None, None, None) == -1
    # small cross entropy: one workgroup covers at most 1024 rows, and it needs the mean's address
    lab = torch.zeros(64, dtype=torch.int64).cuda()
    assert h.ogl_ce_fwd_bwd_mean_gather(p(x), 64, p(lab), 64, None, 1025, 8, f(1.0), None, None, 0, p(x), None, 0, None, None, 0.0, 0.0, 0.0, None) == -1
    assert h.ogl_ce_fwd_bwd_mean_gather(p(x), 64, p(lab), 64, None, 64, 8, f(1.0), None, None, 0, None, None, 0, None, None, 0.0, 0.0, 0.0, None) == -1
