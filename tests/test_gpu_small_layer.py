"""One-workgroup 'pool' layer (csrc/small_layer.hip) against the oracle's layer and against the multi-launch path it replaces.
Run with -m gpu.  Tolerances: forward rtol 1e-4 / atol 1e-5 (fp32 FMA chain vs fp32 MFMA / CPU BLAS), gradients rtol 1e-3 /
atol 1e-5; argmax equal wherever the two winners are not within rounding of each other."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_src,n_dst,S,hin,hout,relu,bias", [
    (832, 32, 25, 32, 40, False, True), (120, 32, 25, 32, 3, False, True), (300, 48, 7, 16, 16, True, True),
    (90, 9, 5, 64, 64, True, False), (40, 40, 3, 8, 5, False, True), (5000, 100, 25, 32, 40, False, True)])
def test_small_pool_layer_matches_oracle_and_unfused(n_src, n_dst, S, hin, hout, relu, bias):
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    rng = np.random.default_rng(n_src + hin)
    assert ops.small_pool_layer_fits(n_src, n_dst, S, hin, hout)
    h = rng.standard_normal((n_src, hin)).astype(np.float32)
    idx = rng.integers(0, n_src, size=(n_dst, S)).astype(np.int32)
    idx[rng.random(n_dst) < 0.15] = -1                                  # destinations without a sampled neighbour
    prm = O.init_layer_params("pool", hin, hout)
    if not bias:
        for k in list(prm):
            if k.endswith(".bias"):
                prm[k] = torch.zeros_like(prm[k])
    gy = rng.standard_normal((n_dst, hout)).astype(np.float32)

    def run(small):
        ops.SMALL_LAYER = small
        try:
            ht = ops.empty_mat(n_src, hin, "cuda").copy_(torch.as_tensor(h)).requires_grad_(True)
            ps = {k: v.clone().cuda().requires_grad_(True) for k, v in prm.items()}
            b = (lambda n: ps[n] if bias else None)
            y = ops.sage_pool_layer(ht, ps["fc_pool.weight"], b("fc_pool.bias"), ps["fc_self.weight"], ps["fc_neigh.weight"],
                                    b("fc_self.bias"), b("fc_neigh.bias"), torch.as_tensor(idx).cuda(), n_dst, relu)
            y.backward(torch.as_tensor(gy).cuda())
            return y.detach().cpu().numpy(), ht.grad.cpu().numpy(), {k: (v.grad.cpu().numpy() if v.grad is not None else None) for k, v in ps.items()}
        finally:
            ops.SMALL_LAYER = True
    y1, dh1, g1 = run(True)
    y0, dh0, g0 = run(False)
    hr = torch.tensor(h, requires_grad=True)
    pr = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    yr = O.sageconv_forward("pool", hr, n_dst, idx, pr, activation=F.relu if relu else None)
    yr.backward(torch.as_tensor(gy))
    np.testing.assert_allclose(y1, yr.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y1, y0, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dh1, hr.grad.numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(dh1, dh0, rtol=1e-3, atol=1e-5)
    for k in prm:
        if k.endswith(".bias") and not bias:
            continue
        np.testing.assert_allclose(g1[k], pr[k].grad.numpy(), rtol=1e-3, atol=1e-5, err_msg=k)
        np.testing.assert_allclose(g1[k], g0[k], rtol=1e-3, atol=1e-5, err_msg=k)
    # inference entry: same values, no autograd node
    with torch.no_grad():
        yi = ops.small_pool_layer_fwd(torch.as_tensor(h).cuda(), prm["fc_pool.weight"].cuda(), prm["fc_pool.bias"].cuda() if bias else None,
                                      prm["fc_self.weight"].cuda(), prm["fc_neigh.weight"].cuda(), prm["fc_self.bias"].cuda() if bias else None,
                                      prm["fc_neigh.bias"].cuda() if bias else None, torch.as_tensor(idx).cuda(), n_dst, relu, want_argmax=False)[0]
    assert np.array_equal(yi.cpu().numpy(), y1)


def test_small_layer_limits():
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    assert ops.small_pool_layer_fits(832, 32, 25, 32, 40)
    assert not ops.small_pool_layer_fits(7060, 512, 25, 600, 41)          # the Reddit layer stays on the GEMM kernels
    assert ops.small_pool_layer_fits(20000, 32, 25, 32, 40) and not ops.small_pool_layer_fits(70000, 32, 25, 32, 40)
    assert not ops.small_pool_layer_fits(4000, 512, 25, 32, 40)           # too many destinations to call it small
    assert not ops.small_pool_layer_fits(100, 32, 25, 65, 8)


@pytest.mark.parametrize("n_src,n_dst,S,K,N", [(7060, 512, 25, 600, 41), (300, 37, 5, 70, 3), (5000, 1000, 10, 128, 64), (900, 4, 25, 257, 40)])
def test_output_layer_backward_in_two_launches(n_src, n_dst, S, K, N):
    """csrc/out_layer.hip against the general launches it replaces (two input-gradient products + the max scatter; two skinny
    weight gradients): same values up to fp32 summation order (the scatter adds in atomic order either way)."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(n_src + N)
    dev = "cuda"
    P = ops.empty_mat(n_src, K, dev).copy_(torch.randn(n_src, K, device=dev).clamp(min=0))
    idx = torch.randint(0, n_src, (n_dst, S), device=dev, dtype=torch.int32)
    idx[::7] = -1
    neigh, argmax = ops.reduce_fwd(P, idx, "max", want_argmax=True)
    h_dst = ops.empty_mat(n_dst, K, dev).copy_(torch.randn(n_dst, K, device=dev))
    dy = ops.empty_mat(n_dst, N, dev).copy_(torch.randn(n_dst, N, device=dev))
    w_self = torch.randn(N, K, device=dev) / K ** 0.5
    w_neigh = torch.randn(N, K, device=dev) / K ** 0.5
    dx, dp = ops.out_layer_bwd_inputs(dy, w_self, w_neigh, argmax, neigh, n_src)
    want_dx = dy.double() @ w_self.double()
    np.testing.assert_allclose(dx.cpu().numpy(), want_dx.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    dneigh = (dy.double() @ w_neigh.double()).float()
    want_dp = ops.reduce_bwd(ops.as_mat(dneigh), None, argmax, "max", n_src, fanout=S, relu_out=neigh)
    np.testing.assert_allclose(dp.cpu().numpy(), want_dp.cpu().numpy(), rtol=1e-5, atol=2e-5)
    dws, dwn, db, db2 = ops.out_layer_bwd_weights(dy, h_dst, neigh)
    np.testing.assert_allclose(dws.cpu().numpy(), (dy.double().T @ h_dst.double()).float().cpu().numpy(), rtol=1e-5, atol=2e-5 * n_dst ** 0.5)
    np.testing.assert_allclose(dwn.cpu().numpy(), (dy.double().T @ neigh.double()).float().cpu().numpy(), rtol=1e-5, atol=2e-5 * n_dst ** 0.5)
    np.testing.assert_allclose(db.cpu().numpy(), dy.double().sum(0).float().cpu().numpy(), rtol=1e-5, atol=1e-4)
    assert torch.equal(db, db2)
    # same summation order as the one-product kernel it doubles
    ref_dw, ref_db = ops.linear_bwd_weight(dy, h_dst, None, None, want_bias=True)
    assert torch.equal(dws, ref_dw) and torch.equal(db, ref_db)


def test_fused_weight_gradients_gather_their_rows_and_small_ce_mean():
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(3)
    dev = "cuda"
    T, M, K, N = 5000, 832, 128, 32
    table = ops.empty_mat(T, K, dev).copy_(torch.randn(T, K, device=dev))
    rows = torch.randint(0, T, (M,), device=dev); rows[::41] = -1
    neigh = ops.empty_mat(M, K, dev).copy_(torch.randn(M, K, device=dev))
    dy = ops.empty_mat(M, N, dev).copy_(torch.randn(M, N, device=dev))
    dws, dwn, db, db2 = ops.out_layer_bwd_weights(dy, table, neigh, x_self_rows=rows)
    ref_s, ref_b = ops.linear_bwd_weight(dy, table, None, rows, want_bias=True)
    ref_n, _ = ops.linear_bwd_weight(dy, neigh, None, None, want_bias=True)
    assert torch.equal(dws, ref_s) and torch.equal(dwn, ref_n) and torch.equal(db, ref_b) and torch.equal(db2, ref_b)
    # cross entropy of a small batch with the mean from the same launch
    for B, Cc in ((32, 40), (1, 3), (128, 41)):
        logits = ops.empty_mat(B, Cc, dev).copy_(torch.randn(B, Cc, device=dev) * 3)
        labels = torch.randint(0, Cc, (B,), device=dev)
        mean, rows_l, dl = ops.ce_fwd_bwd_mean(logits, labels)
        ref_rows, ref_dl = ops.ce_fwd_bwd(logits, labels, 1.0 / B)
        assert torch.equal(rows_l, ref_rows) and torch.equal(dl, ref_dl)
        want = torch.nn.functional.cross_entropy(logits.double(), labels)
        assert abs(float(mean) - float(want)) <= 1e-6 * max(1.0, abs(float(want)))
