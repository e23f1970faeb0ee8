"""Round 6: the staggered multiplier waves of the image GEMM (same bits as the lockstep form) and the gradient ROUTE of the fused
32-seed last layer under every way of running a backward the package does not own (correct gradients or a raised error, never
unwritten memory).  Run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_staggered_multiplier_waves_compute_the_same_bits():
    """k_gemm_x3p with waves 4-7 running each step's last column block behind the NEXT step's opening barrier (ogl_x3_debug_stagger)
    against every wave opening a step on its fragment loads: the same MFMAs on the same operands, every accumulator in the same
    order — bit for bit on every instantiation the train step and the inference passes launch (256 / 192 / 128-row tiles, early-A
    and one-barrier forms, the two-part / addend / image-writing form, the k-major weight gradients incl. the dual product), on
    ragged shapes, one-step and two-step reductions, several tiles per block."""
    import ogl_amd  # noqa: F401
    from ogl_amd import _lib, ops
    ops.set_gemm_mode("auto")
    lib = _lib.lib()
    was, was_ea = lib.ogl_x3_debug_stagger(-1), lib.ogl_x3_debug_early_a(-1)
    try:
        torch.manual_seed(21)
        dev = "cuda"
        x_big = ops.empty_mat(40000, 602, dev).copy_(torch.randn(40000, 602, device=dev))
        x_mid = ops.empty_mat(7061, 600, dev).copy_(torch.randn(7061, 600, device=dev))
        x_few = ops.empty_mat(1500, 31, dev).copy_(torch.randn(1500, 31, device=dev))            # one reduction step (31 + 1)
        x_two = ops.empty_mat(70000, 40, dev).copy_(torch.randn(70000, 40, device=dev))           # two steps, five tiles per block
        w = torch.randn(600, 602, device=dev) / 25; b = torch.randn(600, device=dev)
        w2 = torch.randn(600, 600, device=dev) / 25
        w_few = torch.randn(77, 31, device=dev); b_few = torch.randn(77, device=dev)
        w_two = torch.randn(600, 40, device=dev); b_two = torch.randn(600, device=dev)
        xi_big, xi_mid = ops.x3_split(x_big, append_ones=True), ops.x3_split(x_mid, append_ones=True)
        xi_few, xi_two = ops.x3_split(x_few, append_ones=True), ops.x3_split(x_two, append_ones=True)
        wi, w2i = ops.x3_split(w, append_vec=b), ops.x3_split(w2, append_vec=b)
        wi_few, wi_two = ops.x3_split(w_few, append_vec=b_few), ops.x3_split(w_two, append_vec=b_two)
        rows = torch.randint(0, 40000, (30001,), device=dev)
        dy = ops.empty_mat(30001, 600, dev).copy_(torch.randn(30001, 600, device=dev))
        G = (30001 + 31) // 32
        dyT = ops.x3_split_t(dy, interleave=G)
        dy_img = ops.x3_split(dy)
        x2 = ops.empty_mat(30001, 600, dev).copy_(torch.randn(30001, 600, device=dev))
        x2_img = ops.x3_split(x2)
        add = ops.empty_mat(40000, 600, dev).copy_(torch.randn(40000, 600, device=dev))
        outs = {}
        for ea in (1, 0):
            lib.ogl_x3_debug_early_a(ea)
            for stag in (0, 1):
                lib.ogl_x3_debug_stagger(stag)
                got, names = [], []

                def take(t):
                    got.extend(t if isinstance(t, (list, tuple)) else [t]); names.append(lib.ogl_x3_last_kernel().decode())
                take(ops.linear_fwd_x3(xi_big, rows, wi, relu=True))                                   # 256 x 128, gathered rows, ragged M
                take(ops.linear_fwd_x3(xi_mid, None, w2i, relu=False))                                 # 192 x 128, one round
                take(ops.linear_fwd_x3(xi_few, None, wi_few, relu=True))                               # 128 x 128, three stages, ONE step
                take(ops.linear_fwd_x3(xi_two, None, wi_two, relu=True))                               # two steps per tile, tiles in a row
                y, img = ops.linear_fwd_x3_ext(xi_big, rows, wi, add=add, add_rows=rows, relu=True, want_image=True, image_append_ones=True)
                take([y, img.buf])
                dw, db, _ = ops.linear_bwd_weight_x3k(dyT, xi_big, 30001, 602, x_rows=rows, x_nrows=40000, interleave=G, want_bias=True)
                take([dw, db])
                dw2, db2, _ = ops.linear_bwd_weight_x3k(dy_img, x2_img, 30001, 600, dy_rows=True, want_bias=False)   # k-major x k-major
                take([dw2])
                dual = ops.linear_bwd_weight_x3k_dual(dy_img, xi_big, rows, 40000, 30001, 602, x2_img, 600)
                if dual is not None:
                    ws, stride, ws_ld, nsplit, col2, N = dual
                    take(ws.clone())
                outs[(ea, stag)] = (got, names)
        for ea in (1, 0):
            (g0, n0), (g1, n1) = outs[(ea, 0)], outs[(ea, 1)]
            assert n0 == n1                                                   # (the same instantiations, with and without the stagger)
            for i, (a, c) in enumerate(zip(g0, g1)):
                if a is None:
                    continue
                assert torch.equal(a, c), (ea, i)
        assert len(set(outs[(1, 0)][1])) >= 5, outs[(1, 0)][1]                 # (at least five different instantiations ran)
    finally:
        lib.ogl_x3_debug_stagger(was)
        lib.ogl_x3_debug_early_a(was_ea)
        ops.set_gemm_mode("f32")


def _small_step(T=4000, n0=900, n1=120, B=32, S=25, F_in=500, C_out=3):
    import ogl_amd  # noqa: F401
    from ogl_amd import ops, sampling
    from ogl_amd.graphsage import GraphSAGE
    from ogl_amd.graphsage.sageconv import GatheredRows
    rng = np.random.default_rng(T + n1)
    torch.manual_seed(T + B)
    table = ops.empty_mat(T, F_in, "cuda").copy_(torch.randn(T, F_in, device="cuda"))
    ids0 = torch.as_tensor(rng.choice(T, n0, replace=False).astype(np.int64)).cuda()
    lidx0 = rng.integers(0, n0, size=(n1, S)).astype(np.int32)
    lidx1 = rng.integers(0, n1, size=(B, S)).astype(np.int32)
    lidx1[rng.random(B) < 0.1] = -1
    blocks = [sampling.Block(ids0, ids0[:n1], torch.as_tensor(lidx0).cuda()), sampling.Block(ids0[:n1], ids0[:B], torch.as_tensor(lidx1).cuda())]
    labels = torch.randint(0, C_out, (B,), device="cuda")
    model = GraphSAGE(F_in, 32, C_out, 1, F.relu, 0, "pool").cuda()

    def forward():
        return model.forward_loss(blocks, GatheredRows(table, ids0), labels, rows=True, defer_mean=True)
    return ops, model, forward


def _reference_grads(ops, model, forward):
    """The same step with the last layer's own backward launch (no route)."""
    ops.SMALL_ROUTE = False
    try:
        for p in model.parameters():
            p.grad = None
        loss, _, _ = forward()
        ops.backward(loss)
        torch.cuda.synchronize()
        return {k: v.grad.clone() for k, v in model.named_parameters()}
    finally:
        ops.SMALL_ROUTE = True
        for p in model.parameters():
            p.grad = None


def _close(got, want, names=None):
    for k in (names or want):
        scale = max(1.0, float(want[k].abs().max()))
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k].cpu().numpy(), rtol=1e-4, atol=1e-6 * scale, err_msg=k)


def test_route_is_not_taken_outside_an_owned_backward():
    """``torch.autograd.grad(loss, [last-layer weights])`` (the first layer's node is pruned: nobody would ever consume a route) and a
    plain ``loss.backward()``: the fused last layer finishes with its own launch — correct gradients, nothing pending."""
    ops, model, forward = _small_step()
    want = _reference_grads(ops, model, forward)
    last = [(k, p) for k, p in model.named_parameters() if k.startswith("layers.1.")]
    loss, _, _ = forward()
    assert type(loss.grad_fn).__name__ == "_SmallPoolLossFnBackward" and loss.grad_fn is not None
    gs = torch.autograd.grad(loss, [p for _, p in last])
    torch.cuda.synchronize()
    assert not ops._PENDING_ROUTES
    _close({k: g for (k, _), g in zip(last, gs)}, want, [k for k, _ in last])
    # a backward the package does not own, whole graph
    for p in model.parameters():
        p.grad = None
    loss, _, _ = forward()
    loss.backward()
    torch.cuda.synchronize()
    assert not ops._PENDING_ROUTES
    _close({k: v.grad for k, v in model.named_parameters()}, want)


def test_route_with_gradients_already_in_place_accumulates_correctly():
    """A parameter that already holds a gradient (zero_grad(set_to_none=False), gradient accumulation): AccumulateGrad would add the
    route's still-unwritten tensor — the node must finish with its own launch instead; the sums are exact."""
    ops, model, forward = _small_step()
    want = _reference_grads(ops, model, forward)
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    loss, _, _ = forward()
    ops.backward(loss)
    torch.cuda.synchronize()
    assert not ops._PENDING_ROUTES
    _close({k: v.grad - 1.0 for k, v in model.named_parameters()}, want)


def test_route_under_hooks_is_correct_or_raises():
    """A hook on a last-layer weight sees that weight's real gradient; a tensor hook / retain_grad on the hidden rows (registered by a
    module forward hook, the only way to reach them) switches the route off at the forward."""
    ops, model, forward = _small_step()
    want = _reference_grads(ops, model, forward)
    seen = {}
    w = model.layers[1].fc_self.weight
    handle = w.register_hook(lambda g: seen.setdefault("g", g.clone()))
    try:
        loss, _, _ = forward()
        ops.backward(loss)
        torch.cuda.synchronize()
        # (a gradient hook runs when autograd hands the tensor over — a route's record launch would fill it only afterwards: with a
        # hook on any of the layer's parameters the node finishes with its own launch, and the hook sees the real gradient)
        _close({"layers.1.fc_self.weight": seen["g"]}, want, ["layers.1.fc_self.weight"])
        _close({k: v.grad for k, v in model.named_parameters()}, want)
    finally:
        handle.remove()
    for p in model.parameters():
        p.grad = None
    kept = {}

    def keep(mod, inp, out):
        out.retain_grad()
        kept["h"] = out
    hh = model.layers[0].register_forward_hook(keep)
    try:
        loss, _, _ = forward()
        ops.backward(loss)
        torch.cuda.synchronize()
        assert not ops._PENDING_ROUTES
        _close({k: v.grad for k, v in model.named_parameters()}, want)
        h = kept["h"]
        assert h.grad is not None and bool(torch.isfinite(h.grad).all())
        # the hidden rows' gradient against the unfused layers (oracle-checked elsewhere): finite and the right shape is not enough
        ops.SMALL_ROUTE = False
        try:
            for p in model.parameters():
                p.grad = None
            loss2, _, _ = forward()
            ops.backward(loss2)
            torch.cuda.synchronize()
            h2 = kept["h"]
            np.testing.assert_allclose(h.grad.cpu().numpy(), h2.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)
        finally:
            ops.SMALL_ROUTE = True
    finally:
        hh.remove()


def test_unconsumed_route_raises():
    """A tensor hook that REPLACES the gradient of the hidden rows (so the first layer's node receives a different tensor than the
    route's stand-in) registered after the forward: ``ops.backward`` raises instead of leaving the last layer's gradients unwritten,
    and the optimiser refuses too."""
    ops, model, forward = _small_step()
    from ogl_amd import optim
    grabbed = {}
    hh = model.layers[0].register_forward_hook(lambda m, i, o: grabbed.setdefault("h", o))
    try:
        loss, _, _ = forward()
    finally:
        hh.remove()
    node = loss.grad_fn
    if not getattr(node, "lazy", False):
        pytest.skip("the route was not taken at this shape")
    grabbed["h"].register_hook(lambda g: g * 1.0)             # (after the forward: the node already promised the route)
    with pytest.raises(RuntimeError, match="(?i)route"):
        ops.backward(loss)
    assert not ops._PENDING_ROUTES
    ops._PENDING_ROUTES[123] = dict()
    opt = optim.Adam(model.parameters(), lr=1e-3)
    with pytest.raises(RuntimeError, match="(?i)route"):
        opt.step()
    assert not ops._PENDING_ROUTES
