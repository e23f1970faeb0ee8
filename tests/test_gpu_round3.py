"""Round-3 additions, each against a reference computed another way on the same device or the CPU oracle:
the grid-wide cross-entropy mean (+ its fused zero fill), the summed bias riding in the weight-image launch, the forked
backward (side-stream weight gradients), and the advisor's stale-image scenario (eager inference between REPLAYED optimiser
steps).  Run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,C", [(512, 41), (1024, 40), (1500, 7), (129, 3)])
def test_ce_mean_grid_matches_rows_and_small_kernel(B, C):
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    g = torch.Generator().manual_seed(B)
    logits = (torch.randn(B, C, generator=g) * 3).cuda()
    labels = torch.randint(0, C, (B,), generator=g).cuda()
    rows_ref, dl_ref = ops.ce_fwd_bwd(logits, labels, 1.0 / B)
    slot = ops.request_zeroed(777, 600, logits.device)
    slot[0].fill_(3.0)
    mean, rows, dl = ops.ce_fwd_bwd_mean_grid(logits, labels)
    assert torch.equal(rows, rows_ref) and torch.equal(dl, dl_ref)                 # the same per-row arithmetic
    ref = F.cross_entropy(logits.double().cpu(), labels.cpu()).item()
    assert abs(float(mean) - ref) <= 1e-6 * max(1.0, abs(ref))
    if B <= 1024:                                                                  # the one-workgroup kernel sums in the same order
        m2, _, _ = ops.ce_fwd_bwd_mean(logits, labels)
        assert float(m2) == float(mean)
    z = ops.take_zeroed(slot, 777, 600)                                            # the parked buffer was cleared by that launch
    assert slot[0].abs().sum().item() == 0.0 and z.shape == (777, 600)
    # the counter resets itself: a second launch gives the same mean
    mean2, _, _ = ops.ce_fwd_bwd_mean_grid(logits, labels)
    assert float(mean2) == float(mean)


def test_mean_rows_function_is_differentiable_through_the_mean_only():
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(600, 41, generator=g)).cuda().requires_grad_(True)
    labels = torch.randint(0, 41, (600,), generator=g).cuda()
    mean, rows = ops.cross_entropy_mean_rows(logits, labels)
    assert not rows.requires_grad and mean.requires_grad
    ops.backward(mean)
    ref_in = logits.detach().cpu().double().requires_grad_(True)
    ref = F.cross_entropy(ref_in, labels.cpu())
    ref.backward()
    torch.testing.assert_close(logits.grad.cpu().double(), ref_in.grad, rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(rows.cpu().double(), F.cross_entropy(ref_in.detach(), labels.cpu(), reduction="none"), rtol=1e-5, atol=1e-6)


def test_bias_sum_rides_in_the_weight_image_launch():
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    ops.set_gemm_mode("auto")
    ops.invalidate_weight_images()
    w = torch.randn(600, 600, device="cuda"); b = torch.randn(600, device="cuda")
    b1 = torch.randn(41, device="cuda"); b2 = torch.randn(41, device="cuda")
    ops.weight_images_prepare([("wb", (w, b)), ("bsum", (b1, b2)), ("T", (w,))])
    s = ops.weight_image("bsum", b1, b2)
    assert s is not None and torch.equal(s, b1 + b2)
    img = ops.weight_image("wb", w, b)
    ref = ops.x3_split(w, append_vec=b)
    assert torch.equal(img.buf, ref.buf)
    ops.invalidate_weight_images()


def _reddit_like_small():
    from ogl_amd import synthetic
    feat_size, labels, dyn, n_classes, _ = synthetic.load("reddit", snapshots=2, device="cuda", scale=0.12)
    dyn.evolve()
    return feat_size, labels, dyn.get_graph(), n_classes


def test_forked_backward_gives_the_serial_gradients():
    """The weight gradients launched on the side stream are the serial ones bit for bit (deterministic split-K slabs; the
    fork only changes which stream runs them)."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops, sampling
    from ogl_amd.graphsage import GatheredRows, GraphSAGE
    feat_size, labels, g, n_classes = _reddit_like_small()
    ops.set_gemm_mode("auto")
    torch.manual_seed(2)
    model = GraphSAGE(feat_size, 600, n_classes, 1, F.relu, 0, "pool", edge_feats=0, pool_feats=600).cuda()
    seeds = torch.as_tensor(np.random.default_rng(0).choice(g.n_present, 512, replace=False).astype(np.int64)).cuda()
    grads = {}
    for fork in (False, True):
        ops.FORK_BACKWARD = fork
        ops.invalidate_weight_images()
        sampling.seed(11)
        (input_nodes, sd, blocks), = list(sampling.NodeDataLoader(g, seeds.cpu(), sampling.MultiLayerNeighborSampler([25, 25]), batch_size=512))
        assert blocks[1].number_of_src_nodes() >= 2048        # tall enough for the image kernels (and so for the fork)
        for p in model.parameters():
            p.grad = None
        lab = ops.gather_i64(g.ndata["target"], sd)
        loss = ops.cross_entropy(model(blocks, GatheredRows(g.ndata["feat"], input_nodes)), lab, "mean")
        loss.backward()                                        # plain autograd entry: the fork must join by itself
        grads[fork] = [p.grad.detach().clone() for p in model.parameters()]
        assert ops._SIDE["active"] is False
    ops.FORK_BACKWARD = True
    names = [n for n, _ in model.named_parameters()]
    for n, a, b in zip(names, grads[False], grads[True]):
        if n.startswith("layers.1.fc_self") or n.startswith("layers.1.fc_neigh"):
            assert torch.equal(a, b), n               # upstream of every atomic: bit for bit, whichever stream ran them
        else:
            # downstream of the output layer's atomic max-scatter (dP1) and, for layer 0's fc_pool, of the pool backward's LDS
            # atomics: equal up to their summation order (a missing stream dependency would show as garbage, not as 1e-6)
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_replayed_steps_invalidate_eagerly_built_weight_images():
    """an eager inference pass builds bf16x3 weight images keyed by (data_ptr, version); replayed optimiser steps
    move the weights under both.  A second inference pass must see the NEW weights: compare with a twin trained eagerly."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops, sampling
    from ogl_amd.graphsage import GraphSAGE
    from ogl_amd.graphsage.model import RandomHipSupervisedGraphSage
    feat_size, labels, g, n_classes = _reddit_like_small()
    ops.set_gemm_mode("auto")
    rng = np.random.default_rng(3)
    train_seeds = rng.choice(g.n_present, 4 * 512, replace=False).astype(np.int64)      # n % bs == 0: no eager ragged batch
    eval_seeds = torch.as_tensor(rng.choice(g.n_present, 1024, replace=False).astype(np.int64))
    outs = {}
    for graphs in (False, True):
        torch.manual_seed(5)
        model = GraphSAGE(feat_size, 600, n_classes, 1, F.relu, 0, "pool", edge_feats=0, pool_feats=600).cuda()
        st = RandomHipSupervisedGraphSage(model, 4, 512, labels, 25, cuda=True, batch_full=1024)
        st.use_graphs = graphs
        st.cache_projection = False                     # the per-batch inference path: layer 0 through the dual image product
        st.build_optimizer()
        forms = []
        st.step_hook = lambda info, forms=forms: forms.append(info["form"])

        def infer():
            model.eval()
            sampling.seed(21)
            with torch.no_grad():
                return torch.cat([lg for _, lg in st._inference_batches(g, eval_seeds)]).cpu()
        before = infer()
        model.train()
        sampling.seed(22)
        st._train_batches(g, train_seeds, 512)
        assert forms == (["staged"] * 4 if graphs else ["eager"] * 4), forms
        after = infer()
        outs[graphs] = (before, after)
    assert torch.equal(outs[False][0], outs[True][0])                       # same initial weights, same sampler state
    moved = (outs[False][1] - outs[False][0]).abs().max().item()
    assert moved > 1e-2, moved                                               # four Adam steps changed the logits visibly
    # the replayed model's second pass reflects its four replayed updates (stale images would reproduce `before` in layer 0)
    torch.testing.assert_close(outs[True][1], outs[False][1], rtol=2e-2, atol=0.1 * moved)


def test_fused_inference_batches_equal_per_batch_launches_bit_for_bit():
    """An inference pass against the per-pass tables runs consecutive batches as ONE block (sampling.sample_batches(fuse_rows=...)).
    Every kernel on that path is row-independent — a row's result does not depend on how many rows share the launch — so the fused
    pass reproduces the per-batch pass BIT FOR BIT, and so does any other chunking (what keeps rank-sharded passes bit-identical
    to the one-rank pass: tests/test_gpu_parallel.py)."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops, sampling
    from ogl_amd.graphsage import GraphSAGE
    from ogl_amd.graphsage.model import HipSupervisedGraphSage
    feat_size, labels, g, n_classes = _reddit_like_small()
    ops.set_gemm_mode("auto")
    torch.manual_seed(4)
    model = GraphSAGE(feat_size, 600, n_classes, 1, F.relu, 0, "pool").cuda().eval()
    st = HipSupervisedGraphSage(model, 2, 512, labels, 25, reduction="none", cuda=True, batch_full=256)
    seeds = torch.as_tensor(np.random.default_rng(1).choice(g.n_present, 5 * 256 + 37, replace=False).astype(np.int64))
    outs, chunks = {}, {}
    for rows in (0, 4000, 8000, 12000, 10 ** 9):      # per batch; a few batches per chunk; the whole pass as one chunk
        st.FUSE_INFERENCE_ROWS = rows
        sampling.seed(5)
        with torch.no_grad():
            parts = [(sd.clone(), lg.clone()) for sd, lg in st._inference_batches(g, seeds)]
        assert torch.equal(torch.cat([sd for sd, _ in parts]).cpu(), seeds)
        outs[rows], chunks[rows] = torch.cat([lg for _, lg in parts]), len(parts)
        assert sampling.get_state()["ctr"] == 6                      # one Philox counter per BATCH, whatever the chunking
    assert chunks[0] == 6 and chunks[10 ** 9] == 1 and any(1 < chunks[r] < 6 for r in (4000, 8000, 12000)), chunks
    for r in (4000, 8000, 12000, 10 ** 9):
        assert torch.equal(outs[0], outs[r]), (r, chunks)


@pytest.mark.parametrize("K,N,K2", [(32, 32, 0), (128, 32, 32), (600, 41, 600), (32, 40, 32), (600, 600, 0)])
def test_projection_kernels_are_row_independent(K, N, K2):
    """y[i, :] of every forward projection kernel depends on row i only: the first M rows of a tall product equal the M-row product
    bit for bit, whichever tile configuration the launch picks for its size (on-the-fly, skinny and image kernels)."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    ops.set_gemm_mode("auto")
    torch.manual_seed(0)
    x = torch.randn(40000, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    x2 = torch.randn(12000, K2, device="cuda") if K2 else None
    w2 = torch.randn(N, K2, device="cuda") / K2 ** 0.5 if K2 else None
    rows = torch.randint(0, 40000, (12000,), device="cuda")
    if K2:
        full = ops.linear_fwd(x, w, b, x2=x2, w2=w2, relu=True, x_rows=rows)
    else:
        full = ops.linear_fwd(x[:12000].contiguous(), w, b, relu=True)
    for M in (100, 1000, 2148, 5000, 9216):
        if K2:
            part = ops.linear_fwd(x, w, b, x2=x2[:M].contiguous(), w2=w2, relu=True, x_rows=rows[:M].contiguous())
        else:
            part = ops.linear_fwd(x[:M].contiguous(), w, b, relu=True)
        assert torch.equal(part, full[:M]), (K, N, K2, M)
    if K2 == 0 and K >= 64:
        ximg = ops.x3_split(x[:12000].contiguous(), append_ones=True); wimg = ops.x3_split(w, append_vec=b)
        fimg = ops.linear_fwd_x3(ximg, None, wimg, relu=True)
        for M in (128, 2048, 7000, 9000):
            assert torch.equal(ops.linear_fwd_x3(ximg, None, wimg, relu=True, M=M), fimg[:M]), M


def test_relu_mask_in_the_input_gradient_epilogue_matches_the_separate_pass():
    """ops.FUSE_RELU_BWD (off by default: measured no gain): dh1 leaves the image product already multiplied by [h1 > 0], image
    attached, and the layer-0 combine's backward recognises it — same gradients as the separate ogl_relu_bwd_img pass."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops, sampling
    from ogl_amd.graphsage import GatheredRows, GraphSAGE
    feat_size, labels, g, n_classes = _reddit_like_small()
    ops.set_gemm_mode("auto")
    torch.manual_seed(2)
    model = GraphSAGE(feat_size, 600, n_classes, 1, F.relu, 0, "pool").cuda()
    seeds = torch.as_tensor(np.random.default_rng(0).choice(g.n_present, 512, replace=False).astype(np.int64))
    grads, names = {}, {}
    old = ops.FUSE_RELU_BWD
    try:
        for fuse in (False, True):
            ops.FUSE_RELU_BWD = fuse
            ops.invalidate_weight_images()
            sampling.seed(11)
            (input_nodes, sd, blocks), = list(sampling.NodeDataLoader(g, seeds, sampling.MultiLayerNeighborSampler([25, 25]), batch_size=512))
            for p in model.parameters():
                p.grad = None
            lab = ops.gather_i64(g.ndata["target"], sd)
            ops.profile_start()
            loss = ops.cross_entropy(model(blocks, GatheredRows(g.ndata["feat"], input_nodes)), lab, "mean")
            ops.backward(loss)          # (the unit root gradient keeps dlogits in its padded layout: the output layer's fused backward)
            names[fuse] = [n for n, _, _ in ops.profile_stop()]
            grads[fuse] = [p.grad.detach().clone() for p in model.parameters()]
    finally:
        ops.FUSE_RELU_BWD = old
    assert "ogl_relu_bwd_img" in names[False] and "ogl_relu_bwd_img" not in names[True]
    for a, b in zip(grads[False], grads[True]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)          # (downstream of the output layer's atomic scatter)



def test_adam_with_the_step_count_on_the_device():
    """ogl_adam_step_multi_dev: the step count lives in device memory (k_adam_prepare bumps it and derives the bias corrections in
    front of the update launch) — same results as the host-counted ogl_adam_step_multi at every step, the counter advances by
    exactly one per call, also for parameter sets of more than 32 tensors (several update launches) and for a call with nothing
    to update.  (A one-launch form — every block deriving the corrections itself, the last block bumping the count behind a
    ticket — was parity-green and made the replayed Reddit step 3 % SLOWER: 6 144 blocks each start behind two double-precision
    pow() calls of their first thread; round 3, OGL_ADAM_PREPARE A/B, 1.085 vs 1.045-1.059 ms.)"""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(5)
    for shapes in ([(602, 602), (602,), (600, 602), (41, 600), (41,), (1,), (0,)], [(7, 3)] * 40 + [(1000, 37)], [(0,), (0,)]):
        ps = [torch.randn(*s, device="cuda") for s in shapes]
        qs = [p.clone() for p in ps]
        ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
        ms2, vs2 = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
        step = torch.zeros(1, dtype=torch.int64, device="cuda")
        scal = torch.zeros(2, dtype=torch.float32, device="cuda")
        for t in range(1, 6):
            gs = [torch.randn_like(p) for p in ps]
            ops.adam_step_multi_dev(ps, gs, ms, vs, step, scal)
            ops.adam_step_multi(qs, gs, ms2, vs2, t)
            assert int(step.item()) == t
            want0 = np.float32(1e-3 / (1.0 - 0.9 ** t)); want1 = np.float32(1.0 / np.sqrt(1.0 - 0.999 ** t))
            np.testing.assert_allclose(scal[:2].cpu().numpy(), [want0, want1], rtol=2e-7)
            for a, b in zip(ps, qs):
                np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6, atol=1e-8)
            for a, b in zip(vs, vs2):
                assert torch.equal(a, b)


def test_loss_launch_gathers_its_labels():
    """LazyLabels(table, ids): the grid cross entropy reads label_table[ids[i]] inside its launch (ogl_ce_fwd_bwd_mean_grid with label_ids) —
    same bits as gathering first (ogl_gather_i64 + ogl_ce_fwd_bwd_mean_grid), ids outside the table are rows without a label in both
    forms; the small-batch and 'none' paths materialise the gather themselves."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(3)
    T, B, Cc = 5000, 700, 41
    table = torch.randint(0, Cc, (T,), device="cuda")
    ids = torch.randint(0, T, (B,), device="cuda")
    ids[5] = -1; ids[17] = T + 3                                    # no label
    logits = ops.empty_mat(B, Cc, "cuda").copy_(torch.randn(B, Cc, device="cuda"))
    want = ops.ce_fwd_bwd_mean_grid(logits, ops.gather_i64(table, ids))
    got = ops.ce_fwd_bwd_mean_grid(logits, ops.LazyLabels(table, ids))
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert float(got[1][5]) == 0.0 and float(got[1][17]) == 0.0
    # through the autograd entry points, every batch-size class
    for n in (700, 64, 1):
        lg = logits[:n].clone().requires_grad_(True); lg2 = logits[:n].clone().requires_grad_(True)
        l1 = ops.cross_entropy(lg, ops.gather_i64(table, ids[:n]), "mean"); l1.backward()
        l2 = ops.cross_entropy(lg2, ops.LazyLabels(table, ids[:n]), "mean"); l2.backward()
        assert torch.equal(l1, l2) and torch.equal(lg.grad, lg2.grad)
        m1, r1 = ops.cross_entropy_mean_rows(logits[:n], ops.gather_i64(table, ids[:n]))
        m2, r2 = ops.cross_entropy_mean_rows(logits[:n], ops.LazyLabels(table, ids[:n]))
        assert torch.equal(m1, m2) and torch.equal(r1, r2)
        assert torch.equal(ops.cross_entropy(logits[:n], ops.LazyLabels(table, ids[:n]), "none"),
                           ops.cross_entropy(logits[:n], ops.gather_i64(table, ids[:n]), "none"))


@pytest.mark.parametrize("M,K,K2,N,relu", [(512, 600, 600, 41, False), (832, 500, 500, 32, True), (37, 33, 9, 5, True), (3000, 64, 70, 130, False),
                                            (40, 5000, 4000, 3, False)])
def test_dual_projection_adds_both_biases_in_its_launch(M, K, K2, N, relu):
    """ogl_linear_fwd_dual_bias: fc_self(x) + fc_neigh(x2) with each projection's own bias — (bias + bias2) is formed inside the
    launch, with the rounding of the separate add it replaces: same bits as ogl_linear_fwd on the pre-summed bias (skinny, general
    and split-K shapes)."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(M + N)
    x = ops.empty_mat(M, K, "cuda").copy_(torch.randn(M, K, device="cuda"))
    x2 = ops.empty_mat(M, K2, "cuda").copy_(torch.randn(M, K2, device="cuda"))
    w = torch.randn(N, K, device="cuda") / K ** 0.5; w2 = torch.randn(N, K2, device="cuda") / K2 ** 0.5
    b = torch.randn(N, device="cuda"); b2 = torch.randn(N, device="cuda")
    want = ops.linear_fwd(x, w, b + b2, x2, w2, relu=relu)
    got = ops.linear_fwd(x, w, b, x2, w2, relu=relu, bias2=b2)
    assert torch.equal(want, got)
    ref = x.double() @ w.double().T + x2.double() @ w2.double().T + (b + b2).double()
    if relu:
        ref = ref.clamp_min(0)
    np.testing.assert_allclose(got.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    # through autograd: both biases get their gradient, no ATen add in the forward
    xs = [t.clone().requires_grad_(True) for t in (w, b, w2, b2)]
    ops.profile_start()
    y = ops.linear(x, xs[0], xs[1], x2, xs[2], relu=relu, bias2=xs[3])
    names = [r[0] for r in ops.profile_stop()]
    y.sum().backward()
    assert torch.equal(y.detach(), got) and names == ["ogl_linear_fwd"]
    np.testing.assert_allclose(xs[1].grad.cpu().numpy(), xs[3].grad.cpu().numpy(), rtol=0, atol=0)


def test_small_loss_launch_gathers_labels_and_clears_a_scatter_target():
    """ogl_ce_fwd_bwd_mean_gather (batches up to 1 024 rows, one workgroup): labels read through ids inside the launch, and a small
    pending request_zeroed buffer cleared on the side — same bits as gather + loss + fill as three launches."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    torch.manual_seed(9)
    T, B, Cc = 3000, 64, 7
    table = torch.randint(0, Cc, (T,), device="cuda")
    ids = torch.randint(0, T, (B,), device="cuda"); ids[3] = -5
    logits = ops.empty_mat(B, Cc, "cuda").copy_(torch.randn(B, Cc, device="cuda"))
    want = ops.ce_fwd_bwd_mean(logits, ops.gather_i64(table, ids))
    slot = ops.request_zeroed(832, 32, "cuda")
    slot[0].fill_(7.0)
    got = ops.ce_fwd_bwd_mean(logits, ops.LazyLabels(table, ids))
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert slot[3] is True and float(slot[0].abs().max()) == 0.0
    z = ops.take_zeroed(slot, 832, 32)
    assert z.shape == (832, 32) and float(z.abs().max()) == 0.0
    # a request too large for one workgroup stays pending (its taker clears it)
    big = ops.request_zeroed(5000, 600, "cuda")
    big[0].fill_(1.0)
    ops.ce_fwd_bwd_mean(logits, ops.gather_i64(table, ids))
    assert big[3] is False
    assert float(ops.take_zeroed(big, 5000, 600).abs().max()) == 0.0
    h = ogl_amd._lib.lib()
    assert h.ogl_ce_fwd_bwd_mean_gather(logits.data_ptr(), 8, table.data_ptr(), T, ids.data_ptr(), B, Cc, 1.0, None, None, 0, got[0].data_ptr(),
                                        slot[0].data_ptr(), 1 << 20, None, None, 0.0, 0.0, 0.0, None) == -1
    assert h.ogl_ce_fwd_bwd_mean_gather(logits.data_ptr(), 8, None, T, ids.data_ptr(), B, Cc, 1.0, None, None, 0, got[0].data_ptr(), None, 0, None, None, 0.0, 0.0, 0.0, None) == -1


def test_fill_zero_any_alignment_and_size():
    """ogl_fill_zero: bytes before the first 16-byte boundary, whole 16-byte words, the tail — nothing outside the range is touched."""
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    base = torch.full((1 << 16,), 0x5A, dtype=torch.uint8, device="cuda")
    for off, n in [(0, 0), (0, 1), (3, 5), (1, 15), (16, 16), (7, 4096), (13, 65000), (0, 1 << 16)]:
        base.fill_(0x5A)
        assert ogl_amd._lib.lib().ogl_fill_zero(base.data_ptr() + off, n, None) == 0
        torch.cuda.synchronize()
        want = torch.full_like(base, 0x5A); want[off:off + n] = 0
        assert torch.equal(base, want), (off, n)
    m = ops.empty_mat(333, 70, "cuda", zero=True)
    assert float(m.abs().max()) == 0.0
    assert ogl_amd._lib.lib().ogl_fill_zero(None, 16, None) == -1 and ogl_amd._lib.lib().ogl_fill_zero(base.data_ptr(), -1, None) == -1
