"""Micro-benchmark of the C-ABI GEMM entry points at the Reddit-rung shapes (HIP-event timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa
from ogl_amd import ops


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    if len(sys.argv) > 1:
        ops.set_gemm_mode(sys.argv[1])
        print("gemm mode", ops.get_gemm_mode())
    torch.manual_seed(0)
    n0, n1, B, F, H, C = 63000, 7071, 512, 602, 600, 41
    T = 232965
    table = ops.empty_mat(T, F, "cuda"); table.normal_()
    rows0 = torch.randint(0, T, (n0,), device="cuda")
    wp = torch.randn(F, F, device="cuda") / 25
    ws = torch.randn(H, F, device="cuda") / 25
    wn = torch.randn(H, F, device="cuda") / 25
    b = torch.randn(F, device="cuda")
    p0 = ops.linear_fwd(table, wp, b, relu=True, x_rows=rows0)
    dp0 = ops.empty_mat(n0, F, "cuda"); dp0.normal_()
    x1 = ops.empty_mat(n1, F, "cuda"); x1.normal_()
    dy1 = ops.empty_mat(n1, H, "cuda"); dy1.normal_()
    res = []
    def rep(name, ms, flops):
        res.append((name, ms, flops / ms / 1e9))
        print("%-44s %8.3f ms  %7.1f TFLOP/s" % (name, ms, flops / ms / 1e9), flush=True)
    rep("fwd pool0 [n0,602]x[602,602] rows+relu", timeit(lambda: ops.linear_fwd(table, wp, b, relu=True, x_rows=rows0)), 2.0 * n0 * F * F)
    rep("bwd_weight pool0 (mask, rows)", timeit(lambda: ops.linear_bwd_weight(dp0, table, p0, rows0)), 2.0 * n0 * F * F)
    rep("bwd_weight pool0 (no mask, rows)", timeit(lambda: ops.linear_bwd_weight(dp0, table, None, rows0)), 2.0 * n0 * F * F)
    rep("bwd_input [n0,602]x[602,602]", timeit(lambda: ops.linear_bwd_input(dp0, wp, p0)), 2.0 * n0 * F * F)
    rep("fwd dual [n1,602+602]->600 relu", timeit(lambda: ops.linear_fwd(x1, ws, b[:H], x2=x1, w2=wn, relu=True)), 2.0 * n1 * H * 2 * F)
    rep("fwd [n1,600]x[600,600] relu", timeit(lambda: ops.linear_fwd(dy1, wn[:, :H], b[:H], relu=True)), 2.0 * n1 * H * H)
    rep("bwd_weight [n1] 600x602", timeit(lambda: ops.linear_bwd_weight(dy1, x1)), 2.0 * n1 * H * F)
    rep("bwd_input [n1,600]->602", timeit(lambda: ops.linear_bwd_input(dy1, ws)), 2.0 * n1 * H * F)
    xb = ops.empty_mat(B, H, "cuda"); xb.normal_()
    wc = torch.randn(C, H, device="cuda")
    rep("fwd dual [512,600+600]->41", timeit(lambda: ops.linear_fwd(xb, wc, b[:C], x2=xb, w2=wc)), 2.0 * B * C * 2 * H)
    dyc = ops.empty_mat(B, C, "cuda"); dyc.normal_()
    rep("bwd_weight [512] 41x600", timeit(lambda: ops.linear_bwd_weight(dyc, xb)), 2.0 * B * C * H)
    rep("bwd_input [512,41]->600", timeit(lambda: ops.linear_bwd_input(dyc, wc)), 2.0 * B * C * H)
    # reference point: rocBLAS through torch (not part of the product path)
    xm = table[rows0].contiguous()
    rep("torch (rocBLAS/hipBLASLt) fwd pool0", timeit(lambda: torch.nn.functional.linear(xm, wp, b)), 2.0 * n0 * F * F)
    rep("torch dW pool0", timeit(lambda: dp0.T @ xm), 2.0 * n0 * F * F)


if __name__ == "__main__":
    main()
