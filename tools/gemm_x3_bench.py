"""Micro-benchmark of the pre-split (bf16x3 image) projections against the on-the-fly x6 GEMM at the layer-0 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa
from ogl_amd import ops


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ops.set_gemm_mode("auto")
    torch.manual_seed(0)
    n0, F = 62750, 602
    T = 232965
    table = ops.empty_mat(T, F, "cuda"); table.normal_()
    rows0 = torch.randint(0, T, (n0,), device="cuda")
    wp = torch.randn(F, F, device="cuda") / 25
    b = torch.randn(F, device="cuda")
    dp0 = ops.empty_mat(n0, F, "cuda"); dp0.normal_()
    fl = 2.0 * n0 * F * F
    if os.environ.get("X3_ZERO"):   # DVFS probe: same launches on all-zero operands (the clock the chip holds differs)
        table.zero_(); wp.zero_(); b.zero_(); dp0.zero_()

    def rep(name, ms, flops=None):
        print("%-56s %8.3f ms  %s" % (name, ms, "%7.1f TFLOP/s" % (flops / ms / 1e9) if flops else ""), flush=True)
    t_img = ops.x3_split(table, append_ones=True)
    w_img = ops.x3_split(wp, append_vec=b)
    rep("x3_split table [%d,602] (once per dataset)" % T, timeit(lambda: ops.x3_split(table), 3))
    rep("x3_split weight [602,602] (+bias)", timeit(lambda: ops.x3_split(wp, append_vec=b)))
    rep("fwd pool0 on-the-fly x6 (rows+relu)", timeit(lambda: ops.linear_fwd(table, wp, b, relu=True, x_rows=rows0)), fl)
    rep("fwd pool0 x3 images (rows+relu)", timeit(lambda: ops.linear_fwd_x3(t_img, rows0, w_img, relu=True)), fl)
    rows_c = torch.arange(n0, device="cuda")
    rows_s = torch.sort(rows0).values
    rep("fwd pool0 x3, rows = arange (gather path, contiguous)", timeit(lambda: ops.linear_fwd_x3(t_img, rows_c, w_img, relu=True)), fl)
    rep("fwd pool0 x3, rows = sorted random", timeit(lambda: ops.linear_fwd_x3(t_img, rows_s, w_img, relu=True)), fl)
    rep("fwd pool0 x3, no gather [n0 rows]", timeit(lambda: ops.linear_fwd_x3(t_img, None, w_img, relu=True, M=n0)), fl)
    rep("fwd whole table x3 [%d]" % T, timeit(lambda: ops.linear_fwd_x3(t_img, None, w_img, relu=True), 5), 2.0 * T * F * F)
    rep("fwd whole table on-the-fly x6", timeit(lambda: ops.linear_fwd(table, wp, b, relu=True), 5), 2.0 * T * F * F)
    rep("transpose dy + transpose x[rows] (fp32)", timeit(lambda: (ops.transpose(dp0), ops.transpose(table, rows0))))
    rep("x3_split_t dy + x3_split_t x[rows] (+ones)", timeit(lambda: (ops.x3_split_t(dp0), ops.x3_split_t(table, rows0, ones_row=True))))
    dyT, xT = ops.transpose(dp0), ops.transpose(table, rows0)
    rep("bwd_weight_t pool0 on-the-fly x6", timeit(lambda: ops.linear_bwd_weight_t(dyT, xT)), fl)
    dyI, xI = ops.x3_split_t(dp0), ops.x3_split_t(table, rows0, ones_row=True)
    rep("bwd_weight pool0 x3 images", timeit(lambda: ops.linear_bwd_weight_x3(dyI, xI)), fl)
    n1, H = 7054, 600
    x1 = ops.empty_mat(n1, F, "cuda"); x1.normal_()
    ws_ = torch.randn(H, F, device="cuda") / 25
    rep("fwd [n1,602]->600 on-the-fly x6", timeit(lambda: ops.linear_fwd(x1, ws_, b[:H], relu=True)), 2.0 * n1 * H * F)
    xi1 = ops.x3_split(x1, append_ones=True); wi1 = ops.x3_split(ws_, append_vec=b[:H])
    rep("fwd [n1,602]->600 x3", timeit(lambda: ops.linear_fwd_x3(xi1, None, wi1, relu=True)), 2.0 * n1 * H * F)
    rep("x3_split [n1,602]", timeit(lambda: ops.x3_split(x1)))


def clock_probe():
    """In-kernel shader clock of k_gemm_x3 (MI355X_MICROARCH.md, DVFS give-back item 6): >= 2 s of back-to-back launches,
    then d(s_memtime) / d(s_memrealtime) x 100 MHz per block of the last launch; median over blocks."""
    import time
    from ogl_amd import _lib
    ops.set_gemm_mode("auto")
    torch.manual_seed(0)
    T, F = 232965, 602
    table = ops.empty_mat(T, F, "cuda"); table.normal_()
    wp = torch.randn(F, F, device="cuda") / 25
    if os.environ.get("X3_ZERO"):
        table.zero_(); wp.zero_()
    t_img = ops.x3_split(table, append_ones=True)
    w_img = ops.x3_split(wp, append_vec=torch.zeros(F, device="cuda"))
    probe = 0
    stamps = torch.zeros(1024 + 256 * 8 * 4, dtype=torch.int64, device="cuda")
    _lib.lib().ogl_x3_debug_stamps(stamps.data_ptr(), probe)
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.5:
        for _ in range(20):
            ops.linear_fwd_x3(t_img, None, w_img, relu=True)
        torch.cuda.synchronize(); n += 20
    ms = (time.time() - t0) / n * 1e3
    _lib.lib().ogl_x3_debug_stamps(None, 0)
    ph = stamps.cpu()[1024:].view(256, 8, 4).double()
    st = stamps.cpu()[:1024].view(256, 4).double()
    st = st[st[:, 3] > st[:, 1]]
    ghz = ((st[:, 2] - st[:, 0]) / (st[:, 3] - st[:, 1]) * 0.1)
    cyc = (st[:, 2] - st[:, 0]).median().item()
    fl = 2.0 * T * F * F
    print("whole-table forward [%d x %d -> %d], %s operands: %.3f ms/launch wall, in-kernel clock median %.3f GHz (min %.3f, max %.3f), "
          "%.0f k cycles per block, %.1f TFLOP/s" % (T, F, F, "ZERO" if os.environ.get("X3_ZERO") else "random", ms, ghz.median().item(),
                                                     ghz.min().item(), ghz.max().item(), cyc / 1e3, fl / ms / 1e9), flush=True)
    # matrix-pipe occupancy at the held clock: 6 MFMA terms x padded tiles x 16 cycles per 16x16x32 MFMA per SIMD
    mfma_cycles = 6.0 * (-(-T // 256) * 256) * 640 * 608 / (16 * 16 * 32) * 16 / (256 * 4)
    print("  MFMA cycles needed per SIMD %.0f k -> matrix pipe busy %.1f %% of the in-kernel cycles" % (mfma_cycles / 1e3, 100 * mfma_cycles / cyc))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "clock":
        clock_probe(); sys.exit(0)
    main()
