"""Where a 256 x 128 x 32 step of the layer-0 image products goes (k_gemm_x3p<4, 2, 2, 2, 2, ..., EA>): per-phase cycle sums from a
DIAGNOSTIC build of the library (tools/build_variant.py phase -DOGL_X3_PHASE_STAMPS), for the Reddit-shaped forward product
(gathered table rows) and the k-major weight gradient.  Usage (GPU box): python tools/x3_phase_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ogl_amd  # noqa: E402,F401
from ogl_amd import _lib, ops  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "debug", "libogl_hip_%s.so" % os.environ.get("X3_VARIANT", "phase"))
ops.set_gemm_mode("auto")
torch.manual_seed(0)
dev = "cuda"
M, N, K, T = 62600, 600, 602, 232965
table = ops.empty_mat(T, K, dev).copy_(torch.randn(T, K, device=dev))
x_img = ops.x3_split(table, append_ones=True)
rows = torch.randperm(T, device=dev)[:M].contiguous()
wp = torch.randn(K, K, device=dev) / 25
w_img = ops.x3_split(wp, append_vec=torch.randn(K, device=dev))
dy = ops.empty_mat(M, N, dev).copy_(torch.randn(M, N, device=dev) * (torch.rand(M, N, device=dev) < 0.11))
G = (M + 31) // 32
dyT = ops.x3_split_t(dy, interleave=G)
del dy


def fwd():
    return ops.linear_fwd_x3(x_img, rows, w_img, relu=True)


def dw():
    return ops.linear_bwd_weight_x3k(dyT, x_img, M, K, x_rows=rows, x_nrows=T, interleave=G, want_bias=True, want_bias2=True)


def probe(name, run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record(); run(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    stamps = torch.zeros(1024 + 256 * 32, dtype=torch.int64, device=dev)
    _lib.lib().ogl_x3_debug_stamps(stamps.data_ptr(), 0)
    run(); torch.cuda.synchronize()
    _lib.lib().ogl_x3_debug_stamps(None, 0)
    st = stamps.cpu()
    blk = st[:1024].view(256, 4).double()
    ok = blk[:, 3] > blk[:, 1]
    ghz = ((blk[ok, 2] - blk[ok, 0]) / (blk[ok, 3] - blk[ok, 1]) * 0.1).median().item()
    ph = st[1024:].view(256, 4, 8).double()[ok]
    print("%s (%s): best %.1f us, median %.1f us of 10 launches; %d blocks ran; in-kernel clock %.2f GHz" %
          (name, _lib.lib().ogl_x3_last_kernel().decode(), 1e3 * min(ts), 1e3 * sorted(ts)[5], int(ok.sum()), ghz))
    for role, label, names in ((0, "multiplier wave 0", ("wait at the opening barrier", "fragment loads + column block 0", "wait at `mid`",
                                                          "between steps (epilogues)", "column blocks 1-3")),
                               (1, "multiplier wave 4", ("wait at the opening barrier", "fragment loads + column block 0", "wait at `mid`",
                                                          "between steps (epilogues)", "column blocks 1-3")),
                               (2, "mover wave 8", ("wait at the opening barrier", "issue B(n+1)", "wait at `mid`", "issue A(n+2)",
                                                     "wait for the landing"))):
        p = ph[:, role, :]
        steps = p[:, 5].clamp(min=1)
        per = p[:, :5] / steps[:, None]
        tot = per.sum(1)
        print("  %-18s steps/block %.0f, cycles per step %.0f (= %.2f us at the in-kernel clock)" % (label, steps.median().item(), tot.median().item(),
                                                                                                   tot.median().item() / ghz / 1e3))
        for i, nme in enumerate(names):
            print("      %-34s %7.0f cycles (%4.1f %%)   [min %.0f max %.0f over blocks]" %
                  (nme, per[:, i].median().item(), 100 * per[:, i].median().item() / tot.median().item(), per[:, i].min().item(), per[:, i].max().item()))


probe("forward fc_pool0 [62600 x 603] x [603 -> 602], gathered rows", fwd)
probe("weight gradient dW_pool0 (k-major B)", dw)
