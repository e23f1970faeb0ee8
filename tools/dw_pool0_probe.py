"""The Reddit-shaped layer-0 weight gradient (k-major, 256 x 128 tiles) alone: launch time by HIP events and per-BLOCK durations
(ogl_x3_debug_stamps) — which blocks set the launch's length.  The plan's switches are read once per process
(OGL_BWWK_UNEVEN, OGL_BWWK_UNEVEN_F, OGL_BWWK_SKIP_PAD): run once per setting.  Usage (GPU box): python tools/dw_pool0_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogl_amd  # noqa: E402,F401
from ogl_amd import _lib, ops  # noqa: E402

ops.set_gemm_mode("auto")
torch.manual_seed(0)
dev = "cuda"
M, N, K, T = 62600, 600, 602, 232965
table = ops.empty_mat(T, K, dev).copy_(torch.randn(T, K, device=dev))
x_img = ops.x3_split(table, append_ones=True)
rows = torch.randperm(T, device=dev)[:M].contiguous()
dy = ops.empty_mat(M, N, dev).copy_(torch.randn(M, N, device=dev) * (torch.rand(M, N, device=dev) < 0.11))
G = (M + 31) // 32
dyT = ops.x3_split_t(dy, interleave=G)
del dy


def run():
    return ops.linear_bwd_weight_x3k(dyT, x_img, M, K, x_rows=rows, x_nrows=T, interleave=G, want_bias=True, want_bias2=True)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(10):
    e0.record(); run(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
print("env %s: product + reduction launch, best %.3f ms, median %.3f ms" % ({k: v for k, v in os.environ.items() if k.startswith("OGL_BWWK") or k.startswith("OGL_X3")},
                                                                            min(ts), sorted(ts)[len(ts) // 2]))
stamps = torch.zeros(1024, dtype=torch.int64, device=dev)
_lib.lib().ogl_x3_debug_stamps(stamps.data_ptr(), 0)
run(); torch.cuda.synchronize()
_lib.lib().ogl_x3_debug_stamps(None, 0)
st = stamps.cpu().view(256, 4).double()
ok = st[:, 3] > st[:, 1]
t0 = st[ok][:, 1].min()
dur = (st[:, 3] - st[:, 1]) * 0.01          # us (s_memrealtime: 100 MHz)
start = (st[:, 1] - t0) * 0.01
order = sorted((float(dur[b]), b) for b in range(256) if ok[b])
print("blocks that ran: %d; duration us: min %.1f median %.1f max %.1f; latest start %.1f us" %
      (int(ok.sum()), order[0][0], order[len(order) // 2][0], order[-1][0], float(start[ok].max())))
print("ten longest (us, block = 8 * slot + xcd):", [(round(d, 1), b) for d, b in order[-10:]])
print("ten shortest:", [(round(d, 1), b) for d, b in order[:10]])
# by slot within the XCD chunk (the uneven plan puts whole full slabs first, then whole short slabs, then the dealt rest)
by_slot = {}
for d, b in order:
    by_slot.setdefault(b >> 3, []).append(d)
print("mean duration by slot:", " ".join("%d:%.0f" % (s, sum(v) / len(v)) for s, v in sorted(by_slot.items())))
print("duration us [slot][xcd]:")
for s in range(32):
    print("%2d " % s + " ".join("%6.1f" % float(dur[8 * s + x]) for x in range(8)))
