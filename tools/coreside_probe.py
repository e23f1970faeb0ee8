"""Does an HBM-bound launch with a 77 KB slab per block (the layer-0 pool backward's apply: k_pool_values + k_pool_groups) overlap with a
matrix-pipe-bound product when their blocks CAN share a CU (the exact-fp32 k_gemm: 4 waves, ~34 KB of LDS) and when they cannot (the
split-bf16 image product k_gemm_x3p: 12 waves, 123-144 KB)?  Each launch alone, then both at once on two streams; HIP events around the pair.
Usage (GPU box): python tools/coreside_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogl_amd  # noqa: E402,F401
from ogl_amd import ops  # noqa: E402

torch.manual_seed(0)
rng = np.random.default_rng(0)
n_dst, S, D, n_src = 7060, 25, 602, 62495
idx = torch.as_tensor(rng.integers(0, n_src, (n_dst, S)).astype(np.int32)).cuda()
p = ops.empty_mat(n_src, D, "cuda").copy_(torch.randn(n_src, D, device="cuda").clamp_min(0))
out, argmax = ops.reduce_fwd(p, idx, "max", want_argmax=True)
dout = ops.empty_mat(n_dst, D, "cuda").copy_(torch.randn(n_dst, D, device="cuda"))
plan = ops.pool_bwd_x3_plan(argmax, out, idx, n_src, side=False)
M, K, N = 7060 * 3, 600, 600                                   # three n1-row products' worth of rows: ~the apply's duration per arithmetic
x = ops.empty_mat(M, K, "cuda").copy_(torch.randn(M, K, device="cuda"))
w = torch.randn(N, K, device="cuda") / 25
b = torch.randn(N, device="cuda")
side = torch.cuda.Stream()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]


def apply():
    ops.pool_bwd_x3_apply(dout, idx, plan, n_src)


def timed(fa, fb, reps=20):
    for _ in range(3):
        fa(); fb() if fb else None
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        if fb is None:
            fa()
        else:
            here = torch.cuda.Event(); here.record()
            side.wait_event(here)
            with torch.cuda.stream(side):
                fb()
                done = torch.cuda.Event(); done.record()
            fa()
            torch.cuda.current_stream().wait_event(done)
    ev[1].record(); torch.cuda.synchronize()
    return 1000 * ev[0].elapsed_time(ev[1]) / reps


ONLY = os.environ.get("CORESIDE_ONLY")                      # "alone" / "both": one phase per process (for a kernel trace)
for mode, what in (("f32", "exact-fp32 k_gemm (4 waves, ~34 KB LDS: shares a CU with a 77 KB slab)"),
                   ("auto", "split-bf16 k_gemm_x3p (12 waves, 123-144 KB LDS: owns its CU)")):
    ops.set_gemm_mode(mode)
    xi = ops.x3_split(x, append_ones=True) if mode == "auto" else None
    wi = ops.x3_split(w, append_vec=b) if mode == "auto" else None
    gemm = (lambda: ops.linear_fwd_x3(xi, None, wi, relu=True)) if mode == "auto" else (lambda: ops.linear_fwd(x, w, b, relu=True))
    if ONLY == "alone":
        print(mode, "alone", timed(apply, None), timed(gemm, None)); continue
    if ONLY == "both":
        print(mode, "both", timed(apply, gemm)); continue
    ta, tg = timed(apply, None), timed(gemm, None)
    both = timed(apply, gemm)
    print("%s\n   the pool backward's apply alone %.1f us, the product alone %.1f us, both at once %.1f us (sum %.1f, max %.1f): overlap %.0f %% of the shorter"
          % (what, ta, tg, both, ta + tg, max(ta, tg), 100 * (ta + tg - both) / min(ta, tg)), flush=True)
ops.set_gemm_mode("f32")
