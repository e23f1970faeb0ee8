"""The 'meanpool' first layer's pooled-row gradient at the Reddit rung's shape (7 060 destinations x 25 picks over 62 495 sources, 600
columns), alone, HIP events, alternating: the row-wise segmented backward writing the row-major image (k_seg_reduce + k_seg_fixup) and
the group-wise one writing the transposed group-major image (k_seg_groups).  Usage (GPU box): python tools/seg_t_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogl_amd  # noqa: E402,F401
from ogl_amd import ops  # noqa: E402

torch.manual_seed(0)
rng = np.random.default_rng(0)
n_dst, S, D, n_src = 7060, 25, 600, 62495
# a block builder's numbering: the destinations are the first n_dst sources and frequently sampled vertices come early
w = 1.0 / (1.0 + np.arange(n_src)) ** 0.35
idx = rng.choice(n_src, size=(n_dst, S), p=w / w.sum()).astype(np.int32)
idx_d = torch.as_tensor(idx).cuda()
p = ops.empty_mat(n_src, D, "cuda").copy_(torch.randn(n_src, D, device="cuda").clamp_min(0))
dout = ops.empty_mat(n_dst, D, "cuda").copy_(torch.randn(n_dst, D, device="cuda"))
plan = ops.reduce_bwd_seg_plan(idx_d, D, n_src, side=False, groups=True)
alg = n_dst * S * 4 * D + n_src * 4 * D
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cnt = np.bincount(idx.reshape(-1), minlength=n_src)
print("edges %d, sources with an edge %d, max edges into one source %d" % (idx.size, int((cnt > 0).sum()), int(cnt.max())))
forms = {"rows (k_seg_reduce + k_seg_fixup, row-major image)": lambda: ops.reduce_bwd_seg_apply(dout, idx_d, plan, "mean", mask=p, want_out=False, want_image=True),
         "groups (k_seg_groups, transposed image)": lambda: ops.reduce_bwd_seg_apply_t(dout, idx_d, plan, "mean", mask=p)}
for rep in range(3):
    for name, fn in forms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-55s %.1f us per launch, %.0f GB/s of algorithmic bytes" % (name, 1000 * ms, alg / ms / 1e6), flush=True)
