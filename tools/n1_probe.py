"""Timing probe for the n1-row products: on-the-fly k_gemm vs the image kernel k_gemm_x3p at the Reddit
rung's shapes, images built outside the timed loop (what a fused producer would give)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa
from ogl_amd import ops

torch.cuda.set_device(0)
ops.set_gemm_mode("auto")
n1, F, H = 7060, 602, 600


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


mat = lambda r, c: ops.empty_mat(r, c, "cuda").copy_(torch.randn(r, c, device="cuda"))
h1, neigh, xdst = mat(n1, H), mat(n1, F), mat(n1, F)
wp1, bp1 = mat(H, H), torch.randn(H, device="cuda")
ws, wn, b0 = mat(H, F), mat(H, F), torch.randn(H, device="cuda")
print("F4  k_gemm  [7060,600]->600        %.1f us" % t(lambda: ops.linear_fwd(h1, wp1, bp1, relu=True)))
h1_img = ops.x3_split(h1, append_ones=True)
wp1_img = ops.x3_split(wp1, append_vec=bp1)
print("F4  x3p     images prebuilt        %.1f us   (+ split(h1) %.1f us, + split(W) %.1f us)" % (
    t(lambda: ops.linear_fwd_x3(h1_img, None, wp1_img, relu=True)), t(lambda: ops.x3_split(h1, append_ones=True)),
    t(lambda: ops.x3_split(wp1, append_vec=bp1))))
print("F3  k_gemm  dual [7060,602+602]->600 %.1f us" % t(lambda: ops.linear_fwd(xdst, ws, b0, x2=neigh, w2=wn, relu=True)))
cat = ops.empty_mat(n1, 2 * F, "cuda"); cat.copy_(torch.cat([xdst, neigh], 1))
wcat = ops.empty_mat(H, 2 * F, "cuda"); wcat.copy_(torch.cat([ws, wn], 1))
cat_img = ops.x3_split(cat, append_ones=True); wcat_img = ops.x3_split(wcat, append_vec=b0)
print("F3  x3p     K-concatenated images  %.1f us" % t(lambda: ops.linear_fwd_x3(cat_img, None, wcat_img, relu=True)))
dy = mat(n1, H)
print("B8  k_gemm  via transpose(W)+fwd   %.1f us" % t(lambda: ops.linear_bwd_input(dy, wn)))
dy_img = ops.x3_split(dy); wnT_img = ops.x3_split(ops.transpose(wn))
print("B8  x3p     images prebuilt        %.1f us   (+ transpose+split(W) %.1f us)" % (
    t(lambda: ops.linear_fwd_x3(dy_img, None, wnT_img)), t(lambda: ops.x3_split(ops.transpose(wn)))))
