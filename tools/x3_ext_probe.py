"""Timing of k_gemm_x3p<..., EXT> options at the n1-row shape (which extension costs what)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa
from ogl_amd import ops

torch.cuda.set_device(0)
ops.set_gemm_mode("auto")
M, F, H = int(sys.argv[1]) if len(sys.argv) > 1 else 14000, 602, 600


def t(fn, n=30):
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
neigh, xd = mat(M, F), mat(M, F)
wn, ws, b = mat(H, F), mat(H, F), torch.randn(H, device="cuda")
S0 = mat(M + 1000, H)
rows = torch.randint(0, M + 1000, (M,), device="cuda")
n_img = ops.x3_split(neigh)
x_img = ops.x3_split(xd, append_ones=True)
w1 = ops.x3_split(wn)
wcat1 = ops.x3_split_cat([(wn, None)])
wcat2 = ops.x3_split_cat([(ws, b), (wn, None)])
print("plain x3p                      %.1f us" % t(lambda: ops.linear_fwd_x3(n_img, None, w1, relu=True)))
print("EXT, zero addend (dense rows)  %.1f us" % t(lambda: ops.linear_fwd_x3_ext(n_img, None, wcat1, add=S0, relu=True)))
print("EXT, gathered addend           %.1f us" % t(lambda: ops.linear_fwd_x3_ext(n_img, None, wcat1, add=S0, add_rows=rows, relu=True)))
print("EXT, output image only         %.1f us" % t(lambda: ops.linear_fwd_x3_ext(n_img, None, wcat1, relu=True, want_image=True, image_append_ones=True)))
print("EXT, addend + image            %.1f us" % t(lambda: ops.linear_fwd_x3_ext(n_img, None, wcat1, add=S0, add_rows=rows, relu=True, want_image=True, image_append_ones=True)))
print("EXT, two-part A                %.1f us" % t(lambda: ops.linear_fwd_x3_ext(x_img, None, wcat2, x2_img=n_img, relu=True)))
print("EXT, two-part A + image        %.1f us" % t(lambda: ops.linear_fwd_x3_ext(x_img, None, wcat2, x2_img=n_img, relu=True, want_image=True, image_append_ones=True)))
print("k_gemm addrows                 %.1f us" % t(lambda: ops.linear_fwd_addrows(neigh, wn, S0, add_rows=rows, relu=True)))
print("k_gemm dual                    %.1f us" % t(lambda: ops.linear_fwd(xd, ws, b, x2=neigh, w2=wn, relu=True)))
print("split(out)                     %.1f us" % t(lambda: ops.x3_split(neigh, append_ones=True)))
