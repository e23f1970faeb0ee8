import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ogl_amd
from ogl_amd import ops
ops.set_gemm_mode("auto")
torch.manual_seed(0)
for M in (7054, 7065, 7168):
    N, K = 600, 600
    dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    want = dy.double().T @ x.double()
    for wb in (True, False):
        dw, db = ops.linear_bwd_weight_x3(ops.x3_split_t(dy), ops.x3_split_t(x, None, ones_row=True), want_bias=wb)
        print("M=%d want_bias=%s  x3(T,T): rel %.2e" % (M, wb, float((dw.double() - want).norm() / want.norm())), flush=True)
    dy_img = ops.x3_split(dy)
    dw, db = ops.weight_grad(dy, x, None, want_bias=False, dy_img=dy_img)
    print("M=%d weight_grad(dy_img, no x image): rel %.2e" % (M, float((dw.double() - want).norm() / want.norm())))
    ximg = ops.x3_split(x)
    dw, db = ops.weight_grad(dy, x, None, want_bias=False, dy_img=dy_img, x_img=ximg)
    print("M=%d weight_grad(dy_img, x image): rel %.2e" % (M, float((dw.double() - want).norm() / want.norm())))
