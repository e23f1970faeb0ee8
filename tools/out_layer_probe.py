"""Timing of the output-layer backward launches at the Reddit shape (csrc/out_layer.hip vs the general launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa
from ogl_amd import ops

torch.cuda.set_device(0)
n_src, n_dst, S, K, N = 7060, 512, 25, 600, 41


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


dev = "cuda"
P = ops.empty_mat(n_src, K, dev).copy_(torch.randn(n_src, K, device=dev).clamp(min=0))
idx = torch.randint(0, n_src, (n_dst, S), device=dev, dtype=torch.int32)
neigh, argmax = ops.reduce_fwd(P, idx, "max", want_argmax=True)
h_dst = ops.empty_mat(n_dst, K, dev).copy_(torch.randn(n_dst, K, device=dev))
dy = ops.empty_mat(n_dst, N, dev).copy_(torch.randn(n_dst, N, device=dev))
w_self = torch.randn(N, K, device=dev); w_neigh = torch.randn(N, K, device=dev)
print("out_layer_bwd_inputs (incl. zero fill)  %.1f us" % t(lambda: ops.out_layer_bwd_inputs(dy, w_self, w_neigh, argmax, neigh, n_src)))
print("zero fill alone                         %.1f us" % t(lambda: ops.empty_mat(n_src, K, dev, zero=True)))
print("out_layer_bwd_weights                   %.1f us" % t(lambda: ops.out_layer_bwd_weights(dy, h_dst, neigh)))


def general():
    dn = ops.linear_bwd_input(dy, w_neigh)
    dp = ops.reduce_bwd(dn, None, argmax, "max", n_src, fanout=S, relu_out=neigh)
    dx = ops.linear_bwd_input(dy, w_self)
    return dp, dx
print("general: 2 products + scatter           %.1f us" % t(general))
print("general: 2 weight gradients             %.1f us" % t(lambda: (ops.linear_bwd_weight(dy, h_dst, None, None), ops.linear_bwd_weight(dy, neigh, None, None))))
