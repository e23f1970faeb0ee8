"""Probe: is a row's result independent of how many rows share the launch?  (What keeps rank-sharded inference passes bit-identical
to the one-rank pass whatever the chunking of their batches.)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogl_amd
from ogl_amd import ops
ops.set_gemm_mode("auto")
torch.manual_seed(0)
for K, N, K2 in ((32, 32, 0), (128, 32, 32), (600, 41, 600), (32, 40, 32), (600, 600, 0)):
    x = torch.randn(70000, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    x2 = torch.randn(20000, K2, device="cuda") if K2 else None
    w2 = torch.randn(N, K2, device="cuda") / K2 ** 0.5 if K2 else None
    rows = torch.randint(0, 70000, (20000,), device="cuda")
    full = ops.linear_fwd(x, w, b, x2=x2, w2=w2, relu=True, x_rows=rows if K2 else None) if K2 else ops.linear_fwd(x[:20000].contiguous(), w, b, relu=True)
    for M in (100, 1000, 2148, 5000, 9216):
        if K2:
            part = ops.linear_fwd(x, w, b, x2=x2[:M].contiguous(), w2=w2, relu=True, x_rows=rows[:M].contiguous())
        else:
            part = ops.linear_fwd(x[:M].contiguous(), w, b, relu=True)
        print("K %d N %d K2 %d M %5d equal: %s  %.2e" % (K, N, K2, M, torch.equal(part, full[:M]), float((part - full[:M]).abs().max())))
