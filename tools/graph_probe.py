"""Feasibility probe: torch.cuda.graph capture of C-ABI launches (ctypes, torch's current stream) + per-node replay cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa
from ogl_amd import ops

torch.cuda.set_device(0)
x = ops.empty_mat(4096, 128, "cuda").copy_(torch.randn(4096, 128, device="cuda"))
w = torch.randn(32, 128, device="cuda")
b = torch.randn(32, device="cuda")
idx = torch.randint(0, 4096, (800, 25), dtype=torch.int32, device="cuda")


def body():
    y = ops.linear_fwd(x, w, b, relu=True)
    out, _ = ops.reduce_fwd(y, idx, "max")
    return out


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
ref = body().clone()
g.replay(); torch.cuda.synchronize()
print("replay equals eager:", torch.equal(out, ref))
x.copy_(torch.randn(4096, 128, device="cuda"))
ref2 = body().clone()
g.replay(); torch.cuda.synchronize()
print("replay sees new static input:", torch.equal(out, ref2))

# per-node cost: N tiny kernels
for n in (10, 40):
    def many():
        y = x
        for _ in range(n):
            y = ops.relu_bwd(y, x)
        return y
    many(); torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        o = many()
    for name, fn in (("eager", many), ("graph", g2.replay)):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50):
            fn()
        th = time.perf_counter() - t
        torch.cuda.synchronize(); tt = time.perf_counter() - t
        print("%d kernels %s: host %.1f us/iter, total %.1f us/iter (%.2f us/kernel)" % (n, name, th / 50 * 1e6, tt / 50 * 1e6, tt / 50 / n * 1e6))
