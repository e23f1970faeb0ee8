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

# do hipMemsetAsync nodes recorded into a captured graph re-run on replay?
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for nbytes in (4096, 786432, 8 << 20):
    buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0xFF, nbytes, torch._C._cuda_getCurrentRawStream(0))
        y = x + 1.0
    torch.cuda.synchronize()
    after_capture = int((buf == 0xFF).sum())
    res = []
    for _ in range(3):
        buf.zero_(); torch.cuda.synchronize()
        g3.replay(); torch.cuda.synchronize()
        res.append(int((buf == 0xFF).sum()))
    print("memset node %8d bytes: rc=%d, set after capture %d, set after each replay %s" % (nbytes, rc, after_capture, res))
