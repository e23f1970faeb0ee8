"""Times the layer-0 forward product (gathered table rows [62 495, 602] x the weight image [602, 603]) on each pinned tile of
launch_x3 (0-4: k_gemm_x3p; 5 / 6: the B-direct kernel k_gemm_x3bd), alone, HIP events.  usage: python tools/bd_probe.py [M K N]"""
import os
import sys
import torch
sys.path.insert(0, ".")
import __graft_entry__  # noqa: F401  (registers the package alias)
from ogl_amd import _lib, ops

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (62495, 602, 602)
T = 232965
torch.manual_seed(0)
table = torch.randn(T, K, device="cuda")
rows = torch.randint(0, T, (M,), device="cuda")
w = torch.randn(N, K, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda")
xi, wi = ops.x3_split(table, append_ones=True), ops.x3_split(w, append_vec=b)
ref = None
for cfg in [int(c) for c in os.environ.get("BD_PROBE_CFGS", "0,5,6,0,5,6,2,1").split(",")]:
    assert _lib.lib().ogl_x3_debug_tile(cfg) == 0
    y = ops.linear_fwd_x3(xi, rows, wi, relu=True, x_nrows=T)
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    same = bool(torch.equal(ref, y))
    for _ in range(3):
        ops.linear_fwd_x3(xi, rows, wi, relu=True, x_nrows=T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.linear_fwd_x3(xi, rows, wi, relu=True, x_nrows=T)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    name = _lib.lib().ogl_x3_last_kernel()
    name = name.decode() if isinstance(name, bytes) else name
    print(f"cfg {cfg}  {us:8.1f} us  {2 * 6 * M * (K + 1) * N / us / 1e6:7.1f} TFLOP/s(x6)  bits_equal={same}  {name}", flush=True)
_lib.lib().ogl_x3_debug_tile(-1)
