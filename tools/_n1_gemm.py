import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd
from ogl_amd import ops
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1000
ops.set_gemm_mode("auto")
torch.manual_seed(0)
for M in (3500, 6400, 7054, 8192, 14000):
    for K in (300, 600, 1204):
        x = ops.empty_mat(M, K, "cuda"); x.normal_()
        w = torch.randn(600, K, device="cuda") / 25
        b = torch.randn(600, device="cuda")
        dy = ops.empty_mat(M, 600, "cuda"); dy.normal_()
        t1 = timeit(lambda: ops.linear_fwd(x, w, b, relu=True))
        t2 = timeit(lambda: ops.linear_bwd_input(dy, w))
        print("M=%5d K=%4d  fwd %6.1f us (%5.1f TF)   bwd_input %6.1f us (%5.1f TF)" % (M, K, t1, 2*M*600*K/t1/1e6, t2, 2*M*600*K/t2/1e6), flush=True)
