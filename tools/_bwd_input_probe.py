"""dX = dY . W for the n1-row shapes: ogl_linear_bwd_input (B operand read across its rows) against
transpose(W) + ogl_linear_fwd (both operands reduction-contiguous)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa
from ogl_amd import ops


def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1000


ops.set_gemm_mode("auto")
torch.manual_seed(0)
for M, N, K in ((7054, 600, 602), (7054, 600, 600), (512, 41, 600), (14000, 600, 602)):
    dy = ops.empty_mat(M, N, "cuda"); dy.normal_()
    w = torch.randn(N, K, device="cuda") / 25
    a = ops.linear_bwd_input(dy, w)
    b = ops.linear_fwd(dy, ops.transpose(w), None)
    err = (a - b).abs().max().item()
    t1 = timeit(lambda: ops.linear_bwd_input(dy, w))
    t2 = timeit(lambda: ops.linear_fwd(dy, ops.transpose(w), None))
    t3 = timeit(lambda: ops.transpose(w))
    print("M=%5d N=%3d K=%3d  bwd_input %6.1f us   transpose+fwd %6.1f us (transpose %4.1f)   max|diff| %.2e" % (M, N, K, t1, t2, t3, err), flush=True)
