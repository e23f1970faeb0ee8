"""Run one GEMM shape repeatedly (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa
from ogl_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
torch.manual_seed(0)
n0, F, T = 63000, 602, 232965
table = ops.empty_mat(T, F, "cuda"); table.normal_()
rows0 = torch.randint(0, T, (n0,), device="cuda")
wp = torch.randn(F, F, device="cuda") / 25
b = torch.randn(F, device="cuda")
p0 = ops.linear_fwd(table, wp, b, relu=True, x_rows=rows0)
dp0 = ops.empty_mat(n0, F, "cuda"); dp0.normal_()
xm = ops.gather_rows(table, rows0)
for _ in range(5):
    if which == "fwd":
        ops.linear_fwd(table, wp, b, relu=True, x_rows=rows0)
    elif which == "bww":
        ops.linear_bwd_weight(dp0, table, p0, rows0)
    elif which == "bwwnm":
        ops.linear_bwd_weight(dp0, table, None, rows0)
    elif which == "bwwnr":
        ops.linear_bwd_weight(dp0, xm, None, None)
    else:
        ops.linear_bwd_input(dp0, wp, p0)
torch.cuda.synchronize()
