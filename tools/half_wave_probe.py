"""The max aggregator without argmax over a narrow cached projection table (the arxiv-like priority forward: 169 343 x 128 fp32 = 87 MB,
20 480 destinations x 25 picks per launch) with one row per wave-instruction (k_reduce_fwd_v4) and with two (k_reduce_fwd_max_half),
alone, HIP events, alternating.  Usage (GPU box): python tools/half_wave_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogl_amd  # noqa: E402,F401
from ogl_amd import _lib, ops  # noqa: E402

torch.manual_seed(0)
rng = np.random.default_rng(0)
N, D, n_dst, S = 169343, 128, 20 * 1024 * 7, 25            # (a fused chunk of 20 batches: ~7 layer-0 destinations per seed)
table = ops.empty_mat(N, D, "cuda").copy_(torch.randn(N, D, device="cuda").clamp_min(0))
idx = torch.as_tensor(rng.integers(0, N, size=(n_dst, S))).cuda()
lib = _lib.lib()
alg = n_dst * S * (4 * D + 8) + n_dst * 4 * D
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    for half in (0, 1):
        ops.debug_set("reduce_half", half)
        for _ in range(3):
            ops.reduce_fwd(table, idx, "max")
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ops.reduce_fwd(table, idx, "max")
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("rows per wave-instruction %d: %.1f us per launch, %.0f GB/s of algorithmic bytes (%d destinations x %d picks x %d floats, %d MB table)"
              % (1 + half, 1000 * ms, alg / ms / 1e6, n_dst, S, D, N * D * 4 >> 20), flush=True)
ops.debug_set("reduce_half", 1)
