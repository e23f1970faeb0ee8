"""Loader-phase kernels at the Reddit rung, timed with HIP events: sampler + block build of a 50-batch loader (and a 20-batch one),
hash table vs direct-address table.  Usage (GPU box): python tools/block_build_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogl_amd  # noqa: E402,F401
from ogl_amd import ops, sampling, synthetic  # noqa: E402

arrays = synthetic.make_arrays("reddit", 1.0)
from ogl_amd.graph.dynamic_graph_edge import DynamicGraphEdge  # noqa: E402
dyn = DynamicGraphEdge(arrays["snapshots"], set(), device="cuda")
dyn.build(arrays["feat"], arrays["labels"], True, edge_timestamps={"src": arrays["src"], "dst": arrays["dst"]})
g = dyn.get_graph()
g.set_snapshot(g.n_total, len(arrays["src"]))           # the bench's graph: the last snapshot of the Reddit-like edge stream
rng = np.random.default_rng(0)
smp = sampling.MultiLayerNeighborSampler([25, 25], replace=True, return_eids=True)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for nb in (50, 20):
    seeds = torch.as_tensor(rng.choice(g.n_present, nb * 512, replace=False).astype(np.int64)).cuda()
    batches = [seeds[i * 512:(i + 1) * 512] for i in range(nb)]
    counts = [512] * nb
    starts = [512 * i for i in range(nb)]
    ctrs = list(range(nb))
    p1 = ops.sample_layer_batched(g.handle, seeds, starts, counts, 25, 1, ctrs, 1)
    src1, n1, l1 = ops.build_block_batched_async(seeds, starts, counts, p1, n_ids=g.handle.n)
    n1h = n1.cpu().tolist()
    st0 = [r * 26 for r in starts]
    p0 = ops.sample_layer_batched(g.handle, src1, st0, n1h, 25, 1, ctrs, 0)
    print("nb %d: n1 total %d" % (nb, sum(n1h)))
    print("  sample L0 %.3f ms" % timed(lambda: ops.sample_layer_batched(g.handle, src1, st0, n1h, 25, 1, ctrs, 0)))
    from ogl_amd import _lib
    for direct, lds in ((False, 0), (True, 0), (True, 1)):
        ops.BLOCK_DIRECT = direct
        ops.debug_set("block_min_lds", lds)
        t = timed(lambda: ops.build_block_batched_async(src1, st0, n1h, p0, n_ids=g.handle.n))
        t1 = timed(lambda: ops.build_block_batched_async(seeds, starts, counts, p1, n_ids=g.handle.n))
        print("  build L0 %s %.3f ms   L1 %.3f ms" % (("direct, minima in LDS" if lds else "direct, global atomics") if direct else "hash", t, t1))
    ops.BLOCK_DIRECT = True
    t0 = time.perf_counter()
    for _ in range(5):
        out = smp.sample_batches(g, batches)
        first = out[0]
    torch.cuda.synchronize()
    print("  whole loader (2 read-backs, first batch handed out): %.3f ms wall" % ((time.perf_counter() - t0) / 5 * 1e3))
