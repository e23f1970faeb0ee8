import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import ogl_amd
from ogl_amd import ops, sampling, synthetic
from ogl_amd.graph.dynamic_graph_edge import DynamicGraphEdge
arrays = synthetic.make_arrays("reddit", 1.0)
dyn = DynamicGraphEdge(arrays["snapshots"], set(), device="cuda")
dyn.build(arrays["feat"], arrays["labels"], True, edge_timestamps={"src": arrays["src"], "dst": arrays["dst"]})
g = dyn.get_graph(); g.set_snapshot(g.n_total, len(arrays["src"]))
sampler = sampling.MultiLayerNeighborSampler([25, 25], replace=True)
sampling.seed(1)
rng = np.random.default_rng(0)
seeds = torch.as_tensor(rng.choice(g.n_present, 512 * 3, replace=False))
for input_nodes, sd, blocks in sampling.NodeDataLoader(g, seeds, sampler, batch_size=512):
    li = blocks[0].local_idx.cpu().numpy().astype(np.int64)
    n1, S = li.shape
    n0 = input_nodes.numel()
    d = np.repeat(np.arange(n1), S)
    a = li.reshape(-1)
    ok = a >= 0
    pairs = np.unique((a[ok] >> 5) * n1 + d[ok])
    L = np.bincount(pairs // n1, minlength=(n0 + 31) // 32)
    srcref = np.bincount(a[ok], minlength=n0)
    print("n1", n1, "n0", n0, "groups", len(L), "pairs", len(pairs), "L mean %.1f max %d p99 %d  top10 %s" % (L.mean(), L.max(), np.percentile(L, 99), np.sort(L)[-10:]),
          " per-src refs max", srcref.max(), "argmax group", L.argmax())
