"""Record-fed layer-0 weight gradient vs the round-4 pair (dense dP^T image + k-major product) on a real Reddit-rung block:
HIP-event times of both, alone (no side branch), same inputs; max deviation between the two results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
seeds = torch.as_tensor(rng.choice(g.n_present, 512, replace=False))
(input_nodes, sd, blocks), = list(sampling.NodeDataLoader(g, seeds, sampler, batch_size=512))
idx = blocks[0].local_idx
n1, S = idx.shape
n0 = int(input_nodes.numel())
feat = g.ndata["feat"]
D = K = feat.shape[1]
print("n1", n1, "n0", n0, "D", D, flush=True)
ops.set_gemm_mode("auto")
rimg = ops._static_image(feat)
torch.manual_seed(0)
p = ops.empty_mat(n0, D, "cuda"); p.normal_().clamp_(min=0)
out, argmax = ops.reduce_fwd(p, idx, "max", want_argmax=True)
dout = ops.empty_mat(n1, D, "cuda"); dout.normal_()
G = (n0 + 31) // 32


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return r, float(np.median(ts)), float(np.min(ts))


plan = ops.pool_bwd_x3_plan(argmax, out, idx, n0, side=False)
torch.cuda.synchronize()


def old():
    dyT = ops.pool_bwd_x3_apply(dout, idx, plan, n0)
    return ops.linear_bwd_weight_x3k(dyT, rimg, n0, K, x_rows=input_nodes, x_nrows=feat.shape[0], interleave=G, want_bias=True)[:2]


def new():
    return ops.pool_bwd_x3_dw(dout, idx, plan, n0, rimg, K, x_rows=input_nodes, x_nrows=feat.shape[0], want_bias=True)


(dw0, db0), t0, m0 = timed(old)
(dw1, db1), t1, m1 = timed(new)
sc = float(dw0.abs().max())
print("round-4 pair (values + groups + dW_pool0 + reduce): median %.1f us  min %.1f us" % (t0, m0))
print("record-fed   (values + k_gemm_x3rf + reduce):       median %.1f us  min %.1f us" % (t1, m1))
print("max |dw diff| / max|dw| = %.3g ; db %.3g" % (float((dw0 - dw1).abs().max()) / sc, float((db0 - db1).abs().max()) / float(db0.abs().max())))
ops.profile_start(); old(); new(); rec = ops.profile_stop()
for n, m, ms in rec:
    print("  %-28s %.1f us" % (n, ms * 1e3))
