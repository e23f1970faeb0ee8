"""Which aten ops still launch kernels inside one RBR train step (torch.profiler, Reddit-like shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import ogl_amd  # noqa
from ogl_amd import ops, optim, sampling, synthetic
from ogl_amd.graphsage import GraphSAGE
from ogl_amd.graphsage.sageconv import GatheredRows
from ogl_amd.graph.dynamic_graph_edge import DynamicGraphEdge

ops.set_gemm_mode("auto")
a = synthetic.make_arrays("reddit", 1.0)
dyn = DynamicGraphEdge(a["snapshots"], set(), device="cuda")
dyn.build(a["feat"], a["labels"], True, edge_timestamps={"src": a["src"], "dst": a["dst"]})
g = dyn.get_graph(); g.set_snapshot(g.n_total, len(a["src"]))
torch.manual_seed(1)
model = GraphSAGE(a["f"], 600, a["c"], 1, F.relu, 0, "pool", edge_feats=0, pool_feats=600).cuda()
opt = optim.Adam(model.parameters(), lr=1e-3)
sampler = sampling.MultiLayerNeighborSampler([25, 25], replace=True, return_eids=True); sampling.seed(1)
seeds = torch.as_tensor(np.random.default_rng(0).choice(g.n_present, 512 * 4, replace=False))
batches = list(sampling.NodeDataLoader(g, seeds, sampler, batch_size=512))

def step(b):
    input_nodes, sd, blocks = b
    opt.zero_grad()
    labels = ops.gather_i64(g.ndata["target"], sd)
    loss = ops.cross_entropy(model(blocks, GatheredRows(g.ndata["feat"], input_nodes)), labels, "mean")
    loss.backward(); opt.step()

for b in batches[:3]:
    step(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step(batches[3]); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)) > 0]
for e in sorted(rows, key=lambda e: -getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0))):
    print("%-28s calls %3d  device us %8.1f  shapes %s" % (e.key, e.count, getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)), str(e.input_shapes)[:110]))

