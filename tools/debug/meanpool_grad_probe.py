"""Which half of layers.0.fc_neigh.weight's gradient is off in the full-size meanpool step, and on which code path?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
import ogl_amd
from ogl_amd import ops, optim, sampling, synthetic
from ogl_amd.graphsage import GatheredRows, GraphSAGE
from ogl_amd.graph.dynamic_graph_edge import DynamicGraphEdge
from oracle import oracle as O

a = synthetic.make_arrays("reddit")
dyn = DynamicGraphEdge(a["snapshots"], set(), device="cuda")
dyn.build(a["feat"], a["labels"], True, edge_timestamps={"src": a["src"], "dst": a["dst"]})
g = dyn.get_graph(); h = g.handle
host = dict(indptr=h.indptr.cpu().numpy(), indices=h.indices.cpu().numpy(), keys=h.keys.cpu().numpy())
g.set_snapshot(g.n_total, len(a["src"]))
ops.set_gemm_mode(os.environ.get("GEMM", "auto"))
B, S = 512, 25
deg = O.snapshot_degrees_fast(host["indptr"], host["keys"], g.n_present, g.cut)
cpu = O.CpuModel("meanpool", 602, 600, 41, pool_feats=600, seed=2)
seeds = np.random.default_rng(13).choice(g.n_present, B, replace=False).astype(np.int64)
feat_cpu, lab_cpu = a["feat"], torch.as_tensor(a["labels"]).reshape(-1, 1)
cpu.train_step(feat_cpu, lab_cpu, host["indptr"], host["indices"], deg, seeds, S, 6, 0)
ref = cpu.params[0]["fc_neigh.weight"].grad.numpy()
for lazy in (True, False):
    torch.manual_seed(0)
    cpu2 = O.CpuModel("meanpool", 602, 600, 41, pool_feats=600, seed=2)
    model = GraphSAGE(602, 600, 41, 1, F.relu, 0, "meanpool", edge_feats=0, pool_feats=600).cuda()
    with torch.no_grad():
        for l, prm in zip(model.layers, cpu2.params):
            for k, v in prm.items():
                mod, attr = k.split(".")
                getattr(getattr(l, mod), attr).copy_(v)
    sampling.seed(6)
    (input_nodes, sd, blocks), = list(sampling.NodeDataLoader(g, torch.as_tensor(seeds), sampling.MultiLayerNeighborSampler([S, S]), batch_size=B))
    x = GatheredRows(g.ndata["feat"], input_nodes) if lazy else ops.gather_rows(g.ndata["feat"], input_nodes)
    loss, _, _ = model.forward_loss(blocks, x, ops.gather_i64(g.ndata["target"], sd))
    ops.backward(loss)
    got = model.layers[0].fc_neigh.weight.grad.cpu().numpy()
    for name, sl in (("self half", slice(0, 602)), ("neigh half", slice(602, 1202))):
        print("lazy=%s %s: rel err %.3e  |got| %.3e |ref| %.3e" % (lazy, name, np.linalg.norm(got[:, sl] - ref[:, sl]) / np.linalg.norm(ref[:, sl]),
                                                                np.linalg.norm(got[:, sl]), np.linalg.norm(ref[:, sl])), flush=True)
    print("  loss", float(loss), "n0", int(input_nodes.numel()), "n1", blocks[1].number_of_src_nodes())
