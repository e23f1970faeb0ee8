import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
import ogl_amd
from ogl_amd import ops, optim, sampling, synthetic
from ogl_amd.graphsage import GatheredRows, GraphSAGE
sampling.seed(2); torch.manual_seed(2)
feat_size, _, dyn, n_classes, _ = synthetic.load("arxiv", snapshots=2, device="cuda")
dyn.evolve()
g = dyn.get_graph()
ops.set_gemm_mode(sys.argv[1] if len(sys.argv) > 1 else "auto")
model = GraphSAGE(feat_size, 256, n_classes, 1, F.relu, 0, "pool").cuda()
seeds = torch.as_tensor(np.random.default_rng(0).choice(g.n_present, 512, replace=False))
(input_nodes, sd, blocks), = list(sampling.NodeDataLoader(g, seeds, sampling.MultiLayerNeighborSampler([25, 25]), batch_size=512))
lab = ops.gather_i64(g.ndata["target"], sd)
for it in range(3):
    with torch.no_grad():
        logits = model(blocks, GatheredRows(g.ndata["feat"], input_nodes))
    torch.cuda.synchronize()
    print("no_grad forward", it, "logits absmax", float(logits.abs().max()), "finite", bool(torch.isfinite(logits).all()))
for it in range(3):
    loss, rows, logits = model.forward_loss(blocks, GatheredRows(g.ndata["feat"], input_nodes), lab, rows=True)
    torch.cuda.synchronize()
    print("forward_loss", it, float(loss.detach()), "rows finite", bool(torch.isfinite(rows).all()), "logits absmax", float(logits.abs().max()),
          "n bad rows", int((~torch.isfinite(rows)).sum()))
