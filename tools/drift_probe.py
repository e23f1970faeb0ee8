"""Where does the free-running drift come from?  One Adam step on the device and on the oracle from identical weights; then
(a) per-tensor statistics of W_device - W_oracle, (b) the next batch's loss evaluated by the ORACLE at the device's weights
against the device's own value at them (forward arithmetic), against the oracle's at its own weights (trajectory)."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogl_amd  # noqa
from ogl_amd import ops, optim, sampling, synthetic
from ogl_amd.graph.dynamic_graph_edge import DynamicGraphEdge
from ogl_amd.graphsage import GatheredRows, GraphSAGE
from oracle import oracle as O

a = synthetic.make_arrays("reddit")
dyn = DynamicGraphEdge(a["snapshots"], set(), device="cuda")
dyn.build(a["feat"], a["labels"], True, edge_timestamps={"src": a["src"], "dst": a["dst"]})
g = dyn.get_graph(); g.set_snapshot(g.n_total, len(a["src"]))
h = g.handle
host = dict(indptr=h.indptr.cpu().numpy(), indices=h.indices.cpu().numpy(), keys=h.keys.cpu().numpy())
deg = O.snapshot_degrees_fast(host["indptr"], host["keys"], g.n_present, g.cut)
feat_cpu, lab_cpu = a["feat"], torch.as_tensor(a["labels"]).reshape(-1, 1)
mode = os.environ.get("GEMM", "auto")
ops.set_gemm_mode(mode)
rng = np.random.default_rng(17)
seed_list = [rng.choice(g.n_present, 512, replace=False).astype(np.int64) for _ in range(2)]
cpu = O.CpuModel("pool", 602, 600, 41, seed=3)
model = GraphSAGE(602, 600, 41, 1, F.relu, 0, "pool").cuda()
with torch.no_grad():
    for l, prm in zip(model.layers, cpu.params):
        for k, v in prm.items():
            mod, attr = k.split(".")
            getattr(getattr(l, mod), attr).copy_(v)
opt = optim.Adam(model.parameters(), lr=1e-3)
sampling.seed(9)


def dev_step(seeds, train=True):
    (input_nodes, sd, blocks), = list(sampling.NodeDataLoader(g, torch.as_tensor(seeds), sampling.MultiLayerNeighborSampler([25, 25]), batch_size=512))
    if train:
        opt.zero_grad()
    loss = ops.cross_entropy(model(blocks, GatheredRows(g.ndata["feat"], input_nodes)), ops.gather_i64(g.ndata["target"], sd), "mean")
    if train:
        ops.backward(loss)
        grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
        opt.step()
        return float(loss.detach()), grads
    return float(loss.detach()), None


l0_dev, g_dev = dev_step(seed_list[0])
l0_ref = cpu.train_step(feat_cpu, lab_cpu, host["indptr"], host["indices"], deg, seed_list[0], 25, 9, 0)
print("mode %s  step 0 loss: device %.6f oracle %.6f" % (mode, l0_dev, l0_ref))
for li, prm in enumerate(cpu.params):
    for k, v in prm.items():
        n = "layers.%d.%s" % (li, k)
        gd, gr = g_dev[n].double(), v.grad.double()
        mod, attr = k.split(".")
        wd = getattr(getattr(model.layers[li], mod), attr).detach().cpu().double()
        dw = (wd - v.detach().double()).abs()
        dg = (gd - gr).abs()
        print("%-28s |g| median %.2e  grad abs err: max %.2e median %.2e  rel fro %.2e | weights after Adam: >1e-6: %d  >1e-5: %d  >1e-4: %d of %d, max %.2e"
              % (n, float(gr.abs().median()), float(dg.max()), float(dg.median()), float((gd - gr).norm() / gr.norm()),
                 int((dw > 1e-6).sum()), int((dw > 1e-5).sum()), int((dw > 1e-4).sum()), dw.numel(), float(dw.max())))
# next batch, no training: device at W_d; oracle at W_o; oracle at W_d
l1_dev, _ = dev_step(seed_list[1], train=False)
with torch.no_grad():
    in_ref, sd_ref, blocks_ref = O.sample_blocks(host["indptr"], host["indices"], deg, seed_list[1], [25, 25], 9, 1)
    x = feat_cpu[torch.as_tensor(in_ref)]; y = lab_cpu[torch.as_tensor(sd_ref)]
    l1_ref = float(O.cross_entropy(cpu.forward(x, blocks_ref), y))
    twin = O.CpuModel("pool", 602, 600, 41, seed=3)
    for li, prm in enumerate(twin.params):
        for k in prm:
            mod, attr = k.split(".")
            prm[k].data.copy_(getattr(getattr(model.layers[li], mod), attr).detach().cpu())
    l1_ref_at_dev = float(O.cross_entropy(twin.forward(x, blocks_ref), y))
print("step 1 loss: device(W_d) %.6f  oracle(W_d) %.6f  oracle(W_o) %.6f" % (l1_dev, l1_ref_at_dev, l1_ref))
