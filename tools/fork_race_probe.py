"""Probe: 10 free-running train steps at the Reddit rung against the CPU oracle, repeated with the forked backward on / off
(and optionally after another full-size step in the same process): a stream-ordering bug would show as drift far above the
oracle's own 1-thread-vs-N-thread drift (~4e-5)."""
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
ops.set_gemm_mode("auto")
NSTEP = int(os.environ.get("NSTEP", 6))
rng = np.random.default_rng(17)
seed_list = [rng.choice(g.n_present, 512, replace=False).astype(np.int64) for _ in range(NSTEP)]
cpu = O.CpuModel("pool", 602, 600, 41, seed=3)
init = [{k: v.detach().clone() for k, v in prm.items()} for prm in cpu.params]
want = [cpu.train_step(feat_cpu, lab_cpu, host["indptr"], host["indices"], deg, s, 25, 9, i) for i, s in enumerate(seed_list)]
print("oracle", ["%.5f" % x for x in want], flush=True)
for rep in range(int(os.environ.get("REPS", 6))):
    fork = rep % 2 == 1
    ops.FORK_BACKWARD = fork
    model = GraphSAGE(602, 600, 41, 1, F.relu, 0, "pool").cuda()
    with torch.no_grad():
        for l, prm in zip(model.layers, init):
            for k, v in prm.items():
                mod, attr = k.split(".")
                getattr(getattr(l, mod), attr).copy_(v)
    opt = optim.Adam(model.parameters(), lr=1e-3)
    sampling.seed(9)
    got = []
    for i, seeds in enumerate(seed_list):
        (input_nodes, sd, blocks), = list(sampling.NodeDataLoader(g, torch.as_tensor(seeds), sampling.MultiLayerNeighborSampler([25, 25]), batch_size=512))
        opt.zero_grad()
        loss = ops.cross_entropy(model(blocks, GatheredRows(g.ndata["feat"], input_nodes)), ops.gather_i64(g.ndata["target"], sd), "mean")
        ops.backward(loss)
        opt.step()
        got.append(float(loss))
    drift = np.abs(np.asarray(got) - np.asarray(want)) / np.asarray(want)
    print("rep %d fork=%d drift %s" % (rep, fork, ["%.1e" % x for x in drift]), flush=True)
