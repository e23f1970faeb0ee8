"""Where a captured 32-seed step spends its time: sample graph + read-back vs train graph, host vs device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import ogl_amd  # noqa
from ogl_amd import ops, sampling, synthetic
from ogl_amd.graphsage import GraphSAGE
from ogl_amd.graphsage.model import RandomHipSupervisedGraphSage

name = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
ops.set_gemm_mode("auto")
feat_size, labels, dyn, n_classes, _ = synthetic.load(name, snapshots=2, device="cuda")
dyn.evolve()
g = dyn.get_graph()
model = GraphSAGE(feat_size, 32, n_classes, 1, F.relu, 0, "pool", edge_feats=0, pool_feats=32).cuda()
st = RandomHipSupervisedGraphSage(model, 1, 32, labels, 25, cuda=True, batch_full=1024)
st.build_optimizer(); model.train()
rng = np.random.default_rng(0)
for _ in range(40):
    st._train_batches(g, rng.choice(g.n_present, 32, replace=False), 32)
torch.cuda.synchronize()
cache = st._step_graphs()
smp = list(cache.samplers.values())[0]
N = 200
ts = []
for _ in range(N):
    sd = rng.choice(g.n_present, 32, replace=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n1, n0 = smp.run(sd, 7)
    t1 = time.perf_counter()
    key = [k for k in cache.graphs if k[-1] == min(ogl_amd.stepgraph.round_up(n0, 256), smp.buf.n0_cap)]
    if not key:
        continue
    sg = cache.graphs[key[0]]
    sg.cuda_graph.replay()
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t3 - t2))
a = np.array(ts) * 1e6
print("%s: sample graph + read-back %.1f us | train-graph launch (host) %.1f us | train graph to completion %.1f us | sum %.1f us  (n=%d)" % (
    name, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a.sum(1).mean(), len(a)))
# whole steps through the strategy
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    st._train_batches(g, rng.choice(g.n_present, 32, replace=False), 32)
torch.cuda.synchronize()
print("strategy step: %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
