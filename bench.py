#!/usr/bin/env python
"""Headline benchmark: streamed vertices/sec of the RBR training update on the Reddit-shaped stream
(BASELINE.json configs[3]: depth=2, samples=25, batch=512, F=602/H=600/C=41, pool aggregator),
plus the aggregator's achieved HBM GB/s and the projection GEMM's MFMA TFLOP/s against CDNA4 peaks,
next to the CPU oracle ("port") timed on this node's host cores.

  python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one replay batch of 512 seeds per GPU: 2-hop sampling +
block construction, feature-row gather (fused into the GEMM loaders), 2-layer GraphSAGE forward, cross
entropy, backward, Adam.  Sampling is done per snapshot for `batch_timestep` batches at once and is
inside the timed region; seed selection (choose_vertices) is outside, as in the reference's `delay`
(R/train/graphsage/model.py:108-117).  Inputs (CSR, feature table, labels, weights) are resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PMC_TRAFFIC_FILE = "r06_pmc_traffic.json"   # the committed rocprofv3 --pmc pass `roofline.traffic` is quoted from
PMC_PBR_TRAFFIC_FILE = "r06_pbr_pmc_traffic.json"   # ... and the one of the PBR priority-forward workloads
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense (MI355X_MICROARCH.md); AMD's 5 PF headline includes 2:1 sparsity

def norm_kernel(name):
    """Canonical (kernel, template arguments...) of an image-GEMM instantiation, from either spelling: what ogl_x3_last_kernel() reports
    ('k_gemm_x3p<4, 2, 2, 2, 2, false, true>': arguments as written at the launch site) or what rocprofv3 prints
    ('k_gemm_x3p<4, 2, 2, 2, 2, false, true, false, 0, 0>(X3Args)': every argument).  None when it is not a template-id."""
    import re
    m = re.match(r"\s*(\w+)<([^>]*)>", name or "")
    if not m:
        return None
    a = [x.strip() for x in m.group(2).split(",")]
    if m.group(1) == "k_gemm_x3p" and 5 <= len(a) < 9:
        a += ["false", "false", "false", "false"][len(a) - 5:]  # EXT, BK, AK, EA (csrc/linear_x3.hip; rounds 3-5 also had RH, CH)
    return (m.group(1),) + tuple(a)


WORKLOADS = {
    # name:     dataset  B    S   H    batch_timestep
    "reddit_rbr": dict(dataset="reddit", batch=512, samples=25, hidden=600, batch_timestep=50),
    "arxiv_rbr": dict(dataset="arxiv", batch=32, samples=25, hidden=32, batch_timestep=1),
    "pubmed_rbr": dict(dataset="pubmed", batch=32, samples=25, hidden=32, batch_timestep=2),
    "toy_rbr": dict(dataset="toy", batch=32, samples=5, hidden=16, batch_timestep=2),
    # the reference's OWN settings files (R/settings/*.json: samples / batch_size / embedding_size / batch_timestep / batch_full)
    "pubmed_settings": dict(dataset="pubmed", batch=32, samples=45, hidden=32, batch_timestep=2),
    "arxiv_settings": dict(dataset="arxiv", batch=32, samples=40, hidden=32, batch_timestep=1),
    "bitcoin_settings": dict(dataset="bitcoin", batch=32, samples=45, hidden=256, batch_timestep=60),
    "reddit_settings": dict(dataset="reddit", batch=1024, samples=30, hidden=600, batch_timestep=50),
    "reddit_settings_pbr_forward": dict(dataset="reddit", batch=900, samples=30, hidden=600, batch_timestep=50, forward=True),
    # PBR priority forward: inference over the train set in batches of batch_full, per-seed CE loss
    "reddit_pbr_forward": dict(dataset="reddit", batch=1024, samples=25, hidden=600, batch_timestep=50, forward=True),
    "arxiv_pbr_forward": dict(dataset="arxiv", batch=1024, samples=25, hidden=32, batch_timestep=1, forward=True),
    "toy_pbr_forward": dict(dataset="toy", batch=64, samples=5, hidden=16, batch_timestep=2, forward=True),
    # BASELINE config 3 end to end: the reference's loop body for the PBR strategy on the arxiv-like stream — priority forward over the
    # train set EVERY snapshot (batch_full 1024), one 32-seed update, evolve — whole snapshots on the wall clock
    "arxiv_pbr_snapshot": dict(dataset="arxiv", batch=32, samples=25, hidden=32, batch_timestep=1, snapshot=True, batch_full=1024,
                               priority_forward=1, snapshots=3500, start=3300),
    "toy_pbr_snapshot": dict(dataset="toy", batch=16, samples=5, hidden=16, batch_timestep=2, snapshot=True, batch_full=64,
                             priority_forward=1, snapshots=12, start=2),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="reddit_rbr", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the synthetic graph (debugging only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank); gloo = rehearsal of the N>1 path with several ranks on one GPU")
    ap.add_argument("--gemm", default="auto", choices=["f32", "bf16x6", "auto"],
                    help="projection arithmetic: exact fp32 MFMA, split-bf16 (x6, fp32-accurate) MFMA, or per-layout best")
    ap.add_argument("--aggregator", default="pool", choices=["pool", "meanpool", "mean", "maxpool", "gcn", "lstm"],
                    help="aggregator_type of the model (R/train/__main__.py:124-127 passes 'pool': DGL's SAGEConv; the others are the "
                         "in-repo layer's modes, R/train/graphsage/pytorch/aggregator_dgl.py:128-216, with pool_feats = the hidden size)")
    ap.add_argument("--no-graphs", action="store_true", help="enqueue every launch from Python instead of replaying captured steps")
    ap.add_argument("--graphs", action="store_true", help="replay captured steps for every batch size (default: the strategy's "
                    "'auto' policy — small batches always, large ones only when the host cannot keep ahead of the GPU)")
    ap.add_argument("--no-projection-cache", action="store_true", help="priority forward: recompute fc_pool_0 per batch")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of EACH cpu_baseline leg (multi-thread, one thread)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = the workload's batch per GPU (global batch grows with N); strong = the workload's batch in "
                         "total, cut over the GPUs (RBR: every 512-seed batch; PBR forward: the K batches of the pass)")
    ap.add_argument("--force-dist", action="store_true",
                    help="one rank only: initialise an RCCL process group of world size 1 and run the N-rank code path through it "
                         "(seed shards, gradient buckets + all-reduce calls, sharded passes): what a rank of an N-GPU job costs on "
                         "this GPU, minus the bytes on the links")
    ap.add_argument("--dp-capture", type=int, default=None, choices=[0, 1],
                    help="N > 1 / --force-dist: 1 = the replica's whole step (both RCCL all-reduces + Adam) inside the replayed hipGraph "
                         "(round 4's form; rehearsed on one rank only), 0 = forward + backward replayed, exchange and optimiser eager "
                         "(the default: graphsage/model.py DP_CAPTURE_COLLECTIVES)")
    ap.add_argument("--dp-sharded-update", type=int, default=None, choices=[0, 1],
                    help="N > 1 / --force-dist: 1 = the replica's exchange + optimiser as reduce-scatter -> Adam on this rank's 1 / N of the "
                         "parameters -> all-gather of the weights (parallel.ShardedAdam: the late-exchange lever of DESIGN section 6), "
                         "0 = all-reduce + the identical Adam on every rank (the default)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end snapshot leg (metric iii) of the default line")
    ap.add_argument("--knob", action="append", default=None, metavar="NAME=VALUE",
                    help="A/B: pin a kernel form through ogl_debug_set (ops.KNOBS: x3_tile, x3_stagger, block_min_lds, reduce_half, seg_rows)")
    ap.add_argument("--reduce-half", type=int, default=None, choices=[0, 1],
                    help="A/B: pin the narrow-row max aggregator to one (0) / two (1) neighbour rows per wave-instruction (ogl_debug_set: OGL_KNOB_REDUCE_HALF)")
    ap.add_argument("--variants-timeout", type=int, default=240,
                    help="seconds the exchange variants may take in all before every rank prints / exits with what it has")
    ap.add_argument("--no-variants", action="store_true",
                    help="N > 1 / --force-dist: skip the other two forms of the gradient exchange (`collectives_variants`) after the headline")
    ap.add_argument("--e2e-snapshots", type=int, default=6)
    ap.add_argument("--partition", default="replicated", choices=["replicated", "features"],
                    help="PBR forward, N > 1: 'features' = every rank projects only its vertex range and the projection tables are "
                         "exchanged by one halo all-gather each (the partitioned-feature mode); 'replicated' = every rank projects all rows")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start N ranks as FRESH child processes.  This parent never
    touches HIP (no torch.cuda call except device_count(), which does not initialise the runtime on this image) and does not
    exec: it waits for `python -m torch.distributed.run` and exits with its code; rank 0 of the children prints the JSON line."""
    import socket
    import subprocess
    if args.dist_backend == "nccl":
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            print("bench.py: --gpus %d over RCCL needs %d GPUs, this node shows %d (use --dist-backend gloo to rehearse the "
                  "N-rank control flow on fewer)" % (args.gpus, args.gpus, ndev), file=sys.stderr)
            return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and world > 1 and ndev < world:
        raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs (found %d); use --dist-backend gloo to rehearse on one GPU"
                         % (world, world, ndev))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    torch.cuda.set_device(local_rank % max(ndev, 1))
    ranks_seen = 1
    if args.force_dist:
        if world != 1:
            raise SystemExit("bench.py: --force-dist is a one-rank rehearsal")
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 or args.force_dist:
        # RCCL and gloo both print a banner (library versions / connection report) on STDOUT from their C++ side while the
        # communicator comes up: stdout is kept for the one JSON line, so the group is created — and its first collective
        # run — with file descriptor 1 pointed at stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.dist_backend == "nccl" or args.force_dist:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            nccl = dist.get_backend() == "nccl"
            one = torch.ones(1, dtype=torch.int64, device="cuda" if nccl else "cpu")
            dist.all_reduce(one)                             # the first collective of the run: every rank is really there
            ranks_seen = int(one.item())
            dist.barrier()
            if nccl:
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        if ranks_seen != args.gpus:
            raise SystemExit("bench.py: %d rank(s) answered the first all-reduce, --gpus %d" % (ranks_seen, args.gpus))
    args.ranks_seen = ranks_seen

    import ogl_amd  # noqa: F401
    from ogl_amd import ops, parallel, sampling, synthetic
    from ogl_amd.graphsage import GraphSAGE

    if args.force_dist:
        parallel.force_distributed(True)
    if args.dp_capture is not None:
        from ogl_amd.graphsage import model as _model_mod
        _model_mod.DP_CAPTURE_COLLECTIVES = bool(args.dp_capture)
    if args.dp_sharded_update is not None:
        parallel.SHARDED_UPDATE = bool(args.dp_sharded_update)
    ops.set_gemm_mode(args.gemm)
    if args.reduce_half is not None:
        ops.debug_set("reduce_half", args.reduce_half)
    for kv in args.knob or []:                                 # same-process A/B of a kernel form: --knob seg_rows=0
        name, _, val = kv.partition("=")
        ops.debug_set(name, int(val))
    wl = WORKLOADS[args.workload]
    B, S, H, bt = wl["batch"], wl["samples"], wl["hidden"], wl["batch_timestep"]
    if wl.get("snapshot"):
        if world != 1:
            raise SystemExit("bench.py: the snapshot workloads are one-rank lines")
        print(json.dumps(pbr_snapshot_bench(args, wl)))
        return
    t0 = time.time()
    arrays = synthetic.make_arrays(wl["dataset"], args.scale)
    feat_size, n_classes = arrays["f"], arrays["c"]
    # last snapshot of the stream: every vertex and edge present (the most work per seed)
    if arrays["stream"] == "edge":
        from ogl_amd.graph.dynamic_graph_edge import DynamicGraphEdge
        dyn = DynamicGraphEdge(arrays["snapshots"], set(), device="cuda")
        dyn.build(arrays["feat"], arrays["labels"], True, edge_timestamps={"src": arrays["src"], "dst": arrays["dst"]})
        g = dyn.get_graph()
        g.set_snapshot(g.n_total, len(arrays["src"]))
    else:
        from ogl_amd.graph.dynamic_graph_vertex import DynamicGraphVertex, FullGraphData
        gd = FullGraphData(arrays["n"], np.concatenate([arrays["src"], arrays["dst"]]),
                           np.concatenate([arrays["dst"], arrays["src"]]), arrays["feat"], arrays["labels"])
        dyn = DynamicGraphVertex(gd, arrays["snapshots"], set(), device="cuda")
        dyn.build(vertex_timestamps={int(v): int(t) for t, v in enumerate(arrays["order"])})
        g = dyn.get_graph()
        g.set_snapshot(g.n_total, g.n_total)
    setup_s = time.time() - t0

    torch.manual_seed(1)                                     # identical replicas on every rank
    model = GraphSAGE(feat_size, H, n_classes, 1, F.relu, 0, args.aggregator, edge_feats=0, pool_feats=H).cuda()
    sampling.seed(1)
    split_rng = np.random.default_rng(synthetic.SEEDS["split"])
    train_set = np.sort(split_rng.permutation(g.n_present)[: int(0.85 * g.n_present)])
    strong = args.scaling == "strong" and world > 1
    # The product path: the RBR strategy class (graphsage/model.py).  N ranks: every rank draws the SAME global batch
    # (identical host RNG, asserted by the strategy) and trains on its shard_range slice — weak: B seeds per GPU (global
    # batch B * N), strong: B seeds in total; gradient all-reduce in two buckets overlapped with backward.
    seed_rng = np.random.default_rng(1000)
    B_global = B if (strong or world == 1) else B * world
    B_local = len(range(*parallel.shard_range(B_global, rank, world)))

    if wl.get("forward"):
        forward_bench(args, wl, g, model, train_set, world, rank, arrays, feat_size, n_classes, setup_s)
        return

    from ogl_amd.graphsage.model import RandomHipSupervisedGraphSage
    stats = dict(n0=[], n1=[], forms={})

    def hook(info):
        stats["forms"][info["form"]] = stats["forms"].get(info["form"], 0) + 1
        if "n0" in info:
            stats["n0"].append(info["n0"]); stats["n1"].append(info["n1"])

    def make_strategy(mdl, batch):
        st = RandomHipSupervisedGraphSage(mdl, bt, batch, None, S, cuda=True, batch_full=1024)
        st.use_graphs = False if args.no_graphs else (True if args.graphs else "auto")
        st.build_optimizer()
        mdl.train()
        st.step_hook = hook
        return st

    state = dict(strat=None, batch=B_global)

    def draw(nb):
        return np.concatenate([seed_rng.choice(train_set, state["batch"], replace=False) for _ in range(nb)])

    def run(nsteps, seeds_per_snapshot):
        for seeds in seeds_per_snapshot:          # one snapshot's update: batch_timestep batches of B_global seeds
            state["strat"]._train_batches(g, seeds, state["batch"])

    def plan(nsteps):
        out, left = [], nsteps
        while left > 0:
            nb = min(bt, left)
            out.append(draw(nb)); left -= nb
        return out

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def settle_and_time(steps):
        """Untimed: the auto policy settles and the common size buckets are captured, W warm-up steps; then K steps between barriers,
        MAX over ranks.  Returns seconds."""
        st = state["strat"]
        if st.use_graphs in ("auto", True):
            run(max(6 * bt, 120), plan(max(6 * bt, 120)))
        run(args.warmup, plan(args.warmup))
        sp = plan(steps)
        barrier()
        t_ = time.perf_counter()
        run(steps, sp)
        barrier()
        el = time.perf_counter() - t_
        if world > 1:
            tt_ = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            el = float(tt_.item())
        return el

    def instrumented_collectives(steps):
        """hipEvent pairs around every collective the strategy issues from Python (None when the exchange is recorded inside the graph)."""
        st = state["strat"]
        if getattr(st, "gsync", None) is None or _dp_capture_on():
            return None
        st.gsync.enable_timing(True)
        shd = getattr(st, "sharded", None)
        if shd is not None:
            shd.enable_timing(True)
        csteps = min(steps, max(bt, 20))
        cplan = plan(csteps)
        barrier()
        tc = time.perf_counter()
        run(csteps, cplan)
        barrier()
        c_ms = 1000 * (time.perf_counter() - tc) / csteps
        tm = st.gsync.timings()
        st.gsync.enable_timing(False)
        if shd is not None:                    # (the sharded update: both of its collectives are exposed — Adam sits between them)
            tm.update(shd.timings())
            shd.enable_timing(False)
        exposed = sum(tm[k][0] * tm[k][1] for k in ("single", "all", "late", "reduce_scatter", "all_gather") if k in tm) / csteps
        return {"ms_per_step_instrumented": round(c_ms, 4),
                "per_kind": {k: {"mean_ms": round(v[0], 4), "per_step": round(v[1] / csteps, 2)} for k, v in tm.items()},
                "exposed_exchange_ms_per_step": round(exposed, 4),
                "t1_prime_ms": round(c_ms - exposed, 4),
                "what": "hipEvent pairs on the issuing stream: 'single' / 'all' / 'late' = from the all-reduce call to the point where "
                        "the stream may continue (exposed: Adam waits for it); 'early' = from its launch in the gradient hook to the wait "
                        "in sync() (mostly hidden under the layer-0 backward).  t1_prime = the instrumented step minus the exposed exchange: "
                        "what one rank's step costs with zero link time"}

    strat = state["strat"] = make_strategy(model, B_global)
    state["batch"] = B_global

    if strat.use_graphs in ("auto", True):
        # set-up, not measurement: let the strategy's auto policy see a cold and a warm snapshot and settle on its execution
        # mode, and let replayed steps capture their common size buckets (a capture is ~5 ms: a one-off per bucket over a
        # stream of thousands of snapshots, but a visible share of a 100-step run) before the W warm-up and the K timed steps
        run(max(6 * bt, 120), plan(max(6 * bt, 120)))        # (the policy times snapshots 3-5 and decides on their median)
    run(args.warmup, plan(args.warmup))
    seeds_plan = plan(args.steps)
    stats["n0"], stats["n1"], stats["forms"] = [], [], {}
    graphs_on = strat._graphs_ok() or strat._graphs_ok("staged_dp")
    captures_before = strat._step_graphs().captures if graphs_on else 0
    borrowed_before = strat._step_graphs().borrowed if graphs_on else 0
    barrier()
    t1 = time.perf_counter()
    run(args.steps, seeds_plan)
    barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    forms_timed = dict(stats["forms"])
    captures_timed = (strat._step_graphs().captures - captures_before) if graphs_on else 0
    borrowed_timed = (strat._step_graphs().borrowed - borrowed_before) if graphs_on else 0
    timed_mode = "captured hipGraph replays" if set(forms_timed) & {"sampled", "staged", "staged_dp"} else "eager launches from Python"

    # ---- host side: time to ENQUEUE a step (no synchronisation inside the bracket): the margin between this and
    # ms_per_step is how far the step is from being launch-bound on this box's host cores
    hsteps = min(args.steps, bt)
    hplan = plan(hsteps)
    barrier()
    th = time.perf_counter()
    run(hsteps, hplan)
    host_ms = 1000 * (time.perf_counter() - th) / hsteps
    barrier()
    # ---- N ranks (or --force-dist): device time of the step's collectives, term by term of DESIGN section 6's model T(N) = T1' + L(N) + S(N).
    # An instrumented pass AFTER the timed region: hipEvent pairs around every collective the step issues from Python (the default
    # replica step keeps its exchange outside the replayed graph; --dp-capture 1 records it inside, where events cannot see it).
    collectives = instrumented_collectives(args.steps)
    # ---- N ranks: the ONE-RANK step on every rank of this same invocation (no shard, no exchange: `parallel.local_only`), B seeds per
    # rank — the T1 a scaling curve divides by, measured on the same boxes in the same process group as TN, right BEHIND the headline
    # (a warm chip either way: in front of it the first run of a fresh box read 5-15 % slow)
    one_rank_reference = None
    if world > 1 or args.force_dist:
        with parallel.local_only():
            torch.manual_seed(1)
            sampling.seed(1)
            m1 = GraphSAGE(feat_size, H, n_classes, 1, F.relu, 0, args.aggregator, edge_feats=0, pool_feats=H).cuda()
            state["strat"], state["batch"] = make_strategy(m1, B), B
            el1 = settle_and_time(args.steps)
        one_rank_reference = dict(ms_per_step=round(1000 * el1 / args.steps, 4), vertices_per_s_per_rank=round(args.steps * B / el1, 1),
                                  what="the one-rank step (B = %d seeds, no shard, no exchange) timed on every rank inside this invocation, "
                                       "MAX over ranks: `--gpus 1` of the same command line" % B)
        state["strat"], state["batch"] = strat, B_global
        del m1
    # ---- N ranks: the OTHER forms of the exchange, timed in this same invocation on the same ranks (the driver gives one run per N):
    # (a) = the headline above (all-reduce after the replayed forward + backward, one flat bucket; two overlapped buckets when eager),
    # (b) the sharded update (reduce-scatter -> Adam on 1 / N of the flat parameters -> all-gather), (c) the exchange and the optimiser
    # recorded INSIDE the replayed step graph (RCCL only).  Each on a fresh model + strategy, sequentially, nothing re-executed.
    # They run LAST — after every other measurement of this invocation, with the finished line in hand and a watchdog on every rank: a form
    # that has never run on more than one GPU (the captured exchange) must not be able to cost the run its headline.
    want_variants = (world > 1 or args.force_dist) and not args.no_variants
    collectives_variants = {} if want_variants else None

    def measure_variants():
        from ogl_amd.graphsage import model as _mm
        nccl_on = dist.is_initialized() and dist.get_backend() == "nccl"
        base_flags = (bool(parallel.SHARDED_UPDATE), bool(_mm.DP_CAPTURE_COLLECTIVES))
        collectives_variants["a_allreduce" if base_flags == (False, False) else "headline"] = dict(
            ms_per_step=round(1000 * elapsed / args.steps, 4), sharded_update=base_flags[0], captured_exchange=base_flags[1], collectives=collectives)
        todo = [("a_allreduce", False, False), ("b_sharded_update", True, False), ("c_captured_exchange", False, True)]
        for name, shd_on, cap_on in todo:
            if (shd_on, cap_on) == base_flags:
                continue
            if cap_on and not nccl_on:
                collectives_variants[name] = dict(skipped="collectives are recorded into a hipGraph over RCCL only (backend %s)" % dist.get_backend())
                continue
            parallel.SHARDED_UPDATE, _mm.DP_CAPTURE_COLLECTIVES = shd_on, cap_on
            try:
                torch.manual_seed(1)
                sampling.seed(1)
                mv = GraphSAGE(feat_size, H, n_classes, 1, F.relu, 0, args.aggregator, edge_feats=0, pool_feats=H).cuda()
                state["strat"] = make_strategy(mv, B_global)
                elv = settle_and_time(args.steps)
                collectives_variants[name] = dict(ms_per_step=round(1000 * elv / args.steps, 4), sharded_update=shd_on, captured_exchange=cap_on,
                                                  collectives=instrumented_collectives(args.steps))
                del mv
            except Exception as ex:                          # (recorded, never fatal: the headline is already measured)
                collectives_variants[name] = dict(error="%s: %s" % (type(ex).__name__, str(ex)[:300]))
            finally:
                parallel.SHARDED_UPDATE, _mm.DP_CAPTURE_COLLECTIVES = base_flags
                state["strat"] = strat
        barrier()
    # the OTHER execution mode of large batches, for the record: when the auto policy kept this workload eager (fast host),
    # the same steps replayed as captured graphs (size buckets captured in an untimed warm-up first)
    graph_mode = None
    if world == 1 and timed_mode.startswith("eager") and strat.use_graphs == "auto" and getattr(strat.optimizer, "capturable", False):
        strat.use_graphs = True
        run(max(3 * bt, 120), plan(max(3 * bt, 120)))        # untimed: the common size buckets are captured here
        gsteps = min(args.steps, 2 * bt)
        gplan = plan(gsteps)
        barrier(); tg = time.perf_counter()
        run(gsteps, gplan)
        barrier()
        gtot = 1000 * (time.perf_counter() - tg) / gsteps
        hplan2 = plan(hsteps)                       # host enqueue: ONE snapshot (its loader's read-backs find an idle GPU),
        smp_state = sampling.get_state()            # run twice on the same seeds AND sampler counters: the first pass captures
        run(hsteps, hplan2)                         # any new size bucket, the second replays exactly the same blocks
        sampling.set_state(smp_state)
        barrier(); tg = time.perf_counter()
        run(hsteps, hplan2)
        ghost = 1000 * (time.perf_counter() - tg) / hsteps
        barrier()
        graph_mode = dict(ms_per_step=round(gtot, 4), host_enqueue_ms_per_step=round(ghost, 4), steps=gsteps,
                          note="the same train steps replayed as captured hipGraphs (staged form): what the auto policy switches to "
                               "when a timed snapshot shows host enqueue time > %.2f x GPU time" % strat.STAGED_AUTO_HOST_FRACTION)
        strat.use_graphs = "auto"
    if os.environ.get("OGL_BENCH_CPROFILE") and rank == 0:          # where the host time of a step goes (stderr)
        import cProfile, pstats
        torch.autograd.set_multithreading_enabled(False)            # backward in this thread, so that it is seen
        pr = cProfile.Profile()
        hplan = plan(hsteps)
        pr.enable(); run(hsteps, hplan); pr.disable()
        torch.autograd.set_multithreading_enabled(True)
        barrier()
        pstats.Stats(pr, stream=sys.stderr).sort_stats(os.environ.get("OGL_BENCH_CPROFILE_SORT", "tottime")).print_stats(45)

    # ---- per-kernel HIP-event timing (same workload, separate instrumented pass) -----------------
    prof_steps = min(args.steps, bt)
    prof_plan = plan(prof_steps)
    barrier()
    stats["n0"], stats["n1"] = [], []
    ops.profile_start()                       # per-launch HIP events: this pass runs the same launches EAGERLY (no replay)
    run(prof_steps, prof_plan)
    rec = ops.profile_stop()
    n0_avg = float(np.mean(stats["n0"])) if stats["n0"] else float("nan")
    n1_avg = float(np.mean(stats["n1"])) if stats["n1"] else float("nan")
    if os.environ.get("OGL_BENCH_DUMP_CALLS") and rank == 0:        # every C-ABI call of the first profiled step, in order
        for name, meta, ms in rec[:len(rec) // prof_steps]:
            print("call %-28s %8.4f ms  %s" % (name, ms, meta), file=sys.stderr)
    agg = {}
    for name, meta, ms in rec:
        key = name
        if name in ("ogl_reduce_fwd", "ogl_reduce_fwd_img"):       # (_img: the same pass + the bf16x3 image of its output)
            key = "reduce_fwd_L0" if meta["n_dst"] > B else "reduce_fwd_L1"
        elif name.startswith("ogl_linear"):
            key = name[4:] + ("_pool0" if meta["M"] > B * (1 + S) else "_other")
        elif name == "ogl_small_proj_rows":                         # fc_pool of a 32-seed step's first layer: exact-fp32 MFMA, small tiles
            key = "linear_fwd_f32small_pool0"
        elif name == "ogl_small_first_layer_fwd":                   # the same step's aggregator (max + combine in one launch)
            key = "reduce_fwd_L0"
        a = agg.setdefault(key, dict(ms=0.0, calls=0, bytes=0.0, flops=0.0))
        a["ms"] += ms; a["calls"] += 1
        if meta and meta.get("kernel"):
            a.setdefault("kernel_names", set()).add(meta["kernel"])
        if name in ("ogl_reduce_fwd", "ogl_reduce_fwd_img"):
            E = meta["n_dst"] * meta["fanout"]
            a["bytes"] += E * (4 * meta["d"] + meta["idx_bytes"]) + meta["n_dst"] * 4 * meta["d"] * (2 if meta["argmax"] else 1)
            if name == "ogl_reduce_fwd_img":
                a["bytes"] += meta["n_dst"] * 6 * meta["d"]          # + the image: three bf16 planes per element
        if name == "ogl_small_first_layer_fwd":                      # E gathered rows + indices, the reduced rows + argmax written
            E = meta["n_dst"] * meta["fanout"]
            a["bytes"] += E * (4 * meta["d"] + 4) + meta["n_dst"] * 4 * meta["d"] * 2
            a["small"] = True
        if name == "ogl_small_proj_rows":
            a["flops"] += 2.0 * meta["M"] * meta["N"] * meta["K"]
        # mandatory bytes of the other HBM-bound launches (for roofline.composite_floor_ms; every matrix counted once per pass over it)
        if name in ("ogl_pool_bwd_x3", "ogl_pool_bwd_x3_apply"):     # dout read + the group-major bf16x3 image of dP^T written
            a["bytes"] += meta["n_dst"] * 4 * meta["d"] + (meta["n_src"] + 31) // 32 * 32 * 6 * meta["d"]
        elif name == "ogl_pool_bwd_x3_plan":                         # argmax + pooled rows read, 2-byte (column, slot) ids written
            a["bytes"] += meta["n_dst"] * meta["d"] * (4 + 4 + 2)
        elif name == "ogl_reduce_bwd_seg_apply":                     # SURVEY 8(d) style: E gathered rows + n_src rows written (+ image / mask)
            a["bytes"] += (meta["n_dst"] * meta["fanout"] * 4 * meta["d"] + meta["n_src"] * meta["d"] * ((4 if meta["out"] else 0) +
                           (6 if meta["image"] else 0) + (4 if meta["mask"] else 0)))
            a["bytes_8d"] = a.get("bytes_8d", 0.0) + meta["n_dst"] * meta["fanout"] * 4 * meta["d"] + meta["n_src"] * 4 * meta["d"]
        elif name == "ogl_reduce_bwd_seg_apply_t":                   # the same backward written as the transposed group-major image (k_seg_groups)
            a["bytes"] += (meta["n_dst"] * meta["fanout"] * 4 * meta["d"] + meta["n_src"] * meta["d"] * (6 + (4 if meta["mask"] else 0)))
            a["bytes_8d"] = a.get("bytes_8d", 0.0) + meta["n_dst"] * meta["fanout"] * 4 * meta["d"] + meta["n_src"] * 4 * meta["d"]
        elif name == "ogl_relu_bwd_img":                             # dy, y read; masked dy + its image written
            a["bytes"] += meta["M"] * meta["N"] * (4 + 4 + 4 + 6)
        elif name == "ogl_x3_split":
            a["bytes"] += meta["R"] * meta["K"] * (4 + 6)
        elif name in ("ogl_adam_step_multi", "ogl_adam_step_multi_dev", "ogl_adam_step"):
            a["bytes"] += meta["n"] * 4 * 7                           # p, g, m, v read; p, m, v written
        elif name == "ogl_out_layer_bwd_inputs":                     # dy . W scattered with float atomics into an [n_src, K] target
            a["bytes"] += meta["M"] * meta["K"] * 4 * 2
        elif name == "ogl_fill_zero":
            a["bytes"] += meta["bytes"]
        if name in ("ogl_linear_fwd", "ogl_linear_fwd_x3", "ogl_linear_fwd_x3_ext"):
            a["flops"] += 2.0 * meta["M"] * meta["N"] * (meta["K"] + meta["K2"])
        if name in ("ogl_linear_bwd_input", "ogl_linear_bwd_weight", "ogl_linear_bwd_weight_t", "ogl_linear_bwd_weight_x3",
                    "ogl_linear_bwd_weight_x3k"):
            a["flops"] += 2.0 * meta["M"] * meta["N"] * meta["K"]
    kernels = {k: dict(avg_ms=v["ms"] / v["calls"], ms_per_step=v["ms"] / prof_steps, calls_per_step=v["calls"] / prof_steps,
                       gbs=(v["bytes"] / v["ms"] / 1e6) if v["bytes"] else None,
                       tflops=(v["flops"] / v["ms"] / 1e9) if v["flops"] else None) for k, v in agg.items()}
    ragg = agg.get("reduce_fwd_L0")
    roof_aggr = None
    if ragg:
        ach = ragg["bytes"] / ragg["ms"] / 1e6
        roof_aggr = dict(kernel=("k_small_first_fwd (32-seed step: layer-0 gather + max + combine in one launch, argmax kept; latency-bound at these sizes)"
                                 if ragg.get("small") else
                                 "k_reduce_fwd_v4 (layer-0 gather+max, argmax kept, bf16x3 image of the output written beside it)"), bound="hbm", achieved=round(ach, 1),
                         peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                         avg_launch_ms=round(ragg["ms"] / ragg["calls"], 4),
                         algorithmic_bytes_per_launch=round(ragg["bytes"] / ragg["calls"]),
                         frac_kind="algorithmic bytes / launch time / 8 TB/s (SURVEY 8(d)): cache re-serves count as bytes, so this is NOT a pin rate",
                         note="algorithmic bytes count every gathered row once per gather; a row gathered again while it is still in L2 / "
                         "the 256 MB Infinity Cache does not come from HBM, so the algorithmic rate can approach or exceed the "
                         "8 TB/s pin rate.  `traffic` (PMC FETCH_SIZE x 2 + WRITE_SIZE) is the L2's fabric-side request bytes: "
                         "Infinity-Cache hits are included (MI355X_MICROARCH.md), so traffic / time bounds the HBM rate from above — "
                         "a plain streamed copy tops out at ~6.3 TB/s on this part")
    roof_mean_bwd = None
    sagg = agg.get("ogl_reduce_bwd_seg_apply_t") or agg.get("ogl_reduce_bwd_seg_apply")
    if sagg:
        # the longest launch class dominates: the first layer's launch ('meanpool': the group-wise one, alone in its class; 'mean': the
        # pooled figure over the step's row-wise launches)
        ach8 = sagg["bytes_8d"] / sagg["ms"] / 1e6
        seg_kernel = ("k_seg_groups (mean backward of the first layer as the transposed group-major image: one block per source group, the "
                      "planned lists as scalar loads, no atomics)" if "ogl_reduce_bwd_seg_apply_t" in agg else
                      "k_seg_reduce + k_seg_fixup (mean backward as a planned segmented gather: edges sorted by source, 64-entry tiles, no atomics)")
        roof_mean_bwd = dict(kernel=seg_kernel, bound="hbm", achieved=round(ach8, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(ach8 / HBM_PEAK_GBS, 4), traffic=None, ms_per_step=round(sagg["ms"] / prof_steps, 4),
                             algorithmic_bytes_per_step=round(sagg["bytes_8d"] / prof_steps),
                             bytes_moved_per_step=round(sagg["bytes"] / prof_steps),
                             frac_kind="algorithmic bytes E * 4D read + n_src * 4D written (VERDICT r3 item 3) / launch time / 8 TB/s; the [n_dst, D] "
                                       "gradient matrix the rows are gathered from stays in L2 / MALL, so this is not a pin rate; bytes_moved "
                                       "counts what the launch really writes (the bf16x3 image instead of fp32 rows, + the ReLU mask it reads)")
    gemm_keys = [k for k in agg if k.startswith("linear") and agg[k]["flops"] > 0]
    gflops = sum(agg[k]["flops"] for k in gemm_keys); gms = sum(agg[k]["ms"] for k in gemm_keys)
    # the dominant GEMM = the single LAUNCH (one shape, one kernel) with the longest duration; the "*_other" keys pool
    # several launches of different shapes (their per-launch numbers are in "kernels") and are not one kernel launch
    dom = max(gemm_keys, key=lambda k: agg[k]["ms"] / agg[k]["calls"]) if gemm_keys else None
    roof_gemm = None
    if dom:
        ach = agg[dom]["flops"] / agg[dom]["ms"] / 1e9
        # which arithmetic did this launch run on?  (ops.weight_grad / linear.hip AUTO policy)
        x6 = args.gemm == "bf16x6" or (args.gemm == "auto" and ("_x3" in dom or "bwd_weight_t" in dom or "bwd_weight" not in dom))
        if "f32small" in dom:
            x6 = False                                               # (ogl_small_proj_rows: exact fp32 products whatever the mode)
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if x6 else MFMA_F32_PEAK_TFLOPS
        roof_gemm = dict(kernel="%s (%s; %s)" % ("k_gemm_x3p" if "_x3" in dom else ("k_small_proj_rows" if "f32small" in dom else "k_gemm"), dom, ("split-bf16 x6 on v_mfma_f32_16x16x32_bf16, fp32 accumulate" if "_x3" in dom else
                                                       "split-bf16 x6 on v_mfma_f32_32x32x16_bf16, fp32 accumulate") if x6
                                                      else "v_mfma_f32_32x32x2_f32"),
                         bound="mfma", achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(ach / peak, 4),
                         traffic=None, avg_launch_ms=round(agg[dom]["ms"] / agg[dom]["calls"], 4),
                         algorithmic_flops_per_launch=round(agg[dom]["flops"] / agg[dom]["calls"]),
                         peak_note=("fp32-equivalent roof of the x6 arithmetic: dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per "
                                    "product (vs 157.3 TFLOP/s for the exact-fp32 MFMA)") if x6 else "dense fp32 MFMA peak",
                         all_gemms_tflops=round(gflops / gms / 1e9, 2) if gms else None)
        if x6:
            roof_gemm["power_note"] = ("the 2500 TFLOP/s peak is 1024 SIMDs x 1024 FLOP/clk at the 2.4 GHz maximum clock; under this kernel's load on random "
                                       "operands the part holds 1.9-2.0 GHz (in-kernel s_memtime / s_memrealtime; the same binary on all-zero operands: "
                                       "2.39 GHz and +14-17 % TFLOP/s) with the matrix pipe busy 70-72 % of the in-kernel cycles — "
                                       "profiles/r06_x3_clock_probe.txt, r06_x3_phase_probe.txt, DESIGN.md section 0")

    # ---- the whole step against its COMPOSITE roofline: every GEMM at the MFMA roof of its arithmetic + every HBM-bound launch at the
    # HBM peak, summed (no overlap assumed between the two: the step's launches are dependent) — ms_per_step / this = how far the step
    # as a whole is from its kernels' rooflines
    composite = None
    if gemm_keys:
        x6_roof, f32_roof = BF16_MFMA_PEAK_TFLOPS / 6.0, MFMA_F32_PEAK_TFLOPS
        fl_x6 = sum(v["flops"] for k, v in agg.items() if v["flops"] and args.gemm != "f32" and ("_x3" in k or "bwd_weight" not in k)
                    and "f32small" not in k)
        fl_f32 = sum(v["flops"] for v in agg.values()) - fl_x6
        mfma_ms = (fl_x6 / (x6_roof * 1e9) + fl_f32 / (f32_roof * 1e9)) / prof_steps
        hbm_bytes = sum(v["bytes"] for k, v in agg.items() if v["bytes"] and not v["flops"])
        hbm_ms = hbm_bytes / (HBM_PEAK_GBS * 1e6) / prof_steps
        composite = dict(mfma_floor_ms=round(mfma_ms, 4), hbm_floor_ms=round(hbm_ms, 4), composite_floor_ms=round(mfma_ms + hbm_ms, 4),
                         gemm_gflop_per_step=round(sum(v["flops"] for v in agg.values()) / prof_steps / 1e9, 2),
                         hbm_algorithmic_mb_per_step=round(hbm_bytes / prof_steps / 1e6, 1),
                         counted_hbm_launches=sorted(k for k, v in agg.items() if v["bytes"] and not v["flops"]),
                         what="sum over the step's launches of (GEMM flops / MFMA roof of the launch's arithmetic: %.0f TFLOP/s fp32-equivalent "
                              "for split-bf16 x6, %.1f for exact fp32) + (algorithmic bytes of the HBM-bound launches / %.0f GB/s)"
                              % (x6_roof, f32_roof, HBM_PEAK_GBS))

        if roof_gemm:                                  # (also inside `roofline`: the whole-step fraction beside the dominant launch's)
            roof_gemm["composite_floor_ms"] = composite["composite_floor_ms"]
            roof_gemm["step_frac_of_composite_floor"] = round(composite["composite_floor_ms"] / (1000 * elapsed / args.steps), 4)

    # ---- CPU baseline: the oracle ("port" of the reference path) on this node's host cores ---------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        h = g.handle
        indptr, indices = h.indptr.cpu().numpy(), h.indices.cpu().numpy()
        keys = (h.keys if h.keys is not None else h.indices).cpu().numpy()
        deg = O.snapshot_degrees_fast(indptr, keys, g.n_present, g.cut)
        feat_cpu = g.ndata["feat"].cpu().contiguous()
        lab_cpu = g.ndata["target"].cpu()

        def cpu_leg(threads, batch, min_steps=3):
            """One untimed warm-up step (thread pool, allocator, first-touch), then >= min_steps timed steps within the budget."""
            torch.set_num_threads(threads)
            cpu = O.CpuModel(args.aggregator, feat_size, H, n_classes, pool_feats=H, seed=1)
            cpu.train_step(feat_cpu, lab_cpu, indptr, indices, deg, seed_rng.choice(train_set, batch, replace=False), S, 1, 10 ** 6)
            tc, nstep = time.perf_counter(), 0
            while nstep < min_steps or (time.perf_counter() - tc < args.cpu_seconds and nstep < 8):
                cpu.train_step(feat_cpu, lab_cpu, indptr, indices, deg, seed_rng.choice(train_set, batch, replace=False), S, 1,
                               10 ** 6 + 1 + nstep)
                nstep += 1
            dt = time.perf_counter() - tc
            return dict(value=round(nstep * batch / dt, 2), steps=nstep, batch=batch, seconds=round(dt, 2), threads=threads)

        # the reference's own path is one Python thread driving torch-CPU intra-op threads (n_sampling_workers = 0,
        # R/train/__main__.py:39); more threads than physical cores (or than ~64) only oversubscribe its small GEMMs
        ncpu = os.cpu_count() or 1
        cores = min(ncpu, 64)
        multi = cpu_leg(cores, B)
        multi_all = None
        if ncpu > cores:                              # a bigger host: also all of its threads, and the better of the two counts
            multi_all = cpu_leg(ncpu, B, min_steps=1)       # (oversubscribed hosts take 10+ s per step here: one timed step is enough to see it)
            if multi_all["value"] > multi["value"]:
                multi, multi_all, cores = multi_all, multi, ncpu
        single = cpu_leg(1, B)                        # the same batch as every other leg (>= 3 timed steps, ~2.5 s each)
        cpu_baseline = dict(value=multi["value"], unit="vertices/s", cores=cores, kind="port",
                            sample="1 warm-up + %d timed RBR train steps of %d seeds (same graph, shapes and sampler) in %.1f s; "
                                   "torch-CPU fp32, %d threads (host has %d)" % (multi["steps"], B, multi["seconds"], cores,
                                                                                os.cpu_count() or 1),
                            other_thread_count=(dict(value=multi_all["value"], cores=multi_all["threads"], steps=multi_all["steps"])
                                                if multi_all else None),
                            single_thread=dict(value=single["value"], unit="vertices/s", cores=1,
                                               sample="1 warm-up + %d timed steps of %d seeds in %.1f s, 1 thread"
                                                      % (single["steps"], single["batch"], single["seconds"])))
        torch.set_num_threads(os.cpu_count() or 1)

    # HBM traffic per launch comes from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled
    # as the gfx950 guide prescribes); bench.py cannot collect PMC counters on itself, so it quotes the committed pass.
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
        stamp = "%s, collected %s at HEAD %s — a constant quoted from that pass, NOT measured in this run" % (
            "profiles/" + PMC_TRAFFIC_FILE, pmc.get("collected", "?"), pmc.get("head", "?"))
        if roof_aggr and args.workload == "reddit_rbr":
            roof_aggr["traffic"] = pmc["k_reduce_fwd_v4_L0"]["traffic_bytes"]
            roof_aggr["traffic_source"] = stamp + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: fabric-side requests of the L2 — " \
                                                  "Infinity-Cache hits are counted, so this bounds the HBM bytes from above)"
        if roof_gemm and args.workload == "reddit_rbr" and args.gemm == "auto":
            key = {"linear_fwd_x3_pool0": "k_gemm_x3_fwd_pool0", "linear_bwd_weight_x3_pool0": "k_gemm_x3_bww_pool0",
                   "linear_bwd_weight_x3k_pool0": "k_gemm_x3_bwwk_pool0"}.get(dom)
            if key and key in pmc:
                # the constant is only quoted beside the instantiation it was counted on: a tile / kernel change since the PMC pass
                # makes it stale, and then the line says so instead of carrying the old bytes
                ran = {norm_kernel(n) for n in agg[dom].get("kernel_names", ())}
                counted = norm_kernel(pmc[key].get("kernel"))
                roof_gemm["kernel_instantiation"] = sorted(agg[dom].get("kernel_names", ()))
                if ran == {counted}:
                    roof_gemm["traffic"] = pmc[key]["traffic_bytes"]
                    roof_gemm["traffic_source"] = stamp + " (fabric-side bytes per launch — L2 misses, Infinity-Cache hits included; an " \
                                                          "MFMA-bound kernel; same instantiation as timed here: %s)" % pmc[key].get("kernel")
                else:
                    roof_gemm["traffic_source"] = "DROPPED: profiles/%s counted %s, this run timed %s — re-collect the PMC pass" % (
                        PMC_TRAFFIC_FILE, pmc[key].get("kernel"), sorted(agg[dom].get("kernel_names", ())))
    except Exception:
        pass

    e2e = None
    if rank == 0 and world == 1 and args.workload == "reddit_rbr" and not args.no_e2e and args.scale == 1.0:
        e2e = end_to_end_snapshots(args.e2e_snapshots, gemm=args.gemm)
    if rank == 0:
        value = args.steps * B_global / elapsed
        hbm_copy = None
        if world == 1:          # the box's own streaming figure beside the nominal peak: 1 GiB device copy
            src_b = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
            dst_b = torch.empty_like(src_b)
            dst_b.copy_(src_b); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst_b.copy_(src_b)
            e1.record(); torch.cuda.synchronize()
            torch_gbs = round(10 * 2 * src_b.numel() * 4 / e0.elapsed_time(e1) / 1e6, 1)
            from ogl_amd import _lib as _l
            st_ = torch.cuda.current_stream().cuda_stream
            _l.check(_l.lib().ogl_stream_copy(src_b.data_ptr(), dst_b.data_ptr(), src_b.numel() * 4, st_), "ogl_stream_copy")
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                _l.lib().ogl_stream_copy(src_b.data_ptr(), dst_b.data_ptr(), src_b.numel() * 4, st_)
            e1.record(); torch.cuda.synchronize()
            hbm_copy = dict(gbs=round(10 * 2 * src_b.numel() * 4 / e0.elapsed_time(e1) / 1e6, 1), torch_copy_gbs=torch_gbs,
                            what="1 GiB fp32 device-to-device copy (read + write bytes), 10 repeats, same process: `gbs` = the package's "
                                 "float4 stream-copy kernel (ogl_stream_copy; MI355X_MICROARCH.md quotes ~6.3 TB/s for this form), "
                                 "`torch_copy_gbs` = torch's copy_ of the same buffers.  The roofline denominators stay the 8 TB/s spec figure")
            del src_b, dst_b
        line = {
            "metric": "streamed vertices/sec (RBR train update), %s-shaped stream depth=2 samples=%d" % (wl["dataset"], S),
            "value": round(value, 1), "unit": "vertices/s", "n_gpus": world, "ranks_seen": args.ranks_seen,
            "dist_backend": (args.dist_backend if world > 1 else None), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"gemm_arithmetic": gemm_desc(args.gemm), "workload": "%s: %s-like %s stream, last snapshot (N=%d, CSR nnz=%d), F=%d H=%d C=%d, "
                                   "aggregator=%s, depth=2 samples=%d batch=%s batch_timestep=%d, "
                                   "sample+gather+fwd+CE+bwd+Adam" % (args.workload, wl["dataset"], arrays["stream"], g.n_present,
                                                                        int(h_nnz(g)), feat_size, H, n_classes,
                                                                        {"pool": "pool(max)"}.get(args.aggregator, args.aggregator + " (in-repo layer, pool_feats=%d)" % H), S,
                                                                        ("%d in total (%d on this rank)" % (B, B_local)) if strong else "%d/GPU" % B, bt),
                       "global_batch": B_global,
                       "step_execution": "%s %s%s" % (timed_mode, forms_timed,
                                                      ("; %d new size bucket(s) captured inside the timed region, %d step(s) replayed on the next "
                                                       "larger bucket's graph" % (captures_timed, borrowed_timed))
                                                      if timed_mode.startswith("captured") else "")
                       + ("; auto policy probe: %s" % getattr(strat, "staged_auto_probe", None) if strat.use_graphs == "auto" else ""),
                       "parallelism": "dp%d (seed-sharded replicas, two-bucket grad all-reduce %s)%s%s" % (
                           world, ("INSIDE the replayed step graph: the early bucket's RCCL all-reduce on the side branch under the layer-0 "
                                   "backward, the late bucket and the optimiser (device-side step count) behind it"
                                   if (dist.is_initialized() and dist.get_backend() == "nccl" and _dp_capture_on()) else
                                   "after the replayed forward + backward graph as ONE flat bucket; optimiser eager (the default since round 5; "
                                   "--dp-capture 1 records both into the graph)")
                           if "staged_dp" in forms_timed else "overlapped with backward",
                           "; SHARDED UPDATE: reduce-scatter -> Adam on 1 / N of the flat parameters -> all-gather of the weights "
                           "(--dp-sharded-update 1) instead of all-reduce + full Adam" if getattr(strat, "sharded", None) is not None else "",
                           " — FORCED through a world-size-1 RCCL group (--force-dist)" if args.force_dist else ""),
                       "avg_unique_input_nodes_n0": round(n0_avg, 1), "avg_n1": round(n1_avg, 1), "setup_s": round(setup_s, 1)},
            "roofline": roof_gemm if roof_gemm else roof_aggr,
            "roofline_aggregator": roof_aggr,
            "roofline_mean_backward": roof_mean_bwd,
            "roofline_step": (dict(composite, ms_per_step=round(1000 * elapsed / args.steps, 4),
                                   frac=round(composite["composite_floor_ms"] / (1000 * elapsed / args.steps), 4)) if composite else None),
            "hbm_copy_measured": hbm_copy,
            "host_enqueue_ms_per_step": round(host_ms, 4),
            "collectives": collectives,
            "collectives_variants": collectives_variants,
            "one_rank_reference": one_rank_reference,
            "graph_mode": graph_mode,
            "end_to_end_snapshot": e2e,
            "cpu_baseline": cpu_baseline,
            "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in sorted(kernels.items())},
        }
    else:
        line = None
    if want_variants:
        import threading
        done = {"v": False}

        def emit(note=None):
            if done["v"]:
                return
            done["v"] = True
            if rank == 0:
                line["collectives_variants"] = collectives_variants
                if note:
                    line["collectives_variants_note"] = note
                print(json.dumps(line))
                sys.stdout.flush()

        def bail():                                           # (every rank: a hung collective never returns to Python)
            emit("watchdog: a variant did not finish within %d s; the line carries what was measured before it" % args.variants_timeout)
            os._exit(0)
        dog = threading.Timer(args.variants_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            measure_variants()
        except Exception as ex:
            collectives_variants["error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])
        dog.cancel()
        emit()
    elif rank == 0:
        print(json.dumps(line))
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


def _dp_capture_on():
    from ogl_amd.graphsage import model as _m
    return bool(_m.DP_CAPTURE_COLLECTIVES)


def h_nnz(g):
    return g.handle.nnz


def end_to_end_snapshots(n_snapshots=6, start=4500, gemm="auto"):
    """Metric (iii) of SURVEY 8(d): whole snapshots of the reference's loop (R/train/__main__.py:161-196) on the Reddit-shaped
    stream — RBR update 50 x 512, PBR update 50 x 512 with its priority forward over the train set every 2nd snapshot, the
    no-rehearsal update, one evaluation, then evolve() of both streams and the loop's gc.collect().  Every phase is timed on
    the wall clock between device synchronisations; `delay` is the reference's own per-strategy metric
    (R/train/graphsage/model.py:108-117).  Returns the dict that goes into the bench line."""
    import gc
    import random
    import tempfile
    from ogl_amd import ops, sampling, synthetic
    from ogl_amd.graph import TrainTestGraph
    from ogl_amd.prioritized_replay import LossPriority
    from ogl_amd.utils import Lib_supported, init
    cfg = dict(embedding_size=600, latent_dim=600, samples=25, batch_size=512, batch_timestep=50, delta=4, batch_full=1024,
               priority_forward=2, snapshots=5000)
    np.random.seed(1); random.seed(1); torch.manual_seed(1); sampling.seed(1)
    ops.set_gemm_mode(gemm)
    GraphSAGE, Random, Prioritized, NoReh, _Full, act = init(Lib_supported.HIP, True, 0)
    t0 = time.perf_counter()
    feat_size, labels, graph, n_classes, graph_test = synthetic.load("reddit", snapshots=cfg["snapshots"])
    for _ in range(cfg["delta"]):
        graph_test.evolve()
    for _ in range(start):                                   # the device CSR makes fast-forwarding O(1) per snapshot
        graph.evolve(); graph_test.evolve()
    gu = TrainTestGraph(graph, split=0.15, start_prior_alpha=4, end_prior_alpha=50, scale=1, max_priority=10)
    gu._admit([int(v) for v in range(graph.get_graph().n_present) if v in graph.labelled_vertices])   # the fast-forwarded history
    setup_s = time.perf_counter() - t0

    def mk():
        return GraphSAGE(feat_size, cfg["embedding_size"], n_classes, 1, act, 0, "pool", edge_feats=0, pool_feats=cfg["latent_dim"]).cuda()
    kw = dict(cuda=True, batch_full=cfg["batch_full"], n_workers=0)
    rnd = Random(mk(), cfg["batch_timestep"], cfg["batch_size"], labels, cfg["samples"], **kw)
    pri = Prioritized(mk(), cfg["batch_timestep"], cfg["batch_size"], labels, cfg["samples"], LossPriority(),
                      full_pass=cfg["priority_forward"], **kw)
    nor = NoReh(mk(), cfg["batch_timestep"], cfg["batch_size"], labels, cfg["samples"], **kw)
    for st in (rnd, pri, nor):
        st.build_optimizer()
    forms = {}
    rnd.step_hook = lambda info: forms.__setitem__(info["form"], forms.get(info["form"], 0) + 1)
    out_csv = os.path.join(tempfile.gettempdir(), "ogl_bench_e2e_%d.csv" % os.getpid())
    gc.collect()
    gc.freeze()          # the stream's long-lived host state (id lists, edge tables) leaves the collector's young-to-old walks:
                         # the loop's per-snapshot gc.collect() then only looks at what the snapshot allocated
    phases = []
    sync = torch.cuda.synchronize

    def timed(rec, key, fn):
        sync(); t = time.perf_counter()
        out = fn()
        sync(); rec[key] = rec.get(key, 0.0) + 1000 * (time.perf_counter() - t)
        return out

    # Untimed snapshots first: images, code objects and allocator pools warm up in the first one; the strategies' auto policy times
    # its snapshots 3-5 (eagerly) before it decides how large batches run, and a replayed strategy captures each new size bucket on
    # its second sighting (~5 ms, a one-off per bucket over a stream of thousands of snapshots).  Six timed snapshots right behind ONE
    # warm-up snapshot (round 3) measured that transient: rbr_delay / 50 was 9 % above the micro-benchmark's step.
    WARM = 8
    for _snap in range(n_snapshots + WARM):
        if _snap == WARM:
            forms.clear()                                    # (how the TIMED snapshots' steps ran)
        rec = {}
        sync(); t_snap = time.perf_counter()
        nodes = timed(rec, "rbr_choose_vertices_host", lambda: rnd.choose_vertices(gu))
        rnd.choose_vertices = lambda _gu, _b=nodes: _b
        timed(rec, "rbr_train_gpu", lambda: rnd.train_timestep(gu))
        del rnd.choose_vertices
        rec["rbr_delay"] = 1000 * rnd.delay
        # PBR: choose_vertices = the priority forward (every 2nd snapshot: over the whole train set; else the new arrivals) + the draws
        inner = pri.recompute_priorities
        pf = {}
        pri.recompute_priorities = lambda g_, ts_: timed(pf, "t", lambda: inner(g_, ts_))
        nodes = timed(rec, "pbr_choose_vertices", lambda: pri.choose_vertices(gu))
        pri.recompute_priorities = inner
        rec["pbr_priority_forward_gpu"] = pf.get("t", 0.0)
        rec["pbr_choose_vertices_host"] = rec.pop("pbr_choose_vertices") - rec["pbr_priority_forward_gpu"]
        pri.choose_vertices = lambda _gu, _b=nodes: _b
        timed(rec, "pbr_train_gpu", lambda: pri.train_timestep(gu))
        del pri.choose_vertices
        rec["pbr_delay"] = 1000 * pri.delay
        timed(rec, "noreh_train_gpu", lambda: nor.train_timestep(gu))
        timed(rec, "evaluate_gpu", lambda: rnd.evaluate(gu, out_csv))
        timed(rec, "evolve_host", lambda: (gu.evolve(), graph_test.evolve()))
        timed(rec, "gc_collect_host", gc.collect)
        sync(); rec["wall"] = 1000 * (time.perf_counter() - t_snap)
        rec["train_vertices"] = len(gu.train_set)        # (not get_train_set(): that materialises the list form)
        phases.append(rec)
    gc.unfreeze()
    try:
        os.remove(out_csv)
    except OSError:
        pass
    use = phases[WARM:]
    mean = lambda k: float(np.mean([r.get(k, 0.0) for r in use]))          # noqa: E731
    gpu_keys = [k for k in use[0] if k.endswith("_gpu")]
    host_keys = [k for k in use[0] if k.endswith("_host")]
    gpu_ms, wall = sum(mean(k) for k in gpu_keys), mean("wall")
    seeds = cfg["batch_timestep"] * cfg["batch_size"]
    return dict(
        what="reference loop body per snapshot (R/train/__main__.py:161-196) on the Reddit-like stream at snapshot %d..%d: RBR 50x512 + "
             "PBR 50x512 (priority forward over the train set every 2nd snapshot, batch_full %d) + no-rehearsal + one evaluation of the "
             "test set + evolve of both streams + gc.collect; wall ms between device synchronisations" % (start, start + n_snapshots, cfg["batch_full"]),
        snapshots=len(use), warmup_snapshots=WARM, rbr_ms_per_step_inside_the_loop=round(mean("rbr_delay") / cfg["batch_timestep"], 4),
        rbr_step_execution=dict(getattr(rnd, "staged_auto_probe", None) or {}, forms=dict(forms)),
        wall_ms_per_snapshot=round(wall, 2), gpu_bound_phases_ms=round(gpu_ms, 2),
        gpu_bound_share=round(gpu_ms / wall, 4), host_only_ms=round(wall - gpu_ms, 2),
        phases_ms={k: round(mean(k), 2) for k in gpu_keys + host_keys},
        rbr_delay_ms=round(mean("rbr_delay"), 2), pbr_delay_ms=round(mean("pbr_delay"), 2),
        rbr_trained_vertices_per_s_inside_the_loop=round(seeds / (mean("rbr_delay") / 1000), 1),
        streamed_vertices_per_s_whole_snapshot=round(2 * seeds / (wall / 1000), 1),
        train_set=int(use[-1]["train_vertices"]), setup_s=round(setup_s, 1))


def pbr_snapshot_bench(args, wl):
    """BASELINE config 3 as whole snapshots: what `PrioritizedPytorchSupervisedGraphSage` does per snapshot of the reference's loop
    (R/train/__main__.py:161-196; R/train/graphsage/pytorch/model.py:153-159,210-254): choose_vertices = the priority forward over the
    train set (every `priority_forward`-th snapshot; batches of batch_full, inference, per-seed CE -> LossPriority -> the replay buffer)
    + the prioritised draws, then the train update (batch_timestep batches), then evolve().  --steps = timed snapshots, --warmup =
    untimed ones (at least 6: captures, the auto policy).  Wall clock between device synchronisations, phases itemised."""
    import gc
    import random
    from ogl_amd import ops, sampling, synthetic
    from ogl_amd.graph import TrainTestGraph
    from ogl_amd.prioritized_replay import LossPriority
    from ogl_amd.utils import Lib_supported, init
    np.random.seed(1); random.seed(1); torch.manual_seed(1); sampling.seed(1)
    ops.set_gemm_mode(args.gemm)
    GraphSAGE, _Random, Prioritized, _NoReh, _Full, act = init(Lib_supported.HIP, True, 0)
    t0 = time.perf_counter()
    feat_size, labels, graph, n_classes, _graph_test = synthetic.load(wl["dataset"], snapshots=wl["snapshots"])
    for _ in range(wl["start"]):                             # (the device CSR makes fast-forwarding O(1) per snapshot)
        graph.evolve()
    gu = TrainTestGraph(graph, split=0.15, start_prior_alpha=4, end_prior_alpha=50, scale=1, max_priority=10)
    present = np.asarray(gu.get_subgraph_to_original_map()[np.arange(graph.get_graph().n_present)]).reshape(-1)   # (original ids)
    gu._admit([int(v) for v in present if int(v) in graph.labelled_vertices])
    setup_s = time.perf_counter() - t0
    model = GraphSAGE(feat_size, wl["hidden"], n_classes, 1, act, 0, "pool", edge_feats=0, pool_feats=wl["hidden"]).cuda()
    pri = Prioritized(model, wl["batch_timestep"], wl["batch"], labels, wl["samples"], LossPriority(), full_pass=wl["priority_forward"],
                      cuda=True, batch_full=wl["batch_full"], n_workers=0)
    pri.build_optimizer()
    forms = {}
    pri.step_hook = lambda info: forms.__setitem__(info["form"], forms.get(info["form"], 0) + 1)
    gc.collect(); gc.freeze()
    sync = torch.cuda.synchronize

    def timed(rec, key, fn):
        sync(); t = time.perf_counter()
        out = fn()
        rec[key + "_enqueue"] = rec.get(key + "_enqueue", 0.0) + 1000 * (time.perf_counter() - t)     # the host's share: fn() returned
        sync(); rec[key] = rec.get(key, 0.0) + 1000 * (time.perf_counter() - t)
        return out

    warm = max(6, args.warmup)
    phases = []
    for snap in range(warm + args.steps):
        if snap == warm:
            forms.clear()
        rec = {}
        sync(); t_snap = time.perf_counter()
        inner = pri.recompute_priorities
        pf = {}
        pri.recompute_priorities = lambda g_, ts_: timed(pf, "t", lambda: inner(g_, ts_))
        nodes = timed(rec, "choose_vertices", lambda: pri.choose_vertices(gu))
        pri.recompute_priorities = inner
        rec["priority_forward_gpu"] = pf.get("t", 0.0)
        rec["priority_forward_host_enqueue"] = pf.get("t_enqueue", 0.0)
        rec["choose_vertices_host"] = rec.pop("choose_vertices") - rec["priority_forward_gpu"]
        pri.choose_vertices = lambda _gu, _b=nodes: _b
        timed(rec, "train_gpu", lambda: pri.train_timestep(gu))
        del pri.choose_vertices
        rec["train_vertices"] = len(gu.train_set)        # (not get_train_set(): that materialises the list form)
        timed(rec, "evolve_host", lambda: gu.evolve())
        timed(rec, "gc_collect_host", gc.collect)
        sync(); rec["wall"] = 1000 * (time.perf_counter() - t_snap)
        phases.append(rec)
    gc.unfreeze()
    use = phases[warm:]
    mean = lambda k: float(np.mean([r.get(k, 0.0) for r in use]))          # noqa: E731
    wall = mean("wall")
    fwd_seeds = mean("train_vertices") / wl["priority_forward"]
    upd_seeds = wl["batch_timestep"] * wl["batch"]
    return {
        "metric": "streamed vertices/sec (PBR snapshot: priority forward over the train set + update), %s-shaped stream depth=2 samples=%d"
                  % (wl["dataset"], wl["samples"]),
        "value": round((fwd_seeds + upd_seeds) / (wall / 1000), 1), "unit": "vertices/s", "n_gpus": 1, "steps": args.steps, "warmup": warm,
        "ms_per_step": round(wall, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: the reference loop body of the PBR strategy per snapshot on the %s-like stream at snapshots %d..%d "
                               "(train set %d vertices): priority forward over the train set every %s snapshot (batch_full %d, inference, "
                               "per-seed CE -> LossPriority -> replay buffer on the device), prioritised draw, %d x %d-seed update, evolve, "
                               "gc.collect; a step = one snapshot" % (args.workload, wl["dataset"], wl["start"] + warm,
                                                                        wl["start"] + warm + args.steps, int(use[-1]["train_vertices"]),
                                                                        {1: "", 2: "2nd"}.get(wl["priority_forward"], "n-th"), wl["batch_full"],
                                                                        wl["batch_timestep"], wl["batch"]),
                   "gemm_arithmetic": gemm_desc(args.gemm), "setup_s": round(setup_s, 1)},
        "phases_ms": {k: round(mean(k), 3) for k in ("priority_forward_gpu", "priority_forward_host_enqueue", "choose_vertices_host", "train_gpu",
                                                     "evolve_host", "gc_collect_host")},
        "priority_forward_vertices_per_s": round(fwd_seeds / (mean("priority_forward_gpu") / 1000), 1) if mean("priority_forward_gpu") > 0 else None,
        "update_step_execution": dict(forms),
        "roofline": None, "cpu_baseline": None,
        "note": "an end-to-end line (metric iii of SURVEY 8(d)): its kernels' roofline figures are in the arxiv_pbr_forward and arxiv_rbr lines",
    }


def gemm_desc(mode):
    return {"f32": "fp32 MFMA (exact fp32 fma chain)",
            "bf16x6": "split-bf16 x6 MFMA, fp32 accumulate (fp32-GEMM accuracy, same test tolerances)",
            "auto": "split-bf16 x6 MFMA with fp32 accumulate everywhere it is faster (fp32-GEMM accuracy, same test tolerances): "
                    "every product with >= 2048 rows (layer-0 and n1-row forwards, input gradients, all weight gradients) on pre-split "
                    "bf16x3 images (k_gemm_x3p: producer / consumer waves, LDS-DMA staged; activation images written by the kernels that "
                    "produce the activations, weight gradients read them k-major), exact fp32 (MFMA forward, vector-ALU backward) for "
                    "the 512-row output layer"}[mode]


def forward_bench(args, wl, g, model, train_set, world, rank, arrays, feat_size, n_classes, setup_s):
    """PBR priority forward: K batches of batch_full seeds, eval mode, per-seed CE loss returned to the host once."""
    from ogl_amd import ops, parallel
    from ogl_amd.graphsage.model import HipSupervisedGraphSage
    B, S = wl["batch"], wl["samples"]
    strat = HipSupervisedGraphSage(model, wl["batch_timestep"], 32, None, S, reduction="none", cuda=True, batch_full=B)
    strat.cache_projection = not args.no_projection_cache
    strat.partition_features = args.partition == "features"
    model.eval()
    strong = args.scaling == "strong" and world > 1
    total_batches = lambda nb: nb if strong else nb * world     # noqa: E731  (weak: nb batches PER RANK)

    def run(nb):
        # ONE pass over the (replicated) seed list; whole batches are block-partitioned over the ranks (parallel.batch_shard)
        n_all = total_batches(nb) * B
        seeds = torch.as_tensor(np.resize(train_set, n_all))
        losses = []
        with torch.no_grad():
            for sd, scores in strat._inference_batches(g, seeds, shard=True):
                labels = ops.gather_i64(g.ndata["target"], sd)
                rows, _ = ops.ce_fwd_bwd(scores, labels, want_grad=False)
                losses.append(rows)
        local = torch.cat(losses) if losses else torch.zeros(0, device="cuda")
        if world > 1:                     # the exchange step of the sharded pass: every replay-buffer replica gets every loss
            counts = [parallel.batch_shard(n_all, B, r, world)[3] - parallel.batch_shard(n_all, B, r, world)[2] for r in range(world)]
            local = parallel.all_gather_counts(local, counts)
        return local.cpu()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(max(args.warmup, 1))
    barrier()
    t1 = time.perf_counter()
    out = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ops.profile_start()
    run(min(args.steps, 20))
    rec = ops.profile_stop()
    nprof = min(args.steps, 20)
    agg = {}
    for name, meta, ms in rec:
        key = name
        if name in ("ogl_reduce_fwd", "ogl_reduce_fwd_img"):
            # (layer 0 of a cached pass reads the table through the sampler's global int64 picks; several batches share a launch)
            key = "reduce_fwd_L0" if (meta["idx_bytes"] == 8 or meta["n_dst"] > 64 * B) else "reduce_fwd_L1"
        elif name.startswith("ogl_linear_fwd") and meta["M"] == g.n_present:
            key = name[4:] + "_tables"                     # the per-pass projection tables P0 / S0 over every present vertex
        a = agg.setdefault(key, dict(ms=0.0, calls=0, bytes=0.0, flops=0.0))
        a["ms"] += ms; a["calls"] += 1
        if name in ("ogl_reduce_fwd", "ogl_reduce_fwd_img"):
            E = meta["n_dst"] * meta["fanout"]
            a["bytes"] += E * (4 * meta["d"] + meta["idx_bytes"]) + meta["n_dst"] * 4 * meta["d"] * ((2.5 if meta.get("out", True) else 1.5) if name.endswith("_img") else 1)
        if name in ("ogl_linear_fwd", "ogl_linear_fwd_x3", "ogl_linear_fwd_x3_ext"):
            a["flops"] += 2.0 * meta["M"] * meta["N"] * (meta["K"] + meta["K2"])
    kernels = {k: dict(ms_per_step=round(v["ms"] / nprof, 4), calls_per_step=round(v["calls"] / nprof, 2),
                       gbs=round(v["bytes"] / v["ms"] / 1e6, 1) if v["bytes"] else None,
                       tflops=round(v["flops"] / v["ms"] / 1e9, 2) if v["flops"] else None) for k, v in sorted(agg.items())}
    ragg = agg.get("reduce_fwd_L0")
    roof = None
    if ragg:
        # per-LAUNCH figures are those of the pass's FULL chunks (consecutive batches fused up to FUSE_INFERENCE_ROWS hidden-layer
        # rows; a pass ends with a partial one) — what the PMC pass of this workload is reduced to as well
        l0 = [(n_, m, ms) for n_, m, ms in rec
              if n_ in ("ogl_reduce_fwd", "ogl_reduce_fwd_img") and (m["idx_bytes"] == 8 or m["n_dst"] > 64 * B)]
        top = max(m["n_dst"] for _, m, _ in l0)
        fullc = [(n_, m, ms) for n_, m, ms in l0 if m["n_dst"] >= 0.85 * top]
        full_bytes = sum(m["n_dst"] * m["fanout"] * (4 * m["d"] + m["idx_bytes"]) + m["n_dst"] * 4 * m["d"] * ((2.5 if m.get("out", True) else 1.5) if n_.endswith("_img") else 1)
                         for n_, m, _ in fullc) / len(fullc)
        full_ms = sum(ms for _, _, ms in fullc) / len(fullc)
        per_launch = top / (sum(m["n_dst"] for _, m, _ in l0) / nprof)
        ach = ragg["bytes"] / ragg["ms"] / 1e6
        narrow = fullc[0][1]["d"] <= 128 and args.reduce_half != 0
        roof = dict(kernel="%s (layer-0 max over the cached projection table, int64 global picks; a full chunk = %.1f batches "
                           "of %d seeds per launch)" % ("k_reduce_fwd_max_half: two <= 512-byte rows per wave-instruction" if narrow else "k_reduce_fwd_v4",
                                                        per_launch, B), bound="hbm",
                    achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                    avg_launch_ms=round(full_ms, 4),
                    algorithmic_bytes_per_launch=round(full_bytes),
                    frac_kind="algorithmic bytes / launch time / 8 TB/s (SURVEY 8(d)): rows gathered again within a chunk are re-served by "
                              "L2 / the Infinity Cache and still count, so frac can exceed 1 — it is NOT a pin rate",
                    note="`traffic` (PMC FETCH_SIZE x 2 + WRITE_SIZE) is the L2's fabric-side request bytes per launch: Infinity-Cache (MALL) "
                    "hits are included (MI355X_MICROARCH.md), so traffic / time is an upper bound on the HBM rate (a plain streamed copy "
                    "reaches ~6.3 TB/s on this part; anything above that in traffic / time is MALL-served)")
    tkeys = [k for k in agg if k.endswith("_tables")]
    table_build = None
    if tkeys:
        tms = sum(agg[k]["ms"] for k in tkeys); tfl = sum(agg[k]["flops"] for k in tkeys); tcalls = sum(agg[k]["calls"] for k in tkeys)
        x6 = args.gemm != "f32"
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if x6 else MFMA_F32_PEAK_TFLOPS
        table_build = dict(what="P0 = relu(fc_pool0(X)) and S0 = fc_self0(X) + biases over all %d present vertices: two products, once per PASS "
                                "(weights are fixed during it), amortised over the pass's batches" % g.n_present,
                           launches_per_pass=tcalls, ms_per_pass=round(tms, 4), ms_per_launch=round(tms / max(tcalls, 1), 4),
                           tflops=round(tfl / tms / 1e9, 2) if tms else None, bound="mfma", peak=round(peak, 1),
                           frac=round(tfl / tms / 1e9 / peak, 4) if tms else None)
        table_build["share_of_the_kernel_time_of_a_%d_batch_pass" % nprof] = round(tms / sum(v["ms"] for v in agg.values()), 4)
    # HBM-side bytes per launch of the table-direct aggregator, from the committed rocprofv3 --pmc pass of this workload
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_PBR_TRAFFIC_FILE)))
        ent = pmc.get(args.workload, {}).get("k_reduce_fwd_v4_L0")
        if roof and ent:
            roof["traffic"] = ent["traffic_bytes"]
            roof["traffic_source"] = "profiles/%s, collected %s at HEAD %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload; a " \
                                     "constant quoted from that pass, NOT measured in this run)" % (PMC_PBR_TRAFFIC_FILE, pmc.get("collected", "?"), pmc.get("head", "?"))
    except Exception:
        pass
    if roof:
        # `frac` is a roofline fraction only on bytes that crossed the fabric: the ALGORITHMIC rate (rows gathered again within a chunk are
        # re-served by L2 / the Infinity Cache and still count: it exceeds the pin rate) moves under its own key
        roof["algorithmic"] = dict(achieved=roof["achieved"], frac=roof["frac"], unit="GB/s",
                                   kind="SURVEY 8(d) bytes per launch / launch time: cache re-serves count, so this may exceed 8 TB/s — not a roofline fraction")
        if roof.get("traffic") and roof["avg_launch_ms"] > 0:
            fab = roof["traffic"] / roof["avg_launch_ms"] / 1e6
            roof["achieved"], roof["frac"] = round(fab, 1), round(fab / HBM_PEAK_GBS, 4)
            roof["frac_kind"] = ("fabric-side bytes per full-chunk launch (PMC FETCH_SIZE x 2 + WRITE_SIZE of the committed pass; Infinity-Cache "
                                 "hits included) / this run's launch time / 8 TB/s")
        else:
            roof["achieved"], roof["frac"] = None, None
            roof["frac_kind"] = "no PMC traffic for this workload in profiles/: only the algorithmic rate is known"
    if rank == 0:
        assert out.numel() == total_batches(args.steps) * B and bool(torch.isfinite(out).all())
        print(json.dumps({
            "metric": "streamed vertices/sec (PBR priority forward), %s-shaped stream depth=2 samples=%d" % (wl["dataset"], S),
            "value": round(total_batches(args.steps) * B / elapsed, 1), "unit": "vertices/s", "n_gpus": world,
            "ranks_seen": args.ranks_seen, "dist_backend": (args.dist_backend if world > 1 else None), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"gemm_arithmetic": gemm_desc(args.gemm), "workload": "%s: last snapshot (N=%d), F=%d H=%d C=%d, pool(max), batch_full=%d/GPU, inference sample+forward+CE(none), "
                                   "projection cache %s" % (args.workload, g.n_present, feat_size, wl["hidden"], n_classes, B,
                                                            "on" if strat.cache_projection else "off"),
                       "global_batch": B * world,
                       "parallelism": "dp%d (whole batches of the pass block-partitioned over the ranks%s; projection tables %s)" % (
                           world, ", per-seed losses all-gathered to every rank" if world > 1 else "",
                           "built per vertex range + halo all-gather (partitioned features)" if (strat.partition_features and world > 1)
                           else "built on every rank (replicated features)"),
                       "setup_s": round(setup_s, 1)},
            "roofline": roof, "table_build": table_build, "cpu_baseline": None, "kernels": kernels}))
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
