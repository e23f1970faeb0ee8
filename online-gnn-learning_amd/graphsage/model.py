"""Strategy layer: the classes ``utils.init`` hands to the driver loop.

Mirrors R/train/graphsage/model.py:18-117 (backend-agnostic base: timing, evaluation, CSV rows) and
R/train/graphsage/pytorch/model.py:12-323 (RBR / PBR / no-rehearsal / offline strategies): same
constructor arguments, method names, return values and quiet early-returns.  The per-batch body is
the MI355X path: GPU sampler + block builder, row gather fused into the projection GEMM, HIP
aggregator, HIP cross-entropy and Adam — and the loader samples a whole snapshot's batches up front.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from .. import ops, optim, parallel, sampling, stepgraph
from .sageconv import GatheredRows


# OGL_DP_CAPTURE_COLLECTIVES=1 (or ``model.DP_CAPTURE_COLLECTIVES = True``; bench.py --dp-capture 1): a replica's WHOLE step — both RCCL
# all-reduces, launched from the gradient hooks on the side branch, and the optimiser — is recorded into the replayed hipGraph (round 4:
# 1.011 ms per step against 1.085 for the form below, measured through a world-size-1 RCCL group).  OFF by default since round 5: that
# form has run on ONE rank only (tests/test_gpu_nccl.py), and with two or more ranks each rank decides from its OWN block sizes whether a
# step is a replay, a capture (a size bucket's second sighting) or the eager twin — the collective SEQUENCE is the same in all three, but
# a replayed collective meeting an eagerly enqueued one on a peer is exactly what no multi-GPU run has covered yet, and a mismatch there
# is a hang, not an error.  The default keeps the exchange and the optimiser OUT of the graph (round 3's form: forward + backward
# replayed, ONE flat-bucket all-reduce and Adam enqueued from Python — every rank enqueues its collectives the same way).
DP_CAPTURE_COLLECTIVES = __import__("os").environ.get("OGL_DP_CAPTURE_COLLECTIVES", "0") == "1"
# OGL_SHARDED_FUSED=0: the eager replica step differentiates sum(rows) / n_global through the unfused output layer (rounds 1-4)
SHARDED_FUSED = __import__("os").environ.get("OGL_SHARDED_FUSED", "1") != "0"
# The 32-seed rungs' captured steps sample batch i + 1 on a second stream while batch i trains (stepgraph.sampled_step_pipelined);
# OGL_SAMPLE_PIPELINE=0: sample, read back, train, one after the other.
SAMPLE_PIPELINE = __import__("os").environ.get("OGL_SAMPLE_PIPELINE", "1") != "0"


def _to_numpy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def macro_f1_from_confusion(cm_full):
    """sklearn semantics of R/train/graphsage/model.py:85-87 from a full C x C count matrix: classes that occur neither as
    a label nor as a prediction are dropped (``confusion_matrix`` / ``f1_score(average='macro')`` use the labels present),
    a class with no predicted and no true samples... cannot remain after that; F1 of a class with zero denominator is 0.
    Returns (macro F1, flattened confusion matrix over the present classes)."""
    cm_full = np.asarray(cm_full, dtype=np.int64)
    present = np.nonzero((cm_full.sum(0) + cm_full.sum(1)) > 0)[0]
    cm = cm_full[np.ix_(present, present)]
    tp = np.diag(cm).astype(np.float64)
    denom = cm.sum(0) + cm.sum(1)                     # 2 TP + FP + FN
    f1 = np.where(denom > 0, 2 * tp / np.maximum(denom, 1), 0.0)
    return float(f1.mean()) if len(f1) else 0.0, [int(x) for x in cm.reshape(-1)]


class SupervisedGraphSage:
    """Base class (R/train/graphsage/model.py:18-117)."""

    def __init__(self, graphsage_model, batch_per_timestep, batch_size, labels, samples, n_workers, cuda, batch_full):
        self.graphsage_model = graphsage_model
        self.batch_size = batch_size
        self.batch_per_timestep = batch_per_timestep
        self.samples = samples
        self.n_workers = n_workers
        self.labels = labels
        self._cuda_var = cuda
        self.batch_full = batch_full
        self.amount_of_train = {}
        self.delay = 0.0
        self.fuse_gather = True     # read feature rows straight from the resident table inside the GEMM
        self.cache_projection = True  # inference passes reuse the layer-0 projection tables across batches
        self.partition_features = False  # N ranks: build those tables per vertex range + halo all-gather (parallel.py)
        # one rank: train steps as captured hipGraphs (stepgraph.py).  True / False force it; "auto" (default) captures the
        # small batches always (sampled form: launch- and sync-bound otherwise) and the large, loader-fed ones (staged
        # form) only when a timed snapshot shows the HOST cannot keep ahead of the GPU: replayed nodes run ~1 us further
        # apart than eagerly queued launches (1.28 vs 1.23 ms per Reddit step on a fast host), so eager is the faster
        # mode while the host's enqueue time per step stays under the GPU's time per step
        self.use_graphs = "auto"
        self._staged_auto = None    # "auto": None = undecided, True / False after the probe
        self._staged_seen = 0       # eligible snapshots met so far (the first ones are cold: images, code objects, allocator)
        self._staged_probes = []    # (host seconds, GPU seconds) of the probed snapshots of the current decision
        self._staged_decided_at = 0
        self.staged_auto_log = []   # every decision taken (the run log of the policy)
        self.STAGED_AUTO_HOST_FRACTION = 0.75     # replayed steps cost the GPU the same as eager ones since round 3 (forked branches
                                                  # inside the graph): replay as soon as the host is within 25 % of being the bottleneck
        self.STAGED_AUTO_PROBES = 3       # snapshots timed per decision; the MEDIAN host / GPU ratio decides (one snapshot's wall
                                          # time carries allocator / GC hiccups and back-pressure)
        self.STAGED_AUTO_REPROBE = 500    # eligible snapshots after which the decision is re-taken (0: never) — the graph grows
                                          # and the host's load changes over a 5 000-snapshot stream.  use_graphs = True / False
                                          # is the explicit override (bench.py --graphs / --no-graphs)
        self.step_hook = None       # instrumentation (tests, bench): called after every train step with a dict

    def build_optimizer(self):
        raise NotImplementedError

    def evaluate(self, graph_util, path):
        return self._evaluate_vertices(graph_util, path, np.array(graph_util.get_test_set()))

    def evaluate_next_snapshots(self, temporal_graph, delta, path, at_least=20):
        new_vertices, labelled = temporal_graph.get_added_vertices(delta)
        test = np.array(new_vertices)[np.asarray(labelled, dtype=bool)]
        if len(test) < at_least:
            if parallel.rank_world()[0] == 0:
                with open(path, "a+") as f:
                    f.write(self.get_model() + ";;;\n")
            return
        return self._evaluate_vertices(temporal_graph, path, test)

    def _evaluate_vertices(self, graph_util, path, batch_nids):
        id_to_subgraph = graph_util.get_original_to_subgraph_map()
        subgraph_to_id = graph_util.get_subgraph_to_original_map()
        graph = graph_util.get_graph()
        vertices = id_to_subgraph[batch_nids]
        res = self._eval_confusion(graph, subgraph_to_id, id_to_subgraph, vertices)
        if res is None:
            return
        cm_full, n = res
        if n == 0:
            return
        f1, cm = macro_f1_from_confusion(cm_full)
        if parallel.rank_world()[0] == 0:          # N ranks hold the same (all-reduced) counters: one row per evaluation
            with open(path, "a+") as f:
                f.write(self.get_model() + ";" + str(f1) + ";" + str(self.delay) + ";" + str(cm) + "\n")
        return f1

    def _eval_confusion(self, graph, subgraph_to_id, id_to_subgraph, vertices):
        """Default (backend-agnostic) path: logits to the host, argmax + confusion matrix there."""
        output_data = self._run_custom_eval(graph, subgraph_to_id, id_to_subgraph, vertices)
        if len(output_data) == 0:
            return None
        output_data = np.concatenate(output_data)
        if len(output_data) == 0:
            return None
        vt = torch.as_tensor(np.asarray(vertices), dtype=torch.int64).to(graph.device)
        labels = _to_numpy(ops.gather_i64(graph.ndata["target"], vt))
        pred = output_data.argmax(axis=1)
        C = output_data.shape[1]
        cm = np.zeros((C, C), dtype=np.int64)
        ok = (labels >= 0) & (labels < C)
        np.add.at(cm, (labels[ok], pred[ok]), 1)
        return cm, len(pred)

    def _run_custom_eval(self, graph, subgraph_to_id, id_to_subgraph, test_vertices):
        raise NotImplementedError

    def get_model(self):
        return "base_model"

    def choose_vertices(self, graph_util):
        raise NotImplementedError

    def train_timestep(self, graph_util):
        batch_nodes = self.choose_vertices(graph_util)
        start = time.time()
        id_to_subgraph = graph_util.get_original_to_subgraph_map()
        subgraph_to_id = graph_util.get_subgraph_to_original_map()
        graph = graph_util.get_graph()
        self._run_custom_train(graph, subgraph_to_id, id_to_subgraph, id_to_subgraph[batch_nodes], graph_util)
        torch.cuda.synchronize()            # the reference's delay includes the device work (it syncs per batch)
        self.delay = time.time() - start


class HipSupervisedGraphSage(SupervisedGraphSage):
    """Counterpart of PytorchSupervisedGraphSage (R/train/graphsage/pytorch/model.py:12-108)."""

    def __init__(self, graphsage_model, batch_per_timestep, batch_size, labels, samples, reduction="mean", n_workers=1,
                 cuda=True, batch_full=512):
        super().__init__(graphsage_model, batch_per_timestep, batch_size, labels, samples, n_workers, cuda, batch_full)
        if not cuda:
            raise RuntimeError("the hip backend runs on the GPU only: construct it with cuda=True")
        self.reduction = reduction
        self.xent = lambda scores, labels_: ops.cross_entropy(scores, labels_, reduction)
        self.gsync = None                      # set by build_optimizer() under torch.distributed

    # a batch whose upper-bound input block B (1 + S)^2 stays below this many rows samples inside a captured graph of its own
    # (one 16-byte read-back per step picks the train graph's size bucket); larger batches keep the loader, whose read-backs
    # amortise over the snapshot's batches, and stage each batch into its bucket's train graph (stepgraph.py)
    # (R/settings/pubmed.json and elliptic.json use samples = 45 at batch 32: 32 * 46^2 = 67 712 rows of upper bound — the buffers
    # are index arrays, 0.5 MB; the GEMMs of the step run at the batch's own size bucket, not at the bound)
    SAMPLED_GRAPH_MAX_ROWS = 1 << 18
    # data-parallel replicas replay their forward + backward (form "staged_dp") from this many rows of GLOBAL upper bound
    # n_global (1 + S) on: below it the staged form's size buckets (2 048 / 256 rows) would mostly multiply padding
    STAGED_DP_MIN_ROWS = 4096

    def build_optimizer(self):
        self._sg = None
        # (a replica's whole step — exchange included — is one captured graph too: the optimiser keeps its step count on the device;
        # its own two-part form is for the one-rank step: under data parallelism the early part would run before the all-reduce)
        capturable = bool(self.use_graphs)
        # OGL_DP_SHARDED_UPDATE=1 (parallel.SHARDED_UPDATE; bench.py --dp-sharded-update 1): a replica's exchange + optimiser as
        # reduce-scatter -> Adam on this rank's 1 / N of the parameters -> all-gather of the weights (parallel.ShardedAdam; it REBASES
        # the parameters into one flat buffer, so it comes before everything that keys on their addresses)
        self.sharded = parallel.ShardedAdam(self.graphsage_model.parameters(), lr=0.001) \
            if (parallel.is_distributed() and parallel.SHARDED_UPDATE) else None
        self.optimizer = optim.Adam(self.graphsage_model.parameters(), lr=0.001, capturable=capturable,
                                    early=False if parallel.is_distributed() else None)
        # Under torch.distributed (one process per GPU, identical replicas, identical host RNG seeds on every rank)
        # each rank trains on its shard_range slice of every replay batch; the weighted gradient all-reduce makes the
        # update that of the whole batch, and the sharded PBR passes all-gather their per-seed losses (parallel.py).
        # (the loss of a shard already carries 1 / n_global, hence weight 1; the two-bucket overlap works for ragged and
        # empty shards alike: every rank issues the same two collectives per step)
        self.gsync = parallel.GradSynchronizer(self.graphsage_model.parameters(), overlap=self.sharded is None, weight=1.0) \
            if parallel.is_distributed() else None

    def optimizer_state_dict(self):
        """The state a resumed stream needs beside the model's ``state_dict`` (R/export_model.py:107 saves the weights only): the
        optimiser's moments and step count — from the sharded update when it is on (every rank must call this then: its moments
        are gathered from their owning ranks), else from ``optim.Adam``."""
        if getattr(self, "sharded", None) is not None:
            return dict(kind="sharded", state=self.sharded.state_dict())
        return dict(kind="adam", state=self.optimizer.state_dict())

    def load_optimizer_state_dict(self, sd):
        if sd["kind"] == "sharded":
            if getattr(self, "sharded", None) is None:
                raise RuntimeError("this checkpoint was taken with the sharded update (OGL_DP_SHARDED_UPDATE=1): build the optimiser the same way")
            self.sharded.load_state_dict(sd["state"])
        else:
            self.optimizer.load_state_dict(sd["state"])

    def _exchange_and_step(self, weight=None, single=False):
        """The end of a replica's step: grads <- sum_r weight_r * grad_r, then the optimiser — or, with the sharded update on, the
        reduce-scatter / sharded Adam / all-gather that replaces both (the same collectives on every rank either way)."""
        if getattr(self, "sharded", None) is not None:
            self.sharded.step(self.gsync.weight if weight is None else weight)
            return
        self.gsync.sync(weight=weight, single=single)
        self.optimizer.step()

    def _local_batches(self, graph, seeds, batch_size, shuffle=False):
        """The snapshot's train batches as this rank sees them: yields (input_nodes, local_seeds, blocks, n_global).
        One rank: the reference's NodeDataLoader batches.  N ranks: every batch is cut by ``parallel.shard_range`` in seed
        order (the sampler is keyed per seed, so a vertex draws the same neighbours on whichever rank it lands)."""
        seeds = torch.as_tensor(np.asarray(seeds), dtype=torch.int64).reshape(-1)
        if shuffle:
            seeds = seeds[torch.randperm(seeds.numel())]
        rank, world = parallel.rank_world()
        bs = int(batch_size)
        if bs <= 0:
            raise ValueError("batch_size should be a positive integer value, but got batch_size={}".format(bs))
        parallel.assert_replicated(seeds, "the snapshot's train seeds")          # (a no-op on one rank)
        seeds = seeds.to(graph.device).contiguous()
        full = [seeds[s0:s0 + bs] for s0 in range(0, seeds.numel(), bs)]
        local = [b[slice(*parallel.shard_range(b.numel(), rank, world))] for b in full]
        live = [i for i, b in enumerate(local) if b.numel() > 0]
        ctrs = sampling.reserve_ctrs(len(full))           # every batch of the one-rank loader, also those empty here
        # (a generator: the first batches as soon as they are sampled, the rest sampled on a second stream while those train)
        out = self._sampler().sample_batches_stream(graph, [local[i] for i in live], ctrs=[ctrs[i] for i in live])
        for i, b in enumerate(full):
            if local[i].numel() > 0:
                input_nodes, sd, blocks = next(out)
                yield input_nodes, sd, blocks, b.numel()
            else:
                yield None, local[i], None, b.numel()         # more ranks than seeds in this batch

    def _backward_and_step(self, loss_sum_local, n_local, n_global):
        """d(mean over the GLOBAL batch) -> optimiser step.  ``loss_sum_local`` = sum of this rank's per-seed losses."""
        self.optimizer.zero_grad()
        if n_local > 0:
            (loss_sum_local / n_global).backward()
        if self.gsync is not None:
            for p in self.gsync.params:                            # a rank without seeds contributes zeros
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            self._exchange_and_step()                              # weight 1: the 1 / n_global is already in the loss
            return
        self.optimizer.step()

    def _graphs_ok(self, form="sampled"):
        """Captured steps apply on one rank, with the capturable optimiser build_optimizer() made, without dropout (its
        stream counter is host-side) and outside per-kernel profiling; the staged form under "auto" only after the probe
        of ``_train_batches`` found the host too slow to stay ahead of the GPU."""
        if not (self.use_graphs and ops._PROFILE is None and all(l.feat_drop.p == 0 for l in self.graphsage_model.layers)):
            return False
        if form == "staged_dp":
            # a data-parallel replica: forward + backward captured, ONE all-reduce and the optimiser eager.  The eager replica step
            # is host-bound (measured through a world-size-1 RCCL group, bench.py --force-dist: 1.39 ms of host time per 1.07 ms
            # step — hooks, bucket views, two collectives from Python), the replayed one is not (0.45 ms): "auto" replays.  The
            # decision must be the same on every rank (it fixes the sequence of collectives), so it is taken from configuration
            # and GLOBAL quantities only — never from a rank's own timing or block sizes (train_step: n_global).
            return self.gsync is not None and bool(self.use_graphs)
        if self.gsync is not None or not getattr(self.optimizer, "capturable", False):
            return False
        if form == "staged" and self.use_graphs == "auto":
            return bool(self._staged_auto)
        return True

    def _step_graphs(self):
        if getattr(self, "_sg", None) is None:
            if self.reduction == "mean":
                loss_fn = lambda logits, labels: (ops.cross_entropy(logits, labels, "mean"), None)       # noqa: E731
            else:
                def loss_fn(logits, labels):
                    return ops.cross_entropy_mean_rows(logits, labels)
            self._sg = stepgraph.StepGraphCache(self.graphsage_model, self.optimizer, self.samples, loss_fn,
                                                loss_kind="mean" if self.reduction == "mean" else "mean_rows")
        return self._sg

    def _train_batches(self, graph, train_vertices, batch_size, on_rows=None):
        """The snapshot's train update.  One rank, batches small enough for upper-bound shapes: every full batch is ONE
        captured step that samples for itself (no loader, no read-back); otherwise the loader + ``train_step`` (itself a
        captured step per size bucket on one rank).  ``on_rows(seeds_device, loss_rows)`` receives the per-seed losses
        (PBR).  Counters: one Philox batch counter per batch of the one-rank loader, in order, on either path."""
        bs = int(batch_size)
        n = len(train_vertices)
        if (bs > 0 and self._graphs_ok("sampled") and bs * (1 + self.samples) ** 2 <= self.SAMPLED_GRAPH_MAX_ROWS and n >= bs):
            seeds_np = np.asarray(train_vertices, dtype=np.int64).reshape(-1)
            starts = list(range(0, n, bs))
            ctrs = sampling.reserve_ctrs(len(starts))
            for i, s0 in enumerate(starts):
                sd = seeds_np[s0:s0 + bs]
                if len(sd) == bs:
                    if SAMPLE_PIPELINE and len(starts) >= 2:   # (a one-batch snapshot has nothing to sample ahead: the plain form)
                        j = i + 1
                        nxt = (seeds_np[starts[j]:starts[j] + bs], ctrs[j]) if j < len(starts) else None
                        sg = self._step_graphs().sampled_step_pipelined(graph, sd, ctrs[i], nxt)
                    else:
                        sg = self._step_graphs().sampled_step(graph, sd, ctrs[i])
                    if on_rows is not None:
                        on_rows(sg.buf.seeds.clone(), sg.loss_rows.clone())
                    if self.step_hook is not None:
                        self.step_hook(dict(seeds=sd, loss=sg.loss, grads=sg.grads, form="sampled", ctr=ctrs[i],
                                            n0=sg.last_sizes[0], n1=sg.last_sizes[1]))
                else:                                        # the ragged last batch: eager, with its own counter
                    sdev = torch.as_tensor(sd).to(graph.device)
                    (input_nodes, sdd, blocks), = self._sampler().sample_batches(graph, [sdev], ctrs=[ctrs[i]])
                    loss = self._eager_step(graph, blocks, input_nodes, sdd, on_rows)
                    if self.step_hook is not None:
                        self.step_hook(dict(seeds=sd, loss=loss.detach(), grads=[p.grad for p in self.graphsage_model.parameters()],
                                            form="eager", ctr=ctrs[i], n0=int(input_nodes.numel()), n1=blocks[1].number_of_src_nodes()))
            return
        probe = False
        if self.use_graphs == "auto" and self._graphs_ok("sampled") and bs > 0 and n >= 8 * bs:
            self._staged_seen += 1
            if (self._staged_auto is not None and self.STAGED_AUTO_REPROBE
                    and self._staged_seen - self._staged_decided_at >= self.STAGED_AUTO_REPROBE):
                self._staged_auto, self._staged_probes = None, []        # re-take the decision (this snapshot runs eagerly)
            # never time the first snapshots (cold: images, code objects, allocator pools, Python's own caches — the host
            # side warms up last)
            probe = self._staged_auto is None and self._staged_seen >= 3
        if probe:                                            # time this snapshot's eager update: host enqueue vs GPU
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_host, first = 0.0, True
        for input_nodes, seeds, blocks, n_global in self._local_batches(graph, train_vertices, bs):
            if probe and first:                              # after the loader's own read-backs
                ev0.record(); t0 = time.perf_counter(); first = False
            self.train_step(graph, blocks, input_nodes, seeds, None, n_global, on_rows=on_rows)
        if probe and not first:
            t_host = time.perf_counter() - t0
            ev1.record(); ev1.synchronize()
            t_gpu = ev0.elapsed_time(ev1) / 1e3
            self._staged_probes.append((t_host, t_gpu))
            if len(self._staged_probes) >= self.STAGED_AUTO_PROBES:
                ratios = sorted(h / max(g_, 1e-9) for h, g_ in self._staged_probes)
                med = ratios[len(ratios) // 2]
                self._staged_auto = med > self.STAGED_AUTO_HOST_FRACTION
                self._staged_decided_at = self._staged_seen
                self.staged_auto_probe = dict(host_s=round(float(np.median([h for h, _ in self._staged_probes])), 5),
                                              gpu_s=round(float(np.median([g_ for _, g_ in self._staged_probes])), 5),
                                              host_over_gpu=[round(r, 3) for r in ratios], graphs=self._staged_auto,
                                              decided_at_snapshot=self._staged_seen)
                self.staged_auto_log.append(self.staged_auto_probe)

    def _eager_step(self, graph, blocks, input_nodes, seeds, on_rows=None):
        self.optimizer.zero_grad()
        if hasattr(self.optimizer, "prime"):
            self.optimizer.prime()                     # (its per-step scalars ride in the forward's weight-image launch)
        # (the labels are gathered inside the loss launch; the last layer and the loss are one node where that applies)
        batch_labels = ops.LazyLabels(graph.ndata["target"], seeds)
        # (defer_mean: this step runs the backward itself, right below — the loss VALUE exists from its first launch on)
        loss, rows, _ = self.graphsage_model.forward_loss(blocks, self._inputs(graph, input_nodes), batch_labels,
                                                          rows=self.reduction != "mean", defer_mean=True)
        if rows is not None and on_rows is not None:
            on_rows(seeds, rows.detach())
        if hasattr(self.optimizer, "backward_and_step"):
            self.optimizer.backward_and_step(loss)     # (split-K slabs summed by the optimiser launch, its early part beside the backward)
        else:                                          # any torch optimiser
            ops.backward(loss)
            self.optimizer.step()
        return loss

    def _sampler(self):
        return sampling.MultiLayerNeighborSampler([self.samples for _ in range(2)], replace=True, return_eids=True)

    def _inputs(self, graph, input_nodes):
        if self.fuse_gather:
            return GatheredRows(graph.ndata["feat"], input_nodes)
        return ops.gather_rows(graph.ndata["feat"], input_nodes)

    # Expected unique layer-0 input rows of an inference batch as a fraction of its upper bound B (1 + S)^2: measured
    # 0.18 (Reddit-like, B = 512: n0 = 62.7 k of 346 k) to 0.3 (arxiv- / pubmed-like, B = 1024, before the n_present cap);
    # it only decides when a pass is long enough for the per-pass tables to pay — both paths give the same results.
    UNIQUE_INPUT_FRACTION = 0.3
    # inference passes against the per-pass tables: batches fused into one launch sequence up to this many hidden-layer rows
    # (0 = one launch sequence per batch of batch_full seeds, as the reference's loop).  131 072 rows x 608 floats = 320 MB per
    # intermediate: the products run as several rounds of tiles instead of a fraction of one
    FUSE_INFERENCE_ROWS = 131072

    def _projection_tables(self, graph, layer0):
        """(P0, S0) of ``SAGEConv.project_tables`` for every present vertex, once per pass (weights are fixed during it).
        One rank, or replicated mode: computed locally.  ``partition_features`` under N ranks — the partitioned-feature mode
        of the north star: each rank projects ONLY the rows of its vertex range (the pass reads no other raw feature row on
        that rank: a batch takes its self term from S0 and its neighbour term from P0), written in place into its block of
        the full tables, then ONE all-gather per table exchanges the blocks: the rows outside a rank's range are the halo
        embeddings its sampled neighbourhoods reach into (on a power-law graph every batch does)."""
        feat = graph.ndata["feat"]
        dev = graph.device

        def project(lo, hi, blocks):
            rows = None if (lo == 0 and hi == graph.n_present) else torch.arange(lo, hi, dtype=torch.int64, device=dev)
            layer0.project_tables(feat, rows=rows, out=blocks)

        widths = [layer0.fc_pool.weight.shape[0], layer0._out_feats]
        P, S = parallel.build_row_tables(graph.n_present, widths, project, dev, partition=self.partition_features,
                                         padded_ld=ops.padded_ld)
        return P, S

    def _inference_batches(self, graph, seeds_all, shard=False):
        """Forward-only pass over ``seeds_all`` in batches of ``batch_full``; yields (seeds, logits).

        ``shard`` (N ranks): this rank runs WHOLE batches of the one-rank pass (``parallel.batch_shard``) with their
        sampler counters, so its logits are bit-identical to the one-rank pass's for the same seeds; every rank must
        iterate the generator (the table build below is a collective in partitioned mode) even without a batch of its own.

        When the pass is long enough that its batches would project more input rows than the snapshot holds, the
        layer-0 projection tables are computed ONCE for every present vertex (``_projection_tables``) and every batch
        reduces straight from them: no layer-0 GEMM over the unique inputs, no input-block relabel."""
        layer0 = self.graphsage_model.layers[0]
        n, bf = int(seeds_all.numel()), int(self.batch_full)
        rank, world = parallel.rank_world() if shard else (0, 1)
        nb = -(-n // bf)
        ctrs = sampling.reserve_ctrs(nb)
        b_lo, b_hi, s_lo, s_hi = parallel.batch_shard(n, bf, rank, world)
        per_batch_rows = min(bf * (1 + self.samples) ** 2 * self.UNIQUE_INPUT_FRACTION, graph.n_present)
        use_cache = (self.cache_projection and layer0.fc_pool is not None and not layer0.training
                     and nb * per_batch_rows > graph.n_present)          # a function of global quantities: same on every rank
        with self.graphsage_model.inference_pass():          # weights are fixed for the pass: per-layer constants once
            tables = self._projection_tables(graph, layer0) if use_cache else None
            if b_hi <= b_lo:
                return
            mine = seeds_all[s_lo:s_hi].to(graph.device, non_blocking=True).contiguous()
            batches = [mine[s:s + bf] for s in range(0, mine.numel(), bf)]
            # against the per-pass tables every kernel of a batch's forward is row-independent: consecutive batches run as ONE
            # block while their hidden-layer rows stay under FUSE_INFERENCE_ROWS (each batch still sampled with its own counter)
            fuse = self.FUSE_INFERENCE_ROWS if (use_cache and len(self.graphsage_model.layers) == 2
                                                and self.graphsage_model.layers[1]._aggre_type == "pool") else 0
            for input_nodes, seeds, blocks in self._sampler().sample_batches(graph, batches, relabel_input=not use_cache,
                                                                             ctrs=ctrs[b_lo:b_hi], fuse_rows=fuse):
                x = GatheredRows(graph.ndata["feat"], None, tables) if use_cache else self._inputs(graph, input_nodes)
                yield seeds, self.graphsage_model(blocks, x)

    def _run_custom_eval(self, graph, subgraph_to_id, id_to_subgraph, test_vertices):
        output_data = []
        self.graphsage_model.eval()
        seeds_all = torch.as_tensor(np.asarray(test_vertices), dtype=torch.int64)
        if seeds_all.numel() == 0:
            return output_data
        outs = []
        with torch.no_grad():
            for _, logits in self._inference_batches(graph, seeds_all):
                outs.append(logits)
        # one device->host transfer for the pass instead of one per batch
        return [o.cpu().numpy() for o in outs]

    def _eval_confusion(self, graph, subgraph_to_id, id_to_subgraph, vertices):
        """Device path: argmax + C x C confusion counters are accumulated on the GPU; only C*C int64 cross PCIe."""
        self.graphsage_model.eval()
        seeds_all = torch.as_tensor(np.asarray(vertices), dtype=torch.int64)
        if seeds_all.numel() == 0:
            return None
        n_all = int(seeds_all.numel())
        parallel.assert_replicated(seeds_all, "the evaluation vertices")
        last = self.graphsage_model.layers[-1]
        C_ = int((last.fc_self if last.fc_self is not None else last.fc_neigh).weight.shape[0])
        cm = torch.zeros(C_ * C_, dtype=torch.int64, device=graph.device)
        with torch.no_grad():
            # N ranks: whole batches of the pass are block-partitioned over the ranks, C x C counters summed over them
            for seeds, logits in self._inference_batches(graph, seeds_all, shard=True):
                ops.argmax_confusion(logits, ops.gather_i64(graph.ndata["target"], seeds), cm, want_pred=False)
        cm = cm.cpu()
        if parallel.is_distributed():
            import torch.distributed as dist
            if dist.get_backend() == "gloo":
                dist.all_reduce(cm, op=dist.ReduceOp.SUM)
            else:
                cmd = cm.to(graph.device)
                dist.all_reduce(cmd, op=dist.ReduceOp.SUM)
                cm = cmd.cpu()
        return cm.numpy().reshape(C_, C_), n_all

    def get_model(self):
        return "base_model"

    def train_step(self, graph, blocks, input_nodes, seeds, subgraph_to_id, n_global=None, on_rows=None):
        if self.gsync is None and (n_global is None or n_global == seeds.numel()):
            if self._graphs_ok("staged"):
                n0, n1 = int(input_nodes.numel()), blocks[1].number_of_src_nodes()
                # (under "auto" a size bucket is captured on its second sighting: a rare bucket never costs a 5 ms capture)
                sg = self._step_graphs().staged_step(graph, seeds, blocks, n0, n1, defer_first=self.use_graphs == "auto")
                if sg is not None:
                    loss = sg.loss
                    if on_rows is not None:
                        on_rows(seeds, sg.loss_rows.clone())
                    if self.step_hook is not None:
                        self.step_hook(dict(seeds=seeds, loss=loss, grads=sg.grads, form="staged", n0=n0, n1=n1))
                    return loss
            loss = self._eager_step(graph, blocks, input_nodes, seeds, on_rows)
            if self.step_hook is not None:
                self.step_hook(dict(seeds=seeds, loss=loss.detach(), grads=[p.grad for p in self.graphsage_model.parameters()],
                                    form="eager", n0=int(input_nodes.numel()), n1=blocks[1].number_of_src_nodes()))
            return loss
        # rank-sharded batch: this rank's seeds only; the gradient is that of the mean over the whole batch
        n_local = int(seeds.numel())
        big = self.use_graphs is True or int(n_global) * (1 + self.samples) >= self.STAGED_DP_MIN_ROWS      # (global: same on every rank)
        import torch.distributed as dist
        if (self.gsync is not None and self._graphs_ok("staged_dp") and big and (on_rows is None or self.reduction != "mean")
                and dist.get_backend(self.gsync.group) == "nccl" and DP_CAPTURE_COLLECTIVES       # (module global: read at call time)
                and self.sharded is None):
            # A replica's step as ONE replayed graph (form "staged_dp"): forward, loss, backward, the gradient exchange — the early
            # bucket's RCCL all-reduce launched from the gradient hooks on the side branch, under the layer-0 pool backward and weight
            # gradient; the late bucket (layer 0's fc_pool) behind it — and Adam on the reduced buckets.  The local mean loss's
            # gradients are weighted by n_local / n_global inside the graph (= the gradient of the mean over the whole batch).
            # The sequence of collectives is the same on every rank whatever it runs — a replay, the eager twin below (a size bucket on
            # its first sighting, a ragged or empty shard): before the split is learnt ONE bucket, afterwards early then late.
            w = n_local / float(n_global)
            sg = None
            if n_local > 0:
                n0, n1 = int(input_nodes.numel()), blocks[1].number_of_src_nodes()
                if self.gsync.learnt:            # (the step that learns the bucket split runs eagerly, on every rank: it broadcasts)
                    sg = self._step_graphs().staged_step(graph, seeds, blocks, n0, n1, defer_first=self.use_graphs == "auto",
                                                         dp=(self.gsync, w))
            if sg is not None:
                if on_rows is not None:
                    on_rows(seeds, sg.loss_rows.clone())
                if self.step_hook is not None:
                    self.step_hook(dict(seeds=seeds, loss=sg.loss * w, grads=sg.grads, form="staged_dp", n0=n0, n1=n1))
                return sg.loss * n_local
            # the eager twin: the same launches and the same collectives, enqueued from Python
            self.optimizer.zero_grad()
            loss_e = eager_rows = None
            self.gsync.begin_step(w)
            if n_local > 0:
                batch_labels = ops.LazyLabels(graph.ndata["target"], seeds)
                loss_e, eager_rows, _ = self.graphsage_model.forward_loss(blocks, self._inputs(graph, input_nodes), batch_labels, rows=True,
                                                                          defer_mean=True)
                ops.backward(loss_e)
            else:                                               # more ranks than seeds in this batch: zeros into the same collectives
                for p in self.gsync.params:
                    p.grad = None
            self.gsync.sync(weight=w)
            self.optimizer.step()
            if on_rows is not None:
                on_rows(seeds, eager_rows.detach() if eager_rows is not None else torch.zeros(0, device=graph.device))
            if loss_e is None:
                return None
            if self.step_hook is not None:
                self.step_hook(dict(seeds=seeds, loss=loss_e.detach() * w, grads=[p.grad for p in self.graphsage_model.parameters()],
                                    form="staged_dp_eager", n0=n0, n1=n1))
            return loss_e.detach() * n_local
        if self.gsync is not None and self._graphs_ok("staged_dp") and big and (on_rows is None or self.reduction != "mean"):
            # (a backend whose collectives cannot be recorded into a hipGraph — the gloo rehearsals: forward + backward replayed,
            # the exchange and the optimiser enqueued from Python)
            # replayed replica step: the captured graph ends with the gradients of the LOCAL mean loss in its static tensors; the
            # all-reduce weights them by n_local / n_global (= the gradient of the mean over the whole batch), then the optimiser
            eager_rows = None
            if n_local > 0:
                n0, n1 = int(input_nodes.numel()), blocks[1].number_of_src_nodes()
                with self.gsync.no_sync():                      # (a capture runs autograd for real: its hooks must not launch a collective)
                    sg = self._step_graphs().staged_step(graph, seeds, blocks, n0, n1, apply=False, defer_first=self.use_graphs == "auto")
                    if sg is None:
                        # first sighting of this size bucket: the same forward + backward eagerly (hooks suspended), then the
                        # SAME exchange as the replayed steps — the sequence of collectives does not depend on which form ran
                        self.optimizer.zero_grad()
                        batch_labels = ops.gather_i64(graph.ndata["target"], seeds)
                        scores = self.graphsage_model(blocks, self._inputs(graph, input_nodes))
                        loss_e, eager_rows = ops.cross_entropy_mean_rows(scores, batch_labels)
                        ops.backward(loss_e)
                if sg is not None:
                    for p, gr in zip(self.graphsage_model.parameters(), sg.grads):
                        p.grad = gr                           # (the previous sync left views of its flat bucket there)
            else:                                               # more ranks than seeds in this batch: zeros into the same collective
                sg = None
                for p in self.gsync.params:
                    p.grad = None
            # nothing overlaps the exchange here (the graph has ended), so it is ONE flat bucket: one collective's latency, not two
            self._exchange_and_step(n_local / float(n_global), single=True)
            if sg is None and eager_rows is not None:
                if on_rows is not None:
                    on_rows(seeds, eager_rows.detach())
                if self.step_hook is not None:
                    self.step_hook(dict(seeds=seeds, loss=loss_e.detach() * (n_local / float(n_global)),
                                        grads=[p.grad for p in self.graphsage_model.parameters()], form="staged_dp_eager", n0=n0, n1=n1))
                return loss_e.detach() * n_local
            if sg is None:
                if on_rows is not None:
                    on_rows(seeds, torch.zeros(0, device=graph.device))
                return None
            if on_rows is not None:
                on_rows(seeds, sg.loss_rows.clone())
            if self.step_hook is not None:
                self.step_hook(dict(seeds=seeds, loss=sg.loss * (n_local / float(n_global)), grads=sg.grads, form="staged_dp", n0=n0, n1=n1))
            return sg.loss * n_local
        if self.gsync is not None and SHARDED_FUSED:
            # The eager replica step (form "sharded"; round 5): the SAME launches as the replayed graph records — the fused output layer +
            # loss (GraphSAGE.forward_loss), the gradients weighted by n_local / n_global inside the exchange instead of by an ATen
            # division of the summed rows — so that what bench.py --force-dist itemises per kernel IS the replayed step.
            w = n_local / float(n_global)
            self.optimizer.zero_grad()
            loss_e = rows = None
            self.gsync.begin_step(w)
            if n_local > 0:
                batch_labels = ops.LazyLabels(graph.ndata["target"], seeds)
                loss_e, rows, _ = self.graphsage_model.forward_loss(blocks, self._inputs(graph, input_nodes), batch_labels, rows=True,
                                                                    defer_mean=True)
                ops.backward(loss_e)
            else:                                               # more ranks than seeds in this batch: zeros into the same collectives
                for p in self.gsync.params:
                    p.grad = None
            self._exchange_and_step(w)
            if on_rows is not None:
                on_rows(seeds, rows.detach() if rows is not None else torch.zeros(0, device=graph.device))
            if loss_e is None:
                return None
            if self.step_hook is not None:
                self.step_hook(dict(seeds=seeds, loss=loss_e.detach() * w, grads=[p.grad for p in self.graphsage_model.parameters()],
                                    form="sharded", n0=int(input_nodes.numel()), n1=blocks[1].number_of_src_nodes()))
            return loss_e.detach() * n_local
        loss_sum = rows = None
        if n_local > 0:
            batch_labels = ops.gather_i64(graph.ndata["target"], seeds)
            scores = self.graphsage_model(blocks, self._inputs(graph, input_nodes))
            rows = ops.cross_entropy(scores, batch_labels, "none")
            loss_sum = rows.sum()
        self._backward_and_step(loss_sum, n_local, n_global)
        if on_rows is not None:
            on_rows(seeds, rows.detach() if rows is not None else torch.zeros(0, device=graph.device))
        if self.step_hook is not None and n_local > 0:
            self.step_hook(dict(seeds=seeds, loss=loss_sum.detach() / n_global, grads=[p.grad for p in self.graphsage_model.parameters()],
                                form="sharded", n0=int(input_nodes.numel()), n1=blocks[1].number_of_src_nodes()))
        return loss_sum


class RandomHipSupervisedGraphSage(HipSupervisedGraphSage):
    """RBR: random rehearsal (R/.../pytorch/model.py:110-138)."""

    def __init__(self, model, batch_per_timestep, batch_size, labels, samples, cuda=True, batch_full=512, n_workers=0):
        super().__init__(model, batch_per_timestep, batch_size, labels, samples, n_workers=n_workers, cuda=cuda,
                         batch_full=batch_full)

    def choose_vertices(self, graph_util):
        batch_nodes = []
        for _ in range(self.batch_per_timestep):
            batch_nodes += graph_util.draw_random_train_nodes(self.batch_size)
        return batch_nodes

    def _run_custom_train(self, graph, subgraph_to_id, id_to_subgraph, train_vertices, graph_util):
        self.graphsage_model.train()
        self._train_batches(graph, train_vertices, len(train_vertices) // self.batch_per_timestep)

    def get_model(self):
        return "random"


class PrioritizedHipSupervisedGraphSage(HipSupervisedGraphSage):
    """PBR: prioritised rehearsal (R/.../pytorch/model.py:141-257)."""

    def __init__(self, model, batch_per_timestep, batch_size, labels, samples, priority_strategy, full_pass=2, cuda=True,
                 batch_full=512, n_workers=0):
        super().__init__(model, batch_per_timestep, batch_size, labels, samples, reduction="none", n_workers=n_workers,
                         cuda=cuda, batch_full=batch_full)
        self.time_step = 0
        self.pass_var = 0
        self.full_pass = full_pass
        self._device_trend = None
        self.priority_strategy = priority_strategy

    @property
    def priority_strategy(self):
        """The strategy object the driver passed in.  While its per-vertex state lives in HBM (``_priorities_device``) the host
        object is brought up to date before anyone outside reads it (one device->host copy, only when the state has moved)."""
        dt = self._device_trend
        if dt is not None and dt.dirty and dt.source is self._priority_strategy:
            dt.write_back(self._priority_strategy)
        return self._priority_strategy

    @priority_strategy.setter
    def priority_strategy(self, value):
        self._priority_strategy = value
        self._device_trend = None

    def choose_vertices(self, graph_util):
        if self.time_step % self.full_pass == 0:
            self.pass_var += 1
            # (the whole train set: as an array when the graph state keeps one — no 1e5-element list <-> array conversions)
            self.recompute_priorities(graph_util, graph_util.get_train_array() if hasattr(graph_util, "get_train_array")
                                      else graph_util.get_train_set())
        elif len(graph_util.get_new_train_nodes()) > 1:
            self.recompute_priorities(graph_util, graph_util.get_new_train_nodes())
        batch_nodes = []
        for _ in range(self.batch_per_timestep):
            batch_nodes += graph_util.draw_priority_train_nodes(self.batch_size)
        return batch_nodes

    def _run_custom_train(self, graph, subgraph_to_id, id_to_subgraph, train_vertices, graph_util):
        train_vertices = torch.as_tensor(np.asarray(train_vertices), dtype=torch.int64).reshape(-1)
        n = int(train_vertices.numel())
        bs = n // self.batch_per_timestep
        distributed = self.gsync is not None
        pending = []
        self._train_batches(graph, train_vertices.numpy(), bs, on_rows=lambda sd, rows: pending.append((sd, rows)))
        full_sizes = [min(bs, n - s0) for s0 in range(0, n, bs)] if bs > 0 else []
        # The reference copies every batch's losses to the host right away (a device sync per batch).  Nothing reads the
        # buffer while the snapshot's batches train (they were drawn beforehand), so the per-batch updates are applied in
        # the same order afterwards: identical buffer contents, no pipeline drain between batches — and with the buffer in
        # HBM (TrainTestGraph.device_replay) the losses never leave the device at all.
        if pending:
            if distributed:
                # the exchange step of the sharded PBR update: every rank needs the losses of ALL seeds of every batch to
                # keep its replay-buffer replica identical (one all-gather per snapshot)
                losses = parallel.all_gather_sharded([ls for _, ls in pending], full_sizes)
                sizes = full_sizes
            else:
                losses = [ls for _, ls in pending]
                sizes = [int(ls.numel()) for ls in losses]
            all_seeds = train_vertices.numpy()               # the batches in loader order ARE the seed list, in order
            on_device = (self._device_priorities(graph_util) and all(ls.is_cuda for ls in losses))
            all_loss = None if on_device else torch.cat(losses).cpu().numpy()
            off = 0
            for b, n in enumerate(sizes):
                batch_nodes_seed = np.asarray(subgraph_to_id[all_seeds[off:off + n]])
                if on_device:
                    graph_util.update_priorities_device(batch_nodes_seed, self._priorities_device(batch_nodes_seed, losses[b]))
                else:
                    priorities = self._host_strategy().get_priorities(batch_nodes_seed, all_loss[off:off + n])
                    graph_util.update_priorities_arrays(batch_nodes_seed, np.asarray(priorities, dtype=np.float64))
                off += n
        self.time_step += 1

    def recompute_priorities(self, graph_util, train_set):
        """Priority forward: inference over ``train_set`` in batches of ``batch_full``, per-seed CE loss ->
        ``graph_util.update_priorities`` (R/.../pytorch/model.py:210-254).  Runs under no_grad (the
        reference builds and discards an autograd graph) and returns the losses to the host once."""
        self.graphsage_model.eval()
        id_to_subgraph = graph_util.get_original_to_subgraph_map()
        subgraph_to_id = graph_util.get_subgraph_to_original_map()
        train_arr = train_set if isinstance(train_set, np.ndarray) else np.asarray(list(train_set), dtype=np.int64)
        if train_arr.size == 0:
            return
        seeds_np = np.array(id_to_subgraph[train_arr], dtype=np.int64).reshape(-1)       # (a copy: the train array is read-only, tensors are not)
        seeds_all = torch.from_numpy(seeds_np)
        graph = graph_util.get_graph()
        # N ranks: whole batches of the pass are block-partitioned over the ranks (seed order kept) and the per-seed losses
        # are all-gathered, so every replica of the replay buffer receives every priority (north star: "PBR sharded across
        # the GPUs"); the lengths every rank contributes follow from the partition, only values travel
        world = parallel.rank_world()[1]
        parallel.assert_replicated(seeds_all, "the priority-forward seeds")
        losses = []
        with torch.no_grad():
            for seeds, scores in self._inference_batches(graph, seeds_all, shard=True):
                batch_labels = ops.gather_i64(graph.ndata["target"], seeds)
                loss_rows, _ = ops.ce_fwd_bwd(scores, batch_labels, want_grad=False)
                losses.append(loss_rows)
        local = torch.cat(losses) if losses else torch.zeros(0, device=graph.device)
        if parallel.is_distributed():
            counts = []
            for r in range(world):
                _, _, a, b = parallel.batch_shard(seeds_all.numel(), self.batch_full, r, world)
                counts.append(b - a)
            local = parallel.all_gather_counts(local, counts)
        ids = np.asarray(subgraph_to_id[seeds_np], dtype=np.int64).reshape(-1)
        if self._device_priorities(graph_util) and local.is_cuda:
            graph_util.update_priorities_device(ids, self._priorities_device(ids, local))      # losses -> priorities -> tree, all in HBM
            return
        batch_nids_l = ids.tolist()
        unaggregated_loss = local.cpu().numpy()
        priorities = self._host_strategy().get_priorities(batch_nids_l, unaggregated_loss)
        graph_util.update_priorities_arrays(np.asarray(batch_nids_l), np.asarray(priorities, dtype=np.float64))

    def _device_priorities(self, graph_util):
        """Losses go into the replay structure on the device when it lives there: priority == loss (LossPriority, the strategy
        R/train/__main__.py:141 instantiates) needs nothing in between; TrendPriority / HybridPriority keep their per-vertex state
        in HBM too (priorities.DeviceTrend takes over the host object's arrays on first use)."""
        from ..prioritized_replay.priorities import HybridPriority, LossPriority, TrendPriority
        return getattr(graph_util, "device_replay", False) and type(self._priority_strategy) in (LossPriority, TrendPriority, HybridPriority)

    def _host_strategy(self):
        """The strategy object for a HOST-side update: current (the property writes a live device state back) and from now on the
        only copy — a device state taken earlier is dropped, the next device update takes the host arrays over again."""
        st = self.priority_strategy
        self._device_trend = None
        return st

    def _priorities_device(self, ids_host, losses_dev):
        """get_priorities(ids, losses) with both sides on the device."""
        from ..prioritized_replay.priorities import DeviceTrend, LossPriority
        strategy = self._priority_strategy
        if type(strategy) is LossPriority:
            return losses_dev
        dt = self._device_trend
        if dt is None or dt.source is not strategy:
            dt = self._device_trend = DeviceTrend(strategy, losses_dev.device)
            dt.source = strategy
        # the kernel's preconditions, checked on the host copy the ids come from (no device sync): in range, distinct
        DeviceTrend.check_ids_host(ids_host, dt.n_vertices)
        ids_dev = torch.as_tensor(np.asarray(ids_host, dtype=np.int64)).to(losses_dev.device, non_blocking=True)
        return dt.get_priorities_device(ids_dev, losses_dev)

    def get_model(self):
        return "prioritized"


class FullHipSupervisedGraphSage(HipSupervisedGraphSage):
    """Offline GraphSAGE retrained on the whole train set (R/.../pytorch/model.py:260-290)."""

    def __init__(self, model, batch_per_timestep, batch_size, labels, samples, cuda=True, batch_full=512, n_workers=0):
        super().__init__(model, batch_per_timestep, batch_size, labels, samples, n_workers=n_workers, cuda=cuda,
                         batch_full=batch_full)

    def choose_vertices(self, graph_util):
        return graph_util.get_train_set().copy()

    def _run_custom_train(self, graph, subgraph_to_id, id_to_subgraph, batch_nodes, graph_util):
        self.graphsage_model.train()
        train_set = torch.as_tensor(np.asarray(batch_nodes), dtype=torch.int64)
        for _ in range(self.batch_per_timestep):
            train_set = train_set.view(-1)[torch.randperm(train_set.nelement())].view(train_set.size())
            self._train_batches(graph, train_set.numpy(), self.batch_size)

    def get_model(self):
        return "offline"


class NoRehHipSupervisedGraphSage(HipSupervisedGraphSage):
    """Trains on newly arrived vertices only (R/.../pytorch/model.py:293-323)."""

    def __init__(self, model, batch_per_timestep, batch_size, labels, samples, cuda=True, batch_full=512, n_workers=0):
        super().__init__(model, batch_per_timestep, batch_size, labels, samples, n_workers=n_workers, cuda=cuda,
                         batch_full=batch_full)

    def choose_vertices(self, graph_util):
        return []

    def _run_custom_train(self, graph, subgraph_to_id, id_to_subgraph, batch_nodes, graph_util):
        self.graphsage_model.train()
        for _ in range(self.batch_per_timestep):
            idxs = graph_util.get_new_train_nodes(self.batch_size)
            if len(idxs) < 2:
                return
            batch_nodes = np.asarray(id_to_subgraph[idxs], dtype=np.int64).reshape(-1)
            batch_nodes = batch_nodes[torch.randperm(len(batch_nodes)).numpy()]      # the loader's shuffle=True (torch's CPU stream)
            self._train_batches(graph, batch_nodes, len(batch_nodes))

    def get_model(self):
        return "no_rehersal"
