"""BASELINE.json config 1 — "Pubmed pytorch CPU, no-rehearsal, depth=2 samples=10 batch=32 (plumbing, no GPU)" — through
the oracle on the CPU: the pubmed-like vertex stream (N=19 717, F=500, C=3, 400 snapshots of 49 vertices,
R/train/dataset_utils/pubmed.py:20-21, R/settings/pubmed.json:1), the no-rehearsal loop body
(R/train/graphsage/pytorch/model.py:300-323) for 3 snapshots with S=10, B=32.  Its GPU twin — the HIP strategy class
on the same seeds, losses and weights equal to this loop's — is tests/test_gpu_rungs.py::
test_no_rehearsal_pubmed_matches_oracle_loop.  Here: the stream restatement, the sampler invariants on each snapshot and
the determinism of the loop."""
import numpy as np
import torch

import ogl_amd  # noqa: F401
from ogl_amd import synthetic
from oracle import oracle as O


def _stream(a):
    return O.HostVertexStream(a["n"], a["src"], a["dst"], a["order"], a["snapshots"], a["feat"], a["labels"])


def test_no_rehearsal_loop_on_pubmed_like_stream():
    a = synthetic.make_arrays("pubmed")
    assert (a["n"], a["f"], a["c"], a["snapshots"]) == (19717, 500, 3, 400)
    st = _stream(a)
    assert st.per == 49                                           # int(19717 / 400), dynamic_graph_vertex.py:30
    rng = np.random.default_rng(0)
    seeds, probe = [], _stream(a)
    for t in range(3):
        arr = probe.arrivals()
        assert arr.tolist() == list(range(49 * t, 49 * (t + 1)))
        # snapshot t = induced subgraph on the first 49 (t + 1) arrivals: every kept neighbour is present, degrees are the
        # prefix counts, and they can only grow from one snapshot to the next
        deg = probe.degrees()
        slow = O.snapshot_degrees(probe.indptr, probe.indices, probe.n_present, probe.n_present)
        assert np.array_equal(deg, slow) and deg[probe.n_present:].sum() == 0
        for v in range(probe.n_present):
            nb = probe.indices[probe.indptr[v]:probe.indptr[v] + deg[v]]
            assert (nb < probe.n_present).all()
        # features / labels arrive in arrival order
        assert torch.equal(probe.feat[arr[0]], a["feat"][a["order"][arr[0]]].float())
        assert int(probe.labels[arr[-1]]) == int(a["labels"][a["order"][arr[-1]]])
        sd = rng.permutation(arr)[:32]                            # B = 32 of the 49 arrivals, a fixed shuffle
        seeds.append(sd)
        # sampler at S = 10: exactly 10 picks iff in-degree > 0, every pick a present in-neighbour
        picks = O.sample_layer(probe.indptr, probe.indices, deg, sd, 10, 1, t, 1)
        for i, d in enumerate(sd):
            nb = probe.indices[probe.indptr[d]:probe.indptr[d] + deg[d]]
            assert (picks[i] == -1).all() if deg[d] == 0 else np.isin(picks[i], nb).all()
        probe.evolve()
    model = O.CpuModel("pool", a["f"], 32, a["c"], seed=1)
    w0 = model.params[0]["fc_pool.weight"].detach().clone()
    losses = O.no_rehearsal_stream(st, model, 10, seeds, 1)
    assert len(losses) == 3 and np.isfinite(losses).all() and st.t == 4
    assert not torch.equal(w0, model.params[0]["fc_pool.weight"])
    # deterministic: the same seeds and Philox stream give the same losses bit for bit
    again = O.no_rehearsal_stream(_stream(a), O.CpuModel("pool", a["f"], 32, a["c"], seed=1), 10, seeds, 1)
    assert again == losses
    # fewer than two new train vertices: the reference's quiet early return (model.py:308-309) — no step, stream evolves
    st2, m2 = _stream(a), O.CpuModel("pool", a["f"], 32, a["c"], seed=1)
    assert O.no_rehearsal_stream(st2, m2, 10, [seeds[0][:1]], 1) == [] and st2.t == 2
