"""Inputs of the BASELINE-shape golden cases (one 512 x 25 block over 602 features, pool_feats 600, 600 outputs),
regenerated IDENTICALLY wherever they are needed from a numpy PCG64 stream — ``make_golden.py`` (here, where the reference can be
imported) and the tests (also on the GPU box, where it cannot) — so that only the reference's OUTPUTS have to be committed:
``fullsize_<mode>.npz`` holds y in full, every 4th row of the weight gradients, the bias gradients, and of the input gradient
(7 040 x 602: 17 MB) a fixed random projection [n_src, 8], its row norms and 64 full rows.  Our code, not the reference's."""
import numpy as np

N_DST, N_SRC, FANOUT, FIN, FOUT, POOL = 512, 7040, 25, 602, 600, 600
SEEDS = {"mean": 4101, "meanpool": 4102}
GRAD_ROW_STRIDE = 4
GX_ROWS = 64


def make(mode):
    """dict(local_idx int32 [512, 25] (dst-first block: h_self = x[:512]; ~3 % of the destinations without neighbours, sources
    skewed towards low local ids like a block builder's hubs), x [7040, 602], gy [512, 600], params {name: array},
    proj [602, 8], gx_rows int64 [64])."""
    rng = np.random.Generator(np.random.PCG64(SEEDS[mode]))
    u = rng.random((N_DST, FANOUT))
    li = np.minimum((N_SRC * u ** 2.5).astype(np.int64), N_SRC - 1).astype(np.int32)
    li[rng.random(N_DST) < 0.03] = -1
    x = rng.standard_normal((N_SRC, FIN), dtype=np.float32)
    gy = rng.standard_normal((N_DST, FOUT), dtype=np.float32)

    def xavier(out_f, in_f, gain=np.sqrt(2.0)):
        b = gain * np.sqrt(6.0 / (in_f + out_f))
        return rng.uniform(-b, b, (out_f, in_f)).astype(np.float32)

    def bias(out_f, in_f):
        b = 1.0 / np.sqrt(in_f)
        return rng.uniform(-b, b, out_f).astype(np.float32)

    params = {}
    if mode == "meanpool":
        params["fc_pool.weight"] = xavier(POOL, FIN)
        params["fc_pool.bias"] = bias(POOL, FIN)
        neigh = POOL
    else:
        neigh = FIN
    params["fc_neigh.weight"] = xavier(FOUT, FIN + neigh)
    params["fc_neigh.bias"] = bias(FOUT, FIN + neigh)
    proj = rng.standard_normal((FIN, 8)).astype(np.float32)
    gx_rows = np.sort(rng.choice(N_SRC, GX_ROWS, replace=False)).astype(np.int64)
    return dict(local_idx=li, x=x, gy=gy, params=params, proj=proj, gx_rows=gx_rows)


def digest(gx, inp):
    """What is stored / compared of an input gradient [n_src, 602]: (projection [n_src, 8] in float64, row norms, 64 full rows)."""
    gx = np.asarray(gx, dtype=np.float64)
    return gx @ inp["proj"].astype(np.float64), np.sqrt((gx * gx).sum(1)), gx[inp["gx_rows"]]
