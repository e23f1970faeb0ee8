"""A/B of the image-GEMM tile shapes (ogl_debug_set: OGL_KNOB_X3_TILE) at the step's shapes, one process, alternating runs.
  python tools/tile_probe.py            # prints a JSON line per (shape, tile)"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ogl_amd  # noqa: F401
from ogl_amd import ops, _lib


def timed(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    torch.manual_seed(0)
    shapes = [("fc_pool0 fwd", 62495, 602, 602, 232965, (0, 4)), ("n1 plain", 7060, 600, 600, 7060, (1, 3, 2, 0)),
              ("n1 ext", 7060, 602, 600, 62495, (1, 3, 2, 0)), ("P0 table", 232965, 602, 602, 232965, (0, 4))]
    for name, M, K, N, T, cfgs in shapes:
        tm = ops.empty_mat(T, K, "cuda").copy_(torch.randn(T, K, device="cuda"))
        rows = torch.randperm(T, device="cuda")[:M].sort().values if T > M else None
        w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
        xi, wi = ops.x3_split(tm, append_ones=True), ops.x3_split(w, append_vec=b)
        ext = name.endswith("ext")
        if ext:
            x2 = ops.empty_mat(M, 602, "cuda").copy_(torch.randn(M, 602, device="cuda"))
            x2i = ops.x3_split(x2)
            wcat = ops.x3_split_cat([(w, b), (torch.randn(N, 602, device="cuda") / 25, None)])
            fn = lambda: ops.linear_fwd_x3_ext(xi, rows, wcat, x2_img=x2i, relu=True, x_nrows=T, want_image=True, image_append_ones=True)
        else:
            fn = lambda: ops.linear_fwd_x3(xi, rows, wi, relu=True, x_nrows=T)
        res = {c: [] for c in cfgs}
        for rnd in range(4):
            for c in cfgs:
                ops.debug_set("x3_tile", c)
                timed(fn, 3)
                res[c].append(timed(fn, 20))
        ops.debug_set("x3_tile", -1)
        print(json.dumps({"shape": name, "M": M, "K": K, "N": N, "us_by_tile": {str(c): [round(v, 1) for v in res[c]] for c in cfgs}}), flush=True)
        del tm, xi


if __name__ == "__main__":
    main()
