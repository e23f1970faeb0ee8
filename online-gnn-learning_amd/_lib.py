"""ctypes binding of include/ogl_hip.h.  No fallback: if libogl_hip.so is missing or a call
returns a non-zero status this raises — the product path never routes around the HIP library."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libogl_hip.so")

OGL_OK = 0
REDUCE_MEAN, REDUCE_MAX, REDUCE_SUM = 0, 1, 2
REDUCE_OPS = {"mean": REDUCE_MEAN, "max": REDUCE_MAX, "sum": REDUCE_SUM}

_p = C.c_void_p
_i64 = C.c_int64
_i = C.c_int
_f = C.c_float
_d = C.c_double
_u64 = C.c_uint64

# name -> (restype, argtypes): mirrors include/ogl_hip.h one to one
SIGNATURES = {
    "ogl_version": (_i, []),
    "ogl_source_hash": (C.c_char_p, []),
    "ogl_debug_set": (_i, [_i, _i, _p]),
    "ogl_status_string": (C.c_char_p, [_i]),
    "ogl_last_hip_error": (_i, []),
    "ogl_set_gemm_mode": (_i, [_i]),
    "ogl_graph_create": (_i, [_p, _p, _p, _i64, _i64, C.POINTER(_p)]),
    "ogl_graph_set_snapshot": (_i, [_p, _i64, _i64, _p]),
    "ogl_graph_degrees": (_i, [_p, C.POINTER(_p)]),
    "ogl_graph_copy_degrees": (_i, [_p, _p, _p]),
    "ogl_graph_destroy": (_i, [_p]),
    "ogl_sample_layer": (_i, [_p, _p, _i64, _i, _u64, _u64, _i, _p, _p]),
    "ogl_block_workspace_bytes": (_i64, [_i64, _i]),
    "ogl_build_block": (_i, [_p, _i64, _p, _i, _p, _p, _p, _p, _i64, _p]),
    "ogl_gather_rows": (_i, [_p, _i64, _i64, _p, _i64, _i, _p, _i64, _p]),
    "ogl_fill_zero": (_i, [_p, _i64, _p]),
    "ogl_stream_copy": (_i, [_p, _p, _i64, _p]),
    "ogl_gather_i64": (_i, [_p, _i64, _p, _i64, _p, _p]),
    "ogl_dropout_rows": (_i, [_p, _i64, _p, _i64, _i64, _i, _d, _u64, _u64, _p, _i64, _p]),
    "ogl_reduce_fwd": (_i, [_p, _i64, _i64, _p, _p, _i64, _i, _i, _i, _p, _i64, _p, _p]),
    "ogl_reduce_fwd_img": (_i, [_p, _i64, _i64, _p, _p, _i64, _i, _i, _p, _i64, _p, _p, _p]),
    "ogl_reduce_bwd": (_i, [_p, _i64, _p, _p, _p, _i64, _i64, _i, _i, _i, _i64, _p, _i64, _p]),
    "ogl_reduce_fwd_mean_img": (_i, [_p, _i64, _i64, _p, _p, _i64, _i, _i, _p, _i64, _p, _p]),
    "ogl_reduce_fwd_rows_mean_img": (_i, [_p, _i64, _i64, _p, _p, _i64, _i64, _i, _i, _p, _i64, _p, _p]),
    "ogl_reduce_bwd_seg_workspace_bytes": (_i64, [_i64, _i, _i, _i64]),
    "ogl_reduce_bwd_seg_plan": (_i, [_p, _i64, _i, _i64, _i, _p, _i64, _p]),
    "ogl_reduce_bwd_seg_apply": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _p, _p, _i64, _p]),
    "ogl_reduce_bwd_seg_apply_t": (_i, [_p, _i64, _i64, _i, _i, _i, _i64, _p, _i64, _p, _p, _i64, _p]),
    "ogl_linear_fwd": (_i, [_p, _i64, _p, _i64, _i64, _i, _p, _i64, _i, _p,
                            _p, _i64, _p, _i64, _i, _p, _i64, _i, _p, _i64, _p]),
    "ogl_linear_fwd_dual_bias": (_i, [_p, _i64, _p, _i64, _i64, _i, _p, _i64, _i, _p, _p,
                                      _p, _i64, _p, _i64, _i, _p, _i64, _i, _p, _i64, _p]),
    "ogl_linear_fwd_addrows": (_i, [_p, _i64, _p, _i64, _i64, _i, _p, _i64, _i, _p, _p, _i64, _p, _i64, _i, _p, _i64, _p]),
    "ogl_relu_bwd": (_i, [_p, _i64, _p, _i64, _i64, _i, _p, _i64, _p]),
    "ogl_linear_bwd_input": (_i, [_p, _i64, _i64, _i, _p, _i64, _i, _p, _i64, _p]),
    "ogl_linear_bwd_weight_workspace_bytes": (_i64, [_i64, _i, _i]),
    "ogl_linear_bwd_weight": (_i, [_p, _i64, _p, _i64, _p, _i64, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p]),
    "ogl_linear_bwd_weight_t_workspace_bytes": (_i64, [_i64, _i, _i]),
    "ogl_linear_bwd_weight_t": (_i, [_p, _i64, _p, _i64, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p]),
    "ogl_transpose": (_i, [_p, _i64, _p, _i64, _i64, _i, _p, _i64, _p]),
    "ogl_sample_layer_batched": (_i, [_p, _p, _p, _p, _i, _i, _u64, _p, _i, _p, _p]),
    "ogl_block_workspace_bytes_batched": (_i64, [_p, _i, _i, _i64]),
    "ogl_build_block_batched": (_i, [_p, _p, _p, _i, _p, _i, _i64, _p, _p, _p, _p, _i64, _p]),
    "ogl_pool_bwd_x3_workspace_bytes": (_i64, [_i64, _i, _i, _i64]),
    "ogl_pool_bwd_x3": (_i, [_p, _i64, _p, _p, _i64, _p, _i64, _i, _i, _i64, _p, _p, _i64, _p]),
    "ogl_pool_bwd_x3_plan": (_i, [_p, _p, _i64, _p, _i64, _i, _i, _i64, _p, _i64, _p]),
    "ogl_pool_bwd_x3_apply": (_i, [_p, _i64, _p, _i64, _i, _i, _i64, _p, _p, _i64, _p]),
    "ogl_x3_image_bytes": (_i64, [_i64, _i64]),
    "ogl_x3_split": (_i, [_p, _i64, _p, _i64, _i64, _i, _i, _p, _p, _p]),
    "ogl_x3_split_t": (_i, [_p, _i64, _p, _i64, _i64, _i, _i, _i64, _p, _p]),
    "ogl_x3_debug_stamps": (_i, [_p, _i]),
    "ogl_x3_last_kernel": (C.c_char_p, []),
    "ogl_out_layer_fwd_ce_fits": (_i, [_i64, _i, _i, _i]),
    "ogl_out_layer_fwd_ce": (_i, [_p, _i64, _i64, _p, _i64, _i, _p, _i64, _i, _p, _i64, _p, _i64, _p, _p, _i, _p, _i64, _p, _p, _i64,
                                  _p, _i64, _p, _f, _p, _p, _i64, _p, _p, _p, _i64, _i, _p]),
    "ogl_loss_mean_finish": (_i, [_p, _i64, _p, _p]),
    "ogl_out_layer_fwd_ce_mean": (_i, [_p, _i64, _i64, _p, _i64, _i, _p, _i64, _i, _p, _i64, _p, _i64, _p, _p, _i, _p, _i64, _p, _i64,
                                       _p, _i64, _p, _f, _p, _p, _i64, _p, _p, _i, _p]),
    "ogl_out_layer_bwd_inputs_dense": (_i, [_p, _i64, _i64, _i, _i, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p]),
    "ogl_linear_fwd_x3": (_i, [_p, _i64, _p, _i64, _i64, _i, _p, _i, _i, _p, _i64, _p]),
    "ogl_linear_fwd_x3_ext": (_i, [_p, _i64, _p, _i64, _i, _p, _i64, _p, _i64, _i, _i64, _p, _i, _p, _i64, _p, _i64, _i, _p, _i64, _p, _i, _p, _i64,
                                   _p, _p]),
    "ogl_x3_split_multi": (_i, [_p, _i, _p, _p, _d, _d, _d, _p]),
    "ogl_relu_bwd_img": (_i, [_p, _i64, _p, _i64, _i64, _i, _p, _i64, _p, _p]),
    "ogl_x3_split_into": (_i, [_p, _i64, _i64, _i, _i, _p, _p, _i64, _i64, _p]),
    "ogl_linear_bwd_weight_x3_workspace_bytes": (_i64, [_i64, _i, _i]),
    "ogl_linear_bwd_weight_x3k_workspace_bytes": (_i64, [_i64, _i64, _i, _i, _i]),
    "ogl_linear_bwd_weight_x3k": (_i, [_p, _i64, _p, _i64, _p, _i64, _i64, _i, _i, _i, _p, _i64, _p, _p, _p, _i64, _p]),
    "ogl_linear_bwd_weight_x3": (_i, [_p, _p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p]),
    "ogl_ce_fwd_bwd": (_i, [_p, _i64, _p, _i64, _i, _f, _p, _p, _i64, _p]),
    "ogl_ce_fwd_bwd_mean_grid": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _f, _p, _p, _i64, _p, _p, _p, _i64, _p]),
    "ogl_ce_fwd_bwd_mean_gather": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _f, _p, _p, _i64, _p, _p, _i64, _p, _p, _d, _d, _d, _p]),
    "ogl_adam_step": (_i, [_p, _p, _p, _p, _i64, _i, _d, _d, _d, _d, _p]),
    "ogl_argmax_confusion": (_i, [_p, _i64, _p, _i64, _i, _p, _p, _p]),
    "ogl_sample_layer_dev": (_i, [_p, _p, _i64, _i, _u64, _p, _i, _p, _p]),
    "ogl_out_layer_bwd_inputs": (_i, [_p, _i64, _i64, _i, _i, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _p, _i64, _p, _i64, _p]),
    "ogl_out_layer_bwd_inputs_mean": (_i, [_p, _i64, _i64, _i, _i, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p]),
    "ogl_out_layer_bwd_weights": (_i, [_p, _i64, _i64, _i, _i, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p]),
    "ogl_small_pool_layer_fits": (_i, [_i64, _i64, _i, _i, _i]),
    "ogl_small_pool_layer_fwd": (_i, [_p, _i64, _i64, _p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i, _i, _p, _i64, _p, _p,
                                      _i64, _p, _p]),
    "ogl_small_pool_layer_bwd": (_i, [_p, _i64, _p, _i64, _i, _p, _i64, _i64, _i64, _i, _i, _i, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64,
                                      _p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _p]),
    "ogl_small_pool_loss_fits": (_i, [_i64, _i64, _i, _i, _i]),
    "ogl_small_pool_layer_fwd_ce_bwd": (_i, [_p, _i64, _i64, _p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i, _p, _i64, _p, _f,
                                             _p, _i64, _p, _p, _i64, _p, _p, _p, _i64, _p, _p, _i64, _i, _p, _i64, _p]),
    "ogl_small_pool_layer_bwd_pool": (_i, [_p, _i64, _i64, _i, _i, _p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _i64, _p,
                                           _p, _i64, _p, _p, _i64, _p, _p, _p, _d, _d, _d, _p]),
    "ogl_small_proj_rows": (_i, [_p, _i64, _p, _i64, _i64, _i, _p, _i64, _i, _p, _i, _p, _i64, _p, _p]),
    "ogl_small_first_layer_fits": (_i, [_i64, _i64, _i, _i, _i]),
    "ogl_small_first_layer_fwd": (_i, [_p, _i64, _i64, _p, _i64, _i, _i, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i, _i, _p, _i64,
                                       _p, _p, _i64, _p, _p]),
    "ogl_small_first_layer_bwd": (_i, [_p, _i64, _p, _i64, _i, _i64, _i, _i, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i,
                                       _p, _p, _i64, _i, _p, _i64, _p, _i64, _i64, _p, _p]),
    "ogl_record_weight_grads": (_i, [_p, _i, _p, _i64, _p, _p, _p, _d, _d, _d, _p]),
    "ogl_replay_update": (_i, [_p, _i64, _p, _p, _p, _i64, _d, _d, _d, _d, _d, _p, _p, _p, _p]),
    "ogl_replay_rebuild": (_i, [_p, _i64, _p]),
    "ogl_replay_sample": (_i, [_p, _i64, _i64, _i64, _p, _p, _i64, _p, _p, _p]),
    "ogl_replay_note_keys": (_i, [_p, _i64, _i64, _p, _i64, _p]),
    "ogl_priority_trend": (_i, [_p, _p, _p, _i64, _i64, _p, _p, _p, _p, _d, _d, _p, _p, _p]),
    "ogl_build_block_padded": (_i, [_p, _i64, _p, _i, _p, _i64, _p, _p, _p, _i64, _p]),
    "ogl_linear_bwd_weight_x3k_slabs": (_i, [_p, _i64, _p, _i64, _p, _i64, _i64, _i, _i, _i, _p, _i64, _p, _p, _p, _i64,
                                             C.POINTER(C.c_int), C.POINTER(C.c_int64), _p]),
    "ogl_adam_step_multi_slabs": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _d, _d, _d, _d, _p]),
    "ogl_linear_bwd_weight_x3k_dual_workspace_bytes": (_i64, [_i64, _i, _i, _i, _i]),
    "ogl_linear_bwd_weight_x3k_dual_slabs": (_i, [_p, _i64, _i, _p, _i64, _p, _i64, _i, _i, _p, _i64, _i, _p, _i64, C.POINTER(C.c_int),
                                                C.POINTER(C.c_int64), C.POINTER(C.c_int), _p]),
    "ogl_x3_slab_reduce": (_i, [_p, _i64, _i64, _i, _i64, _i, _i, _p, _i64, _p]),
    "ogl_publish_i64": (_i, [_p, _i, _p, _p, _p]),
    "ogl_sample_blocks_small_workspace_bytes": (_i64, [_i, _i]),
    "ogl_sample_blocks_small_fill": (_i, [_p, _p, _p, _i, _i, C.c_uint64, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "ogl_stage_segments": (_i, [_i, _p, _p, _p, _p, _p, _i64, _p]),
    "ogl_fuse_block_segments": (_i, [_p, _p, _i, _p, _p, _i, _p, _p]),
}

class RecSeg(C.Structure):
    """ogl_rec_seg_t (include/ogl_hip.h): one row group of ogl_record_weight_grads."""
    _fields_ = [("G", _p), ("ldg", _i64), ("arg", _p), ("ldarg", _i64), ("n_idx", _i64), ("ids", _p), ("rows", _p), ("ldr", _i64),
                ("n_rows", _i64), ("F", _i), ("n_dst", _i64), ("n_out", _i), ("dW", _p), ("lddw", _i64), ("db", _p), ("db2", _p),
                ("n_live", _p)]


_lib = None


class OglError(RuntimeError):
    pass


def lib():
    """Load libogl_hip.so once (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libogl_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first; ours binds to the same SONAME)
        h = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)   # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(status, what=""):
    if status != OGL_OK:
        h = lib()
        msg = h.ogl_status_string(int(status)).decode()
        if status == -3:
            msg += " (hipError_t=%d)" % h.ogl_last_hip_error()
        raise OglError("%s failed: %s" % (what or "libogl_hip call", msg))
