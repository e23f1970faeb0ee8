"""The C-ABI library loads on a CPU-only box and exports every symbol include/ogl_hip.h declares
(no compute calls here: those are the -m gpu tests)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ogl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ogl_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    import ogl_amd  # noqa: F401
    from ogl_amd import _lib
    names = _declared()
    assert len(names) >= 20
    h = _lib.lib()
    for n in names:
        assert hasattr(h, n), "libogl_hip.so does not export %s" % n
    # the ctypes table covers the header one to one
    assert sorted(_lib.SIGNATURES) == names
    assert h.ogl_version() == 100
    assert b"OGL_EWORKSPACE" in h.ogl_status_string(-4)


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    import ogl_amd  # noqa: F401
    from ogl_amd import _lib
    h = _lib.lib()
    out = ctypes.c_void_p()
    assert h.ogl_graph_create(None, None, None, 5, 0, ctypes.byref(out)) == -1
    assert h.ogl_sample_layer(None, None, 4, 25, 1, 0, 0, None, None) == -1
    assert h.ogl_block_workspace_bytes(-1, 3) == -1
    assert h.ogl_block_workspace_bytes(512, 25) >= 4 * 3 * 2 * 512 * 26
    assert h.ogl_reduce_fwd(None, 4, 0, None, None, -1, 1, 4, 0, None, 4, None, None) == -1
    assert h.ogl_linear_fwd(None, 2, None, 0, 4, 8, None, 8, 4, None, None, 0, None, 0, 0, None, 0, 0, None, 4, None) == -1
    assert h.ogl_adam_step(None, None, None, None, 10, 0, 1e-3, 0.9, 0.999, 1e-8, None) == -1   # step < 1
    assert h.ogl_linear_bwd_weight_workspace_bytes(233000, 602, 602) > 0
    # round 5, the 32-seed kernels: shape gates and argument checks answer before anything is launched
    assert h.ogl_small_pool_loss_fits(832, 32, 25, 32, 40) == 1 and h.ogl_small_pool_loss_fits(832, 32, 65, 32, 40) == 0
    assert h.ogl_small_pool_loss_fits(4000, 200, 25, 32, 40) == 0                       # more than 128 destinations
    assert h.ogl_small_first_layer_fits(21632, 832, 25, 500, 32) == 1 and h.ogl_small_first_layer_fits(21632, 832, 25, 500, 33) == 0
    assert h.ogl_small_first_layer_fits(21632, 832, 25, 1028, 32) == 0
    assert h.ogl_record_weight_grads(None, 1, None, 0, None, None, None, 0.0, 0.0, 0.0, None) == -1
    seg = _lib.RecSeg()                                                                    # (all NULL: rejected, not dereferenced)
    assert h.ogl_record_weight_grads(ctypes.addressof(seg), 1, None, 0, None, None, None, 0.0, 0.0, 0.0, None) == -1
    assert h.ogl_record_weight_grads(ctypes.addressof(seg), 7, None, 0, None, None, None, 0.0, 0.0, 0.0, None) == -1
    assert h.ogl_small_proj_rows(None, 512, None, 10, 64, 500, None, 500, 500, None, 1, None, 512, None, None) == -1
    assert h.ogl_small_proj_rows(None, 512, None, 10, 0, 500, None, 500, 500, None, 1, None, 512, None, None) == 0    # no rows: nothing to do
    assert h.ogl_small_proj_rows(None, 512, None, 10, 64, 502, None, 502, 500, None, 1, None, 512, None, None) == -1  # K % 4
    assert h.ogl_sample_blocks_small_fill(None, None, None, 32, 25, 1, None, None, None, None, None, None, None, None, 0, 256, None) == -1
    # round 6: the diagnostic knobs behind ONE entry point; the merged entry points' optional parts come in pairs; the mean backward as the
    # transposed image and its addend are refused before anything is launched
    prev = ctypes.c_int(-7)
    assert h.ogl_debug_set(9, 0, None) == -1 and h.ogl_debug_set(0, 3, None) == -1 and h.ogl_debug_set(1, 4, None) == -1
    assert h.ogl_debug_set(3, 1, ctypes.byref(prev)) == 0 and prev.value in (0, 1)
    assert h.ogl_debug_set(4, 1, ctypes.byref(prev)) == 0 and prev.value in (0, 1)
    assert h.ogl_debug_set(0, -1, ctypes.byref(prev)) == 0 and prev.value in (-1, 0, 1, 2)
    assert h.ogl_set_gemm_mode(-1) in (0, 1, 2) and h.ogl_set_gemm_mode(7) == -1        # OGL_GEMM_QUERY changes nothing
    buf = (ctypes.c_char * 64)()
    p64 = ctypes.cast(buf, ctypes.c_void_p)
    assert h.ogl_x3_split_multi(p64, 1, p64, None, 0.0, 0.0, 0.0, None) == -1            # the optimiser's scalars: both addresses or neither
    assert h.ogl_ce_fwd_bwd_mean_gather(p64, 8, p64, 4, None, 4, 8, ctypes.c_float(1.0), None, None, 0, p64, None, 0, p64, None, 0.0, 0.0, 0.0, None) == -1
    assert h.ogl_block_workspace_bytes_batched(None, 1, 25, 0) == -1 and h.ogl_block_workspace_bytes_batched(None, 1, 25, 1000) == -1
    assert h.ogl_reduce_bwd_seg_workspace_bytes(7060, 25, 600, 62495) > 8 * 7060 * 25                       # (incl. the group-major lists)
    assert h.ogl_reduce_bwd_seg_plan(None, 10, 25, 0, 1, p64, 64, None) == -1                               # no sources
    assert h.ogl_reduce_bwd_seg_plan(None, 10, 25, 100, 1, p64, 64, None) == -4                             # workspace too small
    ok = (p64, 608, 10, 25, 600, 0, 100, None, 0, p64, p64, 1 << 30, None)
    assert h.ogl_reduce_bwd_seg_apply_t(*(ok[:4] + (642,) + ok[5:])) == -1                                   # more than 640 columns
    assert h.ogl_reduce_bwd_seg_apply_t(*((ok[0], 607) + ok[2:])) == -1                                      # an odd leading dimension
    assert h.ogl_reduce_bwd_seg_apply_t(*(ok[:9] + (None,) + ok[10:])) == -1                                 # no image
    assert h.ogl_reduce_bwd_seg_apply_t(*(ok[:11] + (64,) + ok[12:])) == -4                                  # workspace too small
    rows = (p64, 608, p64, 10, 25, 600, 0, 100, None, 0, p64, 608, 5, None, 0, p64, p64, 1 << 30, None)
    assert h.ogl_reduce_bwd_seg_apply(*rows) == -1                                                           # an addend without fp32 rows
    assert h.ogl_reduce_bwd_seg_apply(*(rows[:12] + (101,) + (p64, 608) + rows[15:])) == -1                  # more addend rows than sources


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import ogl_amd  # noqa: F401
    from ogl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_cpu_tensor_is_rejected():
    import pytest
    import torch
    import ogl_amd  # noqa: F401
    from ogl_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.as_mat(torch.zeros(3, 4))
