"""MI355X-native streaming-GraphSAGE update path (sample -> gather -> aggregate -> project).

Host side mirrors the reference's ``train/graphsage`` + ``backend=pytorch`` surface; all arithmetic
runs in libogl_hip.so (hand-written HIP for gfx950) through the C-ABI in include/ogl_hip.h.
Import as ``ogl_amd`` (see ogl_amd.py at the repo root).
"""
from . import _lib  # noqa: F401
from . import torch_ops  # noqa: F401  (registers torch.ops.ogl.*)

__version__ = "0.1.0"
