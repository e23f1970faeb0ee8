"""Backend shim: the plugin registration point of the reference (R/train/utils.py:16-38).

``init(library, GPU, GPU_ID)`` returns the same 6-tuple the driver unpacks at
R/train/__main__.py:99 — ``(GraphSAGE, RandomSupervised, PrioritizedSupervised, NoRehSupervised,
FullSupervised, activation)`` — for the new backend ``Lib_supported.HIP``.  A maintainer adds one
``elif`` to the reference's ``init``; ``PYTORCH`` is accepted as an alias so
``python train <dataset> pytorch ...`` lands on the HIP path unchanged.
"""
import enum

import numpy as np
import torch

LIB = None
_GPU_ID = None
_GPU = False


class Lib_supported(enum.Enum):
    PYTORCH = 1
    TF = 2
    TF_STATIC = 3
    HIP = 4


def init(library, GPU=True, GPU_ID=-1):
    global LIB, _GPU_ID, _GPU
    _GPU, _GPU_ID, LIB = GPU, GPU_ID, library
    if library in (Lib_supported.HIP, Lib_supported.PYTORCH):
        from .graphsage.graphsage import GraphSAGE
        from .graphsage.model import (FullHipSupervisedGraphSage, NoRehHipSupervisedGraphSage,
                                      PrioritizedHipSupervisedGraphSage, RandomHipSupervisedGraphSage)
        if not GPU:
            raise RuntimeError("the hip backend needs a GPU (cuda=True); there is no CPU fallback")
        if GPU_ID is not None and GPU_ID >= 0:
            torch.cuda.set_device(GPU_ID)       # the reference passes the bool here (utils.py:31); the id is what it means
        return (GraphSAGE, RandomHipSupervisedGraphSage, PrioritizedHipSupervisedGraphSage, NoRehHipSupervisedGraphSage,
                FullHipSupervisedGraphSage, torch.nn.functional.relu)
    raise NotImplementedError("backend %r is outside this build (TensorFlow backends are out of scope)" % (library,))


def to_nn_lib(data, GPU=True, dtype=None):
    """float64 -> float32 and optional H2D (R/train/utils.py:41-67)."""
    t = torch.tensor(data) if not isinstance(data, torch.Tensor) else data
    if t.dtype == torch.float64:
        t = t.float()
    return t.cuda() if GPU else t


def from_nn_lib_to_list(data):
    return data.tolist()


def from_nn_lib_to_set(data):
    return {v.item() for v in data}


def from_nn_lib_to_numpy(data):
    return data.cpu().numpy() if data.is_cuda else data.numpy()


def from_nn_get_python_value(tensor):
    return tensor.item()


def get_context():
    return torch.device("cuda", torch.cuda.current_device())


def index_tensor(tensor, indices):
    return tensor[indices]


class sparse1d:
    """1-row sparse id map (R/train/utils.py:132-142); the hip graph classes use dense numpy maps, this
    is kept for callers that still hand one in."""

    def __init__(self, sparse_matrix):
        self.mtx = sparse_matrix

    def __getitem__(self, items):
        if hasattr(items, "__len__") and not isinstance(items, str):
            return np.squeeze(self.mtx[0, items].toarray())
        return self.mtx[0, items]

    def __setitem__(self, keys, items):
        self.mtx[0, keys] = items
