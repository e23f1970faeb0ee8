"""`load` for the bitcoin vertex stream — same files and return tuple as R/train/dataset_utils/bitcoin.py:78-113."""
from .common_utils import load_vertex_stream

FILES = ["feat_data.npy", "targets.npy", "graph.adjlist", "vertex_timestamp.json"]


def load(path, snapshots=100, cuda=True, copy_to_gpu=True):
    """-> (feat_size, targets[N,1], dynamic_graph, n_classes, dynamic_graph_test)"""
    return load_vertex_stream(path, "feat_data.npy", "vertex_timestamp.json", snapshots, cuda, copy_to_gpu)
