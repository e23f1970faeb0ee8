"""`load` for the reddit edge stream — same files and return tuple as R/train/dataset_utils/reddit.py:144-177."""
from .common_utils import load_edge_stream

FILES = ["feat_data.npy", "targets.npy", "edges_dataframe.csv"]


def load(path, snapshots=100, cuda=True, copy_to_gpu=True):
    """-> (feat_size, targets[N,1], dynamic_graph, n_classes, dynamic_graph_test)"""
    return load_edge_stream(path, snapshots, cuda, copy_to_gpu)
