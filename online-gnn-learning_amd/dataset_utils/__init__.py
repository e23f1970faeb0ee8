"""On-disk dataset loaders with the reference's ``load(path, snapshots, cuda, copy_to_gpu)`` signature and file
formats (R/train/dataset_utils/{pubmed,arxiv,bitcoin,reddit}.py); SURVEY.md §8(f)-4.  No downloader: there is
no network, missing files raise FileNotFoundError naming them."""
from . import arxiv, bitcoin, pubmed, reddit  # noqa: F401
from .common_utils import load_edge_stream, load_vertex_stream, read_adjlist  # noqa: F401
