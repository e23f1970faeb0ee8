from .snapshot_graph import SnapshotGraph, build_time_ordered_csr  # noqa: F401
from .dynamic_graph import DynamicGraph  # noqa: F401
from .dynamic_graph_vertex import DynamicGraphVertex  # noqa: F401
from .dynamic_graph_edge import DynamicGraphEdge  # noqa: F401
from .train_test_graph import TrainTestGraph  # noqa: F401
