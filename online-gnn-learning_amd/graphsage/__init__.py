from .sageconv import SAGEConv, GatheredRows  # noqa: F401
from .graphsage import GraphSAGE  # noqa: F401
