"""Import-path parity with the reference (``from prioritized_replay.generate_priority import LossPriority``,
R/train/__main__.py:8): the policies live in :mod:`.priorities`."""
from .priorities import GeneratePriority, HybridPriority, LossPriority, TrendPriority  # noqa: F401
