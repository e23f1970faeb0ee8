from .replay_buffer import PrioritizedReplayBuffer, SumTree  # noqa: F401
from .priorities import GeneratePriority, LossPriority, TrendPriority, HybridPriority  # noqa: F401
