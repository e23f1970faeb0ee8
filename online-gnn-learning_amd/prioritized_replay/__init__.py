from .replay_buffer import PrioritizedReplayBuffer, SumTree  # noqa: F401
from .generate_priority import GeneratePriority, LossPriority, TrendPriority, HybridPriority  # noqa: F401
