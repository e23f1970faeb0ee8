"""Priority strategies (surface of R/train/prioritized_replay/generate_priority.py:3-58)."""
import numpy as np


class GeneratePriority:
    def get_priorities(self, batch_nodes_seed, losses):
        raise NotImplementedError


class LossPriority(GeneratePriority):
    """priority = per-seed loss (generate_priority.py:7-9)."""

    def get_priorities(self, batch_nodes_seed, losses):
        return losses


class TrendPriority(GeneratePriority):
    """Exponentially smoothed positive loss trend (generate_priority.py:11-45; the reference's
    ``np.float`` no longer exists in numpy 2 — float64 is what it meant)."""

    def __init__(self, n_vertices, alpha=0.85):
        self.values = np.zeros(n_vertices, dtype=np.float64)
        self.prev_loss = np.zeros(n_vertices, dtype=np.float64)
        self.init = np.ones(n_vertices, dtype=bool)
        self.avg, self.n_items, self.alpha = 0.0, 0, alpha

    def get_priorities(self, batch_nodes_seed, losses):
        ids = np.asarray(batch_nodes_seed)
        fresh = ids[self.init[ids]]
        self.init[fresh] = False
        self.values[fresh] = self.avg
        self.n_items += len(fresh)
        update = np.clip(losses - self.prev_loss[ids], 0, None)
        total = self.avg * self.n_items - np.sum(self.values[ids])
        self.values[ids] = self.values[ids] * self.alpha + update * (1 - self.alpha)
        self.avg = (total + np.sum(self.values[ids])) / self.n_items
        self.prev_loss[ids] = losses
        return self.values[ids]


class HybridPriority(GeneratePriority):
    def __init__(self, n_vertices, alpha=0.85, loss_contrib=0.5):
        self.trend_p, self.loss_p, self.loss_contrib = TrendPriority(n_vertices, alpha), LossPriority(), loss_contrib

    def get_priorities(self, batch_nodes_seed, losses):
        return (self.trend_p.get_priorities(batch_nodes_seed, losses) * (1 - self.loss_contrib)
                + self.loss_p.get_priorities(batch_nodes_seed, losses) * self.loss_contrib)
