"""How a per-seed loss becomes a replay priority.

Three policies with the call shape ``get_priorities(vertex_ids, losses) -> priorities`` the PBR strategy uses
(R/train/graphsage/pytorch/model.py:204,252): the raw loss (what R/train/__main__.py:141 instantiates), an
exponentially smoothed *increase* of the loss, and a convex mix of the two
(behaviour of R/train/prioritized_replay/generate_priority.py:7-58; its ``np.float`` is float64 here because the
alias no longer exists in numpy 2).
"""
import numpy as np


class GeneratePriority:
    def get_priorities(self, batch_nodes_seed, losses):
        raise NotImplementedError


class LossPriority(GeneratePriority):
    def get_priorities(self, batch_nodes_seed, losses):
        return losses                                   # identity: priority == loss


class TrendPriority(GeneratePriority):
    """priority_v <- alpha * priority_v + (1 - alpha) * max(0, loss_v - previous loss_v); a vertex seen for the
    first time starts from the running mean priority of the vertices seen so far."""

    def __init__(self, n_vertices, alpha=0.85):
        self.alpha = alpha
        self.values = np.zeros(n_vertices)              # smoothed trend per vertex
        self.prev_loss = np.zeros(n_vertices)
        self.init = np.ones(n_vertices, dtype=bool)     # True until the vertex is first scored
        self.avg = 0.0                                  # mean of `values` over scored vertices
        self.n_items = 0

    def get_priorities(self, batch_nodes_seed, losses):
        ids = np.asarray(batch_nodes_seed)
        losses = np.asarray(losses, dtype=np.float64)
        newcomers = ids[self.init[ids]]
        self.init[newcomers] = False
        self.values[newcomers] = self.avg
        self.n_items += len(newcomers)
        rise = np.maximum(losses - self.prev_loss[ids], 0.0)
        mass_without_batch = self.avg * self.n_items - self.values[ids].sum()
        self.values[ids] = self.alpha * self.values[ids] + (1.0 - self.alpha) * rise
        self.avg = (mass_without_batch + self.values[ids].sum()) / self.n_items
        self.prev_loss[ids] = losses
        return self.values[ids]


class HybridPriority(GeneratePriority):
    """loss_contrib * loss + (1 - loss_contrib) * trend."""

    def __init__(self, n_vertices, alpha=0.85, loss_contrib=0.5):
        self.loss_contrib = loss_contrib
        self.trend_p = TrendPriority(n_vertices, alpha)
        self.loss_p = LossPriority()

    def get_priorities(self, batch_nodes_seed, losses):
        trend = self.trend_p.get_priorities(batch_nodes_seed, losses)
        return self.loss_contrib * np.asarray(self.loss_p.get_priorities(batch_nodes_seed, losses)) + \
            (1.0 - self.loss_contrib) * trend
