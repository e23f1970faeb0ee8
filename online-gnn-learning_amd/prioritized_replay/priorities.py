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


class DeviceTrend:
    """The state of a ``TrendPriority`` / ``HybridPriority`` in HBM (``ogl_priority_trend``): the PBR passes' per-seed losses become
    priorities without leaving the device.  Built from the host object the driver constructed (R/train/__main__.py:141 style:
    ``TrendPriority(n_vertices)``), whose arrays it takes over; ``to_host()`` writes the state back into an equivalent host object."""

    def __init__(self, strategy, device):
        import torch
        trend = strategy.trend_p if isinstance(strategy, HybridPriority) else strategy
        assert isinstance(trend, TrendPriority)
        self.loss_contrib = float(strategy.loss_contrib) if isinstance(strategy, HybridPriority) else -1.0
        self.alpha = float(trend.alpha)
        self.device = torch.device(device)
        self.values = torch.as_tensor(trend.values, dtype=torch.float64).to(self.device)
        self.prev_loss = torch.as_tensor(trend.prev_loss, dtype=torch.float64).to(self.device)
        self.init = torch.as_tensor(trend.init.astype(np.uint8)).to(self.device)
        self.stats = torch.tensor([float(trend.avg), float(trend.n_items)], dtype=torch.float64, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.n_vertices = int(self.values.numel())
        self.dirty = False            # the device state has moved past the host object's (write_back() brings it up to date)

    def get_priorities_device(self, ids_dev, losses_dev):
        """ids_dev int64 [n] (distinct original vertex ids, on the device), losses_dev float32 / float64 [n] -> float64 [n]."""
        import torch
        from .. import _lib
        from ..ops import _ptr, _stream
        n = int(ids_dev.numel())
        assert ids_dev.dtype == torch.int64 and ids_dev.is_cuda and ids_dev.is_contiguous() and losses_dev.numel() == n
        losses_dev = losses_dev.contiguous()
        # zeros, not empty: the kernel skips a row whose id is outside [0, n_vertices) (and raises `err`, see check()); such a row
        # must not carry whatever the allocation held into the replay tree
        from ..ops import fill_zero
        out = fill_zero(torch.empty(n, dtype=torch.float64, device=self.device))
        self.dirty = True
        l32 = losses_dev if losses_dev.dtype == torch.float32 else None
        l64 = losses_dev if losses_dev.dtype == torch.float64 else None
        if l32 is None and l64 is None:
            l32 = losses_dev.float()
        _lib.check(_lib.lib().ogl_priority_trend(_ptr(ids_dev), _ptr(l32), _ptr(l64), n, self.n_vertices, _ptr(self.values),
                                                 _ptr(self.prev_loss), _ptr(self.init), _ptr(self.stats), self.alpha, self.loss_contrib,
                                                 _ptr(out), _ptr(self.err), _stream()), "ogl_priority_trend")
        return out

    def check(self):
        if int(self.err.item()):
            raise IndexError("a vertex id outside [0, n_vertices) reached the trend priorities")

    @staticmethod
    def check_ids_host(ids_host, n_vertices):
        """The kernel's preconditions on the HOST copy of the ids (they come from the host: subgraph_to_id[seeds]): inside
        [0, n_vertices) and distinct (one thread per row updates that vertex's state: duplicates would race).  O(n), no sync."""
        ids = np.asarray(ids_host, dtype=np.int64).reshape(-1)
        if ids.size == 0:
            return
        lo, hi = int(ids.min()), int(ids.max())
        if lo < 0 or hi >= n_vertices:
            raise IndexError("a vertex id outside [0, %d) reached the trend priorities (min %d, max %d)" % (n_vertices, lo, hi))
        if ids.size > 1 and int(np.bincount(ids - lo, minlength=1).max()) > 1:
            raise ValueError("duplicate vertex ids in one trend-priority batch (the reference's dict update keeps the last one; "
                             "the device kernel requires distinct ids)")

    def write_back(self, strategy):
        """Bring the host object this state was taken from up to date IN PLACE (anything that reads the strategy object on the
        host — logging, a checkpoint, a hand-over to host-side code — sees the current state)."""
        trend = strategy.trend_p if isinstance(strategy, HybridPriority) else strategy
        self.check()
        trend.values[:] = self.values.cpu().numpy()
        trend.prev_loss[:] = self.prev_loss.cpu().numpy()
        trend.init[:] = self.init.cpu().numpy().astype(bool)
        st = self.stats.cpu().numpy()
        trend.avg, trend.n_items = float(st[0]), int(round(st[1]))
        self.dirty = False

    def to_host(self):
        """A host strategy object in the state this one is in (tests; hand-over back to host-side code)."""
        trend = TrendPriority(self.n_vertices, self.alpha)
        trend.values = self.values.cpu().numpy().copy()
        trend.prev_loss = self.prev_loss.cpu().numpy().copy()
        trend.init = self.init.cpu().numpy().astype(bool)
        st = self.stats.cpu().numpy()
        trend.avg, trend.n_items = float(st[0]), int(round(st[1]))
        if self.loss_contrib < 0:
            return trend
        hy = HybridPriority(self.n_vertices, self.alpha, self.loss_contrib)
        hy.trend_p = trend
        return hy
