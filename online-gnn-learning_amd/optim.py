"""Adam on the HIP kernel (ogl_adam_step) behind the torch.optim.Optimizer surface the strategies use
(``torch.optim.Adam(params, lr=0.001)``, R/train/graphsage/pytorch/model.py:24-25)."""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            buckets = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                buckets.setdefault(st["step"], []).append((p.data, g, st["exp_avg"], st["exp_avg_sq"]))
            for step, items in buckets.items():      # normally one bucket: every tensor in ONE launch
                ps, gs, ms, vs = zip(*items)
                ops.adam_step_multi(ps, gs, ms, vs, step, group["lr"], b1, b2, group["eps"])
        return loss
