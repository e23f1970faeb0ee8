"""Adam on the HIP kernel (ogl_adam_step_multi) behind the torch.optim.Optimizer surface the strategies use
(``torch.optim.Adam(params, lr=0.001)``, R/train/graphsage/pytorch/model.py:24-25).

``capturable=True`` keeps ONE step count for the whole parameter set in device memory (``ogl_adam_step_multi_dev``), so
that ``step()`` can be recorded into a hipGraph and replayed (stepgraph.py); the arithmetic is the same."""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = bool(capturable)
        self._step_dev = None
        self._scalars_dev = None

    def _device_state(self, device):
        if self._step_dev is None:
            self._step_dev = torch.zeros(1, dtype=torch.int64, device=device)
            self._scalars_dev = torch.zeros(2, dtype=torch.float32, device=device)
        return self._step_dev, self._scalars_dev

    def prepare_capture(self):
        """Allocate every piece of state outside a capture (moments of all parameters, the device-side step count)."""
        assert self.capturable
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                self._device_state(p.device)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ops.side_join()                     # (a forked backward joins itself when it ends; this is for gradients made by hand)
        ops.invalidate_weight_images()      # the kernels below write the parameters through raw pointers (no version bump)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            if self.capturable:
                items = []
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    st = self.state[p]
                    if "exp_avg" not in st:
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    items.append((p.data, g, st["exp_avg"], st["exp_avg_sq"]))
                if items:
                    step_dev, scal = self._device_state(items[0][0].device)
                    ps, gs, ms, vs = zip(*items)
                    ops.adam_step_multi_dev(ps, gs, ms, vs, step_dev, scal, group["lr"], b1, b2, group["eps"])
                continue
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
