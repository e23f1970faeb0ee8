"""Adam on the HIP kernel (ogl_adam_step_multi_slabs) behind the torch.optim.Optimizer surface the strategies use
(``torch.optim.Adam(params, lr=0.001)``, R/train/graphsage/pytorch/model.py:24-25).

``capturable=True`` keeps ONE step count for the whole parameter set in device memory, so that ``step()`` can be recorded into a
hipGraph and replayed (stepgraph.py); the arithmetic is the same.

Two things a train step's ``zero_grad / backward / step`` (R/.../pytorch/model.py:87,106-107,193,201-202) gains when it goes through
``backward_and_step(loss)`` instead of ``loss.backward(); step()`` — same update, bit for bit:

* **split-K slabs are summed by the optimiser launch** (``consumes_slabs``): inside ``ops.deferred_splitk`` the k-major weight gradients
  skip their reduction launch and leave their partial sums; the Adam kernel adds them in slab order (the reduction launch's bits), writes
  the gradient into ``p.grad`` and applies it.  Four ~8 us launches per Reddit step existed only to sum slabs Adam was about to read.
* **the optimiser runs in two parts** (``early``): the first armed backward pass records the order in which gradients arrive; from then
  on everything but the trailing arrivals (layer 0's ``fc_pool``: its weight gradient is the step's longest launch) is updated from the
  gradient hook of the last early parameter — on the side stream of a forked backward, after everything both streams had enqueued, so
  beside the layer-0 pool backward and weight gradient — and ``step()`` updates the rest.  No launch after that point reads an early
  parameter (the input-gradient products read weight IMAGES built at the start of the step), and the side stream waits for the main
  stream before the update.  The end of the step shrinks from reduce + prepare + Adam over 1.5 M parameters to Adam over 0.36 M."""
import os

import torch

from . import ops

# Off by default — measured (round 4, replayed Reddit step, tools/ab_env.sh OGL_ADAM_EARLY 3, two boxes): 0.993-1.007 ms without, 1.020-1.032 with
# the early part created first at its fork (it took the critical chain's hardware queue: the pool backward started 65 us late); created
# after the critical launch: 1.014-1.028 without, 1.019-1.021 with — the end of the step does shrink (reduce + prepare + Adam 42 us -> Adam
# over fc_pool0 alone 16-18 us), but the side branch's three weight-gradient products end AFTER the layer-0 weight gradient has taken every
# CU (their reductions are gone, they still share the chip with the pool backward), so the early update waits for a CU for 190 us and
# the layer-0 weight gradient runs 250 us instead of 225 beside it.  It pays once that branch ends before the layer-0 weight gradient starts.
EARLY = False


class Adam(torch.optim.Optimizer):
    consumes_slabs = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False, early=None, late_fraction=0.35):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = bool(capturable)
        self._step_dev = None
        self._scalars_dev = None
        self.early = EARLY if early is None else bool(early)
        self.late_fraction = float(late_fraction)
        self.late_min = 0.15
        self._armed = False          # hooks act only inside backward_and_step / backward_learn
        self._learn_only = False
        self._order = []             # arrival order (id(p)) of the armed backward pass
        self._early_ids = None       # None: not learnt yet; frozenset (possibly empty: no early part) afterwards
        self._early_seen = 0
        self._early_done = False     # the early part of the CURRENT step has been applied
        self._by_id = {}
        if self.early:
            for group in self.param_groups:
                for p in group["params"]:
                    if p.requires_grad and p.is_cuda:
                        self._by_id[id(p)] = (p, group)
                        p.register_post_accumulate_grad_hook(self._on_grad)

    # ---- state ------------------------------------------------------------------------------------------------------------
    @property
    def needs_order(self):
        """The early / late split has not been learnt yet (the next armed backward pass records the arrival order)."""
        return self.early and self._early_ids is None and bool(self._by_id)

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

    # ---- the two-part step ------------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        if not self._armed:
            return
        self._order.append(id(p))
        if self._learn_only or not self._early_ids or self._early_done or id(p) not in self._early_ids:
            return
        self._early_seen += 1
        if self._early_seen == len(self._early_ids):
            # every early gradient exists (its kernels are enqueued): update those parameters beside the rest of the backward.  The
            # update waits for everything the MAIN stream has enqueued up to here (the last reader of an early parameter's weight
            # image is behind us) and for the side stream's own queue — but its launches are CREATED after the main stream's next
            # launch (ops._DEFERRED): in a captured step the first-created child of a node keeps its parent's hardware queue, and
            # created first the update took the critical chain's queue — the pool backward then sat behind the side branch's
            # weight gradients on the other one (measured: its first launch 65 us late).
            here = ops.fork_point()

            def run():
                with ops.early_section(here):
                    self._apply([self._by_id[i] for i in self._early_ids], prepare=True)
                self._early_done = True
            ops._DEFERRED.append(run)

    def _learn_split(self):
        order = list(dict.fromkeys(self._order))
        with_grad = [i for i, (p, _) in self._by_id.items() if p.grad is not None]
        if len(order) != len(with_grad) or not order:
            return                                      # (not every gradient arrived through a hook: try again next time)
        # late = the SHORTEST suffix of the arrival order that holds a worthwhile share of the elements (>= late_min: the tensors
        # whose gradients come out of the step's last, longest launches), never more than late_fraction of them
        total = sum(self._by_id[i][0].numel() for i in order)
        late, acc = [], 0
        for i in reversed(order):
            n = self._by_id[i][0].numel()
            if late and (acc >= self.late_min * total or acc + n > self.late_fraction * total):
                break
            late.append(i); acc += n
        early = [i for i in order if i not in set(late)]
        self._early_ids = frozenset(early) if (early and late) else frozenset()

    def _arm(self, learn_only=False):
        self._armed, self._learn_only = True, bool(learn_only)
        self._order, self._early_seen, self._early_done = [], 0, False

    def backward_learn(self, loss):
        """A backward pass that only records the arrival order of the gradients (no update): what a captured step runs once,
        outside its capture, so that the capture itself already has the two-part form."""
        self._arm(learn_only=True)
        try:
            ops.backward(loss)
        finally:
            self._armed = False
        if self.needs_order:
            self._learn_split()

    def prime(self):
        """Call BEFORE the step's forward pass: the step's first launch (the weight images, ops.weight_images_prepare) then also
        advances the device-side step count and computes the bias-correction scalars — the one-thread launch ``step()`` would put
        in front of the update, at the end of the step, goes.  Without such a launch in the forward pass ``step()`` prepares as usual."""
        if self.capturable and len(self.param_groups) == 1:
            g = self.param_groups[0]
            dev = next((p.device for p in g["params"] if p.is_cuda), None)
            if dev is not None:
                step_dev, scal = self._device_state(dev)
                ops.adam_prime(step_dev, scal, g["lr"], g["betas"][0], g["betas"][1])

    def backward_and_step(self, loss):
        """``loss.backward(); self.step()`` with the split-K slabs left to this optimiser and its early part launched from the
        gradient hooks (module docstring).  Gradients must start from zero_grad(): one backward pass per step."""
        with ops.deferred_splitk(self):
            self._arm()
            try:
                ops.backward(loss)
            finally:
                self._armed = False
            self.step()

    # ---- the update -------------------------------------------------------------------------------------------------------
    def _apply(self, items, prepare):
        """Adam on ``items`` = [(parameter, its group)]: one launch per (group, host step count)."""
        by_group = {}
        for p, group in items:
            if p.grad is not None:
                by_group.setdefault(id(group), (group, []))[1].append(p)
        first = True
        for group, params in by_group.values():
            b1, b2 = group["betas"]
            buckets = {}
            for p in params:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                sg = ops.take_slabs(p)
                g = p.grad
                if not g.is_contiguous():
                    if sg is not None:                      # (never for this package's kernels: the slab sum is written into g)
                        ops.slab_reduce(sg, g.contiguous()); sg = None
                    g = g.contiguous()
                key = 0
                if not self.capturable:
                    st["step"] = st.get("step", 0) + 1
                    key = st["step"]
                buckets.setdefault(key, []).append((p.data, g, st["exp_avg"], st["exp_avg_sq"], sg))
            for step, its in buckets.items():               # normally one bucket: every tensor in ONE launch
                ps, gs, ms, vs, sgs = zip(*its)
                if self.capturable:
                    step_dev, scal = self._device_state(ps[0].device)
                    if prepare and first and ops.adam_primed(step_dev):
                        prepare = False                     # (the step's weight-image launch already did: see prime())
                    ops.adam_step_multi_slabs(ps, gs, ms, vs, sgs, step_dev=step_dev, scalars_dev=scal, prepare=prepare and first,
                                              lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"])
                else:
                    ops.adam_step_multi_slabs(ps, gs, ms, vs, sgs, step=step, lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"])
                first = False

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ops.assert_no_pending_gradients()   # (a gradient a fused node promised and nobody wrote must never reach the update)
        ops._flush_deferred()               # (an early part still waiting for "the main stream's next launch": there is none)
        ops.side_join()                     # (a forked backward joins itself when it ends; this is for gradients made by hand)
        ops.invalidate_weight_images()      # the kernels below write the parameters through raw pointers (no version bump)
        done = self._early_ids if self._early_done else frozenset()
        items = [(p, group) for group in self.param_groups for p in group["params"] if p.grad is not None and id(p) not in done]
        self._apply(items, prepare=not self._early_done)
        if self.needs_order and self._order:
            self._learn_split()
        self._early_done, self._order = False, []
        return loss
