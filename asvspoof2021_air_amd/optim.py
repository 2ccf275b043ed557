"""Optimisers of the hot path on the flat arenas (csrc/optim.hip).

``FusedAdam(model)`` == torch.optim.Adam(model.parameters(), lr, betas, eps,
weight_decay) as configured at main_train.py:175-176 (coupled L2 decay; tensors
without a gradient are skipped), as ONE launch over the model's parameter arena.
``FusedSGD(module)`` == torch.optim.SGD(module.parameters(), lr) (main_train.py:272).
Both expose ``param_groups[0]['lr']`` so main_train.py:144-147's step decay works.
"""
from . import ops


class FusedAdam:
    def __init__(self, model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4):
        self.model = model
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]
        self.step_count = 0
        self.tail_steps = 0
        self.m = None
        self.v = None

    def _params(self):
        """The model's parameters, listed once (Module.parameters() walks the module tree on every call)."""
        arena = self.model.arena()
        return [p for _, p, _, _ in arena.entries]

    def zero_grad(self, set_to_none=True):
        for p in self._params():
            p.grad = None

    def step(self, grad_scale=1.0):
        """One Adam step over the gradient arena.  Like torch.optim.Adam, tensors without a gradient are left
        alone: a step() with no backward since zero_grad() is a no-op (the arena still holds the previous
        step's sums, which must not be applied twice)."""
        import torch
        params = self._params()
        if any(not p.requires_grad for p in params):
            raise RuntimeError("FusedAdam updates the whole parameter arena in one launch: frozen parameters "
                               "(requires_grad=False) are not supported")
        if all(p.grad is None for p in params):
            return
        arena = self.model.arena()
        if self.m is None or self.m.device != arena.flat.device:
            self.m = torch.zeros_like(arena.flat)
            self.v = torch.zeros_like(arena.flat)
            self.tail_steps = 0
        g = self.param_groups[0]
        self.step_count += 1
        n = arena.head_total
        ops.adam_step(arena.flat[:n], arena.grad[:n], self.m[:n], self.v[:n], self.step_count,
                      g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], grad_scale)
        if arena.tail_has_grad and arena.total > n:
            # the tail (fc_mu.* / fc7.*, bn7.*: gradients only under the CE base loss) keeps its own step
            # count, so its bias correction matches torch.optim.Adam's per-tensor state when it receives
            # gradients only some of the time
            self.tail_steps += 1
            ops.adam_step(arena.flat[n:], arena.grad[n:], self.m[n:], self.v[n:], self.tail_steps,
                          g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], grad_scale)


class FusedSGD:
    def __init__(self, module, lr=5e-4):
        self.params = [p for p in module.parameters()]
        self.param_groups = [{"lr": lr}]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def step(self, grad_scale=1.0):
        for p in self.params:
            if p.grad is not None:
                ops.sgd_step(p.data, p.grad.contiguous(), self.param_groups[0]["lr"], grad_scale)
