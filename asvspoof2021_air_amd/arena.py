"""Flat parameter / gradient arenas.

MI355X-first memory layout for the train step: every trainable tensor of a
model is a view into ONE contiguous fp32 buffer, and every gradient a view into
a second one.  The optimiser is then a single HBM-bound launch over the arena
(csrc/optim.hip) and the data-parallel gradient exchange a handful of large
RCCL all-reduces over slices of the gradient arena instead of one per tensor.

Parameters keep their ``nn.Parameter`` identity, names and shapes, so
``state_dict`` / ``load_state_dict`` / pickling see the reference's layout
(SURVEY.md §5 checkpoint row).
"""
import torch


class ParamArena:
    def __init__(self, named_params, tail_names=()):
        """named_params: list of (name, nn.Parameter).  Parameters whose name is in
        ``tail_names`` are placed last so the ones that receive gradients form a
        contiguous prefix (fc_mu.* / fc7.*, bn7.* get no gradient under ang_iso:
        SURVEY.md §3b)."""
        head = [(n, p) for n, p in named_params if n not in tail_names]
        tail = [(n, p) for n, p in named_params if n in tail_names]
        self.entries = []  # (name, param, offset, numel)
        off = 0
        for n, p in head + tail:
            cnt = p.numel()
            self.entries.append((n, p, off, cnt))
            off += (cnt + 3) // 4 * 4  # keep every view 16-byte aligned
        self.total = off
        self.head_total = 0
        for n, p, o, c in self.entries:
            if n not in tail_names:
                self.head_total = o + (c + 3) // 4 * 4
        self.flat = None
        self.grad = None
        self.device = None
        self.tail_has_grad = False

    def bind(self, device):
        """(Re)allocate the arenas on ``device`` and re-point every parameter at its view."""
        flat = torch.zeros(self.total, device=device, dtype=torch.float32)
        grad = torch.zeros(self.total, device=device, dtype=torch.float32)
        with torch.no_grad():
            for n, p, o, c in self.entries:
                view = flat[o:o + c].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.flat, self.grad, self.device = flat, grad, device
        return self

    def bound(self):
        if self.flat is None:
            return False
        for n, p, o, c in self.entries:
            if p.data_ptr() != self.flat.data_ptr() + 4 * o or p.device != self.flat.device:
                return False
        return True

    def grad_view(self, name_or_index):
        n, p, o, c = self._entry(name_or_index)
        return self.grad[o:o + c].view(p.shape)

    def _entry(self, key):
        if isinstance(key, int):
            return self.entries[key]
        for e in self.entries:
            if e[0] == key:
                return e
        raise KeyError(key)

    def grad_views(self):
        return {n: self.grad[o:o + c].view(p.shape) for n, p, o, c in self.entries}
