"""Data-parallel gradient exchange: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no multi-GPU path.  Here utterances are
sharded across ranks; each rank runs the whole hot path on its shard with its
own BatchNorm statistics (per-GPU batch == the reference's batch, SURVEY.md
§8e), and the only exchange is a SUM all-reduce of the gradient arena -- a few
large buckets in reverse-layer order launched back to back, so RCCL pipelines
them over the 7 xGMI links -- plus the 256-float loss centre.  The optimiser
divides by the world size (grad_scale), so no extra pass touches the arena.
"""
import os

import torch
import torch.distributed as td

BUCKET_BYTES = 16 << 20  # 49.8 MB of ResNet gradients -> 3-4 buckets


def world_size():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def rank():
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def init_from_env(backend=None, device_index=None):
    """Initialise from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  Returns local rank.
    ``device_index``: GPU of this rank (default LOCAL_RANK)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not td.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local if device_index is None else device_index)
        td.init_process_group(backend=backend)
    return local


def all_mean(value):
    """Mean of a Python scalar over the ranks (validation loss before ``Trainer.save_checkpoint`` compares
    it with the best so far: every rank must take the same early-stopping decision)."""
    if world_size() == 1:
        return float(value)
    dev = "cuda" if td.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item()) / world_size()


def bucket_slices(n_elems, bucket_bytes=BUCKET_BYTES):
    """Reverse-order (last layers first) [start, end) slices of a flat arena."""
    per = max(1, bucket_bytes // 4)
    out = []
    end = n_elems
    while end > 0:
        start = max(0, end - per)
        out.append((start, end))
        end = start
    return out


def allreduce_flat(flat, n_elems, bucket_bytes=BUCKET_BYTES):
    """SUM all-reduce flat[:n_elems] in buckets, asynchronously; returns the work handles."""
    works = []
    for s, e in bucket_slices(n_elems, bucket_bytes):
        works.append(td.all_reduce(flat[s:e], op=td.ReduceOp.SUM, async_op=True))
    return works


class GradBucketer:
    """All-reduce of the gradient arena launched from INSIDE the backward pass (BASELINE configs[3]:
    "gradient all-reduce overlapped with backward").

    The backward pass produces gradients from the end of the arena towards its start (last layers
    first).  The model reports ``ready(lo, events)`` after it has enqueued everything that writes
    arena.grad[lo:]; once a bucket's worth of finished tail has accumulated, its SUM all-reduce is
    issued from a dedicated launch stream that waits on those events only - the main stream (dgrad
    chain) and the side stream (weight gradients) keep running underneath the transfer.  What is left
    at the end of backward (the first layers, < one bucket) is reduced by ``allreduce_grads``."""

    def __init__(self, bucket_bytes=None):
        self.bucket_bytes = bucket_bytes
        self.comm = None
        self.total_launched = 0  # buckets sent from inside backward since construction
        self.reset(None, 0)

    def reset(self, grad, hi):
        self.grad, self.hi, self.lo = grad, hi, hi
        self.events, self.works, self.launched = [], [], 0

    def ready(self, lo, events):
        if self.grad is None or lo >= self.lo:
            return
        self.lo = lo
        self.events = [e for e in events if e is not None]
        if (self.hi - self.lo) * 4 >= (self.bucket_bytes or BUCKET_BYTES):
            self.flush()

    def flush(self):
        if self.grad is None or self.lo >= self.hi:
            return
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=self.grad.device)
        for ev in self.events:
            self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            self.works.append(td.all_reduce(self.grad[self.lo:self.hi], op=td.ReduceOp.SUM, async_op=True))
        self.launched += 1
        self.total_launched += 1
        self.hi = self.lo
        self.events = []


def allreduce_grads(model, loss_module=None):
    """Sum gradients over ranks: the model's gradient arena (skipping the tail that has
    no gradient under ang_iso) and the loss centre.  Regions a GradBucketer already sent during
    backward are only waited for."""
    if world_size() == 1:
        return
    arena = model.arena()
    bucketer = getattr(model, "_bucketer", None)
    works = []
    if bucketer is not None and bucketer.grad is arena.grad and bucketer.launched:
        works = allreduce_flat(arena.grad, bucketer.hi) + bucketer.works
        if arena.tail_has_grad:
            works.append(td.all_reduce(arena.grad[arena.head_total:arena.total], op=td.ReduceOp.SUM, async_op=True))
        bucketer.reset(None, 0)
    else:
        if bucketer is not None:
            bucketer.reset(None, 0)
        n = arena.total if arena.tail_has_grad else arena.head_total
        works = allreduce_flat(arena.grad, n)
    if loss_module is not None:
        for p in loss_module.parameters():
            if p.grad is not None:
                works.append(td.all_reduce(p.grad, op=td.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
