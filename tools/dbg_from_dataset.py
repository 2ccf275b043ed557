"""Where does the from-dataset step go?  Host time of next(batch) / trainer.step per step, and the step time with the
prefetcher's pieces switched one at a time (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torch.utils.data import DataLoader
from asvspoof2021_air_amd import dataset as air_ds
from asvspoof2021_air_amd.resnet import ResNet
from asvspoof2021_air_amd.train import Trainer

B, L = 64, 64000
dev = torch.device("cuda")
torch.manual_seed(688)
tr = Trainer(ResNet(3, 256, resnet_type="18", nclasses=2), feat_len=750, device=dev)
tr.enable_graph()
src = air_ds.SyntheticSource(688, 4 * B, length=L, device=dev, cache_items=None)
for i in range(4 * B):
    src.pcm(i)
ds = air_ds.ASVspoof2019("LA", None, "train", feat_len=750, source=src, return_pcm="batch")
dl = DataLoader(ds, batch_size=B, shuffle=True, drop_last=True, collate_fn=ds.collate_fn, num_workers=0)


def forever(make):
    while True:
        for b in make():
            yield b


def run(name, it, n=60):
    for _ in range(10):
        b = next(it)
        tr.step(b[0], b[3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tn = ts = 0.0
    for _ in range(n):
        a = time.perf_counter()
        b = next(it)
        c = time.perf_counter()
        tr.step(b[0], b[3])
        d = time.perf_counter()
        tn += c - a
        ts += d - c
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-44s %.2f ms/step | host: next %.2f ms, step %.2f ms" % (name, 1e3 * dt / n, 1e3 * tn / n, 1e3 * ts / n), flush=True)


resident = [(torch.randn(B, L, device=dev) * 0.1, (torch.rand(B, device=dev) < 0.5).long()) for _ in range(4)]
run("resident", forever(lambda: [(p, None, None, l) for p, l in resident]))
run("prefetcher depth 2", forever(lambda: air_ds.DevicePrefetcher(dl, dev, depth=2)))
run("prefetcher depth 4", forever(lambda: air_ds.DevicePrefetcher(dl, dev, depth=4)))


def plain():
    for b in dl:
        yield [b[0].to(dev, non_blocking=True), b[1], b[2], b[3].to(dev)]


run("loader + .to() on the compute stream", forever(plain))


def host_only():
    for b in dl:
        yield [resident[0][0], b[1], b[2], resident[0][1]]


run("loader on the host, resident tensors to the step", forever(host_only))

# ---- where does `next` spend its time?  the prefetcher's pieces timed one by one
import collections
pf = air_ds.DevicePrefetcher(dl, dev, depth=2)
T = collections.defaultdict(float)
orig_to = pf._to_device


def timed_to(item, slot, k):
    a = time.perf_counter()
    r = orig_to(item, slot, k)
    T["to_device[%d]" % k] += time.perf_counter() - a
    return r


pf._to_device = timed_to
orig_collate = ds.collate_fn


def timed_collate(samples):
    a = time.perf_counter()
    r = orig_collate(samples)
    T["collate"] += time.perf_counter() - a
    return r


dl2 = DataLoader(ds, batch_size=B, shuffle=True, drop_last=True, collate_fn=timed_collate, num_workers=0)
pf.loader = dl2
orig_getitem = type(ds).__getitem__
n = 0
t0 = time.perf_counter()
for ep in range(8):
    for b in pf:
        a = time.perf_counter()
        tr.step(b[0], b[3])
        T["step"] += time.perf_counter() - a
        n += 1
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("per batch over %d batches: total %.2f ms |" % (n, 1e3 * tot / n), " ".join("%s %.2f" % (k, 1e3 * v / n) for k, v in sorted(T.items())))


# which Event.synchronize blocks?  (dataset pinned ring vs prefetcher slot ring)
orig_sync = torch.cuda.Event.synchronize
acc = collections.defaultdict(lambda: [0, 0.0])
import traceback


def timed_sync(self):
    a = time.perf_counter()
    r = orig_sync(self)
    d = time.perf_counter() - a
    where = traceback.extract_stack(limit=2)[0]
    k = "%s:%d" % (os.path.basename(where.filename), where.lineno)
    acc[k][0] += 1
    acc[k][1] += d
    return r


torch.cuda.Event.synchronize = timed_sync
pf3 = air_ds.DevicePrefetcher(dl, dev, depth=2)
for ep in range(8):
    for b in pf3:
        tr.step(b[0], b[3])
torch.cuda.Event.synchronize = orig_sync
torch.cuda.synchronize()
for k, (n, t) in acc.items():
    print("Event.synchronize at %s: %d calls, %.2f ms each" % (k, n, 1e3 * t / max(1, n)))
