"""Host-side collate cost with the GPU idle and busy, pinned vs pageable destination, torch threads 1 vs default."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from asvspoof2021_air_amd.resnet import ResNet
from asvspoof2021_air_amd.train import Trainer
B, L = 64, 64000
dev = torch.device("cuda")
tr = Trainer(ResNet(3, 256, resnet_type="18", nclasses=2), feat_len=750, device=dev)
tr.enable_graph()
pcm, lab = torch.randn(B, L, device=dev) * 0.1, (torch.rand(B, device=dev) < 0.5).long()
for _ in range(5):
    tr.step(pcm, lab)
torch.cuda.synchronize()
rows = [torch.randn(L) for _ in range(4 * B)]
pinned = torch.empty((B, L), pin_memory=True)
pageable = torch.empty((B, L))
import numpy as np
np_rows = [r.numpy() for r in rows]
np_dst = pageable.numpy()


def collate(dst, k):
    for j in range(B):
        dst[j].copy_(rows[(k * B + j) % len(rows)])


def collate_np(k):
    for j in range(B):
        np_dst[j] = np_rows[(k * B + j) % len(rows)]


def timeit(name, fn, busy):
    if busy:
        for _ in range(40):
            tr.step(pcm, lab)
    t0 = time.perf_counter()
    for k in range(10):
        fn(k)
    dt = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    print("%-40s GPU %s: %.2f ms per batch" % (name, "busy" if busy else "idle", 1e3 * dt), flush=True)


for nt in (torch.get_num_threads(), 1):
    torch.set_num_threads(nt)
    print("torch threads", nt)
    for busy in (False, True):
        timeit("copy_ rows -> pinned", lambda k: collate(pinned, k), busy)
        timeit("copy_ rows -> pageable", lambda k: collate(pageable, k), busy)
        timeit("numpy rows -> pageable", collate_np, busy)
        timeit("torch.stack (pageable)", lambda k: torch.stack(rows[:B]), busy)
print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
