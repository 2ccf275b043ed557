"""Core clock / socket power WHILE a kernel family runs (amd-smi sampled from the host in the middle of a few seconds of
queued launches).  The bench's smi_before / smi_after samples are taken with the GPU idle; this is the figure that says
whether a kernel's MFMA-busy fraction is priced at 2.4 GHz or at a power-capped clock.
Usage (GPU box): python tools/exp_clock_under_load.py"""
import sys
import time
import torch
sys.path.insert(0, ".")
from bench import smi_sample
from asvspoof2021_air_amd import ops_h, ops


def under_load(name, fn, seconds=4.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    n = int(seconds * 1e6 / us)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    time.sleep(seconds * 0.35)
    s1 = smi_sample()
    s2 = smi_sample()
    torch.cuda.synchronize()
    us2 = a.elapsed_time(b) / n * 1e3
    print("%-28s %8.1f us cold-ish, %8.1f us sustained | %s | %s" % (
        name, us, us2, {k: s1.get(k) for k in ("gfx_0_mhz", "socket_power_w", "hotspot_c")},
        {k: s2.get(k) for k in ("gfx_0_mhz", "socket_power_w", "hotspot_c")}), flush=True)
    time.sleep(1.0)


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    print("idle", smi_sample())
    B, T, C4, C = 128, 750, 1536, 512
    x4 = ops_h.from_f32(torch.randn(B, C4, T, device=dev, generator=g))
    w4 = torch.randn(C4, C4, 1, device=dev, generator=g) * 0.05
    o4 = torch.empty_like(x4)
    under_load("bf16 GEMM 1536x1536x96000", lambda: ops_h.conv_pointwise(x4, w4, T, relu=True, out=o4))
    x = ops_h.from_f32(torch.randn(B, C, T, device=dev, generator=g))
    w = torch.randn(C, C, 1, device=dev, generator=g) * 0.05
    o = torch.empty_like(x)
    under_load("bf16 GEMM 512x512x96000", lambda: ops_h.conv_pointwise(x, w, T, relu=True, out=o))
    # f32 Winograd conv, ResNet layer1 shape
    xi = torch.randn(64, 64, 30, 750, device=dev, generator=g)
    wi = torch.randn(64, 64, 3, 3, device=dev, generator=g) * 0.05
    under_load("f32 wino4 conv layer1", lambda: ops.conv2d_fwd(xi, wi, stride=1, padding=1))
    # HBM-bound copy
    big = torch.empty(1 << 28, device=dev)
    big2 = torch.empty_like(big)
    under_load("copy 1 GiB", lambda: big2.copy_(big))
    print("idle", smi_sample())


if __name__ == "__main__":
    main()
