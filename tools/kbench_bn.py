"""Micro-benchmark of the BatchNorm kernels at the ResNet's shapes (bytes moved / time)."""
import sys, torch
from asvspoof2021_air_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SH = {"bn1": (16, 20, 750), "l1": (64, 18, 750), "l2": (128, 9, 375), "l3": (256, 5, 188), "l4": (512, 3, 94)}
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, (C, H, W) in SH.items():
    x = torch.randn(B, C, H, W, device="cuda"); dy = torch.randn_like(x)
    g = torch.rand(C, device="cuda") + 0.5; b = torch.randn(C, device="cuda") * 0.1
    mean, invstd, scale, shift = ops.bn_stats(x, g, b)
    nb = x.numel() * 4
    ts = timeit(lambda: ops.bn_stats(x, g, b))
    ta = timeit(lambda: ops.bn_apply(x, scale, shift, relu=True))
    tb = timeit(lambda: ops.bn_bwd(x, dy, mean, invstd, g, b, relu=True))
    print("%-4s %6.1f MB | stats %.3f ms %.2f TB/s | apply %.3f ms %.2f TB/s | bwd %.3f ms %.2f TB/s (5 passes)" % (
        name, nb / 1e6, ts, nb / ts / 1e9, ta, 2 * nb / ta / 1e9, tb, 5 * nb / tb / 1e9))
# ECAPA-TDNN shapes (B = 128, T = 750: planes start on 8-byte boundaries only)
if len(sys.argv) > 2 and sys.argv[2] == "ecapa":
    for name, (Bx, C, T) in {"e512": (128, 512, 750), "e64": (128, 64, 750), "e128": (128, 128, 750), "e512a": (128, 512, 752)}.items():
        x = torch.randn(Bx, C, T, device="cuda").relu_(); dy = torch.randn_like(x)
        g = torch.rand(C, device="cuda") + 0.5; b = torch.randn(C, device="cuda") * 0.1
        mean, invstd, scale, shift = ops.bn_stats(x, g, b)
        dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda"); dbias = torch.empty(C, device="cuda")
        nb = x.numel() * 4
        ts = timeit(lambda: ops.bn_stats(x, g, b))
        ta = timeit(lambda: ops.bn_apply(x, scale, shift))
        tb = timeit(lambda: ops.bn_bwd(x, dy, mean, invstd, g, b, relu_in=True, dgamma=dg, dbeta=db, dbias=dbias))
        print("%-5s %6.1f MB | stats %.3f ms %.2f TB/s | apply %.3f ms %.2f TB/s | bwd(+dbias) %.3f ms %.2f TB/s (5 passes)" % (
            name, nb / 1e6, ts, nb / ts / 1e9, ta, 2 * nb / ta / 1e9, tb, 5 * nb / tb / 1e9))
