"""Debug: cycle totals per phase of workgroup 0 of the F(4x4,3x3) Winograd conv kernel."""
import ctypes, sys, torch
from asvspoof2021_air_amd import ops, _hip
lib = _hip.lib()
CFG = {"l10": (16, 18, 750, 64), "l1": (64, 18, 750, 64), "l2": (128, 9, 375, 128), "l3": (256, 5, 188, 256), "l4": (512, 3, 94, 512)}
for name in (sys.argv[1:] or list(CFG)):
    Cin, H, W, Cout = CFG[name]
    x = torch.randn(64, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    for _ in range(3): ops.conv2d_fwd(x, w, 1, 1)
    tr = torch.zeros(64, dtype=torch.int64, device="cuda")
    lib.air_dbg_wino4_trace.argtypes = [ctypes.c_void_p]
    lib.air_dbg_wino4_trace(ctypes.c_void_p(tr.data_ptr()))
    ops.conv2d_fwd(x, w, 1, 1)
    torch.cuda.synchronize()
    lib.air_dbg_wino4_trace(ctypes.c_void_p(0))
    t = tr.cpu().tolist()
    print("%s workgroup 0: %d cycles, %.1f us wall (%.2f GHz), drain after epilogue %d per item, k-step loops %d per k-step" % (name, t[33], t[34] / 100.0, t[33] / max(1, t[34]) / 10.0, t[35] // max(1, t[6]), t[36] // max(1, t[7])))
    for wv in range(4):
        v = t[8 * wv: 8 * wv + 8]
        S = max(1, v[7]); n = max(1, v[6])
        print("%s wave %d: items %d ksteps %d | per k-step: wait %d barrier %d dma %d read+mfma %d transform %d (sum %d) | epilogue per item %d (stores %d)" % (
            name, wv, v[6], v[7], v[0] // S, v[1] // S, v[2] // S, v[3] // S, v[4] // S, sum(v[:5]) // S, v[5] // n, t[8 * wv + 32] // n if wv == 0 else -1))
