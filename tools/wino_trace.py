"""Debug: cycle stamps of workgroup 0 of the Winograd conv kernel (prologue / item loop / epilogue)."""
import ctypes, sys, torch
from asvspoof2021_air_amd import ops, _hip
lib = _hip.lib()
name = sys.argv[1] if len(sys.argv) > 1 else "l1"
CFG = {"l1": (64, 18, 750, 64), "l2": (128, 9, 375, 128), "l3": (256, 5, 188, 256), "l4": (512, 3, 94, 512)}
Cin, H, W, Cout = CFG[name]
x = torch.randn(64, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
for _ in range(3): ops.conv2d_fwd(x, w, 1, 1)
tr = torch.zeros(128, dtype=torch.int64, device="cuda")
lib.air_dbg_wino_trace.argtypes = [ctypes.c_void_p]
lib.air_dbg_wino_trace(ctypes.c_void_p(tr.data_ptr()))
ops.conv2d_fwd(x, w, 1, 1)
torch.cuda.synchronize()
lib.air_dbg_wino_trace(ctypes.c_void_p(0))
t = tr.cpu().tolist()
c0, w0 = t[62], t[63]
n = max(i for i in range(60) if t[i]) + 1
print("clock: %.3f cycles per 10ns wall tick" % ((t[n - 1] - c0) / max(1, t[64 + n - 1] - w0)))
nch = (64 // 4) if name == "l1" else Cin // 4
for w in range(4):
    v = t[96 + 4 * w: 100 + 4 * w]
    print("wave", w, "totals: S0 %d  wait %d  barrier %d  S1 %d  (sum %d)" % (v[0], v[1], v[2], v[3], sum(v)))
prev = c0
for i in range(n):
    print(i, "kind", "loopstart" if i == 0 else ("mainloop" if i % 2 == 1 else "epilogue"), "dcycles", t[i] - prev, "wall_us %.2f" % ((t[64 + i] - w0) / 100.0))
    prev = t[i]
