import sys, torch, torch.nn.functional as F
from asvspoof2021_air_amd import ops
B, Cin, H, W, Cout = [int(v) for v in sys.argv[1:6]]
torch.manual_seed(0)
x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, 3, 3) * 0.1
want = F.conv2d(x.double(), w.double(), None, 1, 1)
got = ops.conv2d_fwd(x.cuda(), w.cuda(), 1, 1).cpu().double()
err = (got - want).abs()
print("max err", err.max().item(), "scale", want.abs().max().item())
bad = (err > 1e-3) | torch.isnan(got)
print("nan count", torch.isnan(got).sum().item())
print("bad count", bad.sum().item(), "of", bad.numel())
idx = bad.nonzero()
if len(idx):
    for d, n in enumerate("b co h w".split()):
        vals = idx[:, d].unique()
        print(n, vals.tolist()[:40], "..." if len(vals) > 40 else "")
for t in idx[:12].tolist():
    b, c, h, ww = t
    print(t, "got", got[b, c, h, ww].item(), "want", want[b, c, h, ww].item(), "| neighbours got", got[b, c, h, ww-1:ww+3].tolist(), "want", want[b, c, h, ww-1:ww+3].tolist())
print("searching sources of bad values")
for t in idx[:6].tolist():
    b, c, h, ww = t
    v = got[b, c, h, ww]
    m = ((want - v).abs() < 2e-5).nonzero()
    print(t, "value", v.item(), "matches want at", m.tolist()[:5])
