import torch, torch.nn.functional as F
from asvspoof2021_air_amd import _hip, ops
for (B,Cin,H,W,Cout) in [(1,32,2,32,128),(1,32,4,64,128),(2,32,4,66,128),(3,64,9,75,128)]:
    g=torch.Generator().manual_seed(1)
    x=torch.randn(B,Cin,H,W,generator=g); w=(torch.randn(Cout,Cin,3,3,generator=g)*0.05).double().requires_grad_(True)
    y=F.conv2d(x.double(),w,None,2,1); dy=torch.randn(y.shape,generator=g)
    ref,=torch.autograd.grad(y,w,dy.double())
    with _hip.options(CONV_S2=31):
        got=ops.conv2d_wgrad(x.cuda(),dy.cuda(),tuple(w.shape),2,1).cpu().double()
    e=(got-ref).abs()
    print((B,Cin,H,W,Cout),"Wo",y.shape[3],"scale %.3f"%ref.abs().max().item())
    print(" per tap err:", [["%.1e"%e[:,:,a,b].max().item() for b in range(3)] for a in range(3)])
    print(" per co-tile err:", ["%.1e"%e[32*m:32*m+32].max().item() for m in range(Cout//32)])
