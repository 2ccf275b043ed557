import torch, sys
from asvspoof2021_air_amd import ops
torch.manual_seed(0)
for (Cin,H,W,Cout) in [(16,18,750,64),(64,18,750,64),(128,9,375,128),(256,5,188,256),(512,3,94,512)]:
    B=64
    x=torch.randn(B,Cin,H,W,device="cuda"); w=torch.randn(Cout,Cin,3,3,device="cuda")*0.05
    res=torch.randn(B,Cout,H,W,device="cuda")+0.5
    g=torch.ones(Cout,device="cuda"); b=torch.zeros(Cout,device="cuda")
    for r in (None,res):
        y,rec=ops.conv2d_fwd(x,w,1,1,residual=r,stats=True)
        a=ops.bn_stats(y,g,b); c=ops.bn_stats(y,g,b,stats_in=rec)
        yd=y.double(); m=yd.mean((0,2,3)); v=yd.var((0,2,3),unbiased=False)
        e=lambda t,ref: float(((t.double()-ref).abs()/ref.abs().clamp_min(1e-3)).max())
        print(Cin,H,W,Cout, "res" if r is not None else "   ", "own: mean %.2e istd %.2e | rec: mean %.2e istd %.2e"%(e(a[0],m),e(a[1],1/torch.sqrt(v+1e-5)),e(c[0],m),e(c[1],1/torch.sqrt(v+1e-5))))
    # timing of the two bn_stats flavours
    import time
    for name,kw in (("own",{}),("rec",{"stats_in":rec})):
        for _ in range(3): ops.bn_stats(y,g,b,**kw)
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(20): ops.bn_stats(y,g,b,**kw)
        torch.cuda.synchronize(); print("   bn_stats %s %.1f us"%(name,(time.perf_counter()-t)/20*1e6))
