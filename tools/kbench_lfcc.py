import torch, time
from asvspoof2021_air_amd.feature_extraction import LFCC
m = LFCC(320,160,512,16000,20).cuda(); m.mutate_input=False
for B in (64, 256):
    x = 0.1*torch.randn(B,64000,device='cuda')
    for _ in range(5): y = m(x)
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): y = m(x)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e)/50*1e3
    byts = B*(64000*4+401*60*4)
    print("LFCC B=%d: %.1f us/launch  %.2f TB/s algorithmic  %.0f utt/s" % (B, us, byts/us/1e6, B/us*1e6))
    s.record()
    for _ in range(50): y = m.forward_padded(x, 750)
    e.record(); torch.cuda.synchronize()
    print("  padded: %.1f us" % (s.elapsed_time(e)/50*1e3))
