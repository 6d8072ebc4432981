import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3
hip=_native.Context()
def bench(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
for (B,c,H,W) in [(16,48,256,3072),(16,96,128,1536)]:
    w=torch.randn(c,c,3,3)/np.sqrt(9*c); b=torch.randn(c,device='cuda')
    pk,un=pack_conv3x3(w.numpy()); wp=torch.from_numpy(pk.view(np.int16)).cuda()
    wz=torch.zeros_like(wp)
    for name,x,ww in (("random x, random w", torch.randn(B,c,H,W,device='cuda'), wp), ("zero x, random w", torch.zeros(B,c,H,W,device='cuda'), wp),
                      ("random x, zero w", torch.randn(B,c,H,W,device='cuda'), wz), ("const x=1 (lo=0), random w", torch.ones(B,c,H,W,device='cuda'), wp)):
        out=torch.empty_like(x)
        t=bench(lambda: hip.conv3x3_f16x3(x, ww, b, c, un, relu=True, out=out))
        print(f"c{c}: {name:28s} {t:.3f} ms")
