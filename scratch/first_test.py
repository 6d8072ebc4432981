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
B,c,H,W=16,48,256,3072
spec=torch.randn(B,4,H,W,device='cuda'); w1=(torch.randn(c,4)*0.5).cuda(); b1=(torch.randn(c)*0.3).cuda()
w=torch.randn(c,c,3,3)/np.sqrt(9*c); b=torch.randn(c,device='cuda')
pk,un=pack_conv3x3(w.numpy()); wp=torch.from_numpy(pk.view(np.int16)).cuda()
mid=hip.conv1x1_small(spec,w1,b1,relu=True)
t0=bench(lambda: hip.conv1x1_small(spec,w1,b1,relu=True))
t1=bench(lambda: hip.conv3x3_f16x3(mid, wp, b, c, un, relu=True))
t2=bench(lambda: hip.conv3x3_f16x3_first(spec, w1, b1, wp, b, c, un, relu=True))
print(f"conv1x1 {t0:.3f} ms + conv3x3 {t1:.3f} ms = {t0+t1:.3f} | fused {t2:.3f} ms")
