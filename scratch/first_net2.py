import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.tfc_tdf import TfcTdfNet, TfcTdfSpec, synth_weights
from audio_cut_amd.separation.conv_pack import pack_conv3x3
hip=_native.Context()
def bench(fn,n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
B,c,H,W=16,48,256,3072
w1=(torch.randn(c,4)*0.5).cuda(); b1=(torch.randn(c)*0.3).cuda()
w=torch.randn(c,c,3,3)/np.sqrt(9*c); b=torch.randn(c,device='cuda')
pk,un=pack_conv3x3(w.numpy()); wp=torch.from_numpy(pk.view(np.int16)).cuda()
for name,spec in (("randn", torch.randn(B,4,H,W,device='cuda')), ("zeros", torch.zeros(B,4,H,W,device='cuda')),
                  ("denormal", torch.full((B,4,H,W),1e-41,device='cuda')), ("b1=0 & zeros", None)):
    bb=b1
    if spec is None:
        spec=torch.zeros(B,4,H,W,device='cuda'); bb=torch.zeros_like(b1)
    t2=bench(lambda: hip.conv3x3_f16x3_first(spec, w1, bb, wp, b, c, un, relu=True))
    mid=hip.conv1x1_small(spec,w1,bb,relu=True)
    t1=bench(lambda: hip.conv3x3_f16x3(mid, wp, b, c, un, relu=True))
    print(f"{name:14s} fused {t2:.3f} ms | plain conv on the generated tensor {t1:.3f} ms")
