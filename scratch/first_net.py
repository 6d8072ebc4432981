import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.tfc_tdf import TfcTdfNet, TfcTdfSpec, synth_weights
hip=_native.Context()
spec=TfcTdfSpec(); w=synth_weights(spec, seed=0)
net=TfcTdfNet(w, spec, hip=hip).to(hip.device).eval()
x=(torch.randn(32,4,256,3072,device='cuda')*2)
def bench(n=3):
    for _ in range(2): net.forward_tf(x)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): net.forward_tf(x)
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
for fuse in (True, False, True, False):
    hip.fuse_first_conv=fuse
    net.conv_probe=[]
    t=bench()
    pr=net.conv_probe; net.conv_probe=None
    torch.cuda.synchronize()
    first=[e0.elapsed_time(e1) for e0,e1,_ in pr[:3]]
    print("fuse", fuse, f"forward {t:.2f} ms; first three conv launches ms:", [round(v,3) for v in first], "n probes/forward", len(pr)//5)
