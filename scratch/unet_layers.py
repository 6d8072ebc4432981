import os, sys, time, numpy as np, torch, copy
sys.path.insert(0,'/root/repo')
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights, TfcTdfNet
from oracle.separator import mdx_stft
from oracle import chunking as OC
from audio_cut_amd.testing import signals
torch.set_num_threads(16)
spec=TfcTdfSpec()
w=synth_weights(spec, seed=0)
mix=signals.c2_song(12.3, seed=4)
batch,_,_=OC.mdx_windows(mix[:441000])
x=mdx_stft(batch[:1])[..., :64].contiguous()
net=TfcTdfNet(w,spec).eval()
net64=copy.deepcopy(net).double()
netg=copy.deepcopy(net).cuda()
acts={}
def mk(store,name):
    def hook(m,i,o): store[name]=o.detach().double().cpu()
    return hook
sg,s64,sc={}, {}, {}
for nm,n_,st in (("g",netg,sg),("64",net64,s64),("c",net,sc)):
    for i,b in enumerate(n_.enc): b.register_forward_hook(mk(st,f"enc{i}"))
    n_.bottleneck.register_forward_hook(mk(st,"bott"))
    for i,b in enumerate(n_.dec): b.register_forward_hook(mk(st,f"dec{i}"))
yg=netg(x.cuda()); y64=net64(x.double()); yc=net(x)
for k in s64:
    pk=float(s64[k].abs().max())
    print(k, "gpu err %.2e"%(float((sg[k]-s64[k]).abs().max())/pk), "cpu32 err %.2e"%(float((sc[k]-s64[k]).abs().max())/pk), "peak %.3g std %.3g"%(pk, float(s64[k].std())))
pk=float(y64.abs().max())
print("out gpu %.2e cpu32 %.2e"%(float((yg.cpu().double()-y64).abs().max())/pk, float((yc.double()-y64).abs().max())/pk))
