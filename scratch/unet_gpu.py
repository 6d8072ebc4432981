import os, sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights, TfcTdfNet
from oracle.separator import unet_forward, mdx_stft
from oracle import chunking as OC
from audio_cut_amd.testing import signals
tag=sys.argv[1] if len(sys.argv)>1 else "default"
B=int(sys.argv[2]) if len(sys.argv)>2 else 4
spec=TfcTdfSpec()
w=synth_weights(spec, seed=0)
mix=signals.c2_song(12.3, seed=4)
batch,_,_=OC.mdx_windows(mix[:441000])
x=mdx_stft(batch[:1]).contiguous()   # [1,4,3072,256]
ref_path='/tmp/unet_ref64.npy'
if os.path.exists(ref_path): a64=torch.from_numpy(np.load(ref_path))
else:
    torch.set_num_threads(16)
    w64={k:v.astype(np.float64) for k,v in w.items()}
    t=time.time(); a64=unet_forward(x.double(),w64); print("cpu f64 s", time.time()-t); np.save(ref_path,a64.numpy())
pk=float(a64.abs().max())
net=TfcTdfNet(w,spec).cuda().eval()
xg=x.cuda()
with torch.no_grad():
    y=net(xg)
torch.cuda.synchronize()
print(tag, "err vs f64 (rel peak):", float((y.cpu().double()-a64).abs().max())/pk)
xb=xg.transpose(-1,-2).contiguous().repeat(B,1,1,1)
for it in range(3):
    torch.cuda.synchronize(); t=time.time()
    with torch.no_grad(): yb=net.forward_tf(xb)
    torch.cuda.synchronize(); dt=time.time()-t
    print(tag, f"B={B} iter{it}: {dt*1000:.1f} ms  -> {spec.flops_per_item()*B/dt/1e12:.1f} TFLOP/s, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
