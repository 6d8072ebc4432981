import numpy as np, torch, time, sys
sys.path.insert(0,'/root/repo')
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights, TfcTdfNet, _calibration_spectrogram
from oracle.separator import unet_forward, mdx_stft
from oracle import chunking as OC
from audio_cut_amd.testing import signals
torch.set_num_threads(8)
spec=TfcTdfSpec()
t=time.time(); w=synth_weights(spec, seed=0); print("synth", time.time()-t)
mix=signals.c2_song(12.3, seed=4)
batch,_,_=OC.mdx_windows(mix[:441000])
x=mdx_stft(batch[:1])[..., :64].contiguous()   # [1,4,3072,64]
print("in std", float(x.std()), float(x.abs().max()))
t=time.time(); a32=unet_forward(x,w); print("f32 unfused", time.time()-t)
net=TfcTdfNet(w,spec); b32=net(x)
w64={k:v.astype(np.float64) for k,v in w.items()}
t=time.time(); a64=unet_forward(x.double(),w64); print("f64", time.time()-t)
pk=float(a64.abs().max())
print("out std", float(a64.std()), "peak", pk)
print("f32 unfused vs f64:", float((a32.double()-a64).abs().max())/pk)
print("f32 folded  vs f64:", float((b32.double()-a64).abs().max())/pk)
print("f32 folded vs unfused:", float((b32-a32).abs().max())/pk)
