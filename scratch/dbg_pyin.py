import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.testing import signals
from oracle import librosa_ops as L
hip=_native.Context()
x=signals.voice_with_rests(14.0,seed=3)
sr=44100; fmin=L.note_to_hz('C2'); fmax=L.note_to_hz('C7')
xd=hip.to_device(x)
_, cm = hip.yin_f0(xd, sr, fmin, fmax, 2048, 441, want_cmnd=True)
cmh=cm.cpu().numpy().T.copy()
ref_cm,mp,_=L.cmnd_frames(x,sr,fmin,fmax,2048,441)
print("cmnd max abs diff", np.abs(cmh-ref_cm).max())
obs,vp,nb,bps=L.pyin_observations(cmh,sr,fmin,fmax,mp)
f0,vf,vpg=hip.pyin(xd,sr,fmin,fmax,2048,441)
print("vp diff (oracle obs on GPU cmnd vs GPU kernel)", np.abs(vp-vpg).max(), np.argmax(np.abs(vp-vpg)))
i=int(np.argmax(np.abs(vp-vpg)))
print("frame",i,"oracle vp",vp[i],"gpu vp",vpg[i])
col=cmh[:,i]
tr=L._localmin0(col); tr[0]=col[0]<col[1]
idx=np.flatnonzero(tr); print("n troughs",len(idx), "heights", col[idx][:10], "min", col.min())
