import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
torch.set_num_threads(16)
from audio_cut_amd import _native
from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
from oracle import e2e as OE, refine as OR
hip=_native.Context()
w=synth_weights(TfcTdfSpec(), seed=0)
backend=MDX23HipBackend(weights=w, ctx=hip, max_items_per_forward=32); backend.load_model()
sp=SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
dur=float(sys.argv[1]) if len(sys.argv)>1 else 240.0
mix=signals.c2_song(dur, seed=2)
r=sp.split_track(mix)
print("gpu bounds", len(r["sample_boundaries"]), "cuts", len(r["cuts_samples"]), flush=True)
OR.LEGACY_PROMOTION=True
t=time.time(); ref=OE.run_track(mix,44100,w); print("oracle s", time.time()-t, flush=True)
print("bounds equal:", r["sample_boundaries"]==ref.sample_boundaries)
print("cuts equal:", r["cuts_samples"]==ref.policy.cuts, "flags equal:", r["segment_vocal_flags"]==ref.policy.flags)
v=r["vocal_track"]; print("stem err", float(np.max(np.abs(v-ref.vocal))/np.max(np.abs(ref.vocal))))
if r["sample_boundaries"]!=ref.sample_boundaries:
    print(r["sample_boundaries"]); print(ref.sample_boundaries)
if r["cuts_samples"]!=ref.policy.cuts:
    print(r["cuts_samples"]); print(ref.policy.cuts)
