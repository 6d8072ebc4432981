"""Does any kernel read memory nobody wrote?  Every torch.empty the binding layer makes is filled with NaN (floats) / 0x7f bytes
(integers) before use; a track is separated and split with and without the poison and the results must be identical."""
import sys, os, hashlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
hip = _native.Context("cuda:0")
backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip, max_items_per_forward=32); backend.load_model()
sp = SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
mix = signals.c2_song(60.0, seed=100); dev = hip.to_device(mix)
def run():
    r = sp.split_track(mix, audio_dev=dev)
    return hashlib.sha1(r["vocal_track"].tobytes()).hexdigest()[:10], r["sample_boundaries"], r["cuts_samples"], bool(np.isfinite(r["vocal_track"]).all())
a = run(); print("clean   :", a, flush=True)
real_empty = torch.empty
def poisoned(*args, **kw):
    t = real_empty(*args, **kw)
    if t.is_cuda:
        if t.dtype.is_floating_point: t.fill_(float("nan"))
        else: t.view(torch.uint8).fill_(0x7f)
    return t
torch.empty = poisoned
import audio_cut_amd.separation.backends as B, audio_cut_amd.separation.tfc_tdf as T
b = run(); print("poisoned:", b, flush=True)
c = run(); print("poisoned:", c, flush=True)
torch.empty = real_empty
print("IDENTICAL" if a == b == c else "DIFFERENT: some kernel reads uninitialised memory")
