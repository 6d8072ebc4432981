import sys, numpy as np, traceback
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
hip=_native.Context()
w=synth_weights(TfcTdfSpec(), seed=0)
backend=MDX23HipBackend(weights=w, ctx=hip); backend.load_model()
sp=SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
cases={"silence12": np.zeros(12*44100, np.float32), "short1s": signals.c2_song(1.0, seed=1), "tiny": signals.c2_song(0.28, seed=1)[:12345],
       "exact10s": signals.c2_song(10.0, seed=2), "exact7.5": signals.c2_song(7.5, seed=2), "odd": signals.c2_song(23.7, seed=3)[:-7],
       "loud_clip": np.clip(signals.c2_song(11.0, seed=5)*3, -1, 1).astype(np.float32), "dc": np.full(9*44100, 0.3, np.float32)}
for name,x in cases.items():
    try:
        for mode in ("v2.2_mdd","vpbd_acoustic"):
            r=sp.split_track(x, mode=mode)
            print(name, mode, len(x), "bounds", r["sample_boundaries"][:6], "cuts", r.get("cuts_samples", None) and r["cuts_samples"][:6], r.get("note"))
    except Exception as e:
        print(name, "FAILED", type(e).__name__, e); traceback.print_exc(limit=4)
try:
    sp.split_track(np.zeros(0, np.float32))
    print("empty: returned")
except Exception as e:
    print("empty:", type(e).__name__, e)
