import sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
hip=_native.Context()
backend=MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip, max_items_per_forward=32); backend.load_model()
sp=SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
mix=signals.c2_song(240.0, seed=2)
r=sp.split_track(mix); r=sp.split_track(mix)
st=r["device_state"]; cache=r["feature_cache"]
bounds=r["sample_boundaries"]; voc=r["vocal_track"]
from audio_cut_amd.cutting.refine import CutPoint
for _ in range(2):
    t=time.perf_counter(); sp._apply_boundary_policy(bounds, voc, len(mix), cache, vocal_dev=st.get("vocal")); print("policy ms", (time.perf_counter()-t)*1e3, len(bounds))
pr=cProfile.Profile(); pr.enable(); sp._apply_boundary_policy(bounds, voc, len(mix), cache, vocal_dev=st.get("vocal")); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
