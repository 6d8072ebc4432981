"""Where the HOST spends the tail of a single track (everything after the U-Net is queued): cProfile of SeamlessSplitter.split_track on the
C2 track, one job at a time (no pipelining), third run profiled.  usage: python tools/tail_host_profile.py [seconds=240] [top=45]"""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
hip = _native.Context()
backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip, max_items_per_forward=32); backend.load_model()
sp = SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
mix = signals.c2_song(dur, seed=2)
mix = np.mean(mix, axis=0).astype(np.float32) if mix.ndim == 2 else mix
mix_dev = hip.to_device(mix)
for _ in range(2):
    t0 = time.perf_counter(); r = sp.split_track(mix, audio_dev=mix_dev); torch.cuda.synchronize(); print(f"warm-up {1e3 * (time.perf_counter() - t0):.1f} ms", r["timings"])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); r = sp.split_track(mix, audio_dev=mix_dev); pr.disable()
print(f"profiled run {1e3 * (time.perf_counter() - t0):.1f} ms (cProfile adds overhead)", r["timings"], "policy", r.get("timings_policy_s"))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("audio_cut_amd", top); print(s.getvalue())
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue())
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_callers("method 'cpu'|method 'to' of|synchronize"); print(s.getvalue()[:6000])
