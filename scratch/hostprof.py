import sys, cProfile, pstats, io, time, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
hip=_native.Context("cuda:0")
be=MDX23HipBackend(weights=synth_weights(TfcTdfSpec(),0), ctx=hip); be.load_model()
sp=SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=be))
mix=signals.c2_song(240.0, seed=2); md=hip.to_device(mix)
sp.split_track(mix, audio_dev=md); sp.split_track(mix, audio_dev=md)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
t=time.time(); r=sp.split_track(mix, audio_dev=md); torch.cuda.synchronize(); print("step s", time.time()-t)
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:6500])
