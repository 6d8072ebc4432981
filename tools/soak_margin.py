"""How close is a guard decision, and how accurate is the GPU stem where it is taken?  GPU stems vs the live CPU oracle.
usage: python tools/soak_margin.py [dur,song_seed,weight_seed,generator [sample]]   (default: the round-1 soak track, boundary #8)
Prints the moved boundaries, the oracle's dB series around `sample` (default: the first moved boundary, else 1947297), the two
stems' magnitudes where samples enter / leave the 80 ms window, the GPU stem's relative error in 64-sample bins around the
decision, and the relative error by level over the whole track."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
torch.set_num_threads(16)
from audio_cut_amd import _native
from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
from oracle import e2e as OE, refine as OR
OR.LEGACY_PROMOTION = True
hip = _native.Context()
spec = (sys.argv[1] if len(sys.argv) > 1 else "170,64,17,c1_sine_silence").split(",")
dur, sseed, wseed, gen = float(spec[0]), int(spec[1]), int(spec[2]), spec[3]
w = synth_weights(TfcTdfSpec(), seed=wseed)
mix = getattr(signals, gen)(dur, seed=sseed)
mix = np.mean(mix, axis=0).astype(np.float32) if mix.ndim == 2 else mix
backend = MDX23HipBackend(weights=w, ctx=hip, max_items_per_forward=32); backend.load_model()
sp = SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
r = sp.split_track(mix)
t0 = time.time(); ref = OE.run_track(mix, 44100, w); print("oracle", round(time.time() - t0, 1), "s", flush=True)
gv, ov = r["vocal_track"], ref.vocal
peak = float(np.max(np.abs(ov)))
print("stem err of peak", float(np.max(np.abs(gv - ov))) / peak)
b, bo = r["sample_boundaries"], ref.sample_boundaries
moved = [(i, x, y) for i, (x, y) in enumerate(zip(b, bo)) if x != y]
print("moved (index, gpu, oracle):", moved)
at = int(sys.argv[2]) if len(sys.argv) > 2 else (moved[0][2] if moved else 1947297)
np.set_printoptions(precision=6, linewidth=220)
W = 3528
for wave, name in ((gv, "gpu"), (ov, "oracle")):
    lo = max(0, at - 40000); hi = min(len(mix), at + 40000)
    db = OR.moving_meansq_db(wave[lo:hi], W)
    c = at - lo
    print(name, "dB at", at, "-6..+5:", db[c - 6:c + 6])
    print(name, "|x| entering the window (at+1763-3..+3):", np.abs(wave[at + 1763 - 3:at + 1763 + 4]))
    print(name, "|x| leaving the window (at-1764-3..+3):", np.abs(wave[at - 1764 - 3:at - 1764 + 4]))
print("64-sample bins around the decision: offset, oracle rms, gpu/oracle rms ratio - 1, max |gpu - oracle| / oracle rms")
for off in range(-4096, 4097, 512):
    s0 = at + off
    o = ov[s0:s0 + 64].astype(np.float64); g = gv[s0:s0 + 64].astype(np.float64)
    orms = float(np.sqrt(np.mean(o * o)))
    if orms > 0:
        print(f"  {off:+6d}: {orms:.3e}  {float(np.sqrt(np.mean(g * g))) / orms - 1:+.2e}  {float(np.max(np.abs(g - o))) / orms:.2e}")
    else:
        print(f"  {off:+6d}: oracle exactly 0, gpu rms {float(np.sqrt(np.mean(g * g))):.3e}")
n = len(mix) // 441
g2 = np.sqrt(np.mean((gv[:n * 441].reshape(n, 441).astype(np.float64) - ov[:n * 441].reshape(n, 441)) ** 2, axis=1))
o2 = np.sqrt(np.mean(ov[:n * 441].reshape(n, 441).astype(np.float64) ** 2, axis=1))
lvl = 20 * np.log10(o2 / peak + 1e-300)
for lo_db, hi_db in ((-400, -300), (-300, -200), (-200, -100), (-100, -80), (-80, -60), (-60, -40), (-40, -20), (-20, 0)):
    m = (lvl >= lo_db) & (lvl < hi_db) & (o2 > 0)
    if m.any():
        print(f"level [{lo_db},{hi_db}) dB of peak: {int(m.sum())} bins of 10 ms, median rel err {np.median(g2[m] / o2[m]):.2e}, 95th pct {np.percentile(g2[m] / o2[m], 95):.2e}")
