"""Round 2: the soak track's guard boundary #8 - how close is the decision?  GPU stems vs the live CPU oracle."""
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
w = synth_weights(TfcTdfSpec(), seed=17)
mix = signals.c1_sine_silence(170.0, seed=64)
backend = MDX23HipBackend(weights=w, ctx=hip, max_items_per_forward=32); backend.load_model()
sp = SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
r = sp.split_track(mix)
t0 = time.time(); ref = OE.run_track(mix, 44100, w); print("oracle", time.time() - t0, flush=True)
gv, ov = r["vocal_track"], ref.vocal
peak = float(np.max(np.abs(ov)))
print("stem err of peak", float(np.max(np.abs(gv - ov))) / peak)
b, bo = r["sample_boundaries"], ref.sample_boundaries
print("diff", [(i, x, y) for i, (x, y) in enumerate(zip(b, bo)) if x != y])
np.set_printoptions(precision=12, linewidth=220)
for wave, name in ((gv, "gpu"), (ov, "oracle")):
    lo, hi = 1947297 - 40000, 1947297 + 40000
    db = OR.moving_meansq_db(wave[lo:hi], 3528)
    c = 40000
    print(name, "db around:", db[c - 6:c + 5])
    print(name, "vocal |x| near +1763/-1764:", np.abs(wave[1947297 + 1763 - 3:1947297 + 1763 + 3]), np.abs(wave[1947297 - 1764 - 3:1947297 - 1764 + 3]))
# relative error of the GPU stem by local level (10 ms RMS bins), quiet vs loud
n = len(mix) // 441
g2 = np.sqrt(np.mean((gv[:n * 441].reshape(n, 441).astype(np.float64) - ov[:n * 441].reshape(n, 441)) ** 2, axis=1))
o2 = np.sqrt(np.mean(ov[:n * 441].reshape(n, 441).astype(np.float64) ** 2, axis=1))
for lo_db, hi_db in ((-200, -100), (-100, -80), (-80, -60), (-60, -40), (-40, -20), (-20, 0)):
    lvl = 20 * np.log10(o2 / peak + 1e-300)
    m = (lvl >= lo_db) & (lvl < hi_db)
    if m.any():
        print(f"level [{lo_db},{hi_db}) dB of peak: {int(m.sum())} bins, median rel err {np.median(g2[m] / (o2[m] + 1e-300)):.2e}, median abs err of peak {np.median(g2[m]) / peak:.2e}")
