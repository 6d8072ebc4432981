"""One soak case under several kernel configurations against one oracle run."""
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
dur, sseed, wseed, gen = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
w = synth_weights(TfcTdfSpec(), seed=wseed)
mix = getattr(signals, gen)(dur, seed=sseed)
mix = np.mean(mix, axis=0).astype(np.float32) if mix.ndim == 2 else mix
res = {}
for name, wide, conv, tdf in (("final", True, "f16x3", "f16x3"),):
    hip.conv_wide, hip.conv_impl, hip.tdf_impl = wide, conv, tdf
    backend = MDX23HipBackend(weights=w, ctx=hip, max_items_per_forward=16 if conv == "miopen" else 32); backend.load_model()
    sp = SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
    res[name] = sp.split_track(mix)
    print(name, "done", flush=True)
    del backend, sp
hip.conv_wide, hip.conv_impl, hip.tdf_impl = True, "f16x3", "f16x3"
t0 = time.time(); ref = OE.run_track(mix, 44100, w); print("oracle", time.time() - t0, flush=True)
peak = float(np.max(np.abs(ref.vocal)))
for name, r in res.items():
    b = r["sample_boundaries"]
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(b, ref.sample_boundaries)) if x != y] if len(b) == len(ref.sample_boundaries) else "length differs"
    print(f"{name}: boundaries equal={b == ref.sample_boundaries} cuts equal={r['cuts_samples'] == ref.policy.cuts} stem_err={float(np.max(np.abs(r['vocal_track'] - ref.vocal))) / peak:.2e} diff={diff}")
# candidates near the first differing boundary
r = res["final"]
d = [(x, y) for x, y in zip(r["sample_boundaries"], ref.sample_boundaries) if x != y]
if d:
    t = d[0][1] / 44100.0
    print("oracle candidates near:", [(round(a * 44100, 3), b) for a, b in ref.cut_candidates if abs(a - t) < 0.5])
    print("gpu    candidates near:", [(round(a * 44100, 3), b) for a, b in r["cut_candidates"] if abs(a - t) < 0.5])
    print("oracle pauses near:", [(p.start_time, p.end_time, p.cut_point * 44100) for p in ref.pauses if abs(p.cut_point - t) < 0.5])
    print("gpu    pauses near:", [(p.start_time, p.end_time, p.cut_point * 44100) for p in r["pauses"] if abs(p.cut_point - t) < 0.5])
    print("gpu adjustments near:", [a for a in r.get("guard_adjustments_unfiltered", []) if abs(a.final_time - t) < 0.5])
# what the guard saw around the first differing boundary
r = res["final"]
d = [(x, y) for x, y in zip(r["sample_boundaries"], ref.sample_boundaries) if x != y]
if d:
    x, y = d[0]
    lo, hi = min(x, y) - 3, max(x, y) + 4
    np.set_printoptions(precision=9, linewidth=200)
    print("oracle vocal :", ref.vocal[lo:hi])
    print("gpu    vocal :", r["vocal_track"][lo:hi])
    print("mix          :", mix[lo:hi])
    print("indices      :", list(range(lo, hi)))
