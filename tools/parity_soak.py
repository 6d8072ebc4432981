"""Parity soak: the GPU path against the CPU oracle, live, on tracks / weights no fixture covers.
usage: python tools/parity_soak.py "dur,song_seed,weight_seed[,generator[,silero_seed]]" ...   (one progress line per case)
With a silero_seed the chunked VAD is the Silero network (HIP kernels vs oracle/silero.py) on seeded synthetic weights instead of the
no-weights energy gate.  Their output layer is calibrated PER TRACK on the stem the VAD will see (the first 30 s of the GPU path's
vocal stem from a first, weight-less pass; tests/silero_synth.calibration_affine_on) - a fixed burst calibration found no speech at all
on songs and sung lines (round 3: vad_segments=0 on nine of eighteen tracks, which then exercise the Silero kernels but not the focus
windows).  A Silero-mode track COUNTS as one only with >= 3 VAD segments; the tally at the end says how many did."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(16)
from audio_cut_amd import _native
from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
from audio_cut_amd import config as C
from oracle import e2e as OE, refine as OR, silero as OS
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from guard_plateau import boundary_context, classify_boundaries, map_cuts      # the equivalence class of a boundary decided on numerical dust
OR.LEGACY_PROMOTION = True
hip = _native.Context()
ok_all = True
n_exact = n_equiv = n_fail = 0
n_silero = n_silero_counted = 0
for arg in sys.argv[1:]:
    parts = arg.split(",")
    dur, sseed, wseed = float(parts[0]), int(parts[1]), int(parts[2])
    gen = parts[3] if len(parts) > 3 else "c2_song"
    silero_seed = int(parts[4]) if len(parts) > 4 else None
    w = synth_weights(TfcTdfSpec(), seed=wseed)
    backend = MDX23HipBackend(weights=w, ctx=hip, max_items_per_forward=32); backend.load_model()
    saved = C.snapshot(); vad_fn = None
    mix = getattr(signals, gen)(dur, seed=sseed)
    mix = np.mean(mix, axis=0).astype(np.float32) if mix.ndim == 2 else mix
    if silero_seed is not None:
        from silero_synth import calibration_affine_on
        from audio_cut_amd.testing.silero_synth import synth_silero_weights
        first = SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend)).split_track(mix)     # weight-less pass: the stem
        bursts = gen == "c1_sine_silence"
        affine = calibration_affine_on(first["vocal_track"][: 30 * 44100], silero_seed, spread=5.0 if bursts else 8.0, q=(0.2, 0.8) if bursts else (0.6, 0.95))
        sw = synth_silero_weights(silero_seed, affine=affine)
        wpath = os.path.join(tempfile.mkdtemp(), "silero.npz"); np.savez(wpath, **sw)
        C.set_runtime_config({"advanced_vad.silero_weights_path": wpath})
        vad_fn = OS.silero_vad_fn(44100, sw)
        del first
    try:
        sp = SeamlessSplitter(44100, separator=EnhancedVocalSeparator(44100, backend=backend))
        t0 = time.time(); r = sp.split_track(mix); tg = time.time() - t0
    finally:
        C.restore(saved)
    t0 = time.time(); ref = OE.run_track(mix, 44100, w, **({"vad_fn": vad_fn} if vad_fn is not None else {})); to = time.time() - t0
    cuts_ref = ref.policy.cuts if ref.policy is not None else [0, len(mix)]
    flags_ref = ref.policy.flags if ref.policy is not None else None
    peak = float(np.max(np.abs(ref.vocal))) or 1.0
    stem = float(np.max(np.abs(r["vocal_track"] - ref.vocal))) / peak
    rms = float(np.max(np.abs(np.asarray(r["feature_cache"].rms_series) - np.asarray(ref.cache.rms_series)) / (np.abs(np.asarray(ref.cache.rms_series)) + 1e-7)))
    ok = (r["sample_boundaries"] == ref.sample_boundaries and r["cuts_samples"] == cuts_ref and
          (flags_ref is None or [bool(f) for f in r["segment_vocal_flags"]] == [bool(f) for f in flags_ref]) and stem < 1e-4 and
          (silero_seed is None or r["vad_segments"] == ref.vad_segments))
    # "exact" and "plateau-equivalent" are counted separately (tests/guard_plateau.py: a boundary may differ from the oracle's only
    # where the oracle's own dB series are bit-equal at the two indices - the epsilon plateau of digital silence - with the stem
    # around it within 1e-5 of the peak; never a sample tolerance)
    verdict = "exact" if ok else "MISMATCH"
    if not ok and stem < 1e-4 and len(r["sample_boundaries"]) == len(ref.sample_boundaries) and (silero_seed is None or r["vad_segments"] == ref.vad_segments):
        ctx = boundary_context(ref.vocal, mix, ref.sample_boundaries, 44100)
        ex, equiv, fails = classify_boundaries(r["sample_boundaries"], ref.sample_boundaries, ctx, r["vocal_track"], stem_atol=1e-5 * peak)
        if not fails and equiv and map_cuts(r["cuts_samples"], equiv) == list(cuts_ref):
            verdict = f"plateau-equivalent {equiv}"
    if silero_seed is not None:
        n_silero += 1; n_silero_counted += len(ref.vad_segments) >= 3
    n_exact += verdict == "exact"; n_equiv += verdict.startswith("plateau"); n_fail += verdict == "MISMATCH"
    ok_all &= verdict != "MISMATCH"
    print(f"{gen} {dur:g}s song_seed={sseed} weights_seed={wseed}" + (f" silero_seed={silero_seed} vad_segments={len(ref.vad_segments)}" if silero_seed is not None else "") + f": guard boundaries {len(ref.sample_boundaries)} "
          f"manifest cuts {len(cuts_ref)} pauses {len(ref.pauses)} | {verdict} | stem_err={stem:.2e} rms_series_rel={rms:.2e} | gpu {tg:.2f}s oracle {to:.0f}s", flush=True)
    bt_g = np.asarray(r["feature_cache"].beat_times, dtype=np.float64); bt_o = np.asarray(ref.cache.beat_times, dtype=np.float64)
    beats_equal = bt_g.shape == bt_o.shape and bool(np.array_equal(bt_g, bt_o))
    if not beats_equal:        # the beat grid feeds the layout refiner's beat snap (a manifest cut can move by up to beat_snap_ms)
        print(f"  beat_times differ: gpu {bt_g.size} beats, oracle {bt_o.size}; bpm gpu {getattr(getattr(r['feature_cache'], 'bpm_features', None), 'main_bpm', None)} oracle {getattr(getattr(ref.cache, 'bpm_features', None), 'main_bpm', None)}; "
              f"first difference at {next((i for i in range(min(bt_g.size, bt_o.size)) if bt_g[i] != bt_o[i]), min(bt_g.size, bt_o.size))}: gpu {bt_g[:6]}... oracle {bt_o[:6]}...")
    if not ok:
        print("  gpu   :", r["sample_boundaries"], r["cuts_samples"]); print("  oracle:", ref.sample_boundaries, cuts_ref)
        # how close was each moved decision?  The guard takes argmin of the 80 ms moving-RMS dB series (refine.py:184-214): compare
        # the ORACLE's own series at the two indices.  A difference of ~1e-12 dB means the reference's decision was a tie on
        # numerical dust (digital silence: the mean square is far below the 1e-12 epsilon inside the logarithm).
        if len(r["sample_boundaries"]) == len(ref.sample_boundaries):
            for g_i, o_i in zip(r["sample_boundaries"], ref.sample_boundaries):
                if g_i != o_i:
                    lo = max(0, min(g_i, o_i) - 30000); hi = min(len(mix), max(g_i, o_i) + 30000)
                    dv = OR.moving_meansq_db(ref.vocal[lo:hi], 3528); dm = OR.moving_meansq_db(mix[lo:hi], 3528)
                    print(f"    boundary {o_i} (oracle) vs {g_i} (gpu): oracle vocal dB differs by {abs(dv[g_i - lo] - dv[o_i - lo]):.3e}, "
                          f"mix dB by {abs(dm[g_i - lo] - dm[o_i - lo]):.3e}; vocal level there {dv[o_i - lo]:.6f} dB, |vocal| {abs(float(ref.vocal[o_i])):.2e}, |mix| {abs(float(mix[o_i])):.2e}")
    del backend, sp
print(f"tracks: {n_exact} exact, {n_equiv} plateau-equivalent, {n_fail} mismatched; Silero-mode tracks: {n_silero}, of which {n_silero_counted} with >= 3 VAD segments (only those count as Silero-mode parity)")
print("ALL EXACT" if (ok_all and not n_equiv) else ("NO MISMATCH" if ok_all else "MISMATCH"))
