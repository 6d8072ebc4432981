"""Writes tests/golden/NAME.npz: the CPU oracle's result on one synthetic track, with everything a GPU test needs to judge
a guard boundary that the reference itself decides on numerical dust.

    python tests/golden/make_track_fixture.py --seconds 60 --seed 301 --weights 61 --generator c1_sine_silence --silero 1 \
        --name c1_60s_seed301_w61_silero1_oracle
    python tests/golden/make_track_fixture.py --seconds 240 --seed 2 --silero 2 --mode vpbd_acoustic --name c4_full_oracle

Beyond what make_c2_full.py stores, per guard boundary b (reference `src/audio_cut/cutting/refine.py:172-174`: the guard reads
`20*log10(sqrt(moving mean square) + 1e-12)` over an 80 ms window):
  * `on_plateau[b]`: the oracle's vocal AND mix dB series are both the epsilon plateau 20*log10(1e-6 + 1e-12) ... i.e. the
    mean square in the window is below what float64 resolves against the 1e-12 inside the logarithm ... at the boundary;
    there `np.argmin` returns the first sample of a run of bit-equal values whose start is decided by float32 inverse-FFT
    rounding noise.
  * `db_vocal[b]`, `db_mix[b]`: the oracle's two dB series over [b - HALF, b + HALF], and `stem_window[b]`: its vocal stem there,
    so that a test can assert "the GPU's index reads a bit-equal oracle dB value" without re-running the oracle.

`--mode vpbd_acoustic` (BASELINE configs[3] / SURVEY 8d C4) pushes the oracle's stems / cache / pauses / VAD segments through the
REFERENCE's own VocalPhraseBoundaryDetector (imported from /root/reference over the restated librosa ops, as make_golden.py
does), asserts that the product's host-side VPBD agrees with it, and stores the selected candidate times, the pool counts, the
guard boundaries and the manifest cuts.  Needs /root/reference; runs in the build container only."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))          # silero_synth / guard_plateau as top-level modules (the reference has a `tests` package too)

HALF = 4096


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, required=True)
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--weights", type=int, default=0)
    ap.add_argument("--generator", default="c2_song")
    ap.add_argument("--silero", type=int, default=None, help="seed of the synthetic Silero weights (tests/silero_synth.py); default: energy gate")
    ap.add_argument("--silero-calib", default="bursts")
    ap.add_argument("--mode", default="v2.2_mdd", choices=["v2.2_mdd", "vpbd_acoustic"])
    ap.add_argument("--name", required=True)
    ap.add_argument("--light", action="store_true", help="keep the on_plateau flags but not the per-boundary dB / stem windows (100 KB instead of MBs)")
    a = ap.parse_args()

    if a.mode == "vpbd_acoustic":           # the reference's VPBD host logic, over the restated librosa ops (make_golden.py)
        import types
        REF = Path("/root/reference")
        sys.path.insert(0, str(REF)); sys.path.insert(0, str(REF / "src"))
        from oracle import librosa_ops
        librosa_ops.install_as_librosa()
        for nm in ("soundfile", "pydub"):
            sys.modules.setdefault(nm, types.ModuleType(nm))
        sys.modules["pydub"].AudioSegment = object

    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    from audio_cut_amd.testing import signals
    from oracle import detector as OD, e2e as OE, refine as OR, silero as OS
    OR.LEGACY_PROMOTION = True
    sr = 44100
    mix = getattr(signals, a.generator)(a.seconds, seed=a.seed)
    mix = np.mean(mix, axis=0).astype(np.float32) if mix.ndim == 2 else mix
    w = synth_weights(TfcTdfSpec(), seed=a.weights)
    kw = {}
    affine = (np.nan, np.nan)
    if a.silero is not None:
        from silero_synth import calibration_affine, synth_silero_weights
        affine = calibration_affine(a.silero, a.silero_calib)       # stored: the GPU test and bench.py rebuild the weights from it without the oracle
        kw["vad_fn"] = OS.silero_vad_fn(sr, synth_silero_weights(a.silero, calib=a.silero_calib, affine=affine))
    t0 = time.time()
    ref = OE.run_track(mix, sr, w, **kw)
    print(f"oracle: {time.time() - t0:.1f} s, {len(ref.sample_boundaries)} boundaries, {len(ref.pauses)} pauses, "
          f"{len(ref.vad_segments)} VAD segments", flush=True)
    bounds, policy = ref.sample_boundaries, ref.policy
    extra = {}
    if a.mode == "vpbd_acoustic":
        import tempfile
        from vocal_smart_splitter.core.vocal_phrase_boundary_detector import VocalPhraseBoundaryDetector as RefVPBD
        from audio_cut_amd.core.vocal_phrase_boundary_detector import VocalPhraseBoundaryDetector as OurVPBD

        class OraclePauses:          # the detector the reference's VPBD calls (`vocal_phrase_boundary_detector.py:196-203`)
            def detect_pure_vocal_pauses(self, vocal, enable_mdd_enhancement=False, original_audio=None, feature_cache=None,
                                         vad_segments=None, include_breath_candidates=False, **_):
                return OD.detect_pure_vocal_pauses(vocal, sr, enable_mdd_enhancement=enable_mdd_enhancement, original_audio=original_audio,
                                                   feature_cache=feature_cache, vad_segments=vad_segments)

        with tempfile.TemporaryDirectory() as tmp:
            args = dict(mode="vpbd_acoustic", vocal_track=ref.vocal, original_audio=mix, pure_vocal_detector=OraclePauses(),
                        feature_cache=ref.cache, vad_segments=ref.vad_segments, input_path="x.wav", output_dir=tmp)
            r = RefVPBD(sr).detect(**args)
            o = OurVPBD(sr).detect(**args)
        sel = [(c.t, c.score) for c in r.selected_candidates]
        sup = [(c.t, c.score) for c in r.planner_result.suppressed_candidates]
        assert sel == [(c.t, c.score) for c in o.selected_candidates], "product VPBD host logic != reference"
        assert sup == [(c.t, c.score) for c in o.planner_result.suppressed_candidates]
        assert r.boundary_detection["candidate_counts"] == o.boundary_detection["candidate_counts"]
        pol: list = []
        bounds = OE.finalize_vpbd(mix, ref.vocal, sr, ref.cache, ref.markers, sel, sup, policy_out=pol)
        policy = pol[0] if pol else None
        counts = r.boundary_detection["candidate_counts"]
        extra = dict(vpbd_selected=np.asarray(sel, dtype=np.float64).reshape(-1, 2), vpbd_suppressed=np.asarray(sup, dtype=np.float64).reshape(-1, 2),
                     vpbd_counts=np.asarray([counts[k] for k in ("acoustic", "beat", "merged", "total", "selected", "suppressed")], dtype=np.int64))
        print(f"vpbd_acoustic: {len(sel)} selected / {len(sup)} suppressed candidates, counts {counts}", flush=True)
    print(f"{len(bounds)} guard boundaries, {len(policy.cuts) if policy is not None else 2} manifest cuts", flush=True)

    # per-boundary guard context (refine.py:172-174): the oracle's own dB series around every boundary
    from guard_plateau import boundary_context, plateau_db
    n = len(mix)
    plateau = plateau_db()
    ctx = boundary_context(ref.vocal, mix, bounds, sr, HALF)
    if a.light:
        ctx = {"on_plateau": ctx["on_plateau"]}
    print(f"boundaries on the epsilon plateau ({plateau:.6f} dB): {[int(b) for b, p in zip(bounds, ctx['on_plateau']) if p]}", flush=True)

    sec = sr
    nsec = len(mix) // sec
    cuts = policy.cuts if policy is not None else [0, n]
    np.savez_compressed(
        ROOT / "tests" / "golden" / f"{a.name}.npz", seconds=np.float64(a.seconds), seed=np.int64(a.seed), weight_seed=np.int64(a.weights),
        generator=np.asarray(a.generator), silero_seed=np.int64(-1 if a.silero is None else a.silero), silero_calib=np.asarray(a.silero_calib),
        silero_affine=np.asarray(affine, dtype=np.float64), mode=np.asarray(a.mode), numpy_version=np.asarray(np.__version__),
        sample_boundaries=np.asarray(bounds, dtype=np.int64), cuts=np.asarray(cuts, dtype=np.int64),
        flags=np.asarray(policy.flags if policy is not None else [], dtype=np.int8),
        pieces=np.asarray(policy.pieces if policy is not None else [], dtype=np.int64).reshape(-1, 2),
        pause_cut_points=np.asarray([p.cut_point for p in ref.pauses], dtype=np.float64),
        vocal_rms_per_second=np.sqrt(np.mean(ref.vocal[: nsec * sec].reshape(nsec, sec).astype(np.float64) ** 2, axis=1)),
        vocal_head=ref.vocal[: 4 * sec: 7].astype(np.float32), vocal_peak=np.float64(np.max(np.abs(ref.vocal))),
        vad_segments=np.asarray([[s["start"], s["end"]] for s in ref.vad_segments], dtype=np.float64).reshape(-1, 2),
        cache_rms=np.asarray(ref.cache.rms_series, dtype=np.float32), beat_times=np.asarray(ref.cache.beat_times, dtype=np.float64),
        plateau_db=plateau, **ctx, **extra)
    print(f"wrote tests/golden/{a.name}.npz")


if __name__ == "__main__":
    main()
