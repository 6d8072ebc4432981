#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the reference's own Python.

Runs ONLY in the build container (needs /root/reference; the GPU box never sees it).
What executes here is the reference's *control logic* (imported from /root/reference/src):

  audio_cut.utils.gpu_pipeline.chunk_schedule          (direct import)
  audio_cut.cutting.refine                              (loaded by file path; numpy only)
  audio_cut.config.derive.resolve_threshold/min_pause   (direct import)
  audio_cut.detectors.silero_chunk_vad.SileroChunkVAD   (direct import, injected inference_fn)
  audio_cut.analysis.features_cache.ChunkFeatureBuilder       \\
  vocal_smart_splitter.core.pure_vocal_pause_detector          > over the restated librosa ops
  vocal_smart_splitter.core.vocal_separator (presence markers) |  (oracle.librosa_ops registered
  vocal_smart_splitter.core.seamless_splitter helpers         /   under the name `librosa`)

librosa / soundfile / pydub / silero_vad / demucs are not installed; the float
feature ops are therefore the oracle's restatement and every fixture produced through them is
conditional on it ("parity unpinned" for those ops).  Constructors that would download models
(`VocalPauseDetectorV2.__init__`, `EnhancedVocalSeparator.__init__`) are bypassed with
`object.__new__` + attribute setup.

Each fixture stores seeds/parameters and EXPECTED OUTPUTS only (data, no reference source).
The script also asserts oracle == reference on every case before writing, so a fixture is only
ever written from a run in which the oracle was pinned.

numpy/scipy versions are recorded in every fixture (the reference pins numpy<2.0; this container
has 2.2 — see oracle/refine.py LEGACY_PROMOTION).
"""
from __future__ import annotations

import importlib
import importlib.util
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import scipy

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REF))
sys.path.insert(0, str(REF / "src"))

from oracle import librosa_ops  # noqa: E402

librosa_ops.install_as_librosa()
for _name in ("soundfile", "pydub"):
    if _name not in sys.modules:
        sys.modules[_name] = types.ModuleType(_name)
sys.modules["pydub"].AudioSegment = object  # type: ignore[attr-defined]

from oracle import chunking as OC, detector as OD, features as OF, refine as OR, vad as OV  # noqa: E402
from audio_cut_amd.testing import signals  # noqa: E402

VERSIONS = {"numpy": np.__version__, "scipy": scipy.__version__}
SR = 44100


def _save(name: str, **arrays) -> None:
    path = HERE / name
    np.savez_compressed(path, versions=json.dumps(VERSIONS), **arrays)
    print(f"wrote {path.name} ({path.stat().st_size} bytes)")


def _ref_refine():
    spec = importlib.util.spec_from_file_location("ref_refine", REF / "src/audio_cut/cutting/refine.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_refine"] = mod
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------------------
def golden_chunk_schedule() -> None:
    from audio_cut.utils import gpu_pipeline as gp
    cases = [3.0, 10.0, 10.0001, 17.5, 60.0, 239.99, 240.0, 1800.0]
    rows = []
    for total in cases:
        ref = gp.chunk_schedule(total)
        ora = OC.chunk_plan(total)
        assert len(ref) == len(ora)
        for r, o in zip(ref, ora):
            assert (r.index, r.start_s, r.end_s, r.halo_left_s, r.halo_right_s) == (o.index, o.start_s, o.end_s, o.halo_left_s, o.halo_right_s)
            rows.append([total, r.index, r.start_s, r.end_s, r.halo_left_s, r.halo_right_s])
    alt = gp.chunk_schedule(33.0, chunk_s=8.0, overlap_s=3.0, halo_s=1.0)
    ora = OC.chunk_plan(33.0, 8.0, 3.0, 1.0)
    assert [(r.start_s, r.end_s) for r in alt] == [(o.start_s, o.end_s) for o in ora]
    _save("chunk_schedule.npz", rows=np.array(rows, dtype=np.float64),
          alt=np.array([[r.start_s, r.end_s, r.halo_left_s, r.halo_right_s] for r in alt]))


def _refine_case(seed: int, n_s: float, holes: bool):
    rng = np.random.default_rng(seed)
    n = int(SR * n_s)
    t = np.arange(n) / SR
    env = np.clip(np.sin(2 * np.pi * 0.13 * (seed + 1) * t), 0, None) ** 2
    mix = (rng.standard_normal(n) * 0.1 * env + 0.3 * env * np.sin(np.arange(n) * 0.05)).astype(np.float32)
    voc = (0.7 * mix + rng.standard_normal(n).astype(np.float32) * 0.01 * env).astype(np.float32)
    if holes:
        a = int(n * 0.3); b = a + SR
        mix[a:b] = 0; voc[a:b] = 0
    k = 24
    pts = np.stack([rng.uniform(0, n_s, k), rng.uniform(0, 1, k)], axis=1)
    return mix, voc, pts


REFINE_KW = dict(min_gap_s=1.2, max_keep=200, guard_db=1.5, search_right_ms=450.0, guard_win_ms=80.0)


def golden_refine() -> None:
    R = _ref_refine()
    out = {}
    for case, (seed, n_s, holes, floor_db) in enumerate([(0, 12.0, False, -35.0), (1, 12.0, True, -30.0),
                                                         (2, 20.0, True, -40.0), (3, 8.0, False, -20.0)]):
        mix, voc, pts = _refine_case(seed, n_s, holes)
        rr = R.finalize_cut_points(R.CutContext(sr=SR, mix_wave=mix, vocal_wave=voc),
                                   [R.CutPoint(t=float(t), score=float(s)) for t, s in pts], floor_db=floor_db, **REFINE_KW)
        OR.LEGACY_PROMOTION = False          # what the reference computes under this container's numpy 2.x
        oo = OR.finalize_cut_points(SR, mix, voc, [OR.Cut(float(t), float(s)) for t, s in pts], floor_db=floor_db, **REFINE_KW)
        assert rr.sample_boundaries == oo.sample_boundaries, case
        assert [a.final_time for a in rr.adjustments] == [a.final_time for a in oo.adjustments], case
        OR.LEGACY_PROMOTION = True           # the pinned-numpy (<2.0) semantics: the product's parity target
        ol = OR.finalize_cut_points(SR, mix, voc, [OR.Cut(float(t), float(s)) for t, s in pts], floor_db=floor_db, **REFINE_KW)
        out[f"c{case}_params"] = np.array([seed, n_s, float(holes), floor_db])
        out[f"c{case}_boundaries_live"] = np.array(rr.sample_boundaries, dtype=np.int64)
        out[f"c{case}_final_times_live"] = np.array([a.final_time for a in rr.adjustments])
        out[f"c{case}_boundaries_legacy"] = np.array(ol.sample_boundaries, dtype=np.int64)
        out[f"c{case}_final_times_legacy"] = np.array([a.final_time for a in ol.adjustments])
        # lookup arrays (decimated) for the device kernels' parity tests
        lk = R._prepare_quiet_lookup(voc, SR, 80.0, floor_db)
        out[f"c{case}_db_dec"] = lk.rms_db[::997].copy()
        out[f"c{case}_nq_dec"] = lk.next_quiet[::997].copy()
    # the reference's own known answer (tests/unit/test_cutting_consistency.py:20-46)
    r = R.finalize_cut_points(R.CutContext(sr=10, mix_wave=np.zeros(120, np.float32)),
                              [R.CutPoint(t=4.0, score=0.9), R.CutPoint(t=8.0, score=0.9)],
                              min_gap_s=1.0, enable_mix_guard=False, enable_vocal_guard=False, zero_cross_win_ms=0.0)
    assert r.sample_boundaries == [0, 40, 80, 120]
    _save("refine.npz", **out)


def golden_derive() -> None:
    from audio_cut.config import derive
    from vocal_smart_splitter.utils.config_manager import get_config
    adapt = get_config("pure_vocal_detection.relative_threshold_adaptation", {})
    rows = []
    for bpm in (None, 60.0, 89.9, 90.0, 120.0, 140.0, 141.0, 200.0):
        for mdd in (None, 0.0, 0.37, 1.0):
            r = derive.resolve_threshold(0.26, adapt, derive.AdaptStats(bpm=bpm, global_mdd=mdd))
            o = OD.resolve_threshold(0.26, adapt, bpm, mdd)
            assert (r.peak_ratio, r.rms_ratio) == (o.peak_ratio, o.rms_ratio)
            mp = derive.resolve_min_pause(0.5, 1.0, derive.AdaptStats(bpm=bpm, global_mdd=mdd))
            assert mp == OD.resolve_min_pause(0.5, 1.0, bpm)
            rows.append([-1.0 if bpm is None else bpm, -1.0 if mdd is None else mdd, r.peak_ratio, r.rms_ratio, mp])
    _save("derive.npz", rows=np.array(rows))


def _fake_vad(chunk: np.ndarray):
    """Deterministic injected inference_fn: speech wherever |x| block-mean exceeds a level."""
    blk = 2205
    n = len(chunk) // blk
    act = np.abs(chunk[: n * blk]).reshape(n, blk).mean(axis=1) > 0.02
    out = []
    for a, b, v in OD._runs(act):
        if v:
            out.append({"start": a * blk, "end": b * blk})
    return out


def golden_chunk_vad() -> None:
    from audio_cut.detectors.silero_chunk_vad import SileroChunkVAD
    from audio_cut.utils import gpu_pipeline as gp
    voc = signals.vocal_like(40.0, seed=11)
    ref = SileroChunkVAD(sample_rate=SR, merge_gap_ms=120.0, focus_pad_s=0.2, inference_fn=_fake_vad)
    ora = OV.ChunkVadOracle(SR, 120.0, 0.2, _fake_vad)
    for rp, op in zip(gp.chunk_schedule(40.0), OC.chunk_plan(40.0)):
        a = int(round(rp.start_s * SR)); b = int(round(rp.end_s * SR))
        ref.process_chunk(rp, voc[a:b], SR)
        ora.process_chunk(op, voc[a:b], SR)
    rs = ref.finalize(); os_ = ora.finalize()
    assert rs == os_ and len(rs) > 2
    assert ref.to_focus_windows() == ora.to_focus_windows()
    _save("chunk_vad.npz", segments=np.array([[s["start"], s["end"]] for s in rs]),
          focus=np.array(ref.to_focus_windows()))


def _make_ref_detector():
    from vocal_smart_splitter.core import pure_vocal_pause_detector as pv
    det = object.__new__(pv.PureVocalPauseDetector)
    det.sample_rate = SR
    det.min_pause_duration = 0.5
    det.hop_length = int(SR * 0.01)
    det.frame_length = int(SR * 0.025)
    det.n_fft = 2048
    det._last_feature_cache = None
    det._last_focus_windows = []
    det._cut_point_calculator = types.SimpleNamespace(_detect_speech_timestamps=lambda audio: [])
    return pv, det


def _pauses_array(pauses):
    return np.array([[p.start_time, p.end_time, p.confidence, p.cut_point] for p in pauses], dtype=np.float64).reshape(-1, 4)


def golden_features_and_detector() -> None:
    from audio_cut.analysis import features_cache as fc
    from audio_cut.utils import gpu_pipeline as gp
    from vocal_smart_splitter.core import seamless_splitter as ss
    from vocal_smart_splitter.core import vocal_separator as vs
    pv, det = _make_ref_detector()
    out = {}

    # --- (1) chunked feature cache on a 27 s song (3 chunks) ---
    mix = signals.c2_song(27.0, seed=21)
    rb = fc.ChunkFeatureBuilder(sr=SR)
    ob = OF.ChunkFeatureOracle(SR)
    for rp, op in zip(gp.chunk_schedule(27.0), OC.chunk_plan(27.0)):
        a = int(round(rp.start_s * SR)); b = min(len(mix), int(round(rp.end_s * SR)))
        rb.add_chunk(rp, mix[a:b], SR)
        ob.add_chunk(op, mix[a:b], SR)
    rc = rb.finalize(mix)
    oc = ob.finalize(mix)
    for name in ("rms_series", "spectral_flatness", "onset_envelope", "mdd_series", "beat_times", "tempo_curve"):
        assert np.array_equal(np.asarray(getattr(rc, name)), np.asarray(getattr(oc, name))), name
    assert np.array_equal(rc.onset_frames, oc.onset_frames)
    assert float(rc.bpm_features.main_bpm) == float(oc.bpm_features.main_bpm)
    assert rc.global_mdd == oc.global_mdd
    out["cache_rms"] = rc.rms_series; out["cache_flat"] = rc.spectral_flatness
    out["cache_onset"] = rc.onset_envelope; out["cache_onset_frames"] = rc.onset_frames
    out["cache_mdd"] = rc.mdd_series; out["cache_beat_times"] = rc.beat_times
    out["cache_tempo_curve"] = np.asarray(rc.tempo_curve)
    out["cache_scalars"] = np.array([float(rc.bpm_features.main_bpm), rc.bpm_features.beat_strength,
                                     rc.bpm_features.tempo_variance, rc.global_mdd, rc.rms_max, rc.onset_max])

    # --- (2) detector, C1-style: no cache, no VAD (scripts/e2e_profile.py:50) ---
    x = signals.c1_sine_silence(30.0, seed=1)
    rp_ = det.detect_pure_vocal_pauses(x, enable_mdd_enhancement=True, original_audio=x)
    op_ = OD.detect_pure_vocal_pauses(x, SR, enable_mdd_enhancement=True, original_audio=x)
    assert np.array_equal(_pauses_array(rp_), _pauses_array(op_)) and len(rp_) > 3
    out["c1_pauses"] = _pauses_array(rp_)

    # --- (3) detector with cache + VAD focus windows on a vocal-like stem ---
    voc = signals.vocal_like(27.0, seed=21)
    vad_segments = [{"start": 1.0, "end": 6.2, "duration": 5.2}, {"start": 7.1, "end": 13.0, "duration": 5.9},
                    {"start": 13.6, "end": 20.5, "duration": 6.9}, {"start": 21.4, "end": 26.5, "duration": 5.1}]
    rp2 = det.detect_pure_vocal_pauses(voc, enable_mdd_enhancement=True, original_audio=mix, feature_cache=rc, vad_segments=vad_segments)
    op2 = OD.detect_pure_vocal_pauses(voc, SR, enable_mdd_enhancement=True, original_audio=mix, feature_cache=oc, vad_segments=vad_segments)
    assert np.array_equal(_pauses_array(rp2), _pauses_array(op2)), (_pauses_array(rp2), _pauses_array(op2))
    out["c2_pauses_vad"] = _pauses_array(rp2)
    rp3 = det.detect_pure_vocal_pauses(voc, enable_mdd_enhancement=True, original_audio=mix, feature_cache=rc, vad_segments=[])
    op3 = OD.detect_pure_vocal_pauses(voc, SR, enable_mdd_enhancement=True, original_audio=mix, feature_cache=oc, vad_segments=[])
    assert np.array_equal(_pauses_array(rp3), _pauses_array(op3)) and len(rp3) > 1
    out["c2_pauses_novad"] = _pauses_array(rp3)
    assert list(det._focus_windows_from_vad_segments(vad_segments, pad_s=0.2, min_width_s=0.0)) == OD.focus_windows_from_vad(vad_segments, 0.2, 0.0)

    # --- (4) presence markers, no-vocal runs, finalize (seamless_splitter.py:1706-1879) ---
    sep = object.__new__(vs.VocalSeparator)
    sep.sample_rate = SR
    rm = sep._compute_vocal_presence_markers(voc)
    om = OD.vocal_presence_markers(voc, SR)
    assert rm["vocal_presence_cut_points_sec"] == om["vocal_presence_cut_points_sec"]
    assert rm["vocal_presence_segments"] == om["vocal_presence_segments"]
    out["marker_times"] = np.array(rm["vocal_presence_cut_points_sec"])
    quiet = voc.copy(); quiet[int(8 * SR): int(17 * SR)] *= 1e-3
    fake_self = types.SimpleNamespace(sample_rate=SR, _set_guard_adjustments=lambda adj: None)
    rr_ = ss.SeamlessSplitter._find_no_vocal_runs(fake_self, quiet, 6.0)
    or_ = OD.no_vocal_runs(quiet, SR, 6.0)
    assert rr_ == or_ and len(rr_) >= 1, (rr_, or_)
    out["no_vocal_runs"] = np.array(rr_)
    cands = [(p.cut_point, p.confidence) for p in rp3] + [(float(t), 1.0) for t in rm["vocal_presence_cut_points_sec"] if 0 < t < 27.0]
    rf = ss.SeamlessSplitter._finalize_and_filter_cuts_v2(fake_self, cands, mix, pure_vocal_audio=voc)
    from oracle import e2e as OE
    OR.LEGACY_PROMOTION = False
    of = OE.finalize_and_filter_cuts(cands, mix, voc, SR)
    assert rf.sample_boundaries == of.sample_boundaries, (rf.sample_boundaries, of.sample_boundaries)
    OR.LEGACY_PROMOTION = True
    ol = OE.finalize_and_filter_cuts(cands, mix, voc, SR)
    out["final_boundaries_live"] = np.array(rf.sample_boundaries, dtype=np.int64)
    out["final_boundaries_legacy"] = np.array(ol.sample_boundaries, dtype=np.int64)
    _save("features_detector.npz", **out)


from audio_cut_amd.testing.vpbd_inputs import FixedPauses as _FixedPauses, vpbd_case  # noqa: E402


def golden_dormant_branch() -> None:
    """the multi-feature branch (`enable_relative_energy_mode: false`) of the reference detector run for
    real over the restated librosa ops (pyin, lpc, centroid, ZCR ...), against oracle.detector.detect_multifeature_pauses."""
    from vocal_smart_splitter.utils import config_manager as cm
    from oracle import config as OCfg
    pv, det = _make_ref_detector()
    det.breath_duration_range = [0.1, 0.3]
    det.f0_weight, det.formant_weight, det.spectral_weight, det.duration_weight = 0.3, 0.25, 0.25, 0.2
    det.energy_threshold_db = -40; det.f0_drop_threshold = 0.7
    det.breath_confidence_threshold = 0.3; det.pause_confidence_threshold = 0.7
    key = "pure_vocal_detection.enable_relative_energy_mode"
    cm.set_runtime_config({key: False}); OCfg.set_runtime_config({key: False})
    try:
        saved = {}
        for tag, x in (("voice", signals.voice_with_rests(14.0, seed=3)), ("bursts", signals.c1_sine_silence(12.0, seed=2))):
            rf = det._extract_vocal_features(x)
            of = OD.extract_vocal_features(x, SR)
            for name in ("f0_contour", "f0_confidence", "spectral_centroid", "harmonic_ratio", "zero_crossing_rate", "rms_energy"):
                assert np.array_equal(np.asarray(getattr(rf, name)), np.asarray(getattr(of, name)), equal_nan=True), name
            for a_, b_ in zip(rf.formant_energies, of.formant_energies):
                assert np.array_equal(a_, b_)
            assert det._detect_candidate_pauses(rf) == OD.detect_candidate_pauses(of, SR)
            for breath in (False, True):
                rp = det.detect_pure_vocal_pauses(x, include_breath_candidates=breath)
                op = OD.detect_multifeature_pauses(x, SR, include_breath_candidates=breath)
                assert np.array_equal(_pauses_array(rp), _pauses_array(op)), (breath, _pauses_array(rp), _pauses_array(op))
                assert [p.pause_type for p in rp] == [p.pause_type for p in op]
                saved[f"{tag}_pauses_breath{int(breath)}"] = _pauses_array(rp)
            print(f"  dormant branch [{tag}]: {len(saved[tag + '_pauses_breath0'])} pauses, {len(saved[tag + '_pauses_breath1'])} with breaths")
            assert len(saved[f"{tag}_pauses_breath1"]) >= 1, tag
            saved.update({f"{tag}_f0": rf.f0_contour, f"{tag}_voiced_prob": rf.f0_confidence, f"{tag}_centroid": rf.spectral_centroid,
                          f"{tag}_harmonic_ratio": rf.harmonic_ratio, f"{tag}_zcr": rf.zero_crossing_rate, f"{tag}_rms": rf.rms_energy,
                          f"{tag}_formant0": np.asarray(rf.formant_energies[0]), f"{tag}_formant1": np.asarray(rf.formant_energies[1]),
                          f"{tag}_formant2": np.asarray(rf.formant_energies[2]),     # lengths differ: `:1003-1008` appends only to the tracks that got a peak
                          f"{tag}_candidates": np.asarray(det._detect_candidate_pauses(rf), dtype=np.int64).reshape(-1, 2)})
        _save("dormant_branch.npz", **saved)
    finally:
        cm.reset_runtime_config(); OCfg.reset_runtime_config()


from audio_cut_amd.testing.policy_inputs import policy_case as _policy_case, random_layout_case as _random_layout_case  # noqa: E402


def golden_boundary_policy() -> None:
    """classification, layout refiner, local-valley refinement, weak-tail merge and sample-level
    split of the reference (`seamless_splitter.py:521-669`, `segment_layout_refiner.py`) against oracle.layout."""
    from audio_cut.analysis.features_cache import TrackFeatureCache
    from audio_cut.cutting import segment_layout_refiner as rl
    from audio_cut.cutting.refine import CutPoint
    from vocal_smart_splitter.core import seamless_splitter as ss
    from vocal_smart_splitter.utils.config_manager import get_config as rget
    from oracle import layout as OL
    out = {}
    rng = np.random.default_rng(77)
    # (1) the layout refiner alone on random segmentations (durations 0.3-25 s, both kinds, suppressed points, beats)
    n_checked = 0
    for case in range(60):
        edges, kinds, rms, hop_s, beats, supp, cfg_kwargs, midpoint = _random_layout_case(rng, case)
        k = len(kinds); total = float(edges[-1]); frames = len(rms)
        cache = TrackFeatureCache(sr=SR, hop_length=int(SR * hop_s), hop_s=hop_s, duration_s=total, rms_series=rms,
                                  spectral_flatness=np.zeros(frames, np.float32), onset_envelope=np.zeros(frames, np.float32),
                                  onset_strength=np.zeros(frames, np.float32), onset_frames=np.zeros(0, np.int64), rms_max=float(rms.max()),
                                  onset_max=0.0, bpm_features=None, tempo_curve=None, beat_times=beats, global_mdd=0.5,
                                  mdd_series=np.zeros(frames, np.float32))
        segs = [rl.Segment(float(edges[i]), float(edges[i + 1]), kinds[i]) for i in range(k)]
        rres = rl.refine_layout(segs, [], config=rl.LayoutConfig(**cfg_kwargs), sample_rate=SR,
                                suppressed_cut_points=[CutPoint(t=t, score=sc) for t, sc in supp], features=cache,
                                allow_midpoint_fallback=midpoint)
        osegs, osupp, _ = OL.refine_layout([[float(edges[i]), float(edges[i + 1]), kinds[i]] for i in range(k)], OL.LayoutConfig(**cfg_kwargs),
                                          supp, rms, hop_s, beats, midpoint_fallback=midpoint)
        assert [(s.start, s.end, s.kind) for s in rres.segments] == [(s[0], s[1], s[2]) for s in osegs], case
        assert [(float(p.t), float(p.score)) for p in rres.suppressed_points] == [(float(t), float(sc)) for t, sc in osupp], case
        n_checked += 1
        if case < 12:
            out[f"layout{case}_in"] = np.array([[edges[i], edges[i + 1], 1.0 if kinds[i] == "human" else 0.0] for i in range(k)])
            out[f"layout{case}_rms"] = rms; out[f"layout{case}_supp"] = np.array(supp)
            out[f"layout{case}_cfg"] = np.array([cfg_kwargs[x] for x in ("micro_merge_s", "soft_min_s", "soft_max_s", "min_gap_s", "beat_snap_ms")] + [float(midpoint)])
            out[f"layout{case}_out"] = np.array([[s.start, s.end, 1.0 if s.kind == "human" else 0.0] for s in rres.segments])
    # (2) the sample-domain steps on a synthetic stem, composed in the reference's order (`seamless_splitter.py:521-669`)
    fake = types.SimpleNamespace(sample_rate=SR)
    fake._classify_segments_vocal_presence = lambda *a, **k: ss.SeamlessSplitter._classify_segments_vocal_presence(fake, *a, **k)
    for seed in (11, 12):
        voc, cuts, rms, hop_s, beats, supp = _policy_case(seed)
        n = len(voc)
        cache = TrackFeatureCache(sr=SR, hop_length=int(SR * hop_s), hop_s=hop_s, duration_s=n / SR, rms_series=rms,
                                  spectral_flatness=np.zeros_like(rms), onset_envelope=np.zeros_like(rms), onset_strength=np.zeros_like(rms),
                                  onset_frames=np.zeros(0, np.int64), rms_max=float(rms.max()), onset_max=0.0, bpm_features=None,
                                  tempo_curve=None, beat_times=beats, global_mdd=0.5, mdd_series=np.zeros_like(rms))
        flags = fake._classify_segments_vocal_presence(voc, cuts)
        oflags, _ = OL.classify_segments(voc, cuts, SR)
        assert flags == oflags
        # layout config exactly as `:533-556` derives it
        raw = dict(rget("segment_layout", {}) or {})
        micro = rget("quality_control.segment_min_mix_piece", None)
        if micro is not None:
            raw.setdefault("micro_merge_s", float(micro)); raw.setdefault("enable", bool(float(micro) > 0.0))
        smax = rget("quality_control.segment_max_duration", None)
        if smax is not None:
            raw.setdefault("soft_max_s", float(smax))
        raw.setdefault("min_gap_s", float(rget("quality_control.min_split_gap", 1.0)))
        raw.setdefault("beat_snap_ms", float(rget("segment_layout.beat_snap_ms", 0.0) or 0.0))
        lcfg = rl.derive_layout_config(raw, cache, sample_rate=SR)
        ocfg = OL.layout_config_from_settings()
        assert (lcfg.enable, lcfg.micro_merge_s, lcfg.soft_min_s, lcfg.soft_max_s, lcfg.min_gap_s, lcfg.beat_snap_ms) == \
               (ocfg.enable, ocfg.micro_merge_s, ocfg.soft_min_s, ocfg.soft_max_s, ocfg.min_gap_s, ocfg.beat_snap_ms)
        bsec = [c / float(SR) for c in cuts]
        lres = rl.refine_layout([rl.Segment(bsec[i], bsec[i + 1], "human" if flags[i] else "music") for i in range(len(bsec) - 1)], [],
                                config=lcfg, sample_rate=SR, suppressed_cut_points=[CutPoint(t=t, score=sc) for t, sc in supp], features=cache)
        lb = [lres.segments[0].start] + [sg.end for sg in lres.segments]
        upd = [max(0, min(int(round(t * SR)), n)) for t in lb]
        upd[0] = 0; upd[-1] = n
        upd = sorted(set(upd))
        applied = upd != cuts
        cur = upd
        flags = fake._classify_segments_vocal_presence(voc, cur)
        lcl = rget("quality_control.local_boundary_refine", {}) or {}
        if lcl.get("enable") and len(cur) >= 2:
            ref = ss.SeamlessSplitter._refine_boundaries_local_valley(fake, cur, voc, lcl, min_gap_s=float(rget("quality_control.min_split_gap", 1.0)))
            oref = OL.refine_local_valley(cur, voc, SR, lcl, float(rget("quality_control.min_split_gap", 1.0)))
            assert list(ref) == list(oref)
            if list(ref) != cur:
                cur = list(ref); applied = True
                flags = fake._classify_segments_vocal_presence(voc, cur)
        c2, f2, _ = ss.SeamlessSplitter._merge_short_weak_human_tails_into_following_music(
            fake, cur, flags, [], voc, min_duration_s=float(lcfg.soft_min_s or 0.0), layout_applied=applied)
        oc2, of2 = OL.merge_weak_human_tails(cur, flags, voc, SR, float(lcfg.soft_min_s or 0.0), applied)
        assert (list(c2), list(f2)) == (oc2, of2)
        if list(c2) != cur:
            cur, flags, applied = list(c2), list(f2), True
        fake2 = object.__new__(ss.SeamlessSplitter); fake2.sample_rate = SR
        pieces, mflags, _ = fake2._split_at_sample_level(np.arange(n, dtype=np.int64), cur, segment_flags=flags, debug_entries=None)
        ranges = [(int(p[0]), int(p[-1]) + 1) for p in pieces]
        oranges, omflags = OL.split_at_sample_level(n, cur, flags, SR)
        assert ranges == [tuple(r) for r in oranges] and list(mflags) == list(omflags)
        whole = OL.apply_boundary_policy(cuts, voc, n, SR, suppressed=supp, rms_series=rms, hop_s=hop_s, beat_times=beats)
        assert whole.cuts == cur and whole.flags == list(mflags) and whole.pieces == ranges and whole.layout_applied == applied, seed
        out[f"policy{seed}_cuts_in"] = np.asarray(cuts, dtype=np.int64)
        out[f"policy{seed}_cuts_out"] = np.asarray(cur, dtype=np.int64)
        out[f"policy{seed}_flags"] = np.asarray(mflags, dtype=np.int8)
        out[f"policy{seed}_pieces"] = np.asarray(ranges, dtype=np.int64)
        out[f"policy{seed}_applied"] = np.asarray([int(applied)])
        out[f"policy{seed}_supp"] = np.asarray(supp)
        print(f"  boundary policy [seed {seed}]: {len(cuts)} cuts -> {len(cur)}; layout_applied={applied}")
    print(f"  layout refiner: {n_checked} random segmentations identical")
    _save("boundary_policy.npz", **out)


def golden_vpbd() -> None:
    import tempfile
    from vocal_smart_splitter.core.vocal_phrase_boundary_detector import VocalPhraseBoundaryDetector as RefVPBD
    from audio_cut_amd.core.vocal_phrase_boundary_detector import VocalPhraseBoundaryDetector as OurVPBD
    out = {}
    for case, seed in enumerate((31, 32, 33)):
        cache, pauses, vocal = vpbd_case(seed)
        with tempfile.TemporaryDirectory() as tmp:
            r = RefVPBD(SR).detect(mode="vpbd_acoustic", vocal_track=vocal, original_audio=vocal, pure_vocal_detector=_FixedPauses(pauses),
                                   feature_cache=cache, vad_segments=None, input_path="x.wav", output_dir=tmp)
            o = OurVPBD(SR).detect(mode="vpbd_acoustic", vocal_track=vocal, original_audio=vocal, pure_vocal_detector=_FixedPauses(pauses),
                                   feature_cache=cache, vad_segments=None, input_path="x.wav", output_dir=tmp)
        rs = [(c.t, c.score, c.source.value) for c in r.selected_candidates]
        os_ = [(c.t, c.score, c.source.value) for c in o.selected_candidates]
        assert rs == os_ and len(rs) >= 4, (rs, os_)
        assert r.boundary_detection["candidate_counts"] == o.boundary_detection["candidate_counts"]
        rsup = [(c.t, c.score) for c in r.planner_result.suppressed_candidates]
        osup = [(c.t, c.score) for c in o.planner_result.suppressed_candidates]
        assert rsup == osup
        assert [c.features for c in r.selected_candidates] == [c.features for c in o.selected_candidates]
        out[f"c{case}_selected"] = np.array([[t, s] for t, s, _ in rs])
        out[f"c{case}_suppressed"] = np.array(rsup).reshape(-1, 2)
        out[f"c{case}_counts"] = np.array([r.boundary_detection["candidate_counts"][k] for k in ("acoustic", "beat", "merged", "total", "selected", "suppressed")])
        out[f"c{case}_features"] = np.array([[c.features[k] for k in sorted(c.features)] for c in r.selected_candidates])
    _save("vpbd.npz", **out)


def golden_config() -> None:
    from vocal_smart_splitter.utils.config_manager import get_config
    from oracle import config as ocfg
    flat = {}

    def walk(prefix, node):
        if isinstance(node, dict):
            for k, v in node.items():
                walk(f"{prefix}.{k}" if prefix else k, v)
        else:
            flat[prefix] = node

    walk("", ocfg.snapshot())
    mismatches = {k: (v, get_config(k, "<absent>")) for k, v in flat.items() if get_config(k, "<absent>") != v}
    assert not mismatches, mismatches
    (HERE / "config_effective.json").write_text(json.dumps({"versions": VERSIONS, "values": flat}, indent=1, sort_keys=True))
    print("wrote config_effective.json", len(flat), "keys")


if __name__ == "__main__":
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    golden_config()
    golden_chunk_schedule()
    golden_derive()
    golden_refine()
    golden_chunk_vad()
    golden_features_and_detector()
    golden_dormant_branch()
    golden_boundary_policy()
    golden_vpbd()
    print("all goldens generated; oracle pinned against the reference's control logic")
