"""VPBD candidate pool / scorer / planner: host logic pinned by goldens written from the
reference's own VocalPhraseBoundaryDetector, plus the reference's unit-test known answers.  CPU only."""
import types

import numpy as np

from audio_cut_amd.analysis.boundary_features import LyricsTimeline
from audio_cut_amd.core.vocal_phrase_boundary_detector import VocalPhraseBoundaryDetector
from audio_cut_amd.cutting.beat_candidates import detect_chorus_regions, generate_beat_candidates
from audio_cut_amd.cutting.cut_candidate import CandidateSource, CutCandidate, adapt_legacy_acoustic_candidates
from audio_cut_amd.cutting.global_cut_planner import (GlobalCutPlanner, GlobalCutPlannerConfig, apply_guard_shift_metadata,
                                                      planner_result_to_cut_points)
from audio_cut_amd.testing.vpbd_inputs import FixedPauses, vpbd_case

SR = 44100


def test_vpbd_detect_matches_reference_goldens(golden_dir, tmp_path):
    g = np.load(golden_dir / "vpbd.npz")
    for case, seed in enumerate((31, 32, 33)):
        cache, pauses, vocal = vpbd_case(seed)
        res = VocalPhraseBoundaryDetector(SR).detect(mode="vpbd_acoustic", vocal_track=vocal, original_audio=vocal,
                                                     pure_vocal_detector=FixedPauses(pauses), feature_cache=cache, vad_segments=None,
                                                     input_path="x.wav", output_dir=str(tmp_path))
        sel = np.array([[c.t, c.score] for c in res.selected_candidates])
        assert np.array_equal(sel, g[f"c{case}_selected"])
        assert np.array_equal(np.array([[c.t, c.score] for c in res.planner_result.suppressed_candidates]).reshape(-1, 2), g[f"c{case}_suppressed"])
        counts = res.boundary_detection["candidate_counts"]
        assert [counts[k] for k in ("acoustic", "beat", "merged", "total", "selected", "suppressed")] == g[f"c{case}_counts"].tolist()
        feats = np.array([[c.features[k] for k in sorted(c.features)] for c in res.selected_candidates])
        assert np.array_equal(feats, g[f"c{case}_features"])
        assert (tmp_path / "vpbd_candidate_debug.json").exists()
    asr = VocalPhraseBoundaryDetector(SR).detect(mode="vpbd_asr", vocal_track=vocal, original_audio=vocal,
                                                 pure_vocal_detector=FixedPauses(pauses), feature_cache=cache, vad_segments=None)
    assert asr.boundary_detection["actual_mode"] == "vpbd_acoustic" and asr.lyrics_alignment["fallback_reason"] == "lyrics_alignment_disabled"


def _ref_test_cache():
    return types.SimpleNamespace(
        beat_times=np.arange(0.0, 12.001, 0.5, dtype=np.float32),
        rms_series=np.array([0.05] * 40 + [0.90] * 40 + [0.92] * 40 + [0.94] * 40 + [0.91] * 40 + [0.05] * 40, dtype=np.float32),
        hop_s=0.05, duration_s=12.0)


def test_reference_known_answers_beat_candidates():
    """reference tests/unit/test_beat_candidates.py:42-90."""
    assert detect_chorus_regions([0.1, 0.8, 0.85, 0.9, 0.82, 0.2], energy_threshold=0.5, min_consecutive_bars=4) == {1, 2, 3, 4}
    cache = _ref_test_cache()
    kw = dict(beat_times=cache.beat_times, rms_series=cache.rms_series, hop_s=cache.hop_s, duration_s=cache.duration_s,
              sample_rate=1000, bars_per_cut=2, base_score=0.3)
    cands = generate_beat_candidates(vocal_track=np.zeros(12000, np.float32), **kw)
    assert [c.t for c in cands] == [2.0, 6.0]
    assert {c.source for c in cands} == {CandidateSource.BEAT} and all(c.score == 0.3 for c in cands)
    assert all("vocal_cut_risk" in c.features for c in cands) and all(c.meta["bar_index"] in {1, 3} for c in cands)
    vocal = np.zeros(12000, np.float32); vocal[5920:6080] = 0.9
    risk = {c.t: c.features["vocal_cut_risk"] for c in generate_beat_candidates(vocal_track=vocal, **kw)}
    assert risk[2.0] == 0.0 and risk[6.0] > 0.8


def test_reference_known_answers_feature_wiring():
    """reference tests/unit/test_vpbd_feature_wiring.py:14-52."""
    det = VocalPhraseBoundaryDetector(sample_rate=1000)
    cache = types.SimpleNamespace(beat_times=np.array([], np.float32), mdd_series=np.array([1.0, 0.0, 1.0], np.float32),
                                  rms_series=np.zeros(3, np.float32), hop_s=1.0)
    scored = det._score_candidates(candidates=[CutCandidate(1.0, 0.5, CandidateSource.ACOUSTIC_PAUSE)],
                                   timeline=LyricsTimeline(duration_s=3.0, source="none"), feature_cache=cache)
    assert scored[0].features["mdd_affinity"] > 0.0
    rms = np.full(100, 0.1, np.float32); rms[39:43] = 1.0
    cache = types.SimpleNamespace(beat_times=np.array([], np.float32), mdd_series=np.zeros(100, np.float32), rms_series=rms, hop_s=0.05)
    scored = det._score_candidates(candidates=[CutCandidate(2.0, 0.5, CandidateSource.ACOUSTIC_PAUSE)],
                                   timeline=LyricsTimeline(duration_s=5.0, source="none"), feature_cache=cache)
    assert scored[0].features["vocal_cut_risk"] > 0.8


def test_reference_known_answer_planner_to_refine_shapes():
    """reference tests/unit/test_cutting_consistency.py:20-46 (planner half; the refine half runs on the GPU suite)."""
    plan = GlobalCutPlanner(GlobalCutPlannerConfig(hard_min_s=2.0, hard_max_s=6.0)).plan(
        [CutCandidate(4.0, 0.9, CandidateSource.ACOUSTIC_PAUSE), CutCandidate(8.0, 0.9, CandidateSource.LYRICS_GAP)], duration_s=12.0)
    pts = planner_result_to_cut_points(plan)
    assert [(p.t, p.kind) for p in pts] == [(4.0, "acoustic_pause"), (8.0, "lyrics_gap")]
    assert plan.metadata["selected_count"] == 2 and plan.cut_times == [0.0, 4.0, 8.0, 12.0]
    rescue = GlobalCutPlanner(GlobalCutPlannerConfig()).plan([], duration_s=60.0)
    assert rescue.metadata["planner"] == "rescue" and rescue.cut_times == [0.0, 15.0, 30.0, 45.0, 60.0]


def test_breath_relabelling_and_scaling():
    """reference tests/unit/test_breath_candidates.py:31-76 behaviour: breath pause types re-labelled and scaled."""
    raw = [(1.0, 0.8, {"pause_type": "breath_short"}), (2.0, 0.5, {"pause_type": "energy_valley_mdd"}), (3.0, 0.4)]
    out = adapt_legacy_acoustic_candidates(raw, breath_score_scale=0.6)
    assert [(c.t, c.source) for c in out] == [(1.0, CandidateSource.BREATH), (2.0, CandidateSource.ACOUSTIC_PAUSE), (3.0, CandidateSource.ACOUSTIC_PAUSE)]
    assert abs(out[0].score - 0.48) < 1e-12
    assert [c.t for c in adapt_legacy_acoustic_candidates(raw, breath_score_scale=0.0)] == [2.0, 3.0]
