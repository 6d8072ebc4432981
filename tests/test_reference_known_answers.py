"""Known answers the reference's own unit tests hold for the boundary-policy row, replayed against
the product's host logic and, where the case needs no ASR prior, against the oracle, followed by the reference's cases
for the VPBD pool (§8 a18).  The cases are the reference's inputs and expected outputs
(`tests/unit/test_segment_layout_refiner.py`, `tests/unit/test_cpu_baseline_perfect_reconstruction.py`, and the files
named at the second block); they are data, rebuilt here from their description.  CPU only.
"""
import json
from types import SimpleNamespace as NS
from typing import List

import numpy as np

from audio_cut_amd.analysis.boundary_features import BoundaryFeatureExtractor, BoundaryFeatures, LyricsTimeline
from audio_cut_amd.analysis.features_cache import TrackFeatureCache
from audio_cut_amd.config import reset_runtime_config, set_runtime_config
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.core.vocal_phrase_boundary_detector import VocalPhraseBoundaryDetector
from audio_cut_amd.cutting.cut_candidate import CandidateSource, CutCandidate, adapt_legacy_acoustic_candidates
from audio_cut_amd.cutting.global_cut_planner import (GlobalCutPlanner, GlobalCutPlannerConfig, apply_guard_shift_metadata,
                                                      planner_result_to_cut_points)
from audio_cut_amd.cutting.phrase_boundary_scorer import PhraseBoundaryScorer, write_candidate_debug_json
from audio_cut_amd.cutting.refine import CutAdjustment
from audio_cut_amd.cutting.segment_layout_refiner import LayoutConfig, Segment, refine_layout
from oracle import layout as OL


def _cache(rms: np.ndarray, duration_s: float, hop_s: float) -> TrackFeatureCache:
    z = np.zeros(len(rms), dtype=np.float32)
    return TrackFeatureCache(sr=44100, hop_length=int(44100 * hop_s), hop_s=hop_s, duration_s=duration_s, rms_series=rms,
                             spectral_flatness=z.copy(), onset_envelope=z.copy(), onset_strength=z.copy(),
                             onset_frames=np.array([], dtype=np.int64), rms_max=float(rms.max()), onset_max=0.0,
                             bpm_features=None, tempo_curve=None, beat_times=np.array([], dtype=np.float32), global_mdd=0.0,
                             mdd_series=z.copy())


def _one_notch(duration_s: float, valley_s: float, hop_s: float = 0.5) -> TrackFeatureCache:
    n = int(duration_s / hop_s) + 1
    rms = np.ones(n, dtype=np.float32)
    k = int(round(valley_s / hop_s))
    rms[max(0, k - 1): min(n, k + 2)] = np.array([0.08, 0.01, 0.08], dtype=np.float32)
    return _cache(rms, duration_s, hop_s)


def _vee_valleys(duration_s: float, valley_times: List[float], hop_s: float = 0.05) -> TrackFeatureCache:
    n = int(duration_s / hop_s) + 1
    rms = np.ones(n, dtype=np.float32)
    half = max(2, int(round(1.0 / hop_s)))
    for t in valley_times:
        k = int(round(t / hop_s))
        for i in range(max(0, k - half), min(n, k + half + 1)):
            rms[i] = min(rms[i], 0.01 + 0.4 * abs(i - k) / float(half))
    return _cache(rms, duration_s, hop_s)


def _ends(result) -> List[float]:
    return [round(s.end, 1) for s in result.segments[:-1]]


def test_long_segment_splits_at_valley_next_to_asr_boundary():
    res = refine_layout([Segment(0.0, 30.0, "human"), Segment(30.0, 35.0, "music")], [],
                        config=LayoutConfig(enable=True, soft_max_s=15.0, min_gap_s=1.0), sample_rate=44100,
                        features=_one_notch(35.0, 20.0), asr_boundary_times=[20.1])
    assert 20.1 in _ends(res) and 15.0 not in _ends(res)


def test_long_segment_without_valley_is_left_whole():
    feats = _one_notch(35.0, 20.0)
    feats.rms_series[:] = 1.0
    res = refine_layout([Segment(0.0, 30.0, "human"), Segment(30.0, 35.0, "music")], [],
                        config=LayoutConfig(enable=True, soft_max_s=15.0, min_gap_s=1.0), sample_rate=44100,
                        features=feats, asr_boundary_times=[15.0])
    assert _ends(res) == [30.0]
    segs, _, fresh = OL.refine_layout([[0.0, 30.0, "human"], [30.0, 35.0, "music"]],
                                      OL.LayoutConfig(enable=True, soft_max_s=15.0, min_gap_s=1.0), [], feats.rms_series, 0.5, [])
    assert [round(s[1], 1) for s in segs[:-1]] == [30.0] and fresh == []


def test_phrase_boundary_beats_a_deeper_valley_inside_a_word():
    feats = _one_notch(35.0, 20.0)
    feats.rms_series[:] = 1.0
    a, b = int(round(20.0 / feats.hop_s)), int(round(22.0 / feats.hop_s))
    feats.rms_series[a - 1: a + 2] = np.array([0.08, 0.01, 0.08], dtype=np.float32)
    feats.rms_series[b - 1: b + 2] = np.array([0.12, 0.05, 0.12], dtype=np.float32)
    feats.rms_max = float(feats.rms_series.max())
    res = refine_layout([Segment(0.0, 30.0, "human"), Segment(30.0, 35.0, "music")], [],
                        config=LayoutConfig(enable=True, soft_max_s=15.0, min_gap_s=1.0), sample_rate=44100, features=feats,
                        asr_boundary_times=[22.0], asr_word_intervals=[(19.8, 20.2)])
    assert 22.0 in _ends(res) and 20.0 not in _ends(res)


def test_micro_piece_made_by_a_split_rejoins_its_same_kind_neighbour():
    feats = _vee_valleys(30.0, [9.68, 20.70])
    cfg = dict(enable=True, micro_merge_s=2.0, soft_min_s=5.0, soft_max_s=12.0, min_gap_s=1.0)
    start = [(0.0, 21.95, "human"), (21.95, 23.31, "human"), (23.31, 26.47, "music")]
    want = [(0.0, 9.7, "human"), (9.7, 21.95, "human"), (21.95, 26.47, "music")]
    res = refine_layout([Segment(*s) for s in start], [], config=LayoutConfig(**cfg), sample_rate=44100, features=feats)
    assert [(round(s.start, 2), round(s.end, 2), s.kind) for s in res.segments] == want
    assert all(s.duration >= 2.0 for s in res.segments)
    segs, _, _ = OL.refine_layout([list(s) for s in start], OL.LayoutConfig(**cfg), [], feats.rms_series, feats.hop_s, [])
    assert [(round(s[0], 2), round(s[1], 2), s[2]) for s in segs] == want
    # the two restatements agree to the last bit, not just to the printed precision
    assert [(s.start, s.end) for s in res.segments] == [(s[0], s[1]) for s in segs]


def test_sample_level_split_reconstructs_the_track_exactly():
    splitter = SeamlessSplitter.__new__(SeamlessSplitter)
    splitter.sample_rate = 1000
    audio = np.linspace(-1.0, 1.0, 1001, dtype=np.float32)
    cuts, flags = [0, 123, 456, 789, 1001], [True, False, True, False]
    pieces, out_flags, debug = splitter._split_at_sample_level(audio, cuts, segment_flags=list(flags))
    assert out_flags == flags and debug is None
    assert np.array_equal(np.concatenate(pieces), audio)
    spans, oflags = OL.split_at_sample_level(len(audio), cuts, flags, 1000)
    assert oflags == flags and spans == list(zip(cuts[:-1], cuts[1:]))


# ----------------------------------------------------------------------------------------------------------------------
# VPBD pool row: the reference's unit-test cases for the planner, the scorer, the candidate / feature
# containers and the feature extractor (`tests/unit/test_global_cut_planner.py`, `test_phrase_boundary_scorer.py`,
# `test_boundary_data_models.py`, `test_boundary_features.py`, `test_boundary_features_tolerance.py`,
# `test_vocal_phrase_boundary_detector.py:22-49`).  Words / sentences / voiced regions are duck-typed records here: the
# lyrics providers that would produce them are out of scope, the arithmetic over them is not.
# ----------------------------------------------------------------------------------------------------------------------

def test_planner_dynamic_path_honours_duration_limits():
    plan = GlobalCutPlanner(GlobalCutPlannerConfig(hard_min_s=3.0, hard_max_s=8.0, target_min_s=4.0, target_max_s=7.0)).plan(
        [CutCandidate(4.0, 0.8, CandidateSource.LYRICS_GAP), CutCandidate(10.0, 0.9, CandidateSource.SENTENCE_END),
         CutCandidate(16.0, 0.7, CandidateSource.ACOUSTIC_PAUSE)], duration_s=20.0)
    assert plan.feasible is True and plan.cut_times == [0.0, 4.0, 10.0, 16.0, 20.0]
    assert [c.t for c in plan.selected_candidates] == [4.0, 10.0, 16.0]
    assert plan.metadata["planner"] == "dynamic_programming"


def test_planner_keeps_the_best_candidate_per_second():
    plan = GlobalCutPlanner(GlobalCutPlannerConfig(hard_min_s=2.0, hard_max_s=8.0, max_candidates_per_second=1.0)).plan(
        [CutCandidate(5.10, 0.1, CandidateSource.LYRICS_GAP), CutCandidate(5.20, 0.9, CandidateSource.LYRICS_GAP),
         CutCandidate(5.30, 0.8, CandidateSource.LYRICS_GAP)], duration_s=10.0)
    assert [c.t for c in plan.selected_candidates] == [5.2]
    assert sorted(c.t for c in plan.suppressed_candidates) == [5.1, 5.3]


def test_planner_charges_vocal_risk_and_beat_conflict():
    plan = GlobalCutPlanner(GlobalCutPlannerConfig(hard_min_s=4.0, hard_max_s=9.0, vocal_risk_weight=0.5, beat_conflict_weight=0.4)).plan(
        [CutCandidate(6.0, 0.9, CandidateSource.ACOUSTIC_PAUSE, features={"vocal_cut_risk": 1.0}),
         CutCandidate(7.0, 0.8, CandidateSource.LYRICS_GAP, features={"beat_conflict": 0.0})], duration_s=14.0)
    assert plan.cut_times == [0.0, 7.0, 14.0]


def test_planner_rescue_grid_when_no_path_exists():
    plan = GlobalCutPlanner(GlobalCutPlannerConfig(hard_min_s=3.0, hard_max_s=8.0, rescue_enabled=True)).plan([], duration_s=21.0)
    assert plan.feasible is True and plan.cut_times == [0.0, 7.0, 14.0, 21.0] and plan.metadata["planner"] == "rescue"


def test_planner_result_to_cut_points_and_guard_shift_metadata():
    plan = GlobalCutPlanner(GlobalCutPlannerConfig(hard_min_s=2.0, hard_max_s=8.0)).plan(
        [CutCandidate(4.0, 0.8, CandidateSource.LYRICS_GAP)], duration_s=8.0)
    pts = planner_result_to_cut_points(plan)
    updated = apply_guard_shift_metadata(plan, [CutAdjustment(raw_time=4.0, guard_time=4.1, final_time=4.1, score=0.8,
                                                              guard_shift_ms=100.0, final_shift_ms=100.0)])
    assert [(p.t, p.score, p.kind) for p in pts] == [(4.0, 0.8, "lyrics_gap")]
    assert updated.metadata["guard_shift_ms_by_raw_time"][4.0] == 100.0


def test_scorer_clamps_reads_config_and_annotates(tmp_path):
    scorer = PhraseBoundaryScorer(weights={"acoustic_pause": 0.5, "asr_gap": 0.4, "sentence_end": 0.3, "inside_word_penalty": 1.0,
                                           "singing_penalty": 1.0})
    assert scorer.score(BoundaryFeatures(acoustic_pause=1.0, asr_gap=1.0, sentence_end=1.0)) == 1.0
    assert scorer.score(BoundaryFeatures(acoustic_pause=1.0, inside_word_penalty=1.0, singing_penalty=1.0)) == 0.0
    cfg = PhraseBoundaryScorer.from_config({"weights": {"asr_gap": 0.25, "inside_word_penalty": 0.5}})
    assert cfg.weights["asr_gap"] == 0.25 and cfg.weights["inside_word_penalty"] == 0.5
    assert PhraseBoundaryScorer(weights={"breath": 0.2}).score(BoundaryFeatures(breath=1.0)) == 0.2
    feats = BoundaryFeatures(asr_gap=1.0, sentence_end=1.0)
    scored = PhraseBoundaryScorer(weights={"asr_gap": 0.5, "sentence_end": 0.5}).score_candidate(
        CutCandidate(1.2, 0.0, CandidateSource.LYRICS_GAP), feats)
    assert scored.score == 1.0 and scored.features == feats.to_dict() and "vpbd_score" in scored.reasons
    path = tmp_path / "candidates.json"
    write_candidate_debug_json([CutCandidate(1.2, 0.8, CandidateSource.LYRICS_GAP, reasons=["asr_gap"], features={"asr_gap": 1.0})], path)
    payload = json.loads(path.read_text(encoding="utf-8"))
    assert payload["candidates"][0]["source"] == "lyrics_gap" and payload["candidates"][0]["features"]["asr_gap"] == 1.0


def test_candidate_and_feature_containers_clamp():
    c = CutCandidate(t=1.25, score=1.5, source="lyrics_gap", reasons=["word gap"], features={"asr_gap": 0.9}, meta={"word_left": "hello"})
    assert c.score == 1.0 and c.source == CandidateSource.LYRICS_GAP and c.to_dict()["source"] == "lyrics_gap"
    f = BoundaryFeatures(acoustic_pause=2.0, asr_gap=-1.0, sentence_end=0.5)
    assert f.acoustic_pause == 1.0 and f.asr_gap == 0.0 and f.to_dict()["sentence_end"] == 0.5
    adapted = adapt_legacy_acoustic_candidates([(1.25, 0.7, {"rms_valley_db": -45.0})], source=CandidateSource.ACOUSTIC_PAUSE)
    assert adapted[0].source == CandidateSource.ACOUSTIC_PAUSE and adapted[0].score == 0.7
    assert adapted[0].meta["rms_valley_db"] == -45.0 and "legacy_acoustic" in adapted[0].reasons


def test_feature_extractor_affinity_and_risk_terms():
    """The acoustic terms of the reference's feature-extractor tests (`tests/unit/test_boundary_features*.py`); the lyrics-derived
    terms have no evaluator here (no ASR timeline on this path) and stay 0."""
    f = BoundaryFeatureExtractor(timeline=LyricsTimeline(duration_s=5.0, source="none"), beat_times=[1.80], mdd_times=[1.78],
                                 affinity_tolerance_s=0.05).extract(1.80)
    assert f.beat_affinity == 1.0 and f.mdd_affinity > 0.0
    assert f.asr_gap == f.sentence_end == f.inside_word_penalty == f.singing_penalty == 0.0
    rms = np.full(100, 0.1, dtype=np.float32); rms[39:43] = 1.0
    ex = BoundaryFeatureExtractor(timeline=LyricsTimeline(duration_s=5.0, source="none"), rms_series=rms, hop_s=0.05)
    assert ex.extract(0.25).vocal_cut_risk < 0.2 and ex.extract(2.0).vocal_cut_risk > 0.8
    ex = BoundaryFeatureExtractor(timeline=LyricsTimeline(duration_s=5.0, source="none"), beat_times=[1.0, 2.0], affinity_tolerance_s=0.12)
    assert ex.extract(1.0).beat_conflict == 0.0 and ex.extract(1.35).beat_conflict > 0.8


def test_zero_beat_base_score_switches_beat_candidates_off():
    set_runtime_config({"vpbd.beat_candidates.enable": True, "vpbd.beat_candidates.bars_per_cut": 2, "vpbd.beat_candidates.base_score": 0.0})
    try:
        cache = NS(beat_times=np.arange(0.0, 8.001, 0.5, dtype=np.float32), rms_series=np.full(160, 0.8, dtype=np.float32), hop_s=0.05)
        got = VocalPhraseBoundaryDetector(sample_rate=16000)._build_beat_candidates(
            vocal_track=np.full(16000 * 8, 0.2, dtype=np.float32), feature_cache=cache, duration_s=8.0)
    finally:
        reset_runtime_config()
    assert got == []


def test_pool_merge_dedupes_neighbours_and_keeps_every_source():
    """reference tests/unit/test_candidate_pool_fusion.py:107-129: the better-scored member names the merged candidate."""
    acoustic = CutCandidate(t=6.04, score=0.4, source=CandidateSource.ACOUSTIC_PAUSE, reasons=["legacy_acoustic"], meta={"pause_type": "true_pause"})
    sentence = CutCandidate(t=6.0, score=0.85, source=CandidateSource.SENTENCE_END, reasons=["sentence_end", "punctuation_end"], meta={"text": "phrase."})
    merged = VocalPhraseBoundaryDetector(sample_rate=44100)._merge_candidate_pool([acoustic], [sentence], tolerance_s=0.12)
    assert len(merged) == 1 and merged[0].t == 6.0 and merged[0].source == CandidateSource.SENTENCE_END
    assert set(merged[0].meta["sources"]) == {"acoustic_pause", "sentence_end"} and merged[0].meta["source_count"] == 2


def test_legacy_pool_keeps_a_breath_typed_pause_as_an_acoustic_candidate(tmp_path):
    """reference tests/unit/test_candidate_pool_fusion.py:176-222."""
    class OneBreathMddPause:
        def detect_pure_vocal_pauses(self, *args, **kwargs):
            return [NS(start_time=5.8, end_time=6.2, cut_point=6.0, confidence=0.8, duration=0.4, pause_type="breath_mdd")]

    set_runtime_config({"vpbd.candidate_pool": "legacy", "vpbd.breath_score_scale": 0.0, "lyrics_alignment.enabled": False,
                        "global_planner.hard_min_s": 2.0, "global_planner.hard_max_s": 8.0, "global_planner.target_min_s": 5.0,
                        "global_planner.target_max_s": 7.0})
    try:
        vocal = np.zeros(int(44100 * 12.0), dtype=np.float32)
        res = VocalPhraseBoundaryDetector(sample_rate=44100).detect(
            mode="vpbd_acoustic", vocal_track=vocal, original_audio=vocal.copy(), pure_vocal_detector=OneBreathMddPause(),
            feature_cache=None, vad_segments=[], input_path=str(tmp_path / "sample.wav"), output_dir=str(tmp_path / "out"))
    finally:
        reset_runtime_config()
    counts = res.boundary_detection["candidate_counts"]
    assert res.boundary_detection["candidate_pool"] == "legacy" and counts["acoustic"] == 1 and counts["merged"] == 1
    assert res.selected_candidates[0].source == CandidateSource.ACOUSTIC_PAUSE


def test_segment_file_names_carry_label_lib_marker_and_duration(monkeypatch, tmp_path):
    """reference tests/unit/test_legacy_mode_regression.py:37-70 (exporter row)."""
    from pathlib import Path

    import audio_cut_amd.utils.audio_export as export_module
    stems: List[str] = []

    def record_only(audio, sample_rate, base_path, format_name, *, options=None):
        stems.append(Path(base_path).name)
        return f"{base_path}.{format_name}"

    monkeypatch.setattr(export_module, "export_audio", record_only)
    ex = export_module.SegmentExporter(sample_rate=44100)
    files = ex.export_segments([np.zeros(44100, dtype=np.float32)] * 2, str(tmp_path), segment_is_vocal=[True, False], export_format="wav",
                               export_options={}, lib_flags=[True, False], duration_map={0: 2.0, 1: 3.5})
    assert stems == ["segment_001_human_lib_2.0", "segment_002_music_3.5"]
    assert [Path(f).name for f in files] == ["segment_001_human_lib_2.0.wav", "segment_002_music_3.5.wav"]
    monkeypatch.undo()
    # the real writer: same names on disk, bytes equal to the span form the device path uses
    audio = np.linspace(-0.5, 0.5, 2000, dtype=np.float32)
    a = ex.export_segments([audio[:800], audio[800:]], str(tmp_path / "a"), segment_is_vocal=[True, False], export_format="wav",
                           export_options={}, always_append_duration=True)
    b = ex.export_spans(export_module.PackedTrack(audio, 44100), [(0, 800), (800, 2000)], str(tmp_path / "b"),
                        segment_is_vocal=[True, False], always_append_duration=True)
    assert [Path(f).name for f in a] == [Path(f).name for f in b] == ["segment_001_human_0.0.wav", "segment_002_music_0.0.wav"]
    assert all(Path(x).read_bytes() == Path(y).read_bytes() for x, y in zip(a, b))


# ----------------------------------------------------------------------------------------------------------------------
# Manifest row: `_build_manifest` and the QA report, from the reference's `tests/unit/test_api_manifest.py`,
# `test_qa_report.py` and `test_legacy_mode_regression.py:73-108`.  Input files are written with the stdlib `wave` module.
# ----------------------------------------------------------------------------------------------------------------------
def _silent_wav(path, seconds: float):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(44100); w.writeframes(b"\x00\x00" * int(44100 * seconds))
    return path


def _one_segment_result(seconds: float) -> dict:
    return {"success": True, "export_plan": [], "cut_points_sec": [0.0, seconds], "cut_points_samples": [0, int(44100 * seconds)],
            "segment_labels": ["human"], "segment_durations": [seconds], "segment_vocal_flags": [True]}


def test_manifest_plain_shape_without_vpbd_blocks(tmp_path):
    from audio_cut_amd.api import _build_manifest
    man = _build_manifest(result=_one_segment_result(2.0), input_path=_silent_wav(tmp_path / "song.wav", 2.0), export_dir=tmp_path,
                          mode="v2.2_mdd", sample_rate=44100, channels=1, layout_cfg={})
    assert man["cuts"]["final"] == [0.0, 2.0] and man["version"] == "v2.2_mdd"
    assert "lyrics_alignment" not in man and "boundary_detection" not in man and "lyrics" not in man["segments"][0]
    assert set(man) == {"version", "success", "job", "export_plan", "audio", "layout_cfg", "cuts", "segments", "artifacts", "guard",
                        "separation", "timings_ms", "stats", "qa_report"}
    assert man["guard"] == {"shift_stats": {}, "adjustments": [], "precision_ok": True, "threshold_ms": {}}
    assert man["layout_cfg"] == {"applied": False} and man["audio"]["duration"] == 2.0 and man["audio"]["hash"].startswith("sha256:")


def test_manifest_passes_auto_profile_and_smart_segmentation_through(tmp_path):
    from audio_cut_amd.api import _build_manifest
    src = _silent_wav(tmp_path / "song.wav", 2.0)
    res = _one_segment_result(2.0) | {"auto_profile": {"style": "pop", "confidence": 0.7, "bpm": 108.0, "mdd": 0.38,
                                                       "applied_overrides": ["global_planner.target_min_s"]}}
    man = _build_manifest(result=res, input_path=src, export_dir=tmp_path, mode="vpbd_asr", sample_rate=44100, channels=1, layout_cfg={})
    assert man["auto_profile"]["style"] == "pop" and man["auto_profile"]["applied_overrides"] == ["global_planner.target_min_s"]
    res = {"success": True, "method": "smart_segment_v2", "export_plan": [], "cut_points_sec": [0.0, 2.0, 4.0],
           "cut_points_samples": [0, 88200, 176400], "segment_labels": ["human", "music"], "segment_durations": [2.0, 2.0],
           "segment_vocal_flags": [True, False], "bpm": 120.0, "bar_duration_s": 2.0, "density": "medium", "silence_boundaries": [2.0]}
    man = _build_manifest(result=res, input_path=src, export_dir=tmp_path, mode="librosa_onset", sample_rate=44100, channels=1, layout_cfg={})
    assert man["version"] == "librosa_onset"
    assert man["smart_segmentation"] == {"method": "smart_segment_v2", "bpm": 120.0, "bar_duration_s": 2.0, "density": "medium",
                                         "silence_boundaries": [2.0]}


def test_manifest_annotates_cuts_with_planner_candidates(tmp_path):
    """`_build_final_cuts`: a selected candidate followed through the guard's move names the cut it became."""
    from audio_cut_amd.api import _build_manifest
    res = {"success": True, "export_plan": [], "cut_points_sec": [0.0, 4.1, 8.0], "cut_points_samples": [0, 180810, 352800],
           "segment_labels": ["human", "human"], "segment_durations": [4.1, 3.9], "segment_vocal_flags": [True, True],
           "boundary_detection": {"selected": [{"t": 4.0, "score": 0.8, "source": "breath", "features": {"beat_affinity": 0.9},
                                                "reasons": ["vpbd_score"], "meta": {"sources": ["breath", "beat"]}}],
                                  "planner": {"final_time_by_raw_time": {4.0: 4.1}, "guard_shift_ms_by_raw_time": {4.0: 100.0}}}}
    man = _build_manifest(result=res, input_path=_silent_wav(tmp_path / "song.wav", 8.0), export_dir=tmp_path, mode="vpbd_acoustic",
                          sample_rate=44100, channels=1, layout_cfg={"enable": True})
    assert man["cuts"]["final"][0] == {"t": 0.0} and man["cuts"]["final"][2] == {"t": 8.0}
    assert man["cuts"]["final"][1] == {"t": 4.1, "score": 0.8, "source": "breath", "features": {"beat_affinity": 0.9},
                                       "reasons": ["vpbd_score"], "meta": {"sources": ["breath", "beat"]}, "guard_shift_ms": 100.0}
    qa = man["qa_report"]
    assert qa["breath_cut_ratio"] == 1.0 and qa["beat_aligned_ratio"] == 1.0 and qa["avg_boundary_score"] == 0.8
    assert qa["guard_shift_p50_ms"] == 100.0 and man["layout_cfg"] == {"enable": True, "applied": False}


def test_qa_report_known_answers(tmp_path):
    import pytest
    from audio_cut_amd.api import _build_manifest
    from audio_cut_amd.qa_report import build_qa_report
    report = build_qa_report({
        "audio": {"duration": 20.0}, "segments": [{"duration": 6.0}, {"duration": 10.0}, {"duration": 16.0}],
        "cuts": {"final": [{"t": 0.0}, {"t": 1.2, "score": 0.4, "guard_shift_ms": 10.0, "source": "breath"},
                           {"t": 8.0, "score": 0.8, "guard_shift_ms": 30.0, "source": "beat", "features": {"beat_affinity": 1.0}}, {"t": 20.0}]},
        "lyrics_alignment": {"fallback_reason": "timeout", "timeline": {
            "duration_s": 20.0,
            "words": [{"text": "hello", "start_s": 1.0, "end_s": 1.5, "confidence": 0.8}, {"text": "world", "start_s": 6.0, "end_s": 7.0, "confidence": None}],
            "vad_regions": [{"start_s": 1.0, "end_s": 2.0, "confidence": 0.9, "kind": "singing"}]}}})
    assert report["segments_count"] == 3 and report["median_segment_s"] == 10.0
    assert report["segment_5_15_pass_rate"] == pytest.approx(2 / 3)
    assert report["cut_inside_word_rate"] == 0.5 and report["cut_inside_singing_rate"] == 0.5
    assert report["avg_boundary_score"] == 0.6 and report["lyrics_coverage_ratio"] == 0.075 and report["asr_avg_confidence"] == 0.8
    assert report["guard_shift_p50_ms"] == 20.0 and report["guard_shift_p95_ms"] == 29.0 and report["fallback_reason"] == "timeout"
    assert report["breath_cut_ratio"] == 0.5 and report["beat_aligned_ratio"] == 0.5
    man = _build_manifest(result=_one_segment_result(8.0), input_path=_silent_wav(tmp_path / "song.wav", 8.0), export_dir=tmp_path,
                          mode="v2.2_mdd", sample_rate=44100, channels=1, layout_cfg={})
    assert man["qa_report"]["segments_count"] == 1 and man["qa_report"]["segment_5_15_pass_rate"] == 1.0


def test_guard_shift_stats_known_answer():
    """`_set_guard_adjustments` arithmetic (reference `seamless_splitter.py:2423-2470`) on a hand-made set of adjustments."""
    adj = [CutAdjustment(raw_time=1.0, guard_time=1.1, final_time=1.15, score=1.0, guard_shift_ms=100.0, final_shift_ms=150.0),
           CutAdjustment(raw_time=5.0, guard_time=5.0, final_time=4.95, score=1.0, guard_shift_ms=0.0, final_shift_ms=-50.0),
           CutAdjustment(raw_time=9.0, guard_time=9.3, final_time=9.3, score=1.0, guard_shift_ms=300.0, final_shift_ms=300.0)]
    st = SeamlessSplitter._guard_shift_stats(adj)
    assert st["count"] == 3 and st["max_shift_ms"] == 300.0 and st["avg_shift_ms"] == (150.0 + 50.0 + 300.0) / 3
    assert st["avg_guard_only_shift_ms"] == 225.0 and st["avg_vocal_guard_shift_ms"] == 200.0 and st["avg_mix_guard_shift_ms"] == 50.0
    assert st["p95_shift_ms"] == float(np.percentile([150.0, 50.0, 300.0], 95.0))
    assert SeamlessSplitter._guard_shift_stats([])["count"] == 0
