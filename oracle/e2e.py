"""Oracle end-to-end driver: SURVEY.md §3.1 steps 3-9 (separate -> cache -> detect -> finalize) on the CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
`src/vocal_smart_splitter/core/seamless_splitter.py:300-481` (the v2.2_mdd flow between the loader
and the integer `sample_boundaries`) and `:1792-1879` (`_finalize_and_filter_cuts_v2`).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import detector as D, features as FT, librosa_ops as L, refine as R, vad as V
from .config import get_config


def finalize_and_filter_cuts(cut_candidates: Sequence[Tuple[float, float]], mix: np.ndarray,
                             vocal: Optional[np.ndarray], sr: int) -> R.RefineOutput:
    """seamless_splitter.py:1792-1879 (quirk Q1: floor_percentile 0.5 is read as a fraction -> 50th pct)."""
    if sr <= 0 or mix.size == 0:
        return R.RefineOutput([], [0, len(mix)], [])
    pts = [R.Cut(float(t), float(s)) for t, s in cut_candidates]
    if not pts:
        return R.RefineOutput([], [0, len(mix)], [])
    min_gap = float(get_config("quality_control.min_split_gap", 1.0))
    max_keep = int(get_config("pure_vocal_detection.valley_scoring.max_kept_after_nms", 150))
    guard_on = bool(get_config("quality_control.enforce_quiet_cut.enable", False))
    guard_db = float(get_config("quality_control.enforce_quiet_cut.guard_db", 2.5))
    right_ms = float(get_config("quality_control.enforce_quiet_cut.search_right_ms", 150))
    win_ms = float(get_config("quality_control.enforce_quiet_cut.win_ms", 80))
    floor_db = -60.0
    if guard_on:
        override = get_config("quality_control.enforce_quiet_cut.floor_db_override", None)
        if override is not None:
            floor_db = float(override)
        else:
            cfg = get_config("quality_control.enforce_quiet_cut.floor_percentile", 5)
            pct = float(cfg) / 100.0 if float(cfg) > 1 else float(cfg)
            hop = max(1, int(sr * 0.01))
            r = L.rms(mix, hop_length=hop)[0]
            floor_db = float(np.percentile(20.0 * np.log10(r + 1e-12), max(0.0, min(100.0, pct * 100.0))))
    topk = get_config("quality_control.nms_topk_per_10s", None)
    out = R.finalize_cut_points(
        sr, mix, vocal, pts, use_vocal_guard_first=vocal is not None, min_gap_s=min_gap, max_keep=max_keep,
        topk_per_10s=int(topk) if topk is not None else None,
        nms_window_s=float(get_config("quality_control.nms_window_s", 10.0)), guard_db=guard_db,
        search_right_ms=right_ms, guard_win_ms=win_ms, floor_db=floor_db, enable_mix_guard=guard_on,
        enable_vocal_guard=(guard_on and vocal is not None))
    out.floor_db = floor_db  # type: ignore[attr-defined]
    return out


@dataclass
class TrackResult:
    sample_boundaries: List[int]
    pauses: List[D.Pause]
    cut_candidates: List[Tuple[float, float]]
    vocal: np.ndarray
    instrumental: Optional[np.ndarray]
    cache: Optional[FT.FeatureCache]
    vad_segments: List[Dict[str, float]]
    markers: Dict
    confidence: float
    timings: Dict[str, float] = field(default_factory=dict)
    policy: Optional[object] = None           # oracle.layout.PolicyResult: the manifest-facing cuts after seamless_splitter.py:521-669


def detect_and_finalize(mix: np.ndarray, vocal: np.ndarray, sr: int, cache: Optional[FT.FeatureCache],
                        vad_segments: Optional[List[Dict[str, float]]], markers: Optional[Dict] = None,
                        timings: Optional[Dict[str, float]] = None, policy_out: Optional[list] = None):
    """seamless_splitter.py:413-481 for mode v2.2_mdd."""
    t0 = time.perf_counter()
    pauses = D.detect_pure_vocal_pauses(vocal, sr, enable_mdd_enhancement=True, original_audio=mix,
                                        feature_cache=cache, vad_segments=vad_segments)
    t1 = time.perf_counter()
    cands: List[Tuple[float, float]] = [(float(p.cut_point), float(p.confidence)) for p in pauses]
    if pauses:
        min_music = float(get_config("quality_control.pure_music_min_duration", 0.0))
        if min_music > 0.0:
            for a, b in D.no_vocal_runs(vocal, sr, min_music):
                cands.append((float(a), 1.0)); cands.append((float(b), 1.0))
        dur = len(mix) / sr
        protected = set()
        for t in (markers or {}).get("vocal_presence_cut_points_sec", []):
            if 0.0 < t < dur:
                cands.append((float(t), 1.0))
                protected.add(int(round(t * sr)))
        refined = finalize_and_filter_cuts(cands, mix, vocal, sr)
        suppressed = [(float(c.t), float(c.score)) for c in (refined.suppressed or [])]
        bounds = set(refined.sample_boundaries)
        for s in protected:                      # seamless_splitter.py:501-508
            s = int(min(max(s, 0), len(mix)))
            if s not in (0, len(mix)):
                bounds.add(s)
        bounds = sorted(bounds)
    else:   # seamless_splitter.py:421-433: no candidates -> single segment
        bounds = [0, len(mix)]
        suppressed = None
    t2 = time.perf_counter()
    if timings is not None:
        timings["detect_s"] = t1 - t0
        timings["finalize_s"] = t2 - t1
    if policy_out is not None and suppressed is not None:       # seamless_splitter.py:521-669 (not reached on the single-segment exit)
        from . import layout as LY
        policy_out.append(LY.apply_boundary_policy(bounds, vocal, len(mix), sr, suppressed=suppressed,
                                                   rms_series=None if cache is None else cache.rms_series,
                                                   hop_s=0.05 if cache is None else cache.hop_s,
                                                   beat_times=None if cache is None else cache.beat_times))
    return pauses, cands, bounds


def finalize_vpbd(mix: np.ndarray, vocal: np.ndarray, sr: int, cache: Optional[FT.FeatureCache], markers: Optional[Dict],
                  selected: Sequence[Tuple[float, float]], suppressed_by_planner: Sequence[Tuple[float, float]],
                  policy_out: Optional[list] = None) -> List[int]:
    """seamless_splitter.py:362-408,436-520 for mode vpbd_acoustic: the candidates are the VPBD planner's selection
    (`selected`: (t, score); the planner's positive-score suppressed ones as the rescue set when nothing was selected,
    `:396-401`) instead of the pauses; then the same no-vocal runs, presence markers, guard and boundary policy as v2.2_mdd."""
    cands: List[Tuple[float, float]] = [(float(t), float(s)) for t, s in selected]
    if not cands:
        cands = [(float(t), float(s)) for t, s in suppressed_by_planner if float(s) > 0.0]
    if not cands:
        return [0, len(mix)]
    min_music = float(get_config("quality_control.pure_music_min_duration", 0.0))
    if min_music > 0.0:
        for a, b in D.no_vocal_runs(vocal, sr, min_music):
            cands.append((float(a), 1.0)); cands.append((float(b), 1.0))
    dur = len(mix) / sr
    protected = set()
    for t in (markers or {}).get("vocal_presence_cut_points_sec", []):
        if 0.0 < t < dur:
            cands.append((float(t), 1.0))
            protected.add(int(round(t * sr)))
    refined = finalize_and_filter_cuts(cands, mix, vocal, sr)
    suppressed = [(float(c.t), float(c.score)) for c in (refined.suppressed or [])]
    bounds = set(refined.sample_boundaries)
    for s in protected:
        s = int(min(max(s, 0), len(mix)))
        if s not in (0, len(mix)):
            bounds.add(s)
    bounds = sorted(bounds)
    if policy_out is not None:
        from . import layout as LY
        policy_out.append(LY.apply_boundary_policy(bounds, vocal, len(mix), sr, suppressed=suppressed,
                                                   rms_series=None if cache is None else cache.rms_series,
                                                   hop_s=0.05 if cache is None else cache.hop_s,
                                                   beat_times=None if cache is None else cache.beat_times))
    return bounds


def run_track(mix: np.ndarray, sr: int, weights, *, vad_fn=None, n_levels: int = 5, l: int = 3) -> TrackResult:
    """Full oracle path for one track (needs torch for the CPU U-Net)."""
    from . import separator as S
    timings: Dict[str, float] = {}
    t0 = time.perf_counter()
    feat = FT.ChunkFeatureOracle(sr)
    cvad = V.ChunkVadOracle(sr, float(get_config("advanced_vad.silero_merge_gap_ms", 120.0)),
                            float(get_config("advanced_vad.focus_window_pad_s", 0.2)),
                            vad_fn or V.energy_gate_vad(sr))

    def on_chunk(plan, mix_chunk, vocal_chunk, has_effective):
        cvad.process_chunk(plan, vocal_chunk, sr)
        if has_effective:
            feat.add_chunk(plan, mix_chunk, sr)

    vocal, inst, _ = S.separate_track(mix, sr, weights, on_chunk=on_chunk, n_levels=n_levels, l=l)
    vad_segments = cvad.finalize()
    cache = feat.finalize(mix)
    conf = D.estimate_confidence(vocal, inst, mix)
    markers = D.vocal_presence_markers(vocal, sr)
    timings["separate_s"] = time.perf_counter() - t0
    pol: list = []
    pauses, cands, bounds = detect_and_finalize(mix, vocal, sr, cache, vad_segments, markers, timings, policy_out=pol)
    return TrackResult(bounds, pauses, cands, vocal, inst, cache, vad_segments, markers, conf, timings, pol[0] if pol else None)
