"""VPBD unified candidate pool (mode `vpbd_acoustic`) — mirrors the acoustic path of the reference's
`src/vocal_smart_splitter/core/vocal_phrase_boundary_detector.py:49-385`:
acoustic pauses from `PureVocalPauseDetector` (HIP kernels) + weak beat candidates in high-energy bars +
+-120 ms fusion + feature scoring (MDD valleys, cached-RMS vocal risk, beat affinity/conflict) + the
global DP planner.  Host logic over <= a few hundred candidates; the heavy lifting (stems, caches, pauses)
is already in HBM-resident kernels upstream.

`vpbd_asr` resolves to `vpbd_acoustic` with `fallback_reason="lyrics_alignment_disabled"`, exactly what the
reference does when `lyrics_alignment.enabled` is false (`:78-80`, the shipped default `unified.yaml:28-29`);
the ASR providers themselves are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np

from ..analysis.boundary_features import BoundaryFeatureExtractor, LyricsTimeline
from ..config import get_config
from ..cutting.beat_candidates import generate_beat_candidates
from ..cutting.cut_candidate import CandidateSource, CutCandidate, adapt_legacy_acoustic_candidates
from ..cutting.global_cut_planner import GlobalCutPlanner, GlobalCutPlannerConfig, GlobalCutPlanResult
from ..cutting.phrase_boundary_scorer import PhraseBoundaryScorer, write_candidate_debug_json


@dataclass
class VPBDDetectionResult:
    selected_candidates: List[CutCandidate]
    planner_result: GlobalCutPlanResult
    boundary_detection: Dict[str, Any]
    lyrics_alignment: Dict[str, Any]


def _section(name: str) -> Dict[str, Any]:
    try:
        v = get_config(name, {})
    except Exception:
        return {}
    return dict(v) if isinstance(v, dict) else {}


def _planner_config() -> GlobalCutPlannerConfig:
    c = _section("global_planner")
    return GlobalCutPlannerConfig(
        hard_min_s=float(c.get("hard_min_s", 2.0)), hard_max_s=float(c.get("hard_max_s", 18.0)),
        target_min_s=float(c.get("target_min_s", 5.0)), target_max_s=float(c.get("target_max_s", 12.0)),
        duration_penalty_weight=float(c.get("duration_penalty_weight", 0.15)),
        vocal_risk_weight=float(c.get("vocal_risk_weight", 0.25)), beat_conflict_weight=float(c.get("beat_conflict_weight", 0.15)),
        max_candidates_per_second=float(c.get("max_candidates_per_second", 2.0)), rescue_enabled=bool(c.get("rescue_enabled", True)))


class VocalPhraseBoundaryDetector:
    def __init__(self, sample_rate: int = 44100) -> None:
        self.sample_rate = sample_rate

    def detect(self, *, mode: str, vocal_track: np.ndarray, original_audio: np.ndarray, pure_vocal_detector: Any,
               feature_cache: Optional[Any], vad_segments: Optional[List[Dict[str, float]]], input_path: str = "",
               output_dir: str = "", device_state: Optional[Dict[str, Any]] = None) -> VPBDDetectionResult:
        sr = self.sample_rate
        duration_s = len(original_audio) / float(sr) if sr > 0 else 0.0
        actual_mode, fallback_reason = mode, None
        timeline = LyricsTimeline(duration_s=duration_s, source="none")
        vpbd_cfg = _section("vpbd")
        pool = str(vpbd_cfg.get("candidate_pool", "unified")).strip().lower()
        if pool not in {"unified", "legacy"}:
            pool = "unified"
        unified = pool == "unified"
        lyrics_cfg = _section("lyrics_alignment")
        strict = bool(lyrics_cfg.get("strict", False))
        lyrics_enabled = bool(lyrics_cfg.get("enabled", False)) and mode == "vpbd_asr"
        if mode == "vpbd_asr":
            if lyrics_enabled:
                raise NotImplementedError("lyrics/ASR providers are outside the separate+detect hot path (SURVEY.md §2 #16)")
            actual_mode, fallback_reason = "vpbd_acoustic", "lyrics_alignment_disabled"

        state = device_state or {}
        acoustic = self._build_acoustic_candidates(vocal_track=vocal_track, original_audio=original_audio,
                                                   pure_vocal_detector=pure_vocal_detector, feature_cache=feature_cache,
                                                   vad_segments=vad_segments, enable_mdd=True, include_breath_candidates=unified,
                                                   device_state=state)
        beats = self._build_beat_candidates(vocal_track=vocal_track, feature_cache=feature_cache, duration_s=duration_s,
                                            device_state=state) if unified else []
        merged = self._merge_candidate_pool(acoustic, [], beats)
        scored = self._score_candidates(candidates=merged, timeline=timeline, feature_cache=feature_cache)
        debug_path: Optional[str] = None
        if bool(vpbd_cfg.get("candidate_debug_json", False)) and output_dir:
            debug_path = str(Path(output_dir) / "vpbd_candidate_debug.json")
            write_candidate_debug_json(scored, debug_path)
        plan = GlobalCutPlanner(_planner_config()).plan(scored, duration_s=duration_s)
        lyrics_meta = {"enabled": lyrics_enabled, "provider": str(lyrics_cfg.get("provider", "disabled")), "strict": strict,
                       "fallback_reason": fallback_reason, "word_count": 0, "sentence_count": 0, "vad_region_count": 0,
                       "warnings": [], "timeline": timeline.to_dict()}
        boundary_meta = {
            "mode": mode, "actual_mode": actual_mode, "candidate_pool": pool, "candidate_debug_path": debug_path,
            "candidate_counts": {"acoustic": len(acoustic), "lyrics": 0, "lyrics_pooled": 0, "beat": len(beats),
                                 "merged": len(merged), "total": len(scored), "selected": len(plan.selected_candidates),
                                 "suppressed": len(plan.suppressed_candidates), "lyrics_soft_prior": 0},
            "planner": dict(plan.metadata),
            "selected": [c.to_dict() for c in plan.selected_candidates],
            "suppressed": [c.to_dict() for c in plan.suppressed_candidates],
        }
        return VPBDDetectionResult(list(plan.selected_candidates), plan, boundary_meta, lyrics_meta)

    # -- pool members ---------------------------------------------------------------------------------------
    def _build_acoustic_candidates(self, *, vocal_track, original_audio, pure_vocal_detector, feature_cache, vad_segments,
                                   enable_mdd: bool, include_breath_candidates: bool = True, device_state=None) -> List[CutCandidate]:
        kwargs = {}
        if device_state and hasattr(pure_vocal_detector, "_context"):
            kwargs = {"vocal_dev": device_state.get("vocal"), "original_dev": device_state.get("mix")}
        pauses = pure_vocal_detector.detect_pure_vocal_pauses(
            vocal_track, enable_mdd_enhancement=enable_mdd, original_audio=original_audio, feature_cache=feature_cache,
            vad_segments=vad_segments, include_breath_candidates=include_breath_candidates, **kwargs)
        scale = float(_section("vpbd").get("breath_score_scale", 0.6)) if include_breath_candidates else 0.0
        raw = []
        for p in pauses or []:
            t = float(getattr(p, "cut_point", (p.start_time + p.end_time) / 2.0))
            meta = {"pause_start_s": float(getattr(p, "start_time", t)), "pause_end_s": float(getattr(p, "end_time", t)),
                    "pause_duration_s": float(getattr(p, "duration", 0.0))}
            if include_breath_candidates:
                meta["pause_type"] = str(getattr(p, "pause_type", ""))
            raw.append((t, float(getattr(p, "confidence", 1.0)), meta))
        return adapt_legacy_acoustic_candidates(raw, source=CandidateSource.ACOUSTIC_PAUSE, breath_score_scale=scale)

    def _build_beat_candidates(self, *, vocal_track, feature_cache, duration_s: float, device_state=None) -> List[CutCandidate]:
        cfg = _section("vpbd").get("beat_candidates", {})
        if not isinstance(cfg, dict) or not bool(cfg.get("enable", False)) or feature_cache is None:
            return []
        base = float(cfg.get("base_score", 0.3))
        if base <= 0.0:
            return []
        st = device_state or {}
        return generate_beat_candidates(
            beat_times=getattr(feature_cache, "beat_times", []), rms_series=getattr(feature_cache, "rms_series", []),
            hop_s=float(getattr(feature_cache, "hop_s", 0.0) or 0.0), duration_s=duration_s, sample_rate=self.sample_rate,
            vocal_track=vocal_track, bars_per_cut=int(cfg.get("bars_per_cut", 2)), base_score=base,
            hip=st.get("hip"), vocal_dev=st.get("vocal"))

    def _merge_candidate_pool(self, *groups: List[CutCandidate], tolerance_s: float = 0.12) -> List[CutCandidate]:
        ordered = sorted((c for g in groups for c in g), key=lambda c: (c.t, c.source.value))
        if not ordered:
            return []
        clusters: List[List[CutCandidate]] = [[ordered[0]]]
        for c in ordered[1:]:
            if c.t - clusters[-1][-1].t <= tolerance_s:
                clusters[-1].append(c)
            else:
                clusters.append([c])
        return [self._merge_cluster(cl) for cl in clusters]

    def _merge_cluster(self, cluster: List[CutCandidate]) -> CutCandidate:
        best = max(cluster, key=lambda c: c.score)
        reasons: List[str] = []
        sources: List[str] = []
        source_scores: Dict[str, float] = {}
        members: List[Dict[str, Any]] = []
        for c in cluster:
            for r in c.reasons:
                if r not in reasons:
                    reasons.append(r)
            s = c.source.value
            if s not in sources:
                sources.append(s)
            source_scores[s] = max(float(c.score), source_scores.get(s, 0.0))
            members.append({"t": c.t, "score": c.score, "source": s, "reasons": list(c.reasons)})
        meta = dict(best.meta)
        meta.update({"sources": sources, "source_count": len(sources), "source_scores": source_scores})
        if len(cluster) > 1:
            meta["merged_candidates"] = members
        return replace(best, reasons=reasons, meta=meta)

    # -- scoring ----------------------------------------------------------------------------------------------
    def _score_candidates(self, *, candidates: List[CutCandidate], timeline, feature_cache: Optional[Any]) -> List[CutCandidate]:
        beat_times = getattr(feature_cache, "beat_times", []) if feature_cache is not None else []
        rms_series = getattr(feature_cache, "rms_series", []) if feature_cache is not None else []
        hop_s = float(getattr(feature_cache, "hop_s", 0.0) or 0.0) if feature_cache is not None else 0.0
        extractor = BoundaryFeatureExtractor(
            timeline=timeline, beat_times=beat_times, mdd_times=self._mdd_valley_times(feature_cache), rms_series=rms_series,
            hop_s=hop_s)
        scorer = PhraseBoundaryScorer.from_config(_section("phrase_boundary"))
        out: List[CutCandidate] = []
        for c in candidates:
            feats = extractor.extract(c.t, acoustic_pause=self._acoustic_pause_score(c))
            breath = self._breath_score(c)
            if breath > 0.0:
                feats = replace(feats, breath=breath)
            sc = scorer.score_candidate(c, feats)
            merged = dict(c.features)
            merged.update(sc.features)
            out.append(replace(sc, features=merged))
        return out

    @staticmethod
    def _acoustic_pause_score(c: CutCandidate) -> float:
        kinds = (CandidateSource.ACOUSTIC_PAUSE.value, CandidateSource.MDD_VALLEY.value)
        score = c.score if c.source.value in kinds else 0.0
        ss = c.meta.get("source_scores", {})
        if isinstance(ss, dict):
            for k in kinds:
                try:
                    score = max(score, float(ss.get(k, 0.0)))
                except (TypeError, ValueError):
                    continue
        return score

    @staticmethod
    def _breath_score(c: CutCandidate) -> float:
        score = c.score if c.source == CandidateSource.BREATH else 0.0
        ss = c.meta.get("source_scores", {})
        if isinstance(ss, dict):
            try:
                score = max(score, float(ss.get(CandidateSource.BREATH.value, 0.0)))
            except (TypeError, ValueError):
                pass
        return score

    @staticmethod
    def _mdd_valley_times(feature_cache: Optional[Any]) -> List[float]:
        """local minima of the cached MDD series at or below its 35th percentile (reference `:370-385`)."""
        if feature_cache is None:
            return []
        mdd = np.asarray(getattr(feature_cache, "mdd_series", []), dtype=np.float32)
        hop_s = float(getattr(feature_cache, "hop_s", 0.0) or 0.0)
        if mdd.size < 3 or hop_s <= 0.0:
            return []
        thr = float(np.percentile(mdd, 35))
        mid = mdd[1:-1].astype(np.float64)
        left = mdd[:-2].astype(np.float64); right = mdd[2:].astype(np.float64)
        hit = (mid <= thr) & ((mid < left) | (mid < right))
        return [(int(i) + 1) * hop_s for i in np.flatnonzero(hit)]


__all__ = ["VocalPhraseBoundaryDetector", "VPBDDetectionResult"]
