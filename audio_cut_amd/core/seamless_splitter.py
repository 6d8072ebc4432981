"""The hot-path slice of the reference's orchestrator: SURVEY.md §3.1 steps 2-9 of
`SeamlessSplitter._process_pure_vocal_split` (`src/vocal_smart_splitter/core/seamless_splitter.py:261-481`)
— separate -> feature cache -> pause detection -> pure-music spans + presence markers ->
`_finalize_and_filter_cuts_v2` -> integer `sample_boundaries` — for mode `v2.2_mdd` (and
`v2.1`, which only switches the MDD boost off, `:412`).

What follows in the reference (segment classification, layout refinement, local-valley refinement,
weak-tail merge, export; `:522-770`) is post-path policy and is out of scope this round
(SURVEY.md §8f "next" 1 and 4): `split_audio_seamlessly` returns the boundaries and the metadata
the manifest's `gpu` block needs, not exported files.

`_find_no_vocal_runs` (`:1706-1790`) and `_finalize_and_filter_cuts_v2` (`:1792-1879`) keep their
names and signatures; their RMS(2048/441) passes run on the stems resident in HBM.
"""
from __future__ import annotations

import logging
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _native
from ..analysis.features_cache import TrackFeatureCache, build_feature_cache
from ..config import get_config
from ..cutting.refine import CutContext, CutPoint, CutRefineResult, finalize_cut_points
from ..detectors.pure_vocal_pause_detector import PureVocalPauseDetector, _bool_runs
from .enhanced_vocal_separator import EnhancedVocalSeparator, SeparationResult

logger = logging.getLogger(__name__)


def _fill_false_runs(mask: np.ndarray, max_len: int) -> np.ndarray:
    out = np.asarray(mask, dtype=bool).copy()
    for a, b, v in _bool_runs(out):
        if not v and (b - a) <= max_len:
            out[a:b] = True
    return out


def _remove_true_runs(mask: np.ndarray, max_len: int) -> np.ndarray:
    out = np.asarray(mask, dtype=bool).copy()
    for a, b, v in _bool_runs(out):
        if v and (b - a) <= max_len:
            out[a:b] = False
    return out


class SeamlessSplitter:
    SUPPORTED_MODES = ("v2.2_mdd", "v2.1", "vpbd_acoustic", "vpbd_asr")

    def __init__(self, sample_rate: int = 44100, *, separator: Optional[EnhancedVocalSeparator] = None,
                 device: Optional[str] = None) -> None:
        self.sample_rate = sample_rate
        self.separator = separator if separator is not None else EnhancedVocalSeparator(sample_rate, device=device)
        backend = getattr(self.separator, "_primary_backend", None)
        self._hip: Optional["_native.Context"] = getattr(backend, "hip", None)
        self.pure_vocal_detector = PureVocalPauseDetector(sample_rate, ctx=self._hip)
        from .vocal_phrase_boundary_detector import VocalPhraseBoundaryDetector
        self.vpbd_detector = VocalPhraseBoundaryDetector(sample_rate)
        self._last_guard_adjustments_raw: list = []
        self._last_suppressed_cut_points: list = []

    def _context(self) -> "_native.Context":
        if self._hip is None:
            self._hip = _native.Context()
        return self._hip

    # ------------------------------------------------------------------------------------------
    def split_track(self, original_audio: np.ndarray, mode: str = "v2.2_mdd", *, audio_dev=None) -> Dict:
        """Steps 2-9 of SURVEY.md §3.1 on an in-memory mono float32 track at `sample_rate`."""
        if mode not in self.SUPPORTED_MODES:
            raise NotImplementedError(f"mode {mode!r}: only the v2.2_mdd / v2.1 path is built this round")
        sr = self.sample_rate
        t0 = time.perf_counter()
        sep: SeparationResult = self.separator.separate_for_detection(original_audio, gpu_context=None, audio_dev=audio_dev)
        t_sep = time.perf_counter() - t0
        state = sep.device_state or {}
        vocal_track = sep.vocal_track
        cache: Optional[TrackFeatureCache] = sep.feature_cache
        if cache is None:
            cache = build_feature_cache(original_audio, vocal_track, sr, ctx=self._context(), mix_dev=state.get("mix"))
        markers = sep.quality_metrics or {}
        marker_times = [float(t) for t in markers.get("vocal_presence_cut_points_sec", []) if t is not None]

        t1 = time.perf_counter()
        is_vpbd = mode in {"vpbd_acoustic", "vpbd_asr"}
        result: Dict = {"success": True, "mode": mode, "gpu_meta": dict(sep.gpu_meta or {}),
                        "separation_confidence": sep.separation_confidence, "backend_used": sep.backend_used,
                        "vad_segments": sep.vad_segments, "feature_cache": cache}
        if is_vpbd:
            # reference `:362-408`.  The smart_cut intent / AutoProfile runtime overrides applied at `:349` are
            # product configuration policy (out of scope): VPBD runs on the base configuration.
            vpbd = self.vpbd_detector.detect(mode=mode, vocal_track=vocal_track, original_audio=original_audio,
                                             pure_vocal_detector=self.pure_vocal_detector, feature_cache=cache,
                                             vad_segments=sep.vad_segments, device_state=state)
            t_det = time.perf_counter() - t1
            cut_candidates = [(c.t, c.score) for c in vpbd.selected_candidates]
            rescue = [(c.t, c.score) for c in vpbd.planner_result.suppressed_candidates if float(c.score) > 0.0]
            if not cut_candidates and rescue:
                cut_candidates = rescue
            result.update({"boundary_detection": vpbd.boundary_detection, "lyrics_alignment": vpbd.lyrics_alignment,
                           "vpbd_selected_times": [c.t for c in vpbd.selected_candidates], "num_pauses": len(cut_candidates)})
            if not cut_candidates:
                result.update({"sample_boundaries": [0, len(original_audio)], "note": "no_vpbd_candidates",
                               "timings": {"separate_s": t_sep, "detect_s": t_det, "finalize_s": 0.0}})
                return result
            t2 = time.perf_counter()
        else:
            pauses = self.pure_vocal_detector.detect_pure_vocal_pauses(
                vocal_track, enable_mdd_enhancement=(mode == "v2.2_mdd"), original_audio=original_audio, feature_cache=cache,
                vad_segments=sep.vad_segments, vocal_dev=state.get("vocal"), original_dev=state.get("mix"))
            t_det = time.perf_counter() - t1
            result.update({"num_pauses": len(pauses), "pauses": pauses})
            if not pauses:      # `:421-433`: single segment
                result.update({"sample_boundaries": [0, len(original_audio)], "note": "no_pause_candidates",
                               "timings": {"separate_s": t_sep, "detect_s": t_det, "finalize_s": 0.0}})
                return result
            t2 = time.perf_counter()
            cut_candidates = [(float(p.cut_point), float(p.confidence)) for p in pauses]
        min_pure_music = float(get_config("quality_control.pure_music_min_duration", 0.0))
        if min_pure_music > 0.0:
            for a, b in self._find_no_vocal_runs(vocal_track, min_pure_music, vocal_dev=state.get("vocal")):
                cut_candidates.append((float(a), 1.0))
                cut_candidates.append((float(b), 1.0))
        duration = len(original_audio) / sr
        protected = set()
        for t in marker_times:
            if t <= 0.0 or t >= duration:
                continue
            cut_candidates.append((t, 1.0))
            protected.add(int(round(t * sr)))
        refine = self._finalize_and_filter_cuts_v2(cut_candidates, original_audio, pure_vocal_audio=vocal_track,
                                                   mix_dev=state.get("mix"), vocal_dev=state.get("vocal"))
        self._last_suppressed_cut_points = list(refine.suppressed_points or [])
        bounds = sorted(set(refine.sample_boundaries))
        if protected:       # `:501-508`
            total = len(original_audio)
            aug = set(int(b) for b in bounds)
            for s in protected:
                s = int(min(max(s, 0), total))
                if s not in (0, total):
                    aug.add(s)
            bounds = sorted(aug)
        t_fin = time.perf_counter() - t2
        result.update({"sample_boundaries": bounds, "refine_boundaries": list(refine.sample_boundaries),
                       "cut_candidates": cut_candidates,
                       "guard_adjustments": list(refine.adjustments or []),
                       "timings": {"separate_s": t_sep, "detect_s": t_det, "finalize_s": t_fin}})
        return result

    # ------------------------------------------------------------------------------------------
    def _rms2048_db(self, wave: np.ndarray, dev=None) -> np.ndarray:
        hip = self._context()
        x = dev if dev is not None else hip.to_device(np.ascontiguousarray(wave, dtype=np.float32))
        hop = max(1, int(0.01 * self.sample_rate))
        rms = hip.frame_rms(x, 2048, hop).cpu().numpy()
        return rms, 20.0 * np.log10(rms + 1e-12)

    def _find_no_vocal_runs(self, vocal_audio: np.ndarray, min_duration: float, *, vocal_dev=None):
        """`seamless_splitter.py:1706-1790`."""
        sr = self.sample_rate
        hop = max(1, int(0.01 * sr))
        rms, db = self._rms2048_db(vocal_audio, vocal_dev)
        noise_pct = float(get_config("quality_control.enforce_quiet_cut.floor_percentile", 10))
        voice_pct = float(get_config("pure_vocal_detection.pause_stats_adaptation.voice_percentile_hint", 90))
        noise_db = float(np.percentile(db, np.clip(noise_pct, 0, 50)))
        voice_db = float(np.percentile(db, np.clip(voice_pct, 50, 100)))
        delta_db = float(get_config("pure_vocal_detection.pause_stats_adaptation.delta_db", 3.0))
        thr_db = max(noise_db + delta_db, 0.5 * (noise_db + voice_db))
        active = db > thr_db
        frame_sec = hop / float(sr)
        close_k = max(1, int(int(get_config("pure_vocal_detection.pause_stats_adaptation.morph_close_ms", 150)) / 1000.0 / frame_sec))
        open_k = max(1, int(int(get_config("pure_vocal_detection.pause_stats_adaptation.morph_open_ms", 50)) / 1000.0 / frame_sec))
        inactive = ~_remove_true_runs(_fill_false_runs(active, close_k), open_k)
        times = (np.arange(len(rms)) * hop).astype(int) / float(sr)
        n = len(vocal_audio) if vocal_audio is not None else int(vocal_dev.numel())
        spans = []
        for a, b, v in _bool_runs(inactive):
            if not v:
                continue
            st = float(times[a])
            en = float(times[b]) if b < len(inactive) else float(n / float(sr))
            if en - st >= float(min_duration):
                spans.append((st, en))
        return spans

    def _finalize_and_filter_cuts_v2(self, cut_candidates, audio_for_split: np.ndarray,
                                     pure_vocal_audio: Optional[np.ndarray] = None, *, mix_dev=None, vocal_dev=None) -> CutRefineResult:
        """`seamless_splitter.py:1792-1879` (quirk Q1: floor_percentile 0.5 is read as a fraction)."""
        sr = self.sample_rate
        if sr <= 0 or audio_for_split.size == 0:
            return CutRefineResult([], [0, len(audio_for_split)], [])
        points: List[CutPoint] = []
        if isinstance(cut_candidates, list) and cut_candidates:
            first = cut_candidates[0]
            if isinstance(first, tuple) and len(first) >= 2:
                points = [CutPoint(t=float(c[0]), score=float(c[1])) for c in cut_candidates]
            elif isinstance(first, int):
                points = [CutPoint(t=float(s) / float(sr), score=1.0) for s in cut_candidates]
            else:
                points = [CutPoint(t=float(t), score=1.0) for t in cut_candidates]
        if not points:
            return CutRefineResult([], [0, len(audio_for_split)], [])
        min_gap_s = float(get_config("quality_control.min_split_gap", 1.0))
        try:
            max_keep = int(get_config("pure_vocal_detection.valley_scoring.max_kept_after_nms", 150))
        except Exception:
            max_keep = None
        guard_enabled = bool(get_config("quality_control.enforce_quiet_cut.enable", False))
        guard_db = float(get_config("quality_control.enforce_quiet_cut.guard_db", 2.5))
        search_right_ms = float(get_config("quality_control.enforce_quiet_cut.search_right_ms", 150))
        guard_win_ms = float(get_config("quality_control.enforce_quiet_cut.win_ms", 80))
        floor_db = -60.0
        if guard_enabled:
            override = get_config("quality_control.enforce_quiet_cut.floor_db_override", None)
            if override is not None:
                floor_db = float(override)
            else:
                try:
                    cfg = get_config("quality_control.enforce_quiet_cut.floor_percentile", 5)
                    pct = float(cfg) / 100.0 if float(cfg) > 1 else float(cfg)
                except Exception:
                    pct = 0.05
                mono = audio_for_split if audio_for_split.ndim == 1 else np.mean(audio_for_split, axis=0)
                if mono.size > 0:
                    _, rms_db = self._rms2048_db(mono, mix_dev if audio_for_split.ndim == 1 else None)
                    floor_db = float(np.percentile(rms_db, max(0.0, min(100.0, pct * 100.0))))
        ctx = CutContext(sr=sr, mix_wave=audio_for_split, vocal_wave=pure_vocal_audio, mix_dev=mix_dev, vocal_dev=vocal_dev,
                         hip=self._context())
        use_vocal_guard = pure_vocal_audio is not None
        topk_cfg = get_config("quality_control.nms_topk_per_10s", None)
        result = finalize_cut_points(
            ctx, points, use_vocal_guard_first=use_vocal_guard, min_gap_s=min_gap_s, max_keep=max_keep,
            topk_per_10s=int(topk_cfg) if topk_cfg is not None else None,
            nms_window_s=float(get_config("quality_control.nms_window_s", 10.0)), guard_db=guard_db,
            search_right_ms=search_right_ms, guard_win_ms=guard_win_ms, floor_db=floor_db,
            enable_mix_guard=guard_enabled, enable_vocal_guard=(guard_enabled and use_vocal_guard))
        self._last_guard_adjustments_raw = list(result.adjustments or [])
        bounds = sorted(set(result.sample_boundaries or [0, len(audio_for_split)]))
        return CutRefineResult(result.final_points, bounds, list(result.adjustments or []), result.suppressed_points)


__all__ = ["SeamlessSplitter"]
