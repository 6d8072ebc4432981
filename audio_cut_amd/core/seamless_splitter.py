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


PRECISION_GUARD_AVG_MS = 150.0     # reference `seamless_splitter.py:66-67`
PRECISION_GUARD_P95_MS = 220.0


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
    def split_track(self, original_audio: np.ndarray, mode: str = "v2.2_mdd", *, audio_dev=None, separation_gate=None,
                    unet_stream=None) -> Dict:
        """Steps 2-9 of SURVEY.md §3.1 on an in-memory mono float32 track at `sample_rate`.
        `separation_gate` (a lock shared by the workers of a `batch.TrackPipeline`) and `unet_stream` (the pipeline's one U-Net
        stream): with both, this track's separation is queued on that stream under the lock and the lock is released as soon as it is
        queued - the stream itself keeps the U-Nets of consecutive tracks one after the other, and the next one waits in the queue
        behind the running one.  With the gate alone (round 2's scheme) the lock is held until this track's U-Net has left the GPU and
        released before the host-bound tail."""
        if mode not in self.SUPPORTED_MODES:
            raise NotImplementedError(f"mode {mode!r}: only the v2.2_mdd / v2.1 path is built this round")
        sr = self.sample_rate
        if original_audio is None or len(original_audio) == 0:
            raise ValueError("split_track needs a non-empty mono track")
        t0 = time.perf_counter()
        sep: SeparationResult = self.separator.separate_for_detection(original_audio, gpu_context=None, audio_dev=audio_dev,
                                                                     separation_gate=separation_gate, unet_stream=unet_stream)
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
                        "vad_segments": sep.vad_segments, "feature_cache": cache,
                        "vocal_track": vocal_track, "instrumental_track": sep.instrumental_track, "device_state": state}
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
                result.update(self._single_segment_fields(vocal_track, len(original_audio), state.get("vocal")))
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
                result.update(self._single_segment_fields(vocal_track, len(original_audio), state.get("vocal")))
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
        if is_vpbd:         # `:494-499`: the planner block records where the guards moved each selected candidate
            from ..cutting.global_cut_planner import apply_guard_shift_metadata
            vpbd.boundary_detection["planner"] = dict(apply_guard_shift_metadata(vpbd.planner_result, self._last_guard_adjustments_raw).metadata)
        if protected:       # `:501-508`
            total = len(original_audio)
            aug = set(int(b) for b in bounds)
            for s in protected:
                s = int(min(max(s, 0), total))
                if s not in (0, total):
                    aug.add(s)
            bounds = sorted(aug)
        t_fin = time.perf_counter() - t2
        t3 = time.perf_counter()
        policy = self._apply_boundary_policy(bounds, vocal_track, len(original_audio), cache, vocal_dev=state.get("vocal"))
        result.update(policy)
        result["timings_policy_s"] = time.perf_counter() - t3
        kept = list(self._last_guard_adjustments_raw)               # after the layout refiner's filter (`:602`)
        stats = self._guard_shift_stats(kept)
        result.update({"sample_boundaries": bounds, "refine_boundaries": list(refine.sample_boundaries),
                       "cut_candidates": cut_candidates,
                       "guard_adjustments": kept, "guard_adjustments_unfiltered": list(refine.adjustments or []),
                       "guard_shift_stats": stats,
                       "precision_guard_ok": bool(stats["avg_shift_ms"] <= PRECISION_GUARD_AVG_MS and stats["p95_shift_ms"] <= PRECISION_GUARD_P95_MS),
                       "precision_guard_threshold_ms": {"avg": PRECISION_GUARD_AVG_MS, "p95": PRECISION_GUARD_P95_MS},
                       "timings": {"separate_s": t_sep, "detect_s": t_det, "finalize_s": t_fin}})
        return result

    @staticmethod
    def _guard_shift_stats(adjustments: Sequence) -> Dict[str, float]:
        """`_set_guard_adjustments` (`:2423-2470`): how far the quiet guards moved the kept cuts, in ms."""
        total = np.array([a.final_shift_ms for a in adjustments], dtype=float)
        if total.size == 0:
            return {"avg_shift_ms": 0.0, "max_shift_ms": 0.0, "avg_guard_only_shift_ms": 0.0, "avg_vocal_guard_shift_ms": 0.0,
                    "avg_mix_guard_shift_ms": 0.0, "p95_shift_ms": 0.0, "count": 0}
        vocal = np.array([a.guard_shift_ms for a in adjustments], dtype=float)
        mean_pos = lambda v: float(sum(x for x in v.tolist() if x > 0) / max(1, int((v > 0).sum()))) if (v > 0).any() else 0.0
        return {"avg_shift_ms": float(sum(abs(x) for x in total.tolist()) / total.size), "max_shift_ms": float(np.abs(total).max()),
                "avg_guard_only_shift_ms": mean_pos(total), "avg_vocal_guard_shift_ms": mean_pos(vocal),
                "avg_mix_guard_shift_ms": mean_pos(total - vocal), "p95_shift_ms": float(np.percentile(np.abs(total), 95.0)),
                "count": int(total.size)}

    # ------------------------------------------------------------------------------------------
    def _single_segment_fields(self, vocal_track: np.ndarray, n_samples: int, vocal_dev=None) -> Dict:
        """`_create_single_segment_result` (`:2682-2747`): one segment labelled by `_estimate_vocal_presence` (`:2404-2410`)."""
        self._last_guard_adjustments_raw = []
        self._last_suppressed_cut_points = []
        flags = self._classify_segments_vocal_presence(vocal_track, [0, len(vocal_track)], vocal_dev=vocal_dev) \
            if (vocal_track is not None and getattr(vocal_track, "size", 0)) else []
        has_vocal = bool(flags[0]) if flags else False
        sr = float(self.sample_rate)
        return {"cuts_samples": [0, int(n_samples)], "cuts_sec": [0.0, n_samples / sr], "segment_vocal_flags": [has_vocal],
                "segment_spans": [(0, int(n_samples))], "segment_durations": [n_samples / sr], "segment_layout_applied": False,
                "suppressed_cut_points_sec": [], "single_segment": True}

    # ---- post-path boundary policy (reference `seamless_splitter.py:521-669`) -----------
    def _vocal_on_device(self, vocal_audio: np.ndarray, vocal_dev=None):
        return vocal_dev if vocal_dev is not None else self._context().to_device(np.ascontiguousarray(vocal_audio, dtype=np.float32))

    def _classify_segments_vocal_presence(self, vocal_audio: np.ndarray, cut_points: Sequence[int], marker_segments=None,
                                          pure_music_segments=None, instrumental_audio=None, original_audio=None, *,
                                          vocal_dev=None) -> List[bool]:
        """`:2276-2403`: a segment is `human` when at least `segment_vocal_activity_ratio` of its 50 ms / 20 ms RMS frames
        exceed `segment_vocal_threshold_db`.  All segments' frames come from ONE `ac_segment_frame_rms` launch."""
        n_seg = max(len(cut_points) - 1, 0)
        self._last_segment_classification_debug = []
        if n_seg == 0:
            return []
        sr = self.sample_rate
        if sr <= 0 or vocal_audio is None or getattr(vocal_audio, "size", 0) == 0:
            self._last_segment_classification_debug = [{"index": i, "reason": "fallback_invalid_input", "decision": True} for i in range(n_seg)]
            return [True] * n_seg
        ratio_thr = float(get_config("quality_control.segment_vocal_activity_ratio", 0.10))
        thr_db = float(get_config("quality_control.segment_vocal_threshold_db", -50.0))
        hop = max(1, int(0.02 * sr))
        frame_length = max(hop * 2, int(0.05 * sr))
        n = len(vocal_audio)
        a = np.clip(np.asarray(cut_points[:-1], dtype=np.int64), 0, n)
        b = np.maximum(a, np.clip(np.asarray(cut_points[1:], dtype=np.int64), 0, n))
        hip = self._context()
        # segments already measured in this policy pass (the three classification rounds of `_apply_boundary_policy` mostly
        # see the same segments) come from the cache; the rest go to the GPU in two launches
        cache = getattr(self, "_segment_measure_cache", None)
        key = [(int(a[i]), int(b[i])) for i in range(n_seg)]
        need = [i for i in range(n_seg) if b[i] > a[i] and (cache is None or key[i] not in cache)]
        measured = {}
        if need:
            dev = self._vocal_on_device(vocal_audio, vocal_dev)
            nd = np.asarray(need, dtype=np.int64)
            ss = hip.segment_sumsq_peak(dev, a[nd], b[nd])[0]
            long_ix = nd[(b[nd] - a[nd]) >= frame_length]
            fr = dict(zip(long_ix.tolist(), hip.segment_frame_rms(dev, a[long_ix], b[long_ix], frame_length, hop))) if long_ix.size else {}
            for j, i in enumerate(need):
                measured[key[i]] = (float(ss[j]), fr.get(i))
            if cache is not None:
                cache.update(measured)
        look = (lambda k: cache.get(k)) if cache is not None else (lambda k: measured.get(k))
        frames = {i: look(key[i])[1] for i in range(n_seg) if b[i] > a[i] and look(key[i])[1] is not None}
        sumsq = {i: look(key[i])[0] for i in range(n_seg) if b[i] > a[i]}
        flags: List[bool] = []
        debug: List[Dict] = []
        for i in range(n_seg):
            t0, t1 = int(a[i]) / sr, int(b[i]) / sr
            dur = max(t1 - t0, 1e-6)
            size = int(b[i] - a[i])
            ratio = seconds = 0.0
            rms_db = None
            if i in sumsq:          # np.sqrt(np.mean(np.square(seg)) + 1e-12) of the float32 segment
                rms_db = 20.0 * np.log10(float(np.sqrt(np.float32(sumsq[i] / size) + np.float32(1e-12))))
            if i in frames:
                active = (20.0 * np.log10(frames[i] + 1e-12)) > thr_db
                if active.size > 0:
                    ratio = float(np.mean(active))
                    seconds = float(min(dur, float(active.sum()) * (hop / sr)))
            elif size > 0 and rms_db > thr_db:
                ratio, seconds = 1.0, dur
            decision = ratio >= ratio_thr
            why = "vocal_activity_ratio_gte_threshold" if decision else "vocal_activity_ratio_lt_threshold"
            debug.append({"index": i, "start_s": t0, "end_s": t1, "duration_s": dur, "vocal_activity_ratio": ratio,
                          "vocal_activity_seconds": seconds, "activity_ratio_threshold": ratio_thr, "activity_threshold_db": thr_db,
                          "rms_db": rms_db, "decision": decision, "decision_reason": why, "reason": why,
                          "decision_threshold_db": thr_db, "threshold_source": "vocal_activity_ratio"})
            flags.append(bool(decision))
        self._last_segment_classification_debug = debug
        return flags

    def _refine_boundaries_local_valley(self, sample_boundaries: List[int], vocal_audio: np.ndarray, cfg: Dict, *, min_gap_s: float,
                                        protected_intervals_s=None, vocal_dev=None) -> List[int]:
        """`:2613-2680`: every interior boundary may move to the quietest 5 ms of its +-radius neighbourhood when that is at
        least `min_drop_db` quieter; all neighbourhoods are searched in ONE `ac_local_valley` launch on the boundaries as
        they stand, then the moves are accepted left to right (a move only tightens its neighbours' gap checks, which use
        the already-updated left boundary exactly as the reference's in-place loop does)."""
        if vocal_audio is None or vocal_audio.size == 0 or len(sample_boundaries) <= 2:
            return sample_boundaries
        sr = float(self.sample_rate)
        radius = max(1, int(float(cfg.get("search_radius_ms", 200)) / 1000.0 * sr))
        win = max(1, int(float(cfg.get("window_ms", 20)) / 1000.0 * sr))
        drop_db = float(cfg.get("min_drop_db", 3.0))
        micro = float(get_config("segment_layout.micro_merge_s", 0.0) or 0.0)
        piece = float(get_config("quality_control.segment_min_mix_piece", 0.0) or 0.0)
        min_seg = max(1, int(max(float(min_gap_s), micro, piece) * sr))
        protected = sorted((float(x), float(y)) for x, y in (protected_intervals_s or []) if float(y) > float(x))
        refined = list(sample_boundaries)
        centers = np.asarray(refined[1:-1], dtype=np.int64)
        dev = self._vocal_on_device(vocal_audio, vocal_dev)
        orig_db, min_db, min_idx = self._context().local_valley(dev, centers, radius, win)
        for k, idx in enumerate(range(1, len(refined) - 1)):
            if min_idx[k] < 0 or (orig_db[k] - min_db[k]) < drop_db:
                continue
            center = refined[idx]
            cand = max(0, center - radius) + int(min_idx[k]) + win // 2
            if any(x < cand / sr < y for x, y in protected):
                continue
            if cand <= refined[idx - 1] + min_seg or cand >= refined[idx + 1] - min_seg:
                continue
            refined[idx] = cand
        return refined

    def _merge_short_weak_human_tails_into_following_music(self, cut_points: List[int], segment_vocal_flags: List[bool],
                                                           debug_entries: List[Dict], vocal_audio: np.ndarray, *,
                                                           min_duration_s: float, layout_applied: bool, vocal_dev=None):
        """`:2145-2275`: a human segment shorter than `soft_min_s` whose RMS and peak are under 12 % / 18 % of the median
        long human segment, followed by music, becomes part of that music.  Segment energies come from one
        `ac_segment_sumsq_peak` launch; a merged pair's statistics are the sums / max of its parts."""
        debug = [dict(e) for e in (debug_entries or [])]
        if (not layout_applied or min_duration_s <= 0.0 or len(cut_points) < 3 or len(segment_vocal_flags) != len(cut_points) - 1
                or vocal_audio is None or getattr(vocal_audio, "size", 0) == 0 or self.sample_rate <= 0):
            return list(cut_points), list(segment_vocal_flags), debug
        sr = float(self.sample_rate)
        n = len(vocal_audio)
        pts = [int(p) for p in cut_points]
        flags = [bool(f) for f in segment_vocal_flags]
        while len(debug) < len(flags):
            debug.append({})
        a = np.clip(np.asarray(pts[:-1], dtype=np.int64), 0, n)
        b = np.maximum(a, np.clip(np.asarray(pts[1:], dtype=np.int64), 0, n))
        live = np.flatnonzero(b > a)
        ss = np.zeros(len(a)); pk = np.zeros(len(a))
        if live.size:
            s_live, p_live = self._context().segment_sumsq_peak(self._vocal_on_device(vocal_audio, vocal_dev), a[live], b[live])
            ss[live] = s_live; pk[live] = p_live
        size = (b - a).astype(np.float64)
        seg = [{"ss": float(ss[i]), "pk": float(pk[i]), "n": float(size[i])} for i in range(len(a))]

        def stat(i):
            cnt = seg[i]["n"]
            rms = float(np.sqrt(seg[i]["ss"] / cnt + 1e-12)) if cnt > 0 else 0.0
            return max(0.0, (pts[i + 1] - pts[i]) / sr), rms, (seg[i]["pk"] if cnt > 0 else 0.0)

        st = [stat(i) for i in range(len(flags))]
        ref_r = [r for (d, r, p), f in zip(st, flags) if f and d >= min_duration_s and r > 0.0]
        ref_p = [p for (d, r, p), f in zip(st, flags) if f and d >= min_duration_s and p > 0.0]
        if not ref_r or not ref_p:
            return pts, flags, debug[:len(flags)]
        rr = float(np.median(np.asarray(ref_r, dtype=np.float64)))
        rp = float(np.median(np.asarray(ref_p, dtype=np.float64)))
        w_r = float(get_config("quality_control.short_human_tail_rms_ratio", 0.12) or 0.12)
        w_p = float(get_config("quality_control.short_human_tail_peak_ratio", 0.18) or 0.18)
        why = "merged_short_weak_human_tail_into_following_music"
        i = 0
        while i < len(flags) - 1:
            d, r, p = stat(i)
            if not (flags[i] and not flags[i + 1] and d < min_duration_s and r <= rr * w_r and p <= rp * w_p):
                i += 1
                continue
            t0, t1 = pts[i] / sr, pts[i + 2] / sr
            pts.pop(i + 1)
            seg[i:i + 2] = [{"ss": seg[i]["ss"] + seg[i + 1]["ss"], "pk": max(seg[i]["pk"], seg[i + 1]["pk"]), "n": seg[i]["n"] + seg[i + 1]["n"]}]
            left, right = debug[i], debug[i + 1]
            merged = dict(right or left or {})
            span = max(t1 - t0, 1e-6)
            voiced = min(span, float((left or {}).get("vocal_activity_seconds", 0.0) or 0.0) + float((right or {}).get("vocal_activity_seconds", 0.0) or 0.0))
            origin = []
            for e in (left, right):
                if e:
                    origin.extend(e.get("merged_from_segments", [e.get("index")]))
            merged.update({"index": i, "start_s": t0, "end_s": t1, "duration_s": span, "vocal_activity_seconds": voiced,
                           "vocal_activity_ratio": voiced / span, "decision": False, "decision_reason": why, "reason": why,
                           "merged_from_segments": sorted({int(x) for x in origin if x is not None})})
            flags[i:i + 2] = [False]
            debug[i:i + 2] = [merged]
        for k, e in enumerate(debug[:len(flags)]):
            e["index"] = k
        return pts, flags, debug[:len(flags)]

    def _split_at_sample_level(self, audio: np.ndarray, final_cut_points: List[int], *, segment_flags: Optional[List[bool]] = None,
                               debug_entries: Optional[List[Dict]] = None):
        """`:2006-2144`: slices between consecutive cut points; a slice shorter than 10 ms is glued to the next one (a
        trailing one to the previous).  Slices are views of `audio` unless a merge forces a copy."""
        keep = max(1, int(0.01 * self.sample_rate))
        n = len(audio)
        spans: List[List[int]] = []
        flags: Optional[List[bool]] = [] if segment_flags is not None else None
        pending: Optional[List[int]] = None
        pending_flag: Optional[bool] = None
        for i in range(len(final_cut_points) - 1):
            lo = max(0, min(int(final_cut_points[i]), n)); hi = max(lo, min(int(final_cut_points[i + 1]), n))
            span = [lo, hi] if hi > lo else None
            flag = bool(segment_flags[i]) if (segment_flags is not None and i < len(segment_flags)) else True
            if pending is not None:
                span = [pending[0], span[1]] if span is not None else list(pending)
                flag = bool(pending_flag) or flag
                pending, pending_flag = None, None
            if int(final_cut_points[i + 1]) - int(final_cut_points[i]) >= keep and span is not None:
                spans.append(span)
                if flags is not None:
                    flags.append(flag)
            elif span is not None:
                pending, pending_flag = span, flag
        if pending is not None:
            if spans:
                spans[-1][1] = pending[1]
                if flags is not None:
                    flags[-1] = bool(flags[-1]) or bool(pending_flag)
            else:
                spans.append(pending)
                if flags is not None:
                    flags.append(bool(pending_flag))
        self._last_segment_spans = [tuple(sp) for sp in spans]
        return [audio[lo:hi] for lo, hi in spans], flags, None

    def _apply_boundary_policy(self, bounds: List[int], vocal_track: np.ndarray, n_samples: int,
                               cache: Optional[TrackFeatureCache], *, vocal_dev=None) -> Dict:
        """`:521-669` for the modes without lyrics alignment: classify -> layout refiner -> classify -> local valley ->
        classify -> weak-tail merge -> sample-level split.  Returns the manifest-facing fields."""
        from ..cutting.segment_layout_refiner import Segment as LayoutSegment, derive_layout_config, refine_layout
        sr = self.sample_rate
        cuts = sorted(set(int(c) for c in bounds))
        self._segment_measure_cache = {}
        try:
            return self._boundary_policy_steps(cuts, vocal_track, n_samples, cache, vocal_dev)
        finally:
            self._segment_measure_cache = None

    def _boundary_policy_steps(self, cuts: List[int], vocal_track: np.ndarray, n_samples: int, cache, vocal_dev) -> Dict:
        from ..cutting.segment_layout_refiner import Segment as LayoutSegment, derive_layout_config, refine_layout
        sr = self.sample_rate
        flags = self._classify_segments_vocal_presence(vocal_track, cuts, vocal_dev=vocal_dev)
        raw = dict(get_config("segment_layout", {}) or {})
        micro = get_config("quality_control.segment_min_mix_piece", None)
        if micro is not None:
            raw.setdefault("micro_merge_s", float(micro)); raw.setdefault("enable", bool(float(micro) > 0.0))
        smax = get_config("quality_control.segment_max_duration", None)
        if smax is not None:
            raw.setdefault("soft_max_s", float(smax))
        raw.setdefault("min_gap_s", float(get_config("quality_control.min_split_gap", 1.0)))
        raw.setdefault("beat_snap_ms", float(get_config("segment_layout.beat_snap_ms", 0.0) or 0.0))
        lcfg = derive_layout_config(raw, cache, sample_rate=sr)
        applied = False
        if lcfg.enable and len(cuts) >= 2:
            edges = [c / float(sr) for c in cuts]
            res = refine_layout([LayoutSegment(edges[i], edges[i + 1], "human" if flags[i] else "music") for i in range(len(edges) - 1)],
                                self._last_guard_adjustments_raw, config=lcfg, sample_rate=sr,
                                suppressed_cut_points=self._last_suppressed_cut_points, features=cache)
            if res.segments:
                times = [res.segments[0].start] + [sg.end for sg in res.segments]
                upd = [max(0, min(int(round(t * sr)), n_samples)) for t in times]
                if upd:
                    upd[0] = 0; upd[-1] = n_samples
                upd = sorted(set(upd))
                if upd != cuts:
                    applied = True
                cuts = upd if upd else cuts
                self._last_guard_adjustments_raw = list(res.adjustments)
                self._last_suppressed_cut_points = list(res.suppressed_points or [])
                flags = self._classify_segments_vocal_presence(vocal_track, cuts, vocal_dev=vocal_dev)
        local = get_config("quality_control.local_boundary_refine", {}) or {}
        if local.get("enable") and len(cuts) >= 2:
            ref = self._refine_boundaries_local_valley(cuts, vocal_track, local, min_gap_s=float(get_config("quality_control.min_split_gap", 1.0)),
                                                       vocal_dev=vocal_dev)
            if list(ref) != cuts:
                cuts = list(ref); applied = True
                flags = self._classify_segments_vocal_presence(vocal_track, cuts, vocal_dev=vocal_dev)
        c2, f2, dbg = self._merge_short_weak_human_tails_into_following_music(
            cuts, flags, list(getattr(self, "_last_segment_classification_debug", [])), vocal_track,
            min_duration_s=float(getattr(lcfg, "soft_min_s", 0.0) or 0.0), layout_applied=applied, vocal_dev=vocal_dev)
        if list(c2) != cuts:
            cuts, flags, applied = list(c2), list(f2), True
            self._last_segment_classification_debug = dbg
        _, merged_flags, _ = self._split_at_sample_level(np.empty(n_samples, dtype=np.int8), cuts, segment_flags=flags)
        spans = list(self._last_segment_spans)
        return {"cuts_samples": list(cuts), "cuts_sec": [c / float(sr) for c in cuts], "segment_vocal_flags": list(merged_flags or []),
                "segment_spans": spans, "segment_durations": [(hi - lo) / float(sr) for lo, hi in spans],
                "segment_layout_applied": bool(applied),
                "suppressed_cut_points_sec": [float(c.t) for c in self._last_suppressed_cut_points]}

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
            from ..analysis.prefetch import guard_floor_db
            if get_config("quality_control.enforce_quiet_cut.floor_db_override", None) is not None:
                floor_db = guard_floor_db(np.zeros(0))
            else:
                mono = audio_for_split if audio_for_split.ndim == 1 else np.mean(audio_for_split, axis=0)
                if mono.size > 0:
                    _, rms_db = self._rms2048_db(mono, mix_dev if audio_for_split.ndim == 1 else None)
                    floor_db = guard_floor_db(rms_db)
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
