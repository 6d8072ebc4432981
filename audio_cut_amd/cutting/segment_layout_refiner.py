"""Segment layout policy after the guard — mirror of the reference plug point
`src/audio_cut/cutting/segment_layout_refiner.py` (`Segment`, `LayoutConfig`, `LayoutResult`, `derive_layout_config`,
`refine_layout`; same names, keyword names and defaults), SURVEY.md §8(f) row 1.

Host logic over at most a few dozen segments: merge fragments shorter than `micro_merge_s`, merge segments shorter than
`soft_min_s` into the cheaper neighbour, split segments longer than `soft_max_s` at a suppressed guard point or at the
quietest local minimum of the cached RMS, merge what the splits left too short, enforce `min_gap_s`, snap interior
boundaries to beats within `beat_snap_ms`.  Every pass is one sweep of `_sweep_short` with a pass-specific rule that
picks the neighbour; the reference's sweeps (`:136-313,497-545`) differ only in that rule.
Lyrics/ASR inputs are accepted for signature parity; this build has no lyrics provider, so they are normally empty.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from ..analysis.features_cache import TrackFeatureCache
from .refine import CutAdjustment, CutPoint

_INF = float("inf")


@dataclass
class Segment:
    start: float
    end: float
    kind: str = "human"

    @property
    def duration(self) -> float:
        return max(0.0, self.end - self.start)


@dataclass
class LayoutConfig:
    enable: bool = False
    micro_merge_s: float = 0.0
    soft_min_s: float = 0.0
    soft_max_s: float = 0.0
    min_gap_s: float = 1.0
    beat_snap_ms: float = 0.0


@dataclass
class LayoutResult:
    segments: List[Segment]
    adjustments: List[CutAdjustment]
    suppressed_points: List[CutPoint]


def derive_layout_config(raw_cfg: Optional[dict], features: Optional[TrackFeatureCache], *, sample_rate: float) -> LayoutConfig:
    """`:46-71`: static values only (BPM/MDD adaptation is reserved in the reference as well)."""
    raw = raw_cfg or {}
    val = lambda key, dflt: float(raw.get(key, dflt) or dflt)
    return LayoutConfig(enable=bool(raw.get("enable", False)), micro_merge_s=max(0.0, val("micro_merge_s", 0.0)),
                        soft_min_s=max(0.0, val("soft_min_s", 0.0)), soft_max_s=max(0.0, val("soft_max_s", 0.0)),
                        min_gap_s=max(0.0, val("min_gap_s", 1.0)), beat_snap_ms=max(0.0, val("beat_snap_ms", 0.0)))


def _copy(segments: Sequence[Segment]) -> List[Segment]:
    return [Segment(s.start, s.end, s.kind) for s in segments]


def _chain(segs: List[Segment]) -> None:
    for k in range(1, len(segs)):
        segs[k].start = segs[k - 1].end


# A side chooser gets (segment, left neighbour or None, right neighbour or None) and answers "left", "right" or None.
_Chooser = Callable[[Segment, Optional[Segment], Optional[Segment]], Optional[str]]


def _sweep_short(segments: List[Segment], limit_s: float, choose: _Chooser, *, stop_when_isolated: bool, rechain: bool,
                 keep_left_object: bool = False) -> List[Segment]:
    """Walk the list; a segment shorter than `limit_s` (and not a beat-aligned `_lib` one) is absorbed by the neighbour
    `choose` names.  Absorbing to the left steps back one position, to the right re-examines the merged segment."""
    if limit_s <= 0.0 or len(segments) <= 1:
        return segments
    segs = _copy(segments)
    k = 0
    while len(segs) > 1 and k < len(segs):
        cur = segs[k]
        if "_lib" in cur.kind or cur.duration >= limit_s:
            k += 1
            continue
        left = segs[k - 1] if k > 0 else None
        right = segs[k + 1] if k + 1 < len(segs) else None
        if left is None and right is None:
            if stop_when_isolated:
                break
            k += 1
            continue
        side = choose(cur, left, right)
        if side == "left" and left is not None:
            if keep_left_object:
                left.end = cur.end
            else:
                segs[k - 1] = Segment(left.start, cur.end, left.kind)
            del segs[k]
            k = max(k - 1, 0)
        elif side == "right" and right is not None:
            segs[k:k + 2] = [Segment(cur.start, right.end, right.kind)]
        else:
            k += 1
    if rechain:
        _chain(segs)
    return segs


def _choose_micro(soft_max_s: float) -> _Chooser:
    def choose(cur, left, right):
        if left is None or right is None:
            return "left" if left is not None else "right"
        span_l, span_r = cur.end - left.start, right.end - cur.start
        pen_l = span_l if (soft_max_s <= 0.0 or span_l <= soft_max_s) else _INF
        pen_r = span_r if (soft_max_s <= 0.0 or span_r <= soft_max_s) else _INF
        if pen_l <= pen_r:
            return "right" if (pen_l == _INF and pen_r != _INF) else "left"
        return "left" if (pen_r == _INF and pen_l != _INF) else "right"
    return choose


def _choose_soft_min(soft_max_s: float) -> _Chooser:
    def cost(nb, cur):
        if nb is None:
            return _INF, ""
        total = nb.duration + cur.duration
        over = _INF if (soft_max_s > 0.0 and total > soft_max_s) else total
        return over + (0.0 if nb.kind == cur.kind else total + 1.0), nb.kind

    def choose(cur, left, right):
        (cl, kl), (cr, _) = cost(left, cur), cost(right, cur)
        if cl == cr:
            return "left" if cur.kind == kl else "right"
        return "left" if cl < cr else "right"
    return choose


def _choose_post_split(micro_s: float, soft_max_s: float) -> _Chooser:
    def choose(cur, left, right):
        ranked = []
        for side, nb, span in (("left", left, None if left is None else cur.end - left.start),
                               ("right", right, None if right is None else right.end - cur.start)):
            if nb is None:
                continue
            pen = 0.0 if nb.kind == cur.kind else 10.0
            over = 0.0
            if soft_max_s > 0.0 and span > soft_max_s:
                over = span - soft_max_s
                if nb.kind != cur.kind or over > micro_s:
                    pen += 100.0 + over
            ranked.append(((pen, over, span), side))
        return min(ranked, key=lambda item: item[0])[1] if ranked else None
    return choose


def _choose_min_gap(cur, left, right):
    if left is None or right is None:
        return "left" if left is not None else "right"
    return "left" if (cur.end - left.start) <= (right.end - cur.start) else "right"


def _inside_interval(t: float, intervals: Sequence[Tuple[float, float]]) -> bool:
    for a, b in intervals:
        if a < t < b:
            return True
        if a >= t:
            break
    return False


def _asr_affinity(t: float, boundaries: Sequence[float], tol_s: float = 0.75) -> float:
    best = 0.0
    for bt in boundaries or ():
        d = abs(float(bt) - float(t))
        if d <= tol_s:
            best = max(best, 1.0 - d / max(tol_s, 1e-6))
    return best


def _candidate_score(pt: CutPoint, boundaries, words) -> float:
    t = float(pt.t)
    return float(getattr(pt, "score", 0.0) or 0.0) + 0.5 * _asr_affinity(t, boundaries) - (0.75 if _inside_interval(t, words) else 0.0)


def _find_acoustic_valley_split(seg: Segment, features: Optional[TrackFeatureCache], *, asr_boundary_times: Sequence[float],
                                asr_word_intervals: Sequence[Tuple[float, float]], min_gap_s: float) -> Optional[float]:
    """`:402-471`: the quietest qualifying local minimum of the cached RMS strictly inside the segment (min_gap margins)."""
    if features is None or features.frame_count() <= 2:
        return None
    lo_t = seg.start + max(0.0, min_gap_s)
    hi_t = seg.end - max(0.0, min_gap_s)
    if hi_t <= lo_t:
        return None
    sl = features.frame_slice(lo_t, hi_t)
    rms = np.asarray(features.rms_series[sl], dtype=np.float64)
    if rms.size < 3 or not np.all(np.isfinite(rms)):
        return None
    med = float(np.median(rms))
    spread = float(np.percentile(rms, 75) - np.percentile(rms, 5))
    if med <= 1e-12 or spread <= max(1e-9, med * 0.02):
        return None
    gate = min(float(np.percentile(rms, 25)), med * 0.75)
    first = int(sl.start or 0)
    pick_t: Optional[float] = None
    pick = -1.0
    for j in range(1, rms.size - 1):
        v = float(rms[j])
        if v > gate or v > float(rms[j - 1]) or v > float(rms[j + 1]):
            continue
        t = (first + j) * float(features.hop_s)
        if t <= lo_t or t >= hi_t:
            continue
        sc = max(0.0, (med - v) / max(med, 1e-12)) + 0.5 * _asr_affinity(t, asr_boundary_times) \
            - (0.75 if _inside_interval(t, asr_word_intervals) else 0.0)
        if sc > pick:
            pick, pick_t = sc, float(t)
    for bt in asr_boundary_times or ():
        t = float(bt)
        if t <= lo_t or t >= hi_t or _inside_interval(t, asr_word_intervals):
            continue
        j = int(round((t / float(features.hop_s)) - first))
        if j < 0 or j >= rms.size:
            continue
        v = float(np.min(rms[max(0, j - 2): min(rms.size, j + 3)]))
        if v > med:
            continue
        sc = max(0.0, (med - v) / max(med, 1e-12)) + 0.65
        if sc > pick:
            pick, pick_t = sc, t
    return pick_t if (pick_t is not None and pick >= 0.5) else None


def _split_long(segments: List[Segment], suppressed: List[CutPoint], soft_max_s: float, *, min_gap_s: float,
                features: Optional[TrackFeatureCache], asr_boundary_times, asr_word_intervals, allow_midpoint_fallback: bool):
    """`:316-386`: rescue cuts for segments longer than soft_max_s; a freshly cut segment is examined again."""
    if soft_max_s <= 0.0 or len(segments) <= 0:
        return segments, suppressed, []
    pool = list(suppressed)
    segs = _copy(segments)
    fresh: List[CutAdjustment] = []
    eps = 1e-3
    k = 0
    while k < len(segs):
        cur = segs[k]
        if cur.duration <= soft_max_s:
            k += 1
            continue
        inside = [p for p in pool if (cur.start + eps) < float(p.t) < (cur.end - eps)]
        if inside:
            chosen = max(inside, key=lambda p: _candidate_score(p, asr_boundary_times, asr_word_intervals))
            at = float(chosen.t)
            pool.remove(chosen)
        else:
            at = _find_acoustic_valley_split(cur, features, asr_boundary_times=asr_boundary_times,
                                             asr_word_intervals=asr_word_intervals, min_gap_s=min_gap_s)
            if at is None and allow_midpoint_fallback:
                at = cur.start + cur.duration / 2.0
        if at is None:
            k += 1
            continue
        d_left, d_right = at - cur.start, cur.end - at
        if d_left <= 0.0 or d_right <= 0.0 or (min_gap_s > 0.0 and (d_left < min_gap_s or d_right < min_gap_s)):
            k += 1
            continue
        segs[k:k + 1] = [Segment(cur.start, at, cur.kind), Segment(at, cur.end, cur.kind)]
        fresh.append(CutAdjustment(raw_time=at, guard_time=at, final_time=at, score=1.0, guard_shift_ms=0.0, final_shift_ms=0.0))
    _chain(segs)
    return segs, pool, fresh


def _snap_to_beats(segments: List[Segment], beat_snap_ms: float, *, features: Optional[TrackFeatureCache], min_gap_s: float) -> List[Segment]:
    """`:548-598`: interior boundaries move to the nearest beat within the window unless a side would drop under min_gap."""
    if beat_snap_ms <= 0.0 or features is None:
        return segments
    beats = getattr(features, "beat_times", None)
    if beats is None or len(beats) == 0:
        return segments
    reach = beat_snap_ms / 1000.0
    segs = _copy(segments)
    grid = np.asarray(beats, dtype=np.float64)
    for k in range(1, len(segs)):
        at = segs[k].start
        dist = np.abs(grid - at)
        j = int(np.argmin(dist))                      # first nearest beat, as the reference's strict `<` scan keeps
        if not dist[j] <= reach:
            continue
        near = float(grid[j])
        if (near - segs[k - 1].start) < min_gap_s or (segs[k].end - near) < min_gap_s:
            continue
        segs[k - 1].end = near
        segs[k].start = near
    _chain(segs)
    return segs


def _filter_adjustments(adjustments, segments: List[Segment], sample_rate: float, *, extra_adjustments=None) -> List[CutAdjustment]:
    """`:601-629`: keep the guard adjustments whose final time is still an interior boundary, add the rescue cuts."""
    if not adjustments:
        return []
    edges = ([segments[0].start] + [s.end for s in segments]) if segments else []
    interior = set(edges[1:-1])
    if not interior:
        return []
    tol = max(1.0 / max(sample_rate, 1.0), 1e-4)
    kept = [a for a in adjustments if any(abs(a.final_time - e) <= tol for e in interior)]
    for a in extra_adjustments or ():
        if not any(abs(a.final_time - b.final_time) <= tol for b in kept):
            kept.append(a)
    return kept


def refine_layout(segments: Iterable[Segment], adjustments: Iterable[CutAdjustment], *, config: LayoutConfig, sample_rate: float,
                  suppressed_cut_points: Optional[Iterable[CutPoint]] = None, features: Optional[TrackFeatureCache] = None,
                  asr_boundary_times: Optional[Iterable[float]] = None,
                  asr_word_intervals: Optional[Iterable[Tuple[float, float]]] = None,
                  allow_midpoint_fallback: bool = False) -> LayoutResult:
    """`:74-133`."""
    segs = _copy(list(segments))
    adjustments = list(adjustments or [])
    pool = list(suppressed_cut_points or [])
    boundaries = sorted(float(t) for t in (asr_boundary_times or []))
    words = sorted((float(a), float(b)) for a, b in (asr_word_intervals or []) if float(b) > float(a))
    if not config.enable or len(segs) <= 1:
        return LayoutResult(segs, adjustments, pool)
    segs = _sweep_short(segs, config.micro_merge_s, _choose_micro(config.soft_max_s), stop_when_isolated=True, rechain=False,
                        keep_left_object=True)
    segs = _sweep_short(segs, config.soft_min_s, _choose_soft_min(config.soft_max_s), stop_when_isolated=True, rechain=True)
    segs, pool, fresh = _split_long(segs, pool, config.soft_max_s, min_gap_s=config.min_gap_s, features=features,
                                    asr_boundary_times=boundaries, asr_word_intervals=words,
                                    allow_midpoint_fallback=allow_midpoint_fallback)
    segs = _sweep_short(segs, config.micro_merge_s, _choose_post_split(config.micro_merge_s, config.soft_max_s),
                        stop_when_isolated=False, rechain=True)
    segs = _sweep_short(segs, config.min_gap_s, _choose_min_gap, stop_when_isolated=False, rechain=True)
    segs = _snap_to_beats(segs, config.beat_snap_ms, features=features, min_gap_s=config.min_gap_s)
    return LayoutResult(segs, _filter_adjustments(adjustments, segs, sample_rate, extra_adjustments=fresh), pool)


__all__ = ["Segment", "LayoutConfig", "LayoutResult", "derive_layout_config", "refine_layout"]
