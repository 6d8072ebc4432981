"""Oracle for SURVEY.md §8(f) row 1: the post-path boundary policy that still moves integer cut samples.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, in order of application
(`src/vocal_smart_splitter/core/seamless_splitter.py:521-669`):

* segment human/music classification by vocal activity            `seamless_splitter.py:2276-2403`
* the layout refiner (micro merge, soft-min merge, soft-max rescue split, post-split micro merge, min-gap,
  beat snap)                                                      `src/audio_cut/cutting/segment_layout_refiner.py:74-636`
* local-valley boundary refinement                                `seamless_splitter.py:2613-2680`
* short weak human tail merged into the following music           `seamless_splitter.py:2145-2275`
* sample-level split with sub-10 ms carry                         `seamless_splitter.py:2006-2144`

Segments are `[start_s, end_s, kind]` lists.  The lyrics/ASR inputs of the reference are absent on this path
(`vpbd_asr` is out of scope), so their terms are structurally zero and omitted.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import librosa_ops as L
from .config import get_config

INF = float("inf")


@dataclass
class LayoutConfig:
    enable: bool = False
    micro_merge_s: float = 0.0
    soft_min_s: float = 0.0
    soft_max_s: float = 0.0
    min_gap_s: float = 1.0
    beat_snap_ms: float = 0.0


def layout_config_from_settings() -> LayoutConfig:
    """`seamless_splitter.py:533-550` + `derive_layout_config` (`segment_layout_refiner.py:46-71`)."""
    raw = dict(get_config("segment_layout", {}) or {})
    micro = get_config("quality_control.segment_min_mix_piece", None)
    if micro is not None:
        raw.setdefault("micro_merge_s", float(micro))
        raw.setdefault("enable", bool(float(micro) > 0.0))
    smax = get_config("quality_control.segment_max_duration", None)
    if smax is not None:
        raw.setdefault("soft_max_s", float(smax))
    raw.setdefault("min_gap_s", float(get_config("quality_control.min_split_gap", 1.0)))
    raw.setdefault("beat_snap_ms", float(get_config("segment_layout.beat_snap_ms", 0.0) or 0.0))
    num = lambda key, dflt: float(raw.get(key, dflt) or dflt)
    return LayoutConfig(enable=bool(raw.get("enable", False)), micro_merge_s=max(0.0, num("micro_merge_s", 0.0)),
                        soft_min_s=max(0.0, num("soft_min_s", 0.0)), soft_max_s=max(0.0, num("soft_max_s", 0.0)),
                        min_gap_s=max(0.0, num("min_gap_s", 1.0)), beat_snap_ms=max(0.0, num("beat_snap_ms", 0.0)))


# ----------------------------------------------------------------------------------------------
# classification (`_classify_segments_vocal_presence`)
# ----------------------------------------------------------------------------------------------

def classify_segments(vocal: np.ndarray, cuts: Sequence[int], sr: int) -> Tuple[List[bool], List[Dict]]:
    n_seg = max(len(cuts) - 1, 0)
    if n_seg == 0:
        return [], []
    if sr <= 0 or vocal is None or getattr(vocal, "size", 0) == 0:
        return [True] * n_seg, [{"index": i, "reason": "fallback_invalid_input", "decision": True} for i in range(n_seg)]
    ratio_thr = float(get_config("quality_control.segment_vocal_activity_ratio", 0.10))
    thr_db = float(get_config("quality_control.segment_vocal_threshold_db", -50.0))
    hop = max(1, int(0.02 * sr))
    frame_length = max(hop * 2, int(0.05 * sr))
    flags: List[bool] = []
    debug: List[Dict] = []
    for i in range(n_seg):
        a = max(0, min(int(cuts[i]), len(vocal)))
        b = max(a, min(int(cuts[i + 1]), len(vocal)))
        t0, t1 = a / sr, b / sr
        dur = max(t1 - t0, 1e-6)
        seg = vocal[a:b] if b > a else None
        ratio = 0.0
        seconds = 0.0
        rms_db = None
        if seg is not None and len(seg) >= frame_length:
            fr = L.rms(seg, frame_length=frame_length, hop_length=hop)[0]
            active = (20.0 * np.log10(fr + 1e-12)) > thr_db
            if active.size > 0:
                ratio = float(np.mean(active))
                seconds = float(min(dur, float(active.sum()) * (hop / sr)))
        elif seg is not None and len(seg) > 0:
            rms_db = 20.0 * np.log10(float(np.sqrt(np.mean(np.square(seg)) + 1e-12)))
            if rms_db > thr_db:
                ratio, seconds = 1.0, dur
        if seg is not None and rms_db is None and len(seg) > 0:
            rms_db = 20.0 * np.log10(float(np.sqrt(np.mean(np.square(seg)) + 1e-12)))
        decision = ratio >= ratio_thr
        why = "vocal_activity_ratio_gte_threshold" if decision else "vocal_activity_ratio_lt_threshold"
        debug.append({"index": i, "start_s": t0, "end_s": t1, "duration_s": dur, "vocal_activity_ratio": ratio,
                      "vocal_activity_seconds": seconds, "activity_ratio_threshold": ratio_thr, "activity_threshold_db": thr_db,
                      "rms_db": rms_db, "decision": decision, "decision_reason": why, "reason": why})
        flags.append(bool(decision))
    return flags, debug


# ----------------------------------------------------------------------------------------------
# layout refiner (`refine_layout`)
# ----------------------------------------------------------------------------------------------

def _dur(s) -> float:
    return max(0.0, s[1] - s[0])


def _rechain(segs: List[list]) -> None:
    for i in range(1, len(segs)):
        segs[i][0] = segs[i - 1][1]


def micro_merge(segs: List[list], micro_s: float, soft_max_s: float) -> List[list]:
    """`_apply_micro_merge` (`:136-198`)."""
    if micro_s <= 0.0 or len(segs) <= 1:
        return segs
    segs = [list(s) for s in segs]
    i = 0
    while len(segs) > 1 and i < len(segs):
        s = segs[i]
        if "_lib" in s[2] or _dur(s) >= micro_s:
            i += 1
            continue
        left = segs[i - 1] if i > 0 else None
        right = segs[i + 1] if i + 1 < len(segs) else None
        if left is None and right is None:
            break
        if left is not None and right is not None:
            lc, rc = s[1] - left[0], right[1] - s[0]
            lp = lc if (soft_max_s <= 0.0 or lc <= soft_max_s) else INF
            rp = rc if (soft_max_s <= 0.0 or rc <= soft_max_s) else INF
            go_left = lp <= rp
            if go_left and lp == INF and rp != INF:
                go_left = False
            if (not go_left) and rp == INF and lp != INF:
                go_left = True
        else:
            go_left = left is not None
        if go_left:
            left[1] = s[1]
            segs.pop(i)
            i = max(i - 1, 0)
        else:
            segs[i] = [s[0], right[1], right[2]]
            segs.pop(i + 1)
    return segs


def soft_min_merge(segs: List[list], soft_min_s: float, soft_max_s: float) -> List[list]:
    """`_apply_soft_min_merge` (`:201-261`)."""
    if soft_min_s <= 0.0 or len(segs) <= 1:
        return segs
    segs = [list(s) for s in segs]

    def cost(nb, s) -> float:
        if nb is None:
            return INF
        comb = _dur(nb) + _dur(s)
        over = INF if (soft_max_s > 0.0 and comb > soft_max_s) else comb
        return over + (0.0 if nb[2] == s[2] else comb + 1.0)

    i = 0
    while len(segs) > 1 and i < len(segs):
        s = segs[i]
        if "_lib" in s[2] or _dur(s) >= soft_min_s:
            i += 1
            continue
        left = segs[i - 1] if i > 0 else None
        right = segs[i + 1] if i + 1 < len(segs) else None
        if left is None and right is None:
            break
        lc, rc = cost(left, s), cost(right, s)
        if lc == rc:
            go_left = s[2] == (left[2] if left is not None else "")
        else:
            go_left = lc < rc
        if go_left and left is not None:
            segs[i - 1] = [left[0], s[1], left[2]]
            segs.pop(i)
            i = max(i - 1, 0)
        elif (not go_left) and right is not None:
            segs[i] = [s[0], right[1], right[2]]
            segs.pop(i + 1)
        else:
            i += 1
    _rechain(segs)
    return segs


def post_split_micro_merge(segs: List[list], micro_s: float, soft_max_s: float) -> List[list]:
    """`_apply_post_split_micro_merge` (`:264-313`)."""
    if micro_s <= 0.0 or len(segs) <= 1:
        return segs
    segs = [list(s) for s in segs]
    i = 0
    while len(segs) > 1 and i < len(segs):
        s = segs[i]
        if "_lib" in s[2] or _dur(s) >= micro_s:
            i += 1
            continue
        options = []
        if i > 0:
            options.append(("L", segs[i - 1], s[1] - segs[i - 1][0]))
        if i + 1 < len(segs):
            options.append(("R", segs[i + 1], segs[i + 1][1] - s[0]))
        if not options:
            i += 1
            continue

        def key(opt):
            _, nb, comb = opt
            pen = 0.0 if nb[2] == s[2] else 10.0
            over = 0.0
            if soft_max_s > 0.0 and comb > soft_max_s:
                over = comb - soft_max_s
                if nb[2] != s[2] or over > micro_s:
                    pen += 100.0 + over
            return (pen, over, comb)

        side = min(options, key=key)[0]
        if side == "L" and i > 0:
            left = segs[i - 1]
            segs[i - 1] = [left[0], s[1], left[2]]
            segs.pop(i)
            i = max(i - 1, 0)
        elif side == "R" and i + 1 < len(segs):
            right = segs[i + 1]
            segs[i:i + 2] = [[s[0], right[1], right[2]]]
        else:
            i += 1
    _rechain(segs)
    return segs


def acoustic_valley_split(seg, rms_series: Optional[np.ndarray], hop_s: float, min_gap_s: float) -> Optional[float]:
    """`_find_acoustic_valley_split` (`:402-471`) without ASR inputs: quietest local minimum of the cached RMS."""
    if rms_series is None or len(rms_series) <= 2:
        return None
    start = seg[0] + max(0.0, min_gap_s)
    end = seg[1] - max(0.0, min_gap_s)
    if end <= start:
        return None
    n = len(rms_series)
    fi = lambda t: int(np.clip(int(round(t / hop_s)), 0, max(n - 1, 0)))  # TrackFeatureCache.frame_index (`features_cache.py:62-66`)
    a = fi(start)
    b = min(n, max(a + 1, fi(end) + 1))                                   # TrackFeatureCache.frame_slice (`:68-73`)
    rms = np.asarray(rms_series[a:b], dtype=np.float64)
    if rms.size < 3 or not np.all(np.isfinite(rms)):
        return None
    med = float(np.median(rms))
    spread = float(np.percentile(rms, 75) - np.percentile(rms, 5))
    if med <= 1e-12 or spread <= max(1e-9, med * 0.02):
        return None
    thr = min(float(np.percentile(rms, 25)), med * 0.75)
    best_t, best = None, -1.0
    for j in range(1, rms.size - 1):
        v = float(rms[j])
        if v > thr or v > float(rms[j - 1]) or v > float(rms[j + 1]):
            continue
        t = (a + j) * float(hop_s)
        if t <= start or t >= end:
            continue
        score = max(0.0, (med - v) / max(med, 1e-12))
        if score > best:
            best, best_t = score, float(t)
    if best_t is None or best < 0.5:
        return None
    return best_t


def soft_max_splits(segs: List[list], suppressed: List[Tuple[float, float]], soft_max_s: float, min_gap_s: float,
                    rms_series: Optional[np.ndarray], hop_s: float, midpoint_fallback: bool = False):
    """`_apply_soft_max_splits` (`:316-386`).  `suppressed` = [(t, score)] of cut points the guard rejected; returns
    (segments, remaining suppressed, times of the new splits)."""
    if soft_max_s <= 0.0 or len(segs) <= 0:
        return segs, suppressed, []
    suppressed = list(suppressed)
    segs = [list(s) for s in segs]
    new_cuts: List[float] = []
    tol = 1e-3
    i = 0
    while i < len(segs):
        s = segs[i]
        if _dur(s) <= soft_max_s:
            i += 1
            continue
        cands = [p for p in suppressed if (s[0] + tol) < float(p[0]) < (s[1] - tol)]
        if cands:
            best = max(cands, key=lambda p: float(p[1] or 0.0))
            cut = float(best[0])
            suppressed.remove(best)
        else:
            cut = acoustic_valley_split(s, rms_series, hop_s, min_gap_s)
            if cut is None and midpoint_fallback:
                cut = s[0] + _dur(s) / 2.0
        if cut is None:
            i += 1
            continue
        ld, rd = cut - s[0], s[1] - cut
        if ld <= 0.0 or rd <= 0.0 or (min_gap_s > 0.0 and (ld < min_gap_s or rd < min_gap_s)):
            i += 1
            continue
        segs[i:i + 1] = [[s[0], cut, s[2]], [cut, s[1], s[2]]]
        new_cuts.append(cut)
    _rechain(segs)
    return segs, suppressed, new_cuts


def enforce_min_gap(segs: List[list], min_gap_s: float) -> List[list]:
    """`_enforce_min_gap` (`:497-545`)."""
    if min_gap_s <= 0.0 or len(segs) <= 1:
        return segs
    segs = [list(s) for s in segs]
    i = 0
    while len(segs) > 1 and i < len(segs):
        s = segs[i]
        if "_lib" in s[2] or _dur(s) >= min_gap_s:
            i += 1
            continue
        has_l, has_r = i > 0, i + 1 < len(segs)
        if not (has_l or has_r):
            i += 1
            continue
        if has_l and has_r:
            go_left = (s[1] - segs[i - 1][0]) <= (segs[i + 1][1] - s[0])
        else:
            go_left = has_l
        if go_left:
            left = segs[i - 1]
            segs[i - 1] = [left[0], s[1], left[2]]
            segs.pop(i)
            i = max(i - 1, 0)
        else:
            right = segs[i + 1]
            segs[i:i + 2] = [[s[0], right[1], right[2]]]
    _rechain(segs)
    return segs


def beat_snap(segs: List[list], beat_snap_ms: float, beat_times, min_gap_s: float) -> List[list]:
    """`_apply_beat_snap` + `_snap_to_beat` (`:548-598`)."""
    if beat_snap_ms <= 0.0 or beat_times is None or len(beat_times) == 0:
        return segs
    lim = beat_snap_ms / 1000.0
    segs = [list(s) for s in segs]
    for i in range(1, len(segs)):
        bt = segs[i][0]
        best_t, best_off = None, None
        for beat in beat_times:
            off = abs(float(beat) - bt)
            if off > lim:
                continue
            if best_off is None or off < best_off:
                best_off, best_t = off, float(beat)
        if best_t is None:
            continue
        if (best_t - segs[i - 1][0]) < min_gap_s or (segs[i][1] - best_t) < min_gap_s:
            continue
        segs[i - 1][1] = best_t
        segs[i][0] = best_t
    _rechain(segs)
    return segs


def refine_layout(segs: Sequence[Sequence], cfg: LayoutConfig, suppressed: Sequence[Tuple[float, float]],
                  rms_series: Optional[np.ndarray], hop_s: float, beat_times, midpoint_fallback: bool = False):
    """`refine_layout` (`:74-133`) -> (segments, remaining suppressed points, new split times)."""
    segs = [list(s) for s in segs]
    supp = list(suppressed or [])
    if not cfg.enable or len(segs) <= 1:
        return segs, supp, []
    segs = micro_merge(segs, cfg.micro_merge_s, cfg.soft_max_s)
    segs = soft_min_merge(segs, cfg.soft_min_s, cfg.soft_max_s)
    segs, supp, new_cuts = soft_max_splits(segs, supp, cfg.soft_max_s, cfg.min_gap_s, rms_series, hop_s, midpoint_fallback)
    segs = post_split_micro_merge(segs, cfg.micro_merge_s, cfg.soft_max_s)
    segs = enforce_min_gap(segs, cfg.min_gap_s)
    segs = beat_snap(segs, cfg.beat_snap_ms, beat_times, cfg.min_gap_s)
    return segs, supp, new_cuts


def layout_to_cuts(segs: Sequence[Sequence], sr: int, n_samples: int) -> List[int]:
    """`seamless_splitter.py:586-598`: boundaries in seconds -> clamped integer samples, ends pinned, de-duplicated."""
    times = [segs[0][0]] + [s[1] for s in segs]
    cuts = [max(0, min(int(round(t * sr)), n_samples)) for t in times]
    if cuts:
        cuts[0] = 0
        cuts[-1] = n_samples
    return sorted(set(cuts))


# ----------------------------------------------------------------------------------------------
# local valley refinement (`_refine_boundaries_local_valley`)
# ----------------------------------------------------------------------------------------------

def refine_local_valley(cuts: List[int], vocal: np.ndarray, sr: int, cfg: Dict, min_gap_s: float) -> List[int]:
    if vocal is None or vocal.size == 0 or len(cuts) <= 2:
        return cuts
    fs = float(sr)
    radius = max(1, int(float(cfg.get("search_radius_ms", 200)) / 1000.0 * fs))
    win = max(1, int(float(cfg.get("window_ms", 20)) / 1000.0 * fs))
    drop_db = float(cfg.get("min_drop_db", 3.0))
    micro = float(get_config("segment_layout.micro_merge_s", 0.0) or 0.0)
    piece = float(get_config("quality_control.segment_min_mix_piece", 0.0) or 0.0)
    min_seg = max(1, int(max(float(min_gap_s), micro, piece) * fs))
    out = list(cuts)
    for i in range(1, len(out) - 1):
        c = out[i]
        a = max(0, c - radius)
        b = min(len(vocal), c + radius)
        seg = vocal[a:b]
        if seg.size <= win:
            continue
        sq = np.square(seg.astype(np.float64))
        rms = np.sqrt(np.convolve(sq, np.ones(win, dtype=np.float64) / float(win), mode="valid") + 1e-12)
        if rms.size == 0:
            continue
        db = 20.0 * np.log10(rms + 1e-12)
        o = int(np.clip(c - a - win // 2, 0, db.size - 1))
        v = int(np.argmin(db))
        if (db[o] - db[v]) < drop_db:
            continue
        cand = a + v + win // 2
        if cand <= out[i - 1] + min_seg or cand >= out[i + 1] - min_seg:
            continue
        out[i] = cand
    return out


# ----------------------------------------------------------------------------------------------
# weak-tail merge and the sample-level split
# ----------------------------------------------------------------------------------------------

def merge_weak_human_tails(cuts: List[int], flags: List[bool], vocal: np.ndarray, sr: int, min_duration_s: float,
                           layout_applied: bool) -> Tuple[List[int], List[bool]]:
    """`_merge_short_weak_human_tails_into_following_music` (`:2145-2275`) without the debug bookkeeping."""
    if (not layout_applied or min_duration_s <= 0.0 or len(cuts) < 3 or len(flags) != len(cuts) - 1
            or vocal is None or getattr(vocal, "size", 0) == 0 or sr <= 0):
        return list(cuts), list(flags)
    pts = [int(p) for p in cuts]
    fl = [bool(f) for f in flags]
    fs = float(sr)

    def stats():
        out = []
        for i in range(len(pts) - 1):
            a = max(0, min(pts[i], len(vocal)))
            b = max(a, min(pts[i + 1], len(vocal)))
            seg = vocal[a:b]
            if seg.size:
                r = float(np.sqrt(np.mean(np.square(seg.astype(np.float64))) + 1e-12))
                p = float(np.max(np.abs(seg)))
            else:
                r = p = 0.0
            out.append((max(0.0, (pts[i + 1] - pts[i]) / fs), r, p))
        return out

    st = stats()
    ref_r = [s[1] for s, f in zip(st, fl) if f and s[0] >= min_duration_s and s[1] > 0.0]
    ref_p = [s[2] for s, f in zip(st, fl) if f and s[0] >= min_duration_s and s[2] > 0.0]
    if not ref_r or not ref_p:
        return pts, fl
    rr = float(np.median(np.asarray(ref_r, dtype=np.float64)))
    rp = float(np.median(np.asarray(ref_p, dtype=np.float64)))
    w_r = float(get_config("quality_control.short_human_tail_rms_ratio", 0.12) or 0.12)
    w_p = float(get_config("quality_control.short_human_tail_peak_ratio", 0.18) or 0.18)
    i = 0
    while i < len(fl) - 1:
        d, r, p = stats()[i]
        if fl[i] and not fl[i + 1] and d < min_duration_s and r <= rr * w_r and p <= rp * w_p:
            pts.pop(i + 1)
            fl[i:i + 2] = [False]
        else:
            i += 1
    return pts, fl


def split_at_sample_level(n_samples: int, cuts: Sequence[int], flags: Optional[Sequence[bool]], sr: int):
    """`_split_at_sample_level` (`:2006-2144`) as index ranges (the pieces are consecutive slices of one array, so a
    concatenation of carried audio and the next chunk is the range from the carry's start to the chunk's end).
    Pieces shorter than 10 ms are carried into the next piece, a trailing carry is appended to the last piece.
    Returns ([(start, end)] with end exclusive, merged flags)."""
    keep = max(1, int(0.01 * sr))
    clampi = lambda v: max(0, min(int(v), n_samples))
    out: List[List[int]] = []
    out_flags: Optional[List[bool]] = [] if flags is not None else None
    carry: Optional[List[int]] = None          # [start, end) of audio waiting to be merged forward
    carry_flag: Optional[bool] = None
    for i in range(len(cuts) - 1):
        a, b = clampi(cuts[i]), clampi(cuts[i + 1])
        b = max(a, b)
        piece = [a, b] if b > a else None       # `chunk = audio[start:end]`
        flag = bool(flags[i]) if (flags is not None and i < len(flags)) else True
        if carry is not None:
            piece = [carry[0], piece[1]] if piece is not None else list(carry)
            flag = bool(carry_flag) or flag
            carry, carry_flag = None, None
        if int(cuts[i + 1]) - int(cuts[i]) >= keep and piece is not None:
            out.append(piece)
            if out_flags is not None:
                out_flags.append(flag)
        elif piece is not None:
            carry, carry_flag = piece, flag
    if carry is not None:
        if out:
            out[-1][1] = carry[1]
            if out_flags is not None:
                out_flags[-1] = bool(out_flags[-1]) or bool(carry_flag)
        else:
            out.append(carry)
            if out_flags is not None:
                out_flags.append(bool(carry_flag))
    return [tuple(x) for x in out], out_flags


# ----------------------------------------------------------------------------------------------
# the whole policy
# ----------------------------------------------------------------------------------------------

@dataclass
class PolicyResult:
    cuts: List[int]
    flags: List[bool]
    pieces: List[Tuple[int, int]]
    layout_applied: bool
    suppressed: List[Tuple[float, float]]


def apply_boundary_policy(cuts: Sequence[int], vocal: np.ndarray, n_samples: int, sr: int, *,
                          suppressed: Sequence[Tuple[float, float]], rms_series: Optional[np.ndarray], hop_s: float,
                          beat_times) -> PolicyResult:
    """`seamless_splitter.py:521-669` for modes without lyrics: classify -> layout -> classify -> local valley ->
    classify -> weak-tail merge -> sample-level split."""
    cuts = sorted(set(int(c) for c in cuts))
    flags, _ = classify_segments(vocal, cuts, sr)
    cfg = layout_config_from_settings()
    applied = False
    supp = list(suppressed or [])
    if cfg.enable and len(cuts) >= 2:
        segs = [[cuts[i] / float(sr), cuts[i + 1] / float(sr), "human" if flags[i] else "music"] for i in range(len(cuts) - 1)]
        segs, supp, _ = refine_layout(segs, cfg, supp, rms_series, hop_s, beat_times)
        if segs:
            new_cuts = layout_to_cuts(segs, sr, n_samples)
            if new_cuts != cuts:
                applied = True
            cuts = new_cuts if new_cuts else cuts
            flags, _ = classify_segments(vocal, cuts, sr)
    local = get_config("quality_control.local_boundary_refine", {}) or {}
    if local.get("enable") and len(cuts) >= 2:
        ref = refine_local_valley(cuts, vocal, sr, local, float(get_config("quality_control.min_split_gap", 1.0)))
        if ref != cuts:
            cuts = ref
            applied = True
            flags, _ = classify_segments(vocal, cuts, sr)
    c2, f2 = merge_weak_human_tails(cuts, flags, vocal, sr, float(cfg.soft_min_s or 0.0), applied)
    if c2 != cuts:
        cuts, flags, applied = c2, f2, True
    pieces, merged_flags = split_at_sample_level(n_samples, cuts, flags, sr)
    return PolicyResult(cuts=list(cuts), flags=list(merged_flags or []), pieces=pieces, layout_applied=applied, suppressed=supp)
