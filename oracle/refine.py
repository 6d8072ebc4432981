"""Oracle for SURVEY.md §8 row a17: NMS -> zero-cross snap -> quiet guard -> integer boundaries.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
`src/audio_cut/cutting/refine.py` (all of it; function-level citations below)
in vectorised numpy.  Pinned against the reference module itself by
`tests/golden/make_golden.py` (fixtures `tests/golden/refine_*.npz`) and by the
reference's own known answer `tests/unit/test_cutting_consistency.py:20-46`
(`sample_boundaries == [0, 40, 80, 120]`).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

EPS = 1e-12


@dataclass
class Cut:
    t: float
    score: float
    kind: str = "pause"


@dataclass
class Adjustment:
    raw_time: float
    guard_time: float
    final_time: float
    score: float


@dataclass
class RefineOutput:
    times: List[float]
    sample_boundaries: List[int]
    adjustments: List[Adjustment]
    suppressed: List[Cut] = field(default_factory=list)


def _mono(w: Optional[np.ndarray]) -> Optional[np.ndarray]:
    if w is None or w.ndim == 1:
        return w
    return np.mean(w, axis=0) if w.ndim == 2 else w.reshape(-1)


# The reference adds a python int to a numpy scalar of the wave dtype at refine.py:101
# (`zero_pos = (pos - 1) + frac`).  With float32 audio that sum is float64 under the numpy the
# reference pins (`requirements.txt:6`, numpy<2.0: value-based scalar promotion) but float32 under
# numpy>=2 (NEP 50), where positions beyond 2**23 samples lose their fraction.  The parity target is
# the pinned environment (LEGACY_PROMOTION=True); golden generation against the reference running
# under this container's numpy 2.2 flips the switch to match what that run actually computes.
LEGACY_PROMOTION = True


def zero_cross_snap(wave: Optional[np.ndarray], sr: int, t: float, win_ms: float = 8.0,
                    legacy_promotion: Optional[bool] = None) -> float:
    """refine.py:72-110 — nearest (fractional) zero crossing within +-win_ms, first wins on ties."""
    if legacy_promotion is None:
        legacy_promotion = LEGACY_PROMOTION
    wave = _mono(wave)
    if wave is None or wave.size == 0 or sr <= 0:
        return t
    idx = int(round(t * sr))
    if idx <= 0 or idx >= wave.size:
        return t
    half = max(1, int(round(win_ms / 1000.0 * sr)))
    lo = max(1, idx - half)
    hi = min(wave.size - 1, idx + half)
    if hi <= lo:
        return t
    pos = np.arange(lo, hi + 1)
    left = wave[pos - 1]
    right = wave[pos]
    absl = np.abs(left)
    denom = absl + np.abs(right)
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = np.where(denom > EPS, absl / denom, left.dtype.type(0.5))
    if legacy_promotion or frac.dtype == np.float64:
        interp = (pos - 1).astype(np.float64) + frac.astype(np.float64)
    else:
        interp = ((pos - 1).astype(frac.dtype) + frac).astype(np.float64)
    zero = np.where(left == 0.0, (pos - 1).astype(np.float64),
                    np.where(right == 0.0, pos.astype(np.float64), interp))
    valid = (left == 0.0) | (right == 0.0) | (left * right < 0.0)
    if not np.any(valid):
        return t
    dist = np.where(valid, np.abs(zero - idx), np.inf)
    k = int(np.argmin(dist))
    return float(zero[k]) / float(sr)


def quiet_guard_slow(wave: Optional[np.ndarray], sr: int, t: float, *, max_shift_ms: float, guard_db: float,
                     window_ms: float, floor_db: float) -> float:
    """refine.py:113-157 — edge-padded window RMS over the search span, 'valid' convolution."""
    wave = _mono(wave)
    if wave is None or wave.size == 0 or sr <= 0:
        return t
    idx = max(0, int(round(t * sr)))
    span = max(1, int(round(max_shift_ms / 1000.0 * sr)))
    end = min(wave.size, idx + span)
    if end <= idx + 1:
        return t
    seg = wave[idx:end]
    win = max(1, int(round(window_ms / 1000.0 * sr)))
    if seg.size <= win:
        level = seg
    else:
        padded = np.pad(seg, (0, win - 1), mode="edge")
        level = np.sqrt(np.convolve(padded * padded, np.ones(win) / float(win), mode="valid") + EPS)
    db = 20.0 * np.log10(level + EPS)
    k = int(np.argmin(db))
    if (db[0] - db[k]) < guard_db or db[k] > floor_db:
        return t
    centre = min(wave.size - 1, max(0, idx + k + win // 2))
    return float(centre) / float(sr)


@dataclass
class QuietLookup:
    rms_db: np.ndarray      # float64 [N]
    next_quiet: np.ndarray  # int64 [N]
    floor_db: float


def moving_meansq_db(wave: np.ndarray, win: int) -> np.ndarray:
    """refine.py:170-174 — float64 moving mean of x^2 ('same'), then 20*log10(sqrt(.+eps)+eps).

    Same direct `np.convolve(sq, ones/W, 'same')` call the reference makes (O(N*W) on the CPU).
    """
    sq = np.square(wave.astype(np.float64))
    kernel = np.ones(win, dtype=np.float64) / float(win)
    ms = np.convolve(sq, kernel, mode="same")
    return 20.0 * np.log10(np.sqrt(ms + EPS) + EPS)


def next_leq_scan(db: np.ndarray, floor_db: float) -> np.ndarray:
    """refine.py:175-180 — next_quiet[i] = smallest j >= i with db[j] <= floor, else -1."""
    n = db.size
    hit = db <= floor_db
    idx = np.where(hit, np.arange(n, dtype=np.int64), np.int64(np.iinfo(np.int64).max))
    nxt = np.minimum.accumulate(idx[::-1])[::-1]
    nxt[nxt == np.iinfo(np.int64).max] = -1
    return nxt


def prepare_quiet_lookup(wave: Optional[np.ndarray], sr: int, window_ms: float, floor_db: float) -> Optional[QuietLookup]:
    wave = _mono(wave)
    if wave is None or wave.size == 0 or sr <= 0:
        return None
    win = max(1, int(round(window_ms / 1000.0 * sr)))
    db = moving_meansq_db(wave, win)
    return QuietLookup(db, next_leq_scan(db, floor_db), floor_db)


def quiet_guard_fast(t: float, sr: int, lookup: Optional[QuietLookup], *, max_shift_ms: float, guard_db: float) -> float:
    """refine.py:184-214."""
    if lookup is None or sr <= 0 or lookup.rms_db.size == 0:
        return t
    n = lookup.rms_db.size
    idx = int(np.clip(int(round(t * sr)), 0, n - 1))
    end = min(n, idx + max(1, int(round(max_shift_ms / 1000.0 * sr))))
    if end <= idx:
        return t
    k = idx + int(np.argmin(lookup.rms_db[idx:end]))
    if (lookup.rms_db[idx] - lookup.rms_db[k]) < guard_db:
        return t
    if lookup.rms_db[k] > lookup.floor_db or k == idx:
        return t
    return float(k) / float(sr)


def nms_min_gap(points: Sequence[Cut], min_gap_s: float, topk: Optional[int] = None, *,
                max_per_window: Optional[int] = None, window_s: float = 10.0) -> List[Cut]:
    """refine.py:218-245 — stable score-descending greedy with |dt| < min_gap suppression."""
    order = sorted(points, key=lambda p: p.score, reverse=True)
    kept: List[Cut] = []
    counts = {}
    span = max(window_s, min_gap_s, 1e-6)
    for p in order:
        if any(abs(p.t - q.t) < min_gap_s for q in kept):
            continue
        bucket = None
        if max_per_window is not None:
            bucket = int(p.t // span)
            if counts.get(bucket, 0) >= max_per_window:
                continue
        kept.append(p)
        if bucket is not None:
            counts[bucket] = counts.get(bucket, 0) + 1
        if topk is not None and len(kept) >= topk:
            break
    return sorted(kept, key=lambda p: p.t)


def filter_cut_times(times: Sequence[float], *, duration_s: float, min_gap_s: float, min_boundary_s: float) -> List[float]:
    """refine.py:248-265."""
    out: List[float] = []
    if duration_s <= 0.0:
        return out
    edge = min(min_boundary_s, duration_s / 2.0)
    for t in sorted(times):
        if t <= edge or t >= duration_s - edge:
            continue
        if out and (t - out[-1]) < min_gap_s:
            continue
        out.append(t)
    return out


def finalize_cut_points(sr: int, mix_wave: np.ndarray, vocal_wave: Optional[np.ndarray], raw_points: Sequence[Cut], *,
                        use_vocal_guard_first: bool = True, min_gap_s: float = 1.0, max_keep: Optional[int] = None,
                        topk_per_10s: Optional[int] = None, nms_window_s: float = 10.0, guard_db: float = 2.0,
                        search_right_ms: float = 150.0, guard_win_ms: float = 10.0, floor_db: float = -60.0,
                        enable_mix_guard: bool = True, enable_vocal_guard: bool = True,
                        zero_cross_win_ms: float = 8.0, min_boundary_s: float = 0.5) -> RefineOutput:
    """refine.py:268-410."""
    mix = _mono(mix_wave)
    vocal = _mono(vocal_wave) if vocal_wave is not None else None
    if mix is None or mix.size == 0 or sr <= 0:
        return RefineOutput([], [0, 0 if mix is None else len(mix)], [])
    duration_s = len(mix) / float(sr)
    pts = list(raw_points)
    if not pts:
        return RefineOutput([], [0, len(mix)], [])
    cap = topk_per_10s if (topk_per_10s is not None and topk_per_10s > 0) else None
    pruned = nms_min_gap(pts, min_gap_s, max_keep, max_per_window=cap, window_s=nms_window_s)
    kept_ids = {id(p) for p in pruned}
    suppressed = [Cut(float(p.t), float(p.score), p.kind) for p in pts if id(p) not in kept_ids]

    v_lookup = prepare_quiet_lookup(vocal, sr, guard_win_ms, floor_db) if enable_vocal_guard else None
    m_lookup = prepare_quiet_lookup(mix, sr, guard_win_ms, floor_db) if enable_mix_guard else None

    def _guard(wave, lookup, t):
        fast = quiet_guard_fast(t, sr, lookup, max_shift_ms=search_right_ms, guard_db=guard_db)
        if fast != t:
            return fast
        return quiet_guard_slow(wave, sr, t, max_shift_ms=search_right_ms, guard_db=guard_db,
                                window_ms=guard_win_ms, floor_db=floor_db)

    adjustments: List[Adjustment] = []
    adjusted: List[float] = []
    for p in pruned:
        g = p.t
        if use_vocal_guard_first and vocal is not None:
            g = zero_cross_snap(vocal, sr, g, zero_cross_win_ms)
            if enable_vocal_guard:
                g = _guard(vocal, v_lookup, g)
        m = zero_cross_snap(mix, sr, g, zero_cross_win_ms)
        if enable_mix_guard:
            m = _guard(mix, m_lookup, m)
        m = float(np.clip(m, 0.0, max(duration_s, 0.0)))
        adjustments.append(Adjustment(float(p.t), float(g), m, float(p.score)))
        adjusted.append(m)

    kept_times = filter_cut_times(adjusted, duration_s=duration_s, min_gap_s=min_gap_s, min_boundary_s=min_boundary_s)
    kept_adj: List[Adjustment] = []
    for t in kept_times:
        diffs = [abs(a.final_time - t) for a in adjustments]
        kept_adj.append(adjustments[int(np.argmin(diffs))])
    bounds = sorted({0, len(mix), *(int(round(t * sr)) for t in kept_times)})
    return RefineOutput([float(t) for t in kept_times], bounds, kept_adj, suppressed)
