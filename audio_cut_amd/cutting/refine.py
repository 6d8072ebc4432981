"""Cut-point refinement on HIP kernels — drop-in for the reference's `src/audio_cut/cutting/refine.py`
(`finalize_cut_points(ctx, raw_points, *, use_vocal_guard_first=True, min_gap_s=1.0, max_keep=None,
topk_per_10s=None, nms_window_s=10.0, guard_db=2.0, search_right_ms=150.0, guard_win_ms=10.0,
floor_db=-60.0, enable_mix_guard=True, enable_vocal_guard=True, zero_cross_win_ms=8.0,
min_boundary_s=0.5) -> CutRefineResult`, `:268-285`; dataclasses `:16-60`).

Where the reference spends its time (`:171-180`: an O(N*3528) float64 `np.convolve` per wave and a
2N-iteration Python loop) this build runs `ac_moving_meansq_db_f64` + `ac_next_leq_scan` once per
wave on the resident track and answers every per-candidate question with one batched launch per
stage: `ac_zero_cross_nearest`, `ac_window_argmin_f64`, `ac_quiet_guard_slow`.  The candidates move
through the four stages (vocal snap, vocal guard, mix snap, mix guard) together; the decision
arithmetic on the returned scalars (`:152,208-214`) is the reference's, in float64.

Scalar promotion at `:101` follows numpy<2 (float64 zero positions) — the environment the
reference pins (`requirements.txt:6`); see DESIGN.md.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from .. import _native

_EPS = 1e-12


@dataclass
class CutPoint:
    t: float
    score: float
    kind: str = "pause"


@dataclass
class CutContext:
    sr: int
    mix_wave: np.ndarray
    vocal_wave: Optional[np.ndarray] = None
    # optional device-resident copies (torch tensors) so a track already in HBM is not re-uploaded
    mix_dev: object = None
    vocal_dev: object = None
    hip: object = None


@dataclass
class CutAdjustment:
    raw_time: float
    guard_time: float
    final_time: float
    score: float
    guard_shift_ms: float
    final_shift_ms: float


@dataclass
class CutRefineResult:
    final_points: List[CutPoint]
    sample_boundaries: List[int]
    adjustments: List[CutAdjustment]
    suppressed_points: List[CutPoint] = field(default_factory=list)


@dataclass
class QuietGuardLookup:
    rms_db: object          # float64 [N] device tensor
    next_quiet: object      # int64 [N] device tensor
    floor_db: float


def _ensure_mono(wave: Optional[np.ndarray]) -> Optional[np.ndarray]:
    if wave is None or wave.ndim == 1:
        return wave
    if wave.ndim == 2:
        return np.mean(wave, axis=0)
    return wave.reshape(-1)


def nms_min_gap(points: Iterable[CutPoint], min_gap_s: float, topk: Optional[int] = None, *,
                max_per_window: Optional[int] = None, window_s: float = 10.0) -> List[CutPoint]:
    """Score-descending greedy suppression (reference `:218-245`); `sorted` is stable, ties keep input order."""
    ordered = sorted(points, key=lambda p: p.score, reverse=True)
    kept: List[CutPoint] = []
    counts: Dict[int, int] = {}
    span = max(window_s, min_gap_s, 1e-6)
    for p in ordered:
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


def _filter_cut_times(times: Sequence[float], *, duration_s: float, min_gap_s: float, min_boundary_s: float) -> List[float]:
    out: List[float] = []
    if duration_s <= 0.0:
        return out
    edge = min(min_boundary_s, duration_s / 2.0)
    for t in sorted(times):
        if t <= edge or t >= (duration_s - edge):
            continue
        if out and (t - out[-1]) < min_gap_s:
            continue
        out.append(t)
    return out


class _Wave:
    """One wave resident on the device + its lazily built quiet-guard lookup."""

    def __init__(self, hip: "_native.Context", dev, n: int, sr: int):
        self.hip, self.dev, self.n, self.sr = hip, dev, n, sr
        self.lookup: Optional[QuietGuardLookup] = None

    def prepare_lookup(self, window_ms: float, floor_db: float) -> QuietGuardLookup:
        """reference `_prepare_quiet_lookup` (`:161-181`)."""
        win = max(1, int(round(window_ms / 1000.0 * self.sr)))
        db = self.hip.moving_meansq_db(self.dev, win)
        nq = self.hip.next_leq_scan(db, floor_db)
        self.lookup = QuietGuardLookup(db, nq, floor_db)
        return self.lookup

    # ---- batched stages; `times` is a float64 numpy vector, returns a new vector --------------------
    def zero_cross(self, times: np.ndarray, win_ms: float) -> np.ndarray:
        """reference `align_to_zero_cross` (`:72-110`)."""
        out = times.copy()
        if self.n == 0 or self.sr <= 0:
            return out
        idx = np.array([int(round(t * self.sr)) for t in times], dtype=np.int64)
        live = (idx > 0) & (idx < self.n)
        if not np.any(live):
            return out
        half = max(1, int(round(win_ms / 1000.0 * self.sr)))
        pos = self.hip.zero_cross_nearest(self.dev, idx[live], half)
        hit = ~np.isnan(pos)
        res = out[live]
        res[hit] = pos[hit] / float(self.sr)
        out[live] = res
        return out

    def guard(self, times: np.ndarray, *, max_shift_ms: float, guard_db: float, window_ms: float, floor_db: float,
              use_lookup: bool) -> np.ndarray:
        """reference `_apply_quiet_guard_fast` (`:184-214`) then, for points it leaves alone,
        `apply_quiet_guard` (`:113-157`)."""
        out = times.copy()
        sr, n = self.sr, self.n
        if n == 0 or sr <= 0 or len(times) == 0:
            return out
        shift = max(1, int(round(max_shift_ms / 1000.0 * sr)))
        moved = np.zeros(len(times), dtype=bool)
        if use_lookup and self.lookup is not None:
            idx = np.array([int(np.clip(int(round(t * sr)), 0, n - 1)) for t in times], dtype=np.int64)
            ln = np.minimum(n, idx + shift) - idx
            arg, val = self.hip.window_argmin(self.lookup.rms_db, idx, ln)
            for q in range(len(times)):
                if ln[q] <= 0:
                    continue
                orig_db, tgt_db = val[q, 0], val[q, 1]
                if (orig_db - tgt_db) < guard_db or tgt_db > self.lookup.floor_db or arg[q] == idx[q]:
                    continue
                new_t = float(arg[q]) / float(sr)
                if new_t != times[q]:
                    out[q] = new_t
                    moved[q] = True
        rest = np.flatnonzero(~moved)
        if rest.size:
            win = max(1, int(round(window_ms / 1000.0 * sr)))
            idx = np.array([max(0, int(round(times[q] * sr))) for q in rest], dtype=np.int64)
            ok = np.minimum(n, idx + shift) > idx + 1
            if np.any(ok):
                arg, val = self.hip.quiet_guard_slow(self.dev, idx[ok], shift, win)
                for j, q in enumerate(rest[ok]):
                    if arg[j] < 0:
                        continue
                    orig_db, tgt_db = val[j, 0], val[j, 1]
                    if (orig_db - tgt_db) < guard_db or tgt_db > floor_db:
                        continue
                    centre = min(n - 1, max(0, int(idx[ok][j]) + int(arg[j]) + win // 2))     # `:155`
                    out[q] = float(centre) / float(sr)
        return out


def finalize_cut_points(ctx: CutContext, raw_points: Iterable[CutPoint], *, use_vocal_guard_first: bool = True,
                        min_gap_s: float = 1.0, max_keep: Optional[int] = None, topk_per_10s: Optional[int] = None,
                        nms_window_s: float = 10.0, guard_db: float = 2.0, search_right_ms: float = 150.0,
                        guard_win_ms: float = 10.0, floor_db: float = -60.0, enable_mix_guard: bool = True,
                        enable_vocal_guard: bool = True, zero_cross_win_ms: float = 8.0,
                        min_boundary_s: float = 0.5) -> CutRefineResult:
    sr = ctx.sr
    mix = _ensure_mono(ctx.mix_wave)
    vocal = _ensure_mono(ctx.vocal_wave) if ctx.vocal_wave is not None else None
    if mix is None or mix.size == 0 or sr <= 0:
        return CutRefineResult([], [0, len(mix) if mix is not None else 0], [])
    duration_s = len(mix) / float(sr)
    base = list(raw_points)
    if not base:
        return CutRefineResult([], [0, len(mix)], [])
    cap = topk_per_10s if (topk_per_10s is not None and topk_per_10s > 0) else None
    pruned = nms_min_gap(base, min_gap_s=min_gap_s, topk=max_keep, max_per_window=cap, window_s=nms_window_s)
    kept_ids = {id(p) for p in pruned}
    suppressed = [CutPoint(t=float(p.t), score=float(p.score), kind=p.kind) for p in base if id(p) not in kept_ids]

    hip = ctx.hip if ctx.hip is not None else _native.Context()
    mix_w = _Wave(hip, ctx.mix_dev if ctx.mix_dev is not None else hip.to_device(np.ascontiguousarray(mix, dtype=np.float32)), len(mix), sr)
    voc_w = None
    if vocal is not None:
        voc_w = _Wave(hip, ctx.vocal_dev if ctx.vocal_dev is not None else hip.to_device(np.ascontiguousarray(vocal, dtype=np.float32)), len(vocal), sr)
    if enable_vocal_guard and voc_w is not None and vocal.size:
        voc_w.prepare_lookup(guard_win_ms, floor_db)
    if enable_mix_guard:
        mix_w.prepare_lookup(guard_win_ms, floor_db)

    raw_t = np.array([p.t for p in pruned], dtype=np.float64)
    guard_t = raw_t.copy()
    if use_vocal_guard_first and voc_w is not None:
        guard_t = voc_w.zero_cross(guard_t, zero_cross_win_ms)
        if enable_vocal_guard:
            guard_t = voc_w.guard(guard_t, max_shift_ms=search_right_ms, guard_db=guard_db, window_ms=guard_win_ms,
                                  floor_db=floor_db, use_lookup=True)
    mix_t = mix_w.zero_cross(guard_t, zero_cross_win_ms)
    if enable_mix_guard:
        mix_t = mix_w.guard(mix_t, max_shift_ms=search_right_ms, guard_db=guard_db, window_ms=guard_win_ms,
                            floor_db=floor_db, use_lookup=True)
    mix_t = np.clip(mix_t, 0.0, max(duration_s, 0.0))

    adjustments = [CutAdjustment(raw_time=float(r), guard_time=float(g), final_time=float(m), score=float(p.score),
                                 guard_shift_ms=float((g - r) * 1000.0), final_shift_ms=float((m - r) * 1000.0))
                   for p, r, g, m in zip(pruned, raw_t, guard_t, mix_t)]
    kept_times = _filter_cut_times([float(m) for m in mix_t], duration_s=duration_s, min_gap_s=min_gap_s,
                                   min_boundary_s=min_boundary_s)
    kept_adj: List[CutAdjustment] = []
    for t in kept_times:
        diffs = [abs(a.final_time - t) for a in adjustments]
        kept_adj.append(adjustments[int(np.argmin(diffs))])
    bounds = sorted({0, len(mix), *(int(round(t * sr)) for t in kept_times)})
    return CutRefineResult([CutPoint(t=float(t), score=1.0) for t in kept_times], bounds, kept_adj, suppressed)


__all__ = ["CutPoint", "CutContext", "CutAdjustment", "CutRefineResult", "QuietGuardLookup", "nms_min_gap",
           "finalize_cut_points"]
