"""Weak beat-aligned candidates inside continuous high-energy bars — mirrors the reference's
`src/audio_cut/cutting/beat_candidates.py:16-142` and the energy-only branch of
`src/audio_cut/analysis/chorus_regions.py:15-99`.

`vocal_cut_risk` (reference `:117-142`: window RMS / track peak) reads the vocal stem where it lives:
on the GPU (`ac_sum_squares` over the +-80 ms window, `ac_abs_max`-free peak via the same reduction
helper) when a device copy is supplied, else from the host array with the reference's numpy expression.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Set

import numpy as np

from .cut_candidate import CandidateSource, CutCandidate


def detect_chorus_regions(bar_energies: Iterable[float], energy_threshold: float, *, min_consecutive_bars: int = 4,
                          bar_centroids=None, bar_bandwidths=None) -> Set[int]:
    """Indices of bars that belong to runs of >= `min_consecutive_bars` bars with energy >= threshold.
    (The spectral-fusion branch of the reference needs per-bar centroid/bandwidth, which the VPBD path never passes.)"""
    e = np.asarray(list(bar_energies), dtype=np.float32)
    if e.size == 0:
        return set()
    if bar_centroids or bar_bandwidths:
        raise NotImplementedError("spectral-fusion chorus detection belongs to the hybrid_mdd mode (out of scope)")
    high = e >= float(energy_threshold)
    need = max(1, int(min_consecutive_bars))
    bars: Set[int] = set()
    run_start = None
    for i, flag in enumerate(high):
        if flag:
            if run_start is None:
                run_start = i
        else:
            if run_start is not None and i - run_start >= need:
                bars.update(range(run_start, i))
            run_start = None
    if run_start is not None and len(high) - run_start >= need:
        bars.update(range(run_start, len(high)))
    return bars


def _bar_times(beats: np.ndarray, duration_s: float, beats_per_bar: int = 4) -> List[float]:
    inside = sorted(float(b) for b in beats if 0.0 <= float(b) <= duration_s)
    if not inside:
        return [0.0, duration_s]
    if inside[0] > 1e-6:
        inside.insert(0, 0.0)
    bars = inside[:: max(1, int(beats_per_bar))]
    if bars[-1] < duration_s:
        bars.append(duration_s)
    return bars


def _bar_energies(rms: np.ndarray, hop_s: float, bar_times: List[float]) -> List[float]:
    if hop_s <= 0.0:
        return []
    out: List[float] = []
    for a, b in zip(bar_times, bar_times[1:]):
        i0 = max(0, int(np.floor(a / hop_s)))
        i1 = min(len(rms), max(i0 + 1, int(np.ceil(b / hop_s))))
        out.append(0.0 if i0 >= len(rms) else float(np.mean(rms[i0:i1])))
    return out


def _runs_of_consecutive(indices: List[int]) -> List[List[int]]:
    groups: List[List[int]] = []
    for i in indices:
        if groups and i == groups[-1][-1] + 1:
            groups[-1].append(i)
        else:
            groups.append([i])
    return groups


class _VocalRisk:
    """window RMS (float64 mean of squares) / peak |x| of the vocal stem, clamped to [0, 1]."""

    def __init__(self, vocal_track: Optional[np.ndarray], sample_rate: int, guard_win_ms: float, hip=None, vocal_dev=None):
        self.sr = sample_rate
        self.half = max(1, int(round(sample_rate * float(guard_win_ms) / 1000.0)))
        self.hip, self.dev = hip, vocal_dev
        self.mono = None
        self.n = 0
        self.peak = 0.0
        if vocal_track is None or sample_rate <= 0:
            return
        audio = np.asarray(vocal_track, dtype=np.float32)
        if audio.size == 0:
            return
        self.mono = audio if audio.ndim == 1 else np.mean(audio, axis=-1)
        self.n = len(self.mono)
        if self.dev is not None and self.hip is not None and audio.ndim == 1:
            self.peak = float(self.dev.abs().max().item())
        else:
            self.peak = float(np.max(np.abs(self.mono)))

    def _window(self, t: float):
        c = int(round(float(t) * self.sr))
        return max(0, c - self.half), min(self.n, c + self.half)

    def __call__(self, t: float) -> float:
        if self.mono is None or self.peak <= 1e-9:
            return 0.0
        a, b = self._window(t)
        if a >= b:
            return 0.0
        if self.dev is not None and self.hip is not None:
            ms = self.hip.mean_square(self.dev[a:b])
        else:
            ms = float(np.mean(np.square(self.mono[a:b], dtype=np.float64)))
        return max(0.0, min(1.0, float(np.sqrt(ms)) / self.peak))

    def many(self, times) -> List[float]:
        """`[self(t) for t in times]`; on the device every window's sum of squares comes from ONE launch and one download
        (`ac_window_sum_squares`, bit-identical to the per-window `mean_square`) instead of a host round trip per candidate."""
        times = list(times)
        if (self.mono is None or self.peak <= 1e-9 or self.dev is None or self.hip is None or not times
                or 2 * self.half > self.hip.WINDOW_MEAN_SQUARE_MAX):
            return [self(t) for t in times]
        spans = [self._window(t) for t in times]
        live = [i for i, (a, b) in enumerate(spans) if a < b]
        out = [0.0] * len(times)
        if live:
            ms = self.hip.window_mean_squares(self.dev, [spans[i][0] for i in live], [spans[i][1] for i in live])
            for i, v in zip(live, ms):
                out[i] = max(0.0, min(1.0, float(np.sqrt(float(v))) / self.peak))
        return out


def generate_beat_candidates(*, beat_times: Iterable[float], rms_series: Iterable[float], hop_s: float, duration_s: float,
                             sample_rate: int, vocal_track: Optional[np.ndarray], bars_per_cut: int = 2,
                             base_score: float = 0.3, energy_threshold: Optional[float] = None,
                             min_consecutive_bars: int = 4, guard_win_ms: float = 80.0, hip=None, vocal_dev=None) -> List[CutCandidate]:
    beats = np.asarray(list(beat_times), dtype=np.float32)
    rms = np.asarray(list(rms_series), dtype=np.float32)
    duration_s = float(duration_s)
    if beats.size < 2 or rms.size == 0 or duration_s <= 0.0:
        return []
    bars = _bar_times(beats, duration_s)
    if len(bars) < 2:
        return []
    energies = _bar_energies(rms, float(hop_s), bars)
    if not energies:
        return []
    thr = float(energy_threshold) if energy_threshold is not None else float(np.mean(energies))
    high = detect_chorus_regions(energies, thr, min_consecutive_bars=min_consecutive_bars)
    if not high:
        return []
    risk = _VocalRisk(vocal_track, sample_rate, guard_win_ms, hip=hip, vocal_dev=vocal_dev)
    every = max(1, int(bars_per_cut))
    picked = []                          # (time, bar) of every candidate first: their vocal-risk windows are measured together
    for group in _runs_of_consecutive(sorted(high)):
        for k, bar in enumerate(group):
            if k % every != 0 or bar >= len(bars) - 1:
                continue
            t = float(bars[bar])
            if t <= 0.0 or t >= duration_s:
                continue
            picked.append((t, bar))
    risks = risk.many([t for t, _ in picked])
    return [CutCandidate(t=t, score=float(base_score), source=CandidateSource.BEAT, reasons=["high_energy_beat"],
                         features={"vocal_cut_risk": r}, meta={"bar_index": int(bar), "bars_per_cut": int(bars_per_cut)})
            for (t, bar), r in zip(picked, risks)]


__all__ = ["generate_beat_candidates", "detect_chorus_regions"]
