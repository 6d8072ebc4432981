"""Tempo / beat / onset-peak logic over device-computed onset envelopes.

The per-sample and per-frame arithmetic (STFT, mel, log-power differences, the windowed
autocorrelation tempogram) runs in HIP kernels (`_native.Context.stft2048_features`,
`.onset_strength`, `.tempogram_reduce`).  What is left here is scalar host logic over a few
thousand frames: the log-normal tempo prior + argmax, librosa's peak picking, and the beat
tracker whose dynamic programme is sequential by construction and runs in C++ on the host
(`ac_host_beat_dp`).  Algorithms follow the published librosa 0.10 definitions that the reference
calls at `features_cache.py:184-187,283-294`, `adaptive_vad_enhancer.py:61-67,143-156`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import scipy.signal

from .. import _native
from ..config import get_config


@dataclass
class BPMFeatures:
    """Field-for-field mirror of `adaptive_vad_enhancer.py:16-25`."""

    main_bpm: float
    bpm_category: str
    beat_strength: float
    bpm_confidence: float
    tempo_variance: float
    adaptive_factors: Dict = None
    beat_positions: np.ndarray = None


# ---------------------------------------------------------------------------- onset peaks
def onset_detect(envelope: np.ndarray, sr: int, hop_length: int) -> np.ndarray:
    """librosa.onset.onset_detect defaults (normalise to [0,1], 30 ms / 100 ms windows, delta 0.07)."""
    env = np.asarray(envelope)
    if not env.any() or not np.all(np.isfinite(env)):
        return np.array([], dtype=int)
    env = env - np.min(env)
    env = env / (np.max(env) + np.finfo(env.dtype).tiny)
    pre_max = int(np.ceil(0.03 * sr // hop_length)); post_max = int(np.ceil(0.00 * sr // hop_length + 1))
    pre_avg = int(np.ceil(0.10 * sr // hop_length)); post_avg = int(np.ceil(0.10 * sr // hop_length + 1))
    wait = int(np.ceil(0.03 * sr // hop_length)); delta = 0.07
    n = env.shape[0]
    # sliding max / mean over [i - pre, i + post), truncated at the edges.  Interior frames (full windows)
    # are evaluated through a strided window view — row-wise np.max / np.mean give the same values as the
    # per-frame slices — and only the <= pre+post edge frames go through the scalar loop.
    is_max = np.empty(n, dtype=bool)
    means = np.empty(n, dtype=env.dtype)

    def _edge(i: int) -> None:
        is_max[i] = env[i] == np.max(env[max(0, i - pre_max): i + post_max])
        means[i] = np.mean(env[max(0, i - pre_avg): i + post_avg])

    w_max, w_avg = pre_max + post_max, pre_avg + post_avg
    lo = max(pre_max, pre_avg)
    hi = n - max(post_max, post_avg) + 1            # frames [lo, hi) have full windows
    if hi > lo and n >= max(w_max, w_avg):
        from numpy.lib.stride_tricks import sliding_window_view
        mx = sliding_window_view(env, w_max)[lo - pre_max: hi - pre_max].max(axis=1)
        is_max[lo:hi] = env[lo:hi] == mx
        means[lo:hi] = sliding_window_view(env, w_avg)[lo - pre_avg: hi - pre_avg].mean(axis=1)
        for i in list(range(0, lo)) + list(range(hi, n)):
            _edge(i)
    else:
        for i in range(n):
            _edge(i)
    cand = np.flatnonzero(is_max & (env >= means + delta) & (env != 0))
    peaks = []
    last = -np.inf
    for i in cand:
        if i > last + wait:
            peaks.append(int(i))
            last = i
    return np.array(peaks, dtype=int)


# ---------------------------------------------------------------------------- tempo
def tempo_frequencies(n_bins: int, hop_length: int, sr: float) -> np.ndarray:
    out = np.zeros(int(n_bins), dtype=np.float64)
    out[0] = np.inf
    out[1:] = 60.0 * sr / (hop_length * np.arange(1.0, n_bins))
    return out


def tempo_logprior(win: int, hop_length: int, sr: float, start_bpm: float = 120.0, std_bpm: float = 1.0,
                   max_tempo: float = 320.0) -> Tuple[np.ndarray, np.ndarray]:
    bpms = tempo_frequencies(win, hop_length, sr)
    with np.errstate(divide="ignore"):
        lp = -0.5 * ((np.log2(bpms) - np.log2(start_bpm)) / std_bpm) ** 2
    lp[: int(np.argmax(bpms < max_tempo))] = -np.inf
    return bpms, lp


def tempo_from_device(ctx: "_native.Context", env_dev, sr: int, hop_length: int, *, start_bpm: float = 120.0,
                      want_curve: bool = False):
    """(tempo of the time-averaged tempogram, optional per-frame tempo curve) — one kernel launch."""
    win = int(np.floor(int(8.0 * sr) // hop_length))          # time_to_frames(8 s)
    bpms, lp = tempo_logprior(win, hop_length, sr, start_bpm)
    mean, arg = ctx.tempogram_reduce(env_dev, win, lp, want_argmax=want_curve)
    tg = mean.cpu().numpy()
    best = int(np.argmax(np.log1p(1e6 * tg) + lp))
    curve = np.take(bpms, arg.cpu().numpy()) if want_curve else None
    return float(bpms[best]), curve


# ---------------------------------------------------------------------------- beat tracking
def _local_score(env: np.ndarray, period: int) -> np.ndarray:
    window = np.exp(-0.5 * (np.arange(-period, period + 1) * 32.0 / period) ** 2)
    return scipy.signal.convolve(env / env.std(ddof=1), window, "same")


def _last_beat(cumscore: np.ndarray) -> int:
    padded = np.pad(cumscore, (1, 1), mode="edge")
    is_max = (cumscore > padded[:-2]) & (cumscore >= padded[2:])
    med = np.median(cumscore[np.argwhere(is_max)])
    return int(np.argwhere(cumscore * is_max * 2 > med).max())


def beat_frames(env: np.ndarray, bpm: float, sr: int, hop_length: int, tightness: float = 100.0, trim: bool = True) -> np.ndarray:
    """librosa's dynamic-programming beat tracker on a host envelope (DP in C++: ac_host_beat_dp)."""
    period = round(60.0 * (float(sr) / hop_length) / bpm)
    score = _local_score(np.asarray(env), period)
    back, cum = _native.host_beat_dp(score, float(period), float(tightness))
    beats = [_last_beat(cum)]
    while back[beats[-1]] >= 0:
        beats.append(int(back[beats[-1]]))
    beats = np.array(beats[::-1], dtype=int)
    smooth = scipy.signal.convolve(score[beats], scipy.signal.windows.hann(5), "same")
    thr = 0.5 * ((smooth ** 2).mean() ** 0.5) if trim else 0.0
    valid = np.argwhere(smooth > thr)
    return beats[valid.min(): valid.max()]


def beat_track_from_device(ctx, env_dev, sr: int, hop_length: int, *, start_bpm: float = 120.0, tightness: float = 100.0):
    env = env_dev.cpu().numpy()
    if not env.any():
        return 0.0, np.array([], dtype=int), env
    bpm, _ = tempo_from_device(ctx, env_dev, sr, hop_length, start_bpm=start_bpm)
    return bpm, beat_frames(env, bpm, sr, hop_length, tightness), env


# ---------------------------------------------------------------------------- BPMAnalyzer
_BANDS = (("slow", 50, 80), ("medium", 80, 120), ("fast", 120, 160), ("very_fast", 160, 200))


def _classify(bpm: float) -> str:
    for name, lo, hi in _BANDS:
        if lo <= bpm < hi:
            return name
    return "very_slow" if bpm < 50 else "extreme_fast"


def _adaptive_factors(bpm: float, stability: float, variance: float) -> Dict:
    """`adaptive_vad_enhancer.py:189-253`."""
    base = "vocal_pause_splitting.bpm_adaptive_settings.pause_duration_multipliers."
    if bpm < 70:
        f = {"threshold_modifier": -0.05, "min_pause_modifier": get_config(base + "slow_song_multiplier", 1.5),
             "min_speech_modifier": 1.2, "sensitivity": "high"}
    elif bpm < 100:
        f = {"threshold_modifier": 0.0, "min_pause_modifier": get_config(base + "medium_song_multiplier", 1.0),
             "min_speech_modifier": 1.0, "sensitivity": "medium"}
    elif bpm < 140:
        f = {"threshold_modifier": 0.1, "min_pause_modifier": get_config(base + "fast_song_multiplier", 0.7),
             "min_speech_modifier": 0.8, "sensitivity": "low"}
    else:
        f = {"threshold_modifier": 0.15, "min_pause_modifier": get_config(base + "fast_song_multiplier", 0.7),
             "min_speech_modifier": 0.6, "sensitivity": "very_low"}
    f["threshold_modifier"] += (1.0 - stability) * 0.1
    f["threshold_modifier"] += variance * 0.05
    f.update({"bpm_value": bpm, "stability_score": stability, "variance_score": variance,
              "recommended_window_size": 12.0 if bpm < 70 else (10.0 if bpm < 120 else 8.0),
              "beat_sync_important": bpm > 100})
    return f


def default_bpm_features() -> BPMFeatures:
    """`adaptive_vad_enhancer.py:272-299`: what the reference returns when analysis raises."""
    return BPMFeatures(110.0, "medium", 0.6, 0.5, 0.2, {
        "threshold_modifier": 0.0, "min_pause_modifier": 1.0, "min_speech_modifier": 1.0, "sensitivity": "medium",
        "bpm_value": 110.0, "stability_score": 0.6, "variance_score": 0.2, "recommended_window_size": 10.0,
        "beat_sync_important": False}, np.array([]))


class BPMAnalyzer:
    """`adaptive_vad_enhancer.py:27-299` on device features.  `extract_bpm_features_device` takes the
    track already resident in HBM; the hop-512 median onset envelope that the reference builds twice
    (`:61` via beat_track and `:143`) is built once."""

    def __init__(self, sample_rate: int = 44100, ctx: Optional["_native.Context"] = None):
        self.sample_rate = sample_rate
        self._ctx = ctx

    def extract_bpm_features_device(self, ctx, wave_dev) -> BPMFeatures:
        sr = self.sample_rate
        try:
            _, mel = ctx.stft2048_features(wave_dev, 512, want_flat=False, want_mel=True)
            env_dev = ctx.onset_strength(mel, 512, "median")
            del mel
            env = env_dev.cpu().numpy()
            if not env.any():
                tempo, beats = 0.0, np.array([], dtype=int)
                variance = self._variance(ctx, env_dev, env)
            else:
                win = int(np.floor(int(8.0 * sr) // 512))
                bpms, lp = tempo_logprior(win, 512, sr, 120.0)
                mean, arg = ctx.tempogram_reduce(env_dev, win, lp, want_argmax=True)
                tg = mean.cpu().numpy()
                tempo = float(bpms[int(np.argmax(np.log1p(1e6 * tg) + lp))])
                beats = beat_frames(env, tempo, sr, 512, 100)
                curve = np.take(bpms, arg.cpu().numpy())
                variance = self._variance_from_curve(curve)
            stability = self._stability(beats)
            return BPMFeatures(tempo, _classify(tempo), stability, 0.8, variance,
                               _adaptive_factors(tempo, stability, variance), beats)
        except _native.NativeError:
            raise
        except Exception:
            return default_bpm_features()

    def extract_bpm_features(self, audio: np.ndarray) -> BPMFeatures:
        ctx = self._ctx or _native.Context()
        return self.extract_bpm_features_device(ctx, ctx.to_device(np.asarray(audio, dtype=np.float32)))

    @staticmethod
    def _stability(beats: np.ndarray) -> float:
        if len(beats) < 3:
            return 0.5
        iv = np.diff(beats)
        if len(iv) < 2:
            return 0.5
        m = np.mean(iv)
        if m == 0:
            return 0.5
        return float(np.clip(1.0 - np.std(iv) / m, 0.0, 1.0))

    @staticmethod
    def _variance_from_curve(curve: np.ndarray) -> float:
        if len(curve) > 1:
            arr = np.asarray(curve, dtype=np.float64)
            return float(np.clip(float(np.std(arr)) / (float(np.mean(arr)) + 1e-8), 0.0, 1.0))
        return 0.1

    def _variance(self, ctx, env_dev, env) -> float:
        try:
            win = int(np.floor(int(8.0 * self.sample_rate) // 512))
            bpms, lp = tempo_logprior(win, 512, self.sample_rate, 120.0)
            _, arg = ctx.tempogram_reduce(env_dev, win, lp, want_argmax=True)
            return self._variance_from_curve(np.take(bpms, arg.cpu().numpy()))
        except Exception:
            return 0.1
