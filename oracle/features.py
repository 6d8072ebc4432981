"""Oracle for SURVEY.md §8 rows a8-a12: chunked TrackFeatureCache and BPMAnalyzer.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
`src/audio_cut/analysis/features_cache.py:40-91,94-195,254-398` (CPU branch — the
parity target, quirk Q4) and
`src/vocal_smart_splitter/core/adaptive_vad_enhancer.py:27-299` (BPMAnalyzer only)
on top of `oracle.librosa_ops`.  The control logic here is pinned by running the
reference's own `ChunkFeatureBuilder` over the same restated librosa ops
(tests/golden/make_golden.py); the float ops themselves are parity-unpinned.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from . import librosa_ops as L
from .chunking import Plan
from .config import get_config

EPS = 1e-12


@dataclass
class BPMInfo:
    main_bpm: float
    bpm_category: str
    beat_strength: float
    bpm_confidence: float
    tempo_variance: float
    adaptive_factors: Optional[Dict] = None
    beat_positions: Optional[np.ndarray] = None


@dataclass
class FeatureCache:
    sr: int
    hop_length: int
    hop_s: float
    duration_s: float
    rms_series: np.ndarray
    spectral_flatness: np.ndarray
    onset_envelope: np.ndarray
    onset_strength: np.ndarray
    onset_frames: np.ndarray
    rms_max: float
    onset_max: float
    bpm_features: Optional[BPMInfo]
    tempo_curve: Optional[np.ndarray]
    beat_times: np.ndarray
    global_mdd: float
    mdd_series: np.ndarray

    def frame_count(self) -> int:
        return len(self.rms_series)

    def frame_index(self, t: float) -> int:
        if self.hop_s <= 0:
            return 0
        return int(np.clip(int(round(t / self.hop_s)), 0, max(self.frame_count() - 1, 0)))

    def frame_slice(self, start_time: float, end_time: float, pad_frames: int = 0) -> slice:
        a = max(0, self.frame_index(start_time) - pad_frames)
        b = self.frame_index(end_time) + pad_frames + 1
        return slice(a, min(self.frame_count(), max(a + 1, b)))


# ---------------------------------------------------------------------------
# BPMAnalyzer (adaptive_vad_enhancer.py:48-299)
# ---------------------------------------------------------------------------

_BPM_BANDS = (("slow", 50, 80), ("medium", 80, 120), ("fast", 120, 160), ("very_fast", 160, 200))


def _classify_bpm(bpm: float) -> str:
    for name, lo, hi in _BPM_BANDS:
        if lo <= bpm < hi:
            return name
    return "very_slow" if bpm < 50 else "extreme_fast"


def _adaptive_factors(bpm: float, stability: float, variance: float) -> Dict:
    """adaptive_vad_enhancer.py:189-253 (pause multipliers are code defaults: the keys are absent from the YAML)."""
    if bpm < 70:
        f = {"threshold_modifier": -0.05, "min_pause_modifier": 1.5, "min_speech_modifier": 1.2, "sensitivity": "high"}
    elif bpm < 100:
        f = {"threshold_modifier": 0.0, "min_pause_modifier": 1.0, "min_speech_modifier": 1.0, "sensitivity": "medium"}
    elif bpm < 140:
        f = {"threshold_modifier": 0.1, "min_pause_modifier": 0.7, "min_speech_modifier": 0.8, "sensitivity": "low"}
    else:
        f = {"threshold_modifier": 0.15, "min_pause_modifier": 0.7, "min_speech_modifier": 0.6, "sensitivity": "very_low"}
    f["threshold_modifier"] += (1.0 - stability) * 0.1
    f["threshold_modifier"] += variance * 0.05
    f.update({
        "bpm_value": bpm, "stability_score": stability, "variance_score": variance,
        "recommended_window_size": 12.0 if bpm < 70 else (10.0 if bpm < 120 else 8.0),
        "beat_sync_important": bpm > 100,
    })
    return f


def default_bpm_info() -> BPMInfo:
    """adaptive_vad_enhancer.py:272-299."""
    return BPMInfo(110.0, "medium", 0.6, 0.5, 0.2, {
        "threshold_modifier": 0.0, "min_pause_modifier": 1.0, "min_speech_modifier": 1.0, "sensitivity": "medium",
        "bpm_value": 110.0, "stability_score": 0.6, "variance_score": 0.2, "recommended_window_size": 10.0,
        "beat_sync_important": False}, np.array([]))


def beat_stability(beats: np.ndarray) -> float:
    """adaptive_vad_enhancer.py:99-126."""
    if len(beats) < 3:
        return 0.5
    iv = np.diff(beats)
    if len(iv) < 2:
        return 0.5
    m = np.mean(iv)
    if m == 0:
        return 0.5
    return float(np.clip(1.0 - np.std(iv) / m, 0.0, 1.0))


def tempo_variance_from_env(env512: np.ndarray, sr: int) -> float:
    """adaptive_vad_enhancer.py:128-168 given the median-aggregated onset envelope at hop 512."""
    curve = L.tempo(env512, sr=sr, hop_length=512, aggregate=None)
    if len(curve) > 1:
        arr = np.asarray(curve, dtype=np.float64)
        return float(np.clip(float(np.std(arr)) / (float(np.mean(arr)) + 1e-8), 0.0, 1.0))
    return 0.1


def bpm_features(audio: np.ndarray, sr: int) -> BPMInfo:
    """adaptive_vad_enhancer.py:48-97.  Both `beat_track(y=...)` and `_calculate_tempo_variance`
    build the identical median-aggregated hop-512 onset envelope; it is computed once here."""
    try:
        env = L.onset_strength(audio, sr=sr, hop_length=512, aggregate=np.median)
        tempo, beats = L.beat_track(onset_envelope=env, sr=sr, hop_length=512, start_bpm=120.0, tightness=100)
        stab = beat_stability(beats)
        try:
            var = tempo_variance_from_env(env, sr)
        except Exception:
            var = 0.1
        return BPMInfo(tempo, _classify_bpm(tempo), stab, 0.8, var, _adaptive_factors(tempo, stab, var), beats)
    except Exception:
        return default_bpm_info()


# ---------------------------------------------------------------------------
# per-chunk features + stitching (features_cache.py:94-318)
# ---------------------------------------------------------------------------

def mdd_series(rms: np.ndarray, flat: np.ndarray, onset: np.ndarray) -> np.ndarray:
    """features_cache.py:321-335."""
    we = get_config("musical_dynamic_density.energy_weight", 0.5)
    ws = get_config("musical_dynamic_density.spectral_weight", 0.3)
    wo = get_config("musical_dynamic_density.onset_weight", 0.2)
    series = we * (rms / (np.max(rms) + EPS)) + ws * (1.0 - np.clip(flat, 0.0, 1.0)) + wo * (onset / (np.max(onset) + EPS))
    return np.clip(series, 0.0, 1.0)


def chunk_features(mix_chunk: np.ndarray, sr: int, hop: int, frame_len: int) -> Dict[str, np.ndarray]:
    """features_cache.py:181-195."""
    r = L.rms(mix_chunk, frame_length=frame_len, hop_length=hop)[0]
    fl = L.spectral_flatness(mix_chunk, hop_length=hop)[0]
    env = L.onset_strength(mix_chunk, sr=sr, hop_length=hop)
    peaks = L.onset_detect(env, sr=sr, hop_length=hop)
    times = L.frames_to_time(np.arange(len(r)), sr=sr, hop_length=hop)
    return {"rms": r.astype(np.float32, copy=False), "flat": fl.astype(np.float32, copy=False),
            "onset_env": env.astype(np.float32, copy=False), "onset_frames": np.asarray(peaks, dtype=np.int64),
            "frame_times": times.astype(np.float32, copy=False)}


class ChunkFeatureOracle:
    def __init__(self, sr: int, hop_s: float = 0.05):
        self.sr = sr
        self.hop_length = max(1, int(round(sr * hop_s)))
        self.hop_s = float(self.hop_length) / float(sr)
        self.frame_length = max(self.hop_length * 2, int(round(sr * 0.1)))
        self._rms: List[np.ndarray] = []
        self._flat: List[np.ndarray] = []
        self._env: List[np.ndarray] = []
        self._times: List[np.ndarray] = []
        self._onset_frames: List[int] = []
        self._segments: List[np.ndarray] = []

    def add_chunk(self, plan: Plan, mix_chunk: np.ndarray, sr: int) -> None:
        """features_cache.py:122-179."""
        if mix_chunk.size == 0:
            return
        mix_chunk = np.asarray(mix_chunk, dtype=np.float32)
        f = chunk_features(mix_chunk, sr, self.hop_length, self.frame_length)
        times = f["frame_times"] + plan.start_s          # float32 + python float -> float32
        lo, hi = plan.effective_start_s, plan.effective_end_s
        keep = (times >= lo) & (times < hi)
        if not np.any(keep):
            return
        self._rms.append(f["rms"][keep])
        self._flat.append(f["flat"][keep])
        self._env.append(f["onset_env"][keep])
        self._times.append(times[keep])
        base = int(round(plan.start_s / self.hop_s))
        for k in f["onset_frames"]:
            ft = times[k] if k < len(times) else plan.start_s
            if lo <= ft < hi:
                self._onset_frames.append(base + int(k))
        es = int(round(lo * sr)); ee = int(round(hi * sr)); cs = int(round(plan.start_s * sr))
        if ee > es:
            self._segments.append(mix_chunk[es - cs: es - cs + (ee - es)])

    def bpm_input(self, full_mix: np.ndarray) -> np.ndarray:
        """features_cache.py:278 — the concatenation of the (overlapping) effective regions."""
        return np.concatenate(self._segments) if self._segments else full_mix

    def finalize(self, full_mix: np.ndarray) -> FeatureCache:
        """features_cache.py:254-318."""
        if not self._rms:
            return whole_track_cache(full_mix, self.sr, self.hop_s)
        rms = np.concatenate(self._rms)
        flat = np.concatenate(self._flat)
        env = np.concatenate(self._env)
        times = np.concatenate(self._times)
        idx = np.round(times / self.hop_s).astype(int)
        uniq, first = np.unique(idx, return_index=True)
        rms = rms[first].astype(np.float32, copy=False)
        flat = flat[first].astype(np.float32, copy=False)
        env = env[first].astype(np.float32, copy=False)
        marked = set(self._onset_frames)
        onset_frames = np.array(sorted(i for i in uniq if i in marked), dtype=int)
        return _assemble(self.sr, self.hop_length, self.hop_s, full_mix, self.bpm_input(full_mix), rms, flat, env, onset_frames)


def _assemble(sr, hop_length, hop_s, full_mix, bpm_wave, rms, flat, env, onset_frames) -> FeatureCache:
    bpm = bpm_features(bpm_wave, sr)
    curve = L.tempo(env, sr=sr, hop_length=hop_length, aggregate=None)
    _, beat_frames = L.beat_track(onset_envelope=env, sr=sr, hop_length=hop_length)
    beat_times = L.frames_to_time(beat_frames, sr=sr, hop_length=hop_length)
    strength = env.copy()
    mdd = mdd_series(rms, flat, strength)
    return FeatureCache(
        sr=sr, hop_length=hop_length, hop_s=hop_s, duration_s=len(full_mix) / float(sr),
        rms_series=rms, spectral_flatness=flat, onset_envelope=env, onset_strength=strength,
        onset_frames=onset_frames, rms_max=float(np.max(rms) if rms.size else 0.0),
        onset_max=float(np.max(strength) if strength.size else 0.0), bpm_features=bpm,
        tempo_curve=curve, beat_times=beat_times, global_mdd=float(np.mean(mdd)), mdd_series=mdd)


def whole_track_cache(mix: np.ndarray, sr: int, hop_s: float = 0.05) -> FeatureCache:
    """features_cache.py:355-398 (`_build_feature_cache_numpy`; the torch variant at :413-480 discards
    its STFT and calls the same librosa ops)."""
    hop = max(1, int(round(sr * hop_s)))
    frame_len = max(hop * 2, int(round(sr * 0.1)))
    r = L.rms(mix, frame_length=frame_len, hop_length=hop)[0]
    fl = L.spectral_flatness(mix, hop_length=hop)[0]
    env = L.onset_strength(mix, sr=sr, hop_length=hop)
    peaks = L.onset_detect(env, sr=sr, hop_length=hop)
    return _assemble(sr, hop, hop_s, mix, mix, r, fl, env, np.asarray(peaks))
