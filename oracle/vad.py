"""Oracle for SURVEY.md §8 row a13: chunked VAD bookkeeping + the default energy-gate VAD.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
`src/audio_cut/detectors/silero_chunk_vad.py:27-186` (halo clipping incl. quirk Q6 at `:106-109`,
cross-chunk merge `:119-136`, focus windows `:152-182`); pinned against that module itself
(importable here) with an injected `inference_fn` by tests/golden/make_golden.py.

Silero VAD weights / the `silero_vad` package are unavailable offline, so the
default `inference_fn` of this build is an *energy gate* shaped like Silero's contract
(`vocal_pause_detector.py:175-296`: windows of 512 samples @16 kHz, hysteresis with
threshold 0.35 / neg-threshold 0.20, min speech 250 ms, min silence 700 ms, pad 150 ms, result in
track-rate sample indices).  `speech_timestamps` restates the published post-processing of
`silero_vad.get_speech_timestamps` (third-party, **parity unpinned**); `energy_probs` is this
build's own deterministic stand-in for the network and is declared as such in DESIGN.md.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .chunking import Plan

VadFn = Callable[[np.ndarray], Sequence[Dict[str, int]]]


def vad_window(sr: int) -> int:
    return int(round(512 * sr / 16000.0))


def energy_probs(audio: np.ndarray, sr: int, floor_db: float = -60.0, ceil_db: float = -30.0) -> np.ndarray:
    """Per-window pseudo speech probability: clip((rms_dB - floor) / (ceil - floor), 0, 1); last window zero-padded."""
    win = vad_window(sr)
    n = len(audio)
    n_win = (n + win - 1) // win
    padded = np.zeros(n_win * win, dtype=np.float32)
    padded[:n] = audio
    rms = np.sqrt(np.mean(np.square(padded.reshape(n_win, win).astype(np.float64)), axis=1)).astype(np.float32)
    db = 20.0 * np.log10(rms.astype(np.float64) + 1e-12)
    return np.clip((db - floor_db) / (ceil_db - floor_db), 0.0, 1.0)


def speech_timestamps(probs: np.ndarray, n_samples: int, win: int, sr: int, threshold: float = 0.35,
                      min_speech_ms: float = 250, min_silence_ms: float = 700, pad_ms: float = 150) -> List[Dict[str, int]]:
    """Hysteresis post-processing of silero_vad.get_speech_timestamps (max_speech_duration = inf)."""
    min_speech = sr * min_speech_ms / 1000.0
    pad = sr * pad_ms / 1000.0
    min_silence = sr * min_silence_ms / 1000.0
    neg = max(threshold - 0.15, 0.01)          # silero_vad.get_speech_timestamps: neg_threshold = max(threshold - 0.15, 0.01)
    triggered = False
    speeches: List[Dict[str, int]] = []
    cur: Dict[str, int] = {}
    temp_end = 0
    for i, p in enumerate(probs):
        pos = win * i
        if p >= threshold and temp_end:
            temp_end = 0
        if p >= threshold and not triggered:
            triggered = True
            cur = {"start": pos}
            continue
        if p < neg and triggered:
            if not temp_end:
                temp_end = pos
            if pos - temp_end < min_silence:
                continue
            cur["end"] = temp_end
            if cur["end"] - cur["start"] > min_speech:
                speeches.append(cur)
            cur = {}
            temp_end = 0
            triggered = False
    if cur and (n_samples - cur["start"]) > min_speech:
        cur["end"] = n_samples
        speeches.append(cur)
    for i, sp in enumerate(speeches):
        if i == 0:
            sp["start"] = int(max(0, sp["start"] - pad))
        if i != len(speeches) - 1:
            gap = speeches[i + 1]["start"] - sp["end"]
            if gap < 2 * pad:
                sp["end"] += int(gap // 2)
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - gap // 2))
            else:
                sp["end"] = int(min(n_samples, sp["end"] + pad))
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - pad))
        else:
            sp["end"] = int(min(n_samples, sp["end"] + pad))
    return speeches


def energy_gate_vad(sr: int) -> VadFn:
    win = vad_window(sr)

    def _fn(audio: np.ndarray) -> List[Dict[str, int]]:
        probs = energy_probs(np.asarray(audio, dtype=np.float32), sr)
        return speech_timestamps(probs, len(audio), win, sr)

    return _fn


class ChunkVadOracle:
    """silero_chunk_vad.py:27-186."""

    def __init__(self, sample_rate: int, merge_gap_ms: float = 120.0, focus_pad_s: float = 0.2,
                 inference_fn: Optional[VadFn] = None):
        self.sample_rate = sample_rate
        self.merge_gap_ms = merge_gap_ms
        self.focus_pad_s = focus_pad_s
        self.inference_fn = inference_fn or (lambda audio: [])
        self._segments: List[Tuple[float, float]] = []
        self._track_duration_s = 0.0

    def process_chunk(self, plan: Plan, vocal_chunk: np.ndarray, sr: int) -> None:
        if vocal_chunk.size == 0:
            return
        stamps = self.inference_fn(vocal_chunk)
        lo, hi, base = plan.effective_start_s, plan.effective_end_s, plan.start_s
        self._track_duration_s = max(self._track_duration_s, float(plan.end_s))
        for ts in stamps:
            a = int(ts.get("start", 0)); b = int(ts.get("end", 0))
            if b <= a:
                continue
            s = base + a / float(self.sample_rate)
            e = base + b / float(self.sample_rate)
            if e <= lo or s >= hi:
                continue
            s_adj = s if (s < lo < e) else max(s, lo)       # quirk Q6: a straddling span keeps its start
            e_adj = min(e, hi)
            if e_adj - s_adj <= 1e-6:
                continue
            self._segments.append((s_adj, e_adj))
        self._segments.sort(key=lambda it: it[0])

    def merged(self) -> List[Tuple[float, float]]:
        out: List[Tuple[float, float]] = []
        gap = float(self.merge_gap_ms) / 1000.0
        for s, e in self._segments:
            if e <= s:
                continue
            if out and s - out[-1][1] <= gap:
                out[-1] = (out[-1][0], max(out[-1][1], e))
            else:
                out.append((s, e))
        return out

    def finalize(self) -> List[Dict[str, float]]:
        return [{"start": float(s), "end": float(e), "duration": float(max(0.0, e - s))} for s, e in self.merged()]

    def to_focus_windows(self, pad_s: Optional[float] = None, min_width_s: float = 0.0) -> List[Tuple[float, float]]:
        segs = self.merged()
        if not segs:
            return []
        pad = max(0.0, float(self.focus_pad_s if pad_s is None else pad_s))
        track_end = max(self._track_duration_s, max(e for _, e in segs))
        wins = sorted(((max(0.0, s - pad), min(track_end, e + pad)) for s, e in segs), key=lambda it: it[0])
        out: List[Tuple[float, float]] = []
        for s, e in wins:
            if e - s <= 0.0:
                continue
            if not out or s > out[-1][1]:
                out.append((s, e))
            else:
                out[-1] = (out[-1][0], max(out[-1][1], e))
        if min_width_s > 0.0:
            out = [(s, e) for s, e in out if (e - s) >= min_width_s]
        return out
