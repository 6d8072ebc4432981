"""Chunked VAD bookkeeping — drop-in for the reference's `src/audio_cut/detectors/silero_chunk_vad.py`
(`SileroChunkVAD(sample_rate, merge_gap_ms=120.0, focus_pad_s=0.2, inference_fn=None)`,
`process_chunk / finalize / to_focus_windows / build_focus_windows`, `VadFn` contract `:24,95-102`).

The default `inference_fn` is `detectors.silero_vad.default_vad`: the Silero VAD network on the HIP kernels
(`SileroHipVad`: 44.1 -> 16 kHz resampling, 4096 bucket padding, 512-sample windows with state carry,
`vocal_pause_detector.py:175-296`) when a weights file is configured (`advanced_vad.silero_weights_path` /
`AUDIOCUT_SILERO_WEIGHTS`; the `silero_vad` package and its weights cannot be obtained offline), else the
explicit NO-WEIGHTS mode `EnergyGateVad`: a HIP framed-RMS kernel over 512-samples-at-16-kHz-equivalent windows.
Both end in Silero's published hysteresis post-processing with the reference's parameters
(`vocal_pause_detector.py:208-213`: threshold 0.35, min speech 250 ms, min silence 700 ms, pad 150 ms).
Callers may still inject their own `inference_fn`, exactly as the reference allows (`silero_chunk_vad.py:34`).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _native
from ..config import get_config
from ..utils.gpu_pipeline import ChunkPlan

logger = logging.getLogger(__name__)

VadFn = Callable[[np.ndarray], Sequence[Dict[str, int]]]


def speech_timestamps(probs: np.ndarray, n_samples: int, win: int, sr: int, threshold: float, min_speech_ms: float,
                      min_silence_ms: float, pad_ms: float) -> List[Dict[str, int]]:
    """Hysteresis of silero_vad.get_speech_timestamps (neg_threshold = threshold - 0.15, no max duration)."""
    min_speech = sr * min_speech_ms / 1000.0
    pad = sr * pad_ms / 1000.0
    min_silence = sr * min_silence_ms / 1000.0
    neg = max(threshold - 0.15, 0.01)          # silero_vad.get_speech_timestamps: neg_threshold = max(threshold - 0.15, 0.01)
    triggered = False
    spans: List[Dict[str, int]] = []
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
                spans.append(cur)
            cur = {}
            temp_end = 0
            triggered = False
    if cur and (n_samples - cur["start"]) > min_speech:
        cur["end"] = n_samples
        spans.append(cur)
    for i, sp in enumerate(spans):
        if i == 0:
            sp["start"] = int(max(0, sp["start"] - pad))
        if i != len(spans) - 1:
            gap = spans[i + 1]["start"] - sp["end"]
            if gap < 2 * pad:
                sp["end"] += int(gap // 2)
                spans[i + 1]["start"] = int(max(0, spans[i + 1]["start"] - gap // 2))
            else:
                sp["end"] = int(min(n_samples, sp["end"] + pad))
                spans[i + 1]["start"] = int(max(0, spans[i + 1]["start"] - pad))
        else:
            sp["end"] = int(min(n_samples, sp["end"] + pad))
    return spans


class PrecomputedChunk:
    """A chunk whose VAD window RMS values are already on the host (`EnergyGateVad.batch_rms`)."""

    def __init__(self, rms: np.ndarray, n_samples: int) -> None:
        self.rms, self.n = rms, int(n_samples)

    def numel(self) -> int:
        return self.n


class EnergyGateVad:
    """Default `inference_fn`: per-window RMS on the GPU -> pseudo probability -> Silero hysteresis.
    Accepts a host chunk (the `VadFn` contract) or a device tensor (the separator's fast path)."""

    def __init__(self, sample_rate: int, ctx: Optional["_native.Context"] = None, floor_db: float = -60.0, ceil_db: float = -30.0):
        self.sample_rate = sample_rate
        self.win = int(round(512 * sample_rate / 16000.0))
        self.floor_db, self.ceil_db = floor_db, ceil_db
        self._ctx = ctx

    def precompute(self, packed_dev, offsets: Sequence[int], lengths: Sequence[int]) -> List["PrecomputedChunk"]:
        """The batched interface `EnhancedVocalSeparator` looks for (shared with `SileroHipVad`)."""
        return self.batch_rms(packed_dev, offsets, lengths)

    def batch_rms(self, packed_dev, offsets: Sequence[int], lengths: Sequence[int]) -> List["PrecomputedChunk"]:
        """All chunks of a track in ONE launch and one download: `packed_dev` holds the chunks back to back."""
        ctx = self._ctx or _native.Context()
        self._ctx = ctx
        a = np.asarray(offsets, dtype=np.int64)
        b = a + np.asarray(lengths, dtype=np.int64)
        rms = ctx.segment_frame_rms(packed_dev, a, b, self.win, self.win, center=False)
        return [PrecomputedChunk(r, int(n)) for r, n in zip(rms, lengths)]

    def probs(self, chunk) -> Tuple[np.ndarray, int]:
        import torch
        if isinstance(chunk, PrecomputedChunk):
            db = 20.0 * np.log10(chunk.rms.astype(np.float64) + 1e-12)
            return np.clip((db - self.floor_db) / (self.ceil_db - self.floor_db), 0.0, 1.0), chunk.n
        ctx = self._ctx or _native.Context()
        self._ctx = ctx
        x = chunk if isinstance(chunk, torch.Tensor) else ctx.to_device(np.asarray(chunk, dtype=np.float32))
        n = x.numel()
        n_win = (n + self.win - 1) // self.win
        if n_win * self.win != n:                       # zero-pad the last window
            padded = torch.zeros(n_win * self.win, dtype=torch.float32, device=x.device)
            padded[:n] = x
            x = padded
        rms = ctx.frame_rms(x, self.win, self.win, center=False).cpu().numpy()
        db = 20.0 * np.log10(rms.astype(np.float64) + 1e-12)
        return np.clip((db - self.floor_db) / (self.ceil_db - self.floor_db), 0.0, 1.0), n

    def __call__(self, chunk) -> List[Dict[str, int]]:
        probs, n = self.probs(chunk)
        return speech_timestamps(
            probs, n, self.win, self.sample_rate,
            threshold=get_config("advanced_vad.silero_prob_threshold_down", 0.35),
            min_speech_ms=get_config("advanced_vad.silero_min_speech_ms", 250),
            min_silence_ms=get_config("advanced_vad.silero_min_silence_ms", 700),
            pad_ms=get_config("advanced_vad.silero_speech_pad_ms", 150))


@dataclass
class SileroChunkVAD:
    """Incremental VAD on the ChunkPlan schedule, global timeline out (reference `:27-186`)."""

    sample_rate: int
    merge_gap_ms: float = 120.0
    focus_pad_s: float = 0.2
    inference_fn: Optional[VadFn] = None

    _segments: List[Tuple[float, float]] = field(default_factory=list, init=False)
    _track_duration_s: float = field(default=0.0, init=False)
    _finalized: Optional[List[Dict[str, float]]] = field(default=None, init=False)

    def _ensure_inference_fn(self) -> VadFn:
        if self.inference_fn is None:
            from .silero_vad import default_vad
            self.inference_fn = default_vad(self.sample_rate)
        return self.inference_fn

    def process_chunk(self, plan: ChunkPlan, vocal_chunk, sr: int, *, stream=None) -> None:
        if vocal_chunk is None or (hasattr(vocal_chunk, "numel") and vocal_chunk.numel() == 0) or getattr(vocal_chunk, "size", 1) == 0:
            return
        if sr != self.sample_rate:
            raise ValueError(f"SileroChunkVAD sr mismatch: expected {self.sample_rate}, got {sr}")
        fn = self._ensure_inference_fn()
        try:
            stamps = fn(vocal_chunk)
        except _native.NativeError:
            raise
        except Exception as exc:  # reference: log and skip the chunk (`:86-88`)
            logger.error("SileroChunkVAD chunk inference failed: %s", exc, exc_info=True)
            return
        lo, hi, base = plan.effective_start_s, plan.effective_end_s, plan.start_s
        self._track_duration_s = max(self._track_duration_s, float(plan.end_s))
        for ts in stamps:
            a = int(ts.get("start", 0)); b = int(ts.get("end", 0))
            if b <= a:
                continue
            s = base + (a / float(self.sample_rate))
            e = base + (b / float(self.sample_rate))
            if e <= lo or s >= hi:
                continue
            s_adj = s if (s < lo < e) else max(s, lo)        # quirk Q6 (`:106-109`)
            e_adj = min(e, hi)
            if e_adj - s_adj <= 1e-6:
                continue
            self._segments.append((s_adj, e_adj))
        self._segments.sort(key=lambda it: it[0])
        self._finalized = None

    def _merge_segments(self) -> List[Tuple[float, float]]:
        merged: List[Tuple[float, float]] = []
        gap = float(self.merge_gap_ms) / 1000.0
        for s, e in self._segments:
            if e <= s:
                continue
            if merged and s - merged[-1][1] <= gap:
                merged[-1] = (merged[-1][0], max(merged[-1][1], e))
            else:
                merged.append((s, e))
        return merged

    def finalize(self) -> List[Dict[str, float]]:
        if self._finalized is None:
            self._finalized = [{"start": float(s), "end": float(e), "duration": float(max(0.0, e - s))}
                               for s, e in self._merge_segments()]
        return list(self._finalized or [])

    def to_focus_windows(self, *, pad_s: Optional[float] = None, min_width_s: float = 0.0) -> List[Tuple[float, float]]:
        segs = self._merge_segments() if self._finalized is None else [
            (float(d.get("start", 0.0)), float(d.get("end", 0.0))) for d in self._finalized]
        if not segs:
            return []
        pad = max(0.0, float(self.focus_pad_s if pad_s is None else pad_s))
        min_w = max(0.0, float(min_width_s))
        track_end = max(self._track_duration_s, max(e for _, e in segs))
        wins = []
        for s, e in segs:
            l, r = max(0.0, s - pad), min(track_end, e + pad)
            if r - l > 0.0:
                wins.append((l, r))
        wins.sort(key=lambda it: it[0])
        out: List[Tuple[float, float]] = []
        for s, e in wins:
            if not out or s > out[-1][1]:
                out.append((s, e))
            else:
                out[-1] = (out[-1][0], max(out[-1][1], e))
        if min_w > 0.0:
            out = [(s, e) for s, e in out if (e - s) >= min_w]
        return out

    def build_focus_windows(self) -> List[Tuple[float, float]]:
        return self.to_focus_windows(pad_s=self.focus_pad_s)


__all__ = ["SileroChunkVAD", "EnergyGateVad", "VadFn", "speech_timestamps"]
