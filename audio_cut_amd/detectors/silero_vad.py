"""Silero VAD on the GPU - the `inference_fn` behind `SileroChunkVAD`.

Replaces `VocalPauseDetectorV2._init_silero_vad` + `_detect_speech_timestamps`
(`src/vocal_smart_splitter/core/vocal_pause_detector.py:74-123,175-296`):

  reference, per chunk on the CPU                               here, all chunks of a track at once
  -----------------------------------------------------------  ------------------------------------------------------------
  librosa.resample 44.1 kHz -> 16 kHz (`:189`)                   ac_resample_poly_segments (one launch, scipy.signal.resample_poly
                                                                 design: soxr_hq cannot be pinned offline)
  np.pad to the 4096 bucket (`:192-196`)                         the segmented resampler writes into a bucket-padded layout
  silero_vad.get_speech_timestamps -> model(window) x N (`:221`) ac_silero_frontend (windows in parallel) + ac_silero_lstm (one
                                                                 workgroup per chunk walks time) + ac_silero_out
  hysteresis / min durations / padding (inside silero_vad)       `speech_timestamps` on the host over the downloaded probabilities
  clamp to the unpadded length, `int(idx * sr / 16000)` (`:268-296`) the same integers

Weights.  The network is the published Silero VAD v5 16 kHz model; its tensors (TorchScript state-dict names, an optional
`_model.` prefix is dropped) are

  stft.forward_basis_buffer [258, 1, 256]; encoder.{0..3}.reparam_conv.weight / .bias ([128,129,3], [64,128,3], [64,64,3],
  [128,64,3]); decoder.rnn.weight_ih / weight_hh [512, 128], bias_ih / bias_hh [512]; decoder.decoder.2.weight [1, 128, 1], .bias [1]

read from an operator-supplied `.npz` (those names as keys) or `.onnx` whose initializers carry those names
(`separation/onnx_weights.read_onnx_graph`).  The `silero_vad` package and its weight files are not available offline, so the
mapping from the published ONNX export (which wraps the 16 kHz and 8 kHz graphs in an `If` node) has only been exercised against
files written by the in-tree writer (tests/onnx_writer.py); an `.npz` made with
`numpy.savez(path, **{k: v.numpy() for k, v in torch.jit.load('silero_vad.jit').state_dict().items()})` is the supported route.
Without a weights file the explicit no-weights mode is `EnergyGateVad` (detectors/silero_chunk_vad.py).
"""
from __future__ import annotations

import logging
import math
import threading
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

from .. import _native
from ..config import get_config

logger = logging.getLogger(__name__)

WINDOW = 512
CONTEXT = 64
SR16 = 16000

TENSOR_SHAPES: Dict[str, tuple] = {
    "stft.forward_basis_buffer": (258, 1, 256),
    "encoder.0.reparam_conv.weight": (128, 129, 3), "encoder.0.reparam_conv.bias": (128,),
    "encoder.1.reparam_conv.weight": (64, 128, 3), "encoder.1.reparam_conv.bias": (64,),
    "encoder.2.reparam_conv.weight": (64, 64, 3), "encoder.2.reparam_conv.bias": (64,),
    "encoder.3.reparam_conv.weight": (128, 64, 3), "encoder.3.reparam_conv.bias": (128,),
    "decoder.rnn.weight_ih": (512, 128), "decoder.rnn.weight_hh": (512, 128),
    "decoder.rnn.bias_ih": (512,), "decoder.rnn.bias_hh": (512,),
    "decoder.decoder.2.weight": (1, 128, 1), "decoder.decoder.2.bias": (1,),
}


def validate_silero_weights(raw: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Canonical names, float32, shapes checked; raises ValueError naming what is missing or mis-shaped."""
    w: Dict[str, np.ndarray] = {}
    for key, val in raw.items():
        name = str(key)
        for prefix in ("_model.", "model."):
            if name.startswith(prefix):
                name = name[len(prefix):]
        w[name] = np.asarray(val)
    out: Dict[str, np.ndarray] = {}
    for name, shape in TENSOR_SHAPES.items():
        if name not in w:
            raise ValueError(f"Silero VAD weights: tensor {name!r} is missing (have {sorted(w)[:6]} ...)")
        arr = np.ascontiguousarray(w[name], dtype=np.float32)
        if arr.shape != shape:
            raise ValueError(f"Silero VAD weights: {name} has shape {arr.shape}, expected {shape}")
        out[name] = arr
    return out


def load_silero_weights(path) -> Dict[str, np.ndarray]:
    path = Path(path)
    if path.suffix.lower() == ".npz":
        with np.load(path) as z:
            return validate_silero_weights({k: z[k] for k in z.files})
    if path.suffix.lower() == ".onnx":
        from ..separation.onnx_weights import read_onnx_graph
        _, inits = read_onnx_graph(path)
        return validate_silero_weights(inits)
    raise ValueError(f"Silero VAD weights: unsupported file type {path.suffix!r} (use .npz or .onnx)")


def configured_weights_path() -> Optional[Path]:
    """`advanced_vad.silero_weights_path` (config) or `AUDIOCUT_SILERO_WEIGHTS` (environment); None = no-weights mode."""
    cand = get_config("advanced_vad.silero_weights_path", None) or os.environ.get("AUDIOCUT_SILERO_WEIGHTS")
    if not cand:
        return None
    p = Path(str(cand)).expanduser()
    if not p.exists():
        raise FileNotFoundError(f"Silero VAD weights file not found: {p}")
    return p


def speech_timestamps(probs: np.ndarray, n_samples: int, win: int, sr: int, threshold: float, min_speech_ms: float,
                      min_silence_ms: float, pad_ms: float) -> List[Dict[str, int]]:
    from .silero_chunk_vad import speech_timestamps as _st
    return _st(probs, n_samples, win, sr, threshold, min_speech_ms, min_silence_ms, pad_ms)


class PrecomputedProbs:
    """One chunk whose window probabilities are already on the host (`SileroHipVad.precompute`)."""

    def __init__(self, probs: np.ndarray, n_track_rate: int, n16: int, n16_padded: int) -> None:
        self.probs, self.n, self.n16, self.n16_padded = probs, int(n_track_rate), int(n16), int(n16_padded)

    def numel(self) -> int:
        return self.n


class SileroHipVad:
    """`VadFn` (host chunk in, `{'start','end'}` in track-rate samples relative to the chunk out) on the HIP kernels; also takes
    a device tensor, or a `PrecomputedProbs` from `precompute` (all chunks of a track in four launches and one download)."""

    def __init__(self, sample_rate: int, weights: Dict[str, np.ndarray], ctx: Optional["_native.Context"] = None) -> None:
        self.sample_rate = int(sample_rate)
        self.weights = validate_silero_weights(weights)
        self._ctx = ctx
        self._packed: Optional[dict] = None
        self._adaptive = threading.local()                           # see set_adaptive_params: per THREAD, the object is shared by the workers of a TrackPipeline
        g = math.gcd(SR16, self.sample_rate)
        self._up, self._down = SR16 // g, self.sample_rate // g

    def set_adaptive_params(self, vad_threshold: Optional[float] = None, min_pause_duration: Optional[float] = None,
                            speech_pad_ms: Optional[float] = None) -> None:
        """The reference's adaptive branch (`vocal_pause_detector.py:198-206`): when `current_adaptive_params` is set,
        `get_speech_timestamps` runs with its `vad_threshold`, `min_pause_duration` (seconds -> min silence) and `speech_pad_ms`
        instead of the static configuration.  Call with no arguments to go back to the static parameters.
        The parameters belong to a TRACK: one `SileroHipVad` (packed weights on the device) serves every worker thread of a
        `batch.TrackPipeline`, so what is set here is seen by the calling thread only; `__call__(chunk, adaptive=...)` passes them
        for one call."""
        self._adaptive.params = self.adaptive_dict(vad_threshold, min_pause_duration, speech_pad_ms)

    @staticmethod
    def adaptive_dict(vad_threshold: Optional[float] = None, min_pause_duration: Optional[float] = None,
                      speech_pad_ms: Optional[float] = None) -> Optional[Dict[str, float]]:
        if vad_threshold is None and min_pause_duration is None and speech_pad_ms is None:
            return None
        if vad_threshold is None or min_pause_duration is None or speech_pad_ms is None:
            raise ValueError("adaptive VAD parameters come as a set: vad_threshold, min_pause_duration, speech_pad_ms")
        return {"threshold": float(vad_threshold), "min_silence_ms": float(int(float(min_pause_duration) * 1000)),
                "pad_ms": float(int(speech_pad_ms))}

    @property
    def adaptive_params(self) -> Optional[Dict[str, float]]:
        return getattr(self._adaptive, "params", None)

    # -- weights in kernel order ------------------------------------------------------------------------
    def _pack(self) -> dict:
        if self._packed is None:
            ctx = self._ctx = self._ctx or _native.Context()
            w = self.weights
            conv = lambda name: np.ascontiguousarray(w[name].transpose(1, 2, 0).reshape(-1, w[name].shape[0]))   # [ci * 3 + tap][co]
            self._packed = {
                "basis_t": ctx.to_device(np.ascontiguousarray(w["stft.forward_basis_buffer"][:, 0, :].T)),          # [256][258]
                "c1": ctx.to_device(conv("encoder.0.reparam_conv.weight")), "b1": ctx.to_device(w["encoder.0.reparam_conv.bias"]),
                "c2": ctx.to_device(conv("encoder.1.reparam_conv.weight")), "b2": ctx.to_device(w["encoder.1.reparam_conv.bias"]),
                "c3": ctx.to_device(conv("encoder.2.reparam_conv.weight")), "b3": ctx.to_device(w["encoder.2.reparam_conv.bias"]),
                "c4": ctx.to_device(conv("encoder.3.reparam_conv.weight")), "b4": ctx.to_device(w["encoder.3.reparam_conv.bias"]),
                "wih_t": ctx.to_device(np.ascontiguousarray(w["decoder.rnn.weight_ih"].T)),                          # [128][512]
                "whh_t": ctx.to_device(np.ascontiguousarray(w["decoder.rnn.weight_hh"].T)),
                "bias_sum": ctx.to_device((w["decoder.rnn.bias_ih"] + w["decoder.rnn.bias_hh"]).astype(np.float32)),
                "w_out": ctx.to_device(np.ascontiguousarray(w["decoder.decoder.2.weight"].reshape(128))),
                "b_out": float(w["decoder.decoder.2.bias"][0]),
            }
        return self._packed

    # -- batched fast path --------------------------------------------------------------------------------
    def precompute(self, packed_dev, offsets: Sequence[int], lengths: Sequence[int]) -> List[PrecomputedProbs]:
        """`packed_dev` holds the chunks' vocals back to back at the track rate (`TrackSeparation.chunk_vocal`)."""
        ctx = self._ctx = self._ctx or _native.Context()
        bucket = int(get_config("advanced_vad.silero_length_bucket", 4096))
        step = math.lcm(bucket, WINDOW) if bucket > 0 else WINDOW           # every chunk ends on a window boundary: the last window zero-padded
        x16, out_off, n16 = ctx.resample_poly_segments(packed_dev, offsets, lengths, self._up, self._down, bucket=step)
        padded_bucket = [n + ((-n) % bucket if bucket > 0 else 0) for n in n16]             # what the reference pads to (`:192-196`)
        counts = [(n + WINDOW - 1) // WINDOW for n in padded_bucket]
        win_start: List[int] = []
        seg_first: List[int] = []
        for off, cnt in zip(out_off, counts):
            seg_first.append(len(win_start))
            base = int(off)
            win_start.extend([-(base) - 1] + [base + WINDOW * k for k in range(1, cnt)] if cnt else [])
        if not win_start:
            return [PrecomputedProbs(np.zeros(0, np.float32), n, 0, 0) for n in lengths]
        probs = ctx.silero_probs(x16, np.asarray(win_start, dtype=np.int64), np.asarray(seg_first, dtype=np.int32),
                                 np.asarray(counts, dtype=np.int32), self._pack()).cpu().numpy()
        return [PrecomputedProbs(probs[f: f + c], n, m, pb) for f, c, n, m, pb in zip(seg_first, counts, lengths, n16, padded_bucket)]

    # -- VadFn ---------------------------------------------------------------------------------------------
    def __call__(self, chunk, adaptive: Optional[Dict[str, float]] = None) -> List[Dict[str, int]]:
        """`adaptive`: an `adaptive_dict(...)` for this call only (else the calling thread's `set_adaptive_params`, else static)."""
        import torch
        if not isinstance(chunk, PrecomputedProbs):
            ctx = self._ctx = self._ctx or _native.Context()
            x = chunk if isinstance(chunk, torch.Tensor) else ctx.to_device(np.asarray(chunk, dtype=np.float32).reshape(-1))
            if x.numel() == 0:
                return []
            chunk = self.precompute(x, [0], [int(x.numel())])[0]
        ap = adaptive if adaptive is not None else self.adaptive_params
        stamps = speech_timestamps(
            chunk.probs, chunk.n16_padded, WINDOW, SR16,
            threshold=ap["threshold"] if ap else float(get_config("advanced_vad.silero_prob_threshold_down", 0.35)),
            min_speech_ms=float(get_config("advanced_vad.silero_min_speech_ms", 250)),
            min_silence_ms=ap["min_silence_ms"] if ap else float(get_config("advanced_vad.silero_min_silence_ms", 700)),
            pad_ms=ap["pad_ms"] if ap else float(get_config("advanced_vad.silero_speech_pad_ms", 150)))
        out: List[Dict[str, int]] = []
        scale = self.sample_rate / SR16
        for ts in stamps:                                                   # vocal_pause_detector.py:268-296
            a = int(max(0, min(ts.get("start", 0), chunk.n16))); b = int(max(0, min(ts.get("end", 0), chunk.n16)))
            if b > a:
                out.append({"start": int(a * scale), "end": int(b * scale)})
        return out


def default_vad(sample_rate: int, ctx: Optional["_native.Context"] = None):
    """What `SileroChunkVAD` / `EnhancedVocalSeparator` use when no `inference_fn` is injected: the Silero network when a weights
    file is configured, else the explicit no-weights mode (energy gate with Silero's hysteresis)."""
    from .silero_chunk_vad import EnergyGateVad
    path = configured_weights_path()
    if path is not None:
        logger.info("[SileroVAD] weights from %s", path)
        return SileroHipVad(sample_rate, load_silero_weights(path), ctx)
    logger.warning("[SileroVAD] no weights file configured (advanced_vad.silero_weights_path / AUDIOCUT_SILERO_WEIGHTS): "
                   "running the no-weights energy-gate VAD")
    return EnergyGateVad(sample_rate, ctx)


__all__ = ["SileroHipVad", "PrecomputedProbs", "load_silero_weights", "validate_silero_weights", "default_vad", "TENSOR_SHAPES"]
