"""Oracle for SURVEY.md §8 row a13, the Silero half: `VocalPauseDetectorV2._detect_speech_timestamps`
(`src/vocal_smart_splitter/core/vocal_pause_detector.py:175-296`) with the network restated in PyTorch on the CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).

What the reference does per chunk (`:189-296`): resample 44.1 kHz -> 16 kHz (`librosa.resample`, soxr_hq - not available
offline; this build's parity definition for resampling is `scipy.signal.resample_poly`), zero-pad to a multiple of
`advanced_vad.silero_length_bucket` = 4096, `silero_vad.get_speech_timestamps(audio, model, threshold=0.35,
min_speech_duration_ms=250, min_silence_duration_ms=700, speech_pad_ms=150)`, clamp to the unpadded length, rescale with
`int(idx * sr / 16000)`.

Third-party, **parity unpinned** (neither the `silero_vad` package nor its weights exist in this container): the network
is the published Silero VAD v5 16 kHz model as this build restates it -
  x[576] = 64 samples of context (the tail of the previous window, zeros at the start) + the 512-sample window
  -> reflect-pad 64 on the right -> STFT as a strided convolution (`forward_basis_buffer` [258, 1, 256], stride 128): 4 frames
  -> magnitude sqrt(re^2 + im^2) [129, 4]
  -> encoder: Conv1d(129,128,3,p=1) ReLU, Conv1d(128,64,3,p=1,stride 2) ReLU, Conv1d(64,64,3,p=1,stride 2) ReLU,
              Conv1d(64,128,3,p=1) ReLU   -> [128, 1]
  -> decoder: LSTMCell(128,128) carrying (h, c) across the windows of one call, ReLU, Conv1d(128,1,1), Sigmoid -> probability
and `get_speech_timestamps`' windowing: 512-sample windows, the last one zero-padded, state reset at the start of a call.
The weight names are those of the published TorchScript module's state dict (`_model.` prefix dropped).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F

from .config import get_config
from .vad import VadFn, speech_timestamps

Weights = Dict[str, np.ndarray]
WINDOW = 512
CONTEXT = 64
SR16 = 16000


def silero_probs(weights: Weights, audio16k: np.ndarray) -> np.ndarray:
    """Speech probability of every 512-sample window of one call (state and context start at zero)."""
    t = lambda name: torch.from_numpy(np.asarray(weights[name], dtype=np.float32))
    x = torch.from_numpy(np.asarray(audio16k, dtype=np.float32))
    n_win = (x.numel() + WINDOW - 1) // WINDOW
    x = F.pad(x, (0, n_win * WINDOW - x.numel()))
    h = torch.zeros(1, 128); c = torch.zeros(1, 128)
    ctx = torch.zeros(CONTEXT)
    probs = []
    with torch.no_grad():
        for w in range(n_win):
            cur = x[w * WINDOW:(w + 1) * WINDOW]
            inp = torch.cat([ctx, cur]).view(1, 1, -1)
            ctx = cur[-CONTEXT:]
            inp = F.pad(inp, (0, 64), mode="reflect")
            spec = F.conv1d(inp, t("stft.forward_basis_buffer"), stride=128)            # [1, 258, 4]
            mag = torch.sqrt(spec[:, :129] ** 2 + spec[:, 129:] ** 2)
            y = F.relu(F.conv1d(mag, t("encoder.0.reparam_conv.weight"), t("encoder.0.reparam_conv.bias"), padding=1))
            y = F.relu(F.conv1d(y, t("encoder.1.reparam_conv.weight"), t("encoder.1.reparam_conv.bias"), padding=1, stride=2))
            y = F.relu(F.conv1d(y, t("encoder.2.reparam_conv.weight"), t("encoder.2.reparam_conv.bias"), padding=1, stride=2))
            y = F.relu(F.conv1d(y, t("encoder.3.reparam_conv.weight"), t("encoder.3.reparam_conv.bias"), padding=1))
            feat = y[:, :, 0]                                                              # [1, 128]
            gates = F.linear(feat, t("decoder.rnn.weight_ih"), t("decoder.rnn.bias_ih")) + F.linear(h, t("decoder.rnn.weight_hh"), t("decoder.rnn.bias_hh"))
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            logit = F.conv1d(F.relu(h).view(1, 128, 1), t("decoder.decoder.2.weight"), t("decoder.decoder.2.bias"))
            probs.append(float(torch.sigmoid(logit).view(())))
    return np.asarray(probs, dtype=np.float32)


def resample_to_16k(audio: np.ndarray, sr: int) -> np.ndarray:
    from . import resample as RS            # soxr-HQ-specification low-pass (librosa.resample's default res_type), oracle/resample.py
    return RS.resample(np.asarray(audio, dtype=np.float32), SR16, int(sr))


def detect_speech_timestamps(audio: np.ndarray, sr: int, weights: Weights, adaptive=None) -> List[Dict[str, int]]:
    """vocal_pause_detector.py:175-296.  `adaptive` = (vad_threshold, min_pause_duration [s], speech_pad_ms): the
    `current_adaptive_params` branch (`:198-206`); None = the static parameters of the live path."""
    a16 = resample_to_16k(audio, sr)
    n16 = a16.shape[0]
    bucket = int(get_config("advanced_vad.silero_length_bucket", 4096))
    if bucket > 0 and (-n16) % bucket:
        a16 = np.pad(a16, (0, (-n16) % bucket), mode="constant")
    probs = silero_probs(weights, a16)
    if adaptive is not None:
        thr, min_sil, pad = float(adaptive[0]), float(int(float(adaptive[1]) * 1000)), float(int(adaptive[2]))
    else:
        thr = float(get_config("advanced_vad.silero_prob_threshold_down", 0.35))
        min_sil = float(get_config("advanced_vad.silero_min_silence_ms", 700))
        pad = float(get_config("advanced_vad.silero_speech_pad_ms", 150))
    stamps = speech_timestamps(probs, len(a16), WINDOW, SR16, threshold=thr,
                               min_speech_ms=float(get_config("advanced_vad.silero_min_speech_ms", 250)),
                               min_silence_ms=min_sil, pad_ms=pad)
    out: List[Dict[str, int]] = []
    for ts in stamps:
        a = int(max(0, min(ts.get("start", 0), n16))); b = int(max(0, min(ts.get("end", 0), n16)))
        if b > a:
            out.append({"start": a, "end": b})
    scale = sr / SR16
    for ts in out:
        ts["start"] = int(ts["start"] * scale); ts["end"] = int(ts["end"] * scale)
    return out


def silero_vad_fn(sr: int, weights: Weights) -> VadFn:
    return lambda audio: detect_speech_timestamps(np.asarray(audio, dtype=np.float32), sr, weights)
