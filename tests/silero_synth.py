"""Seeded synthetic weights of the Silero VAD v5 16 kHz architecture - TEST HELPER.

The real weights (`silero_vad` package / torch.hub, `vocal_pause_detector.py:74-123`) cannot be fetched offline.  These have the
published shapes and state-dict names (`oracle/silero.py`), a true windowed Fourier basis in `stft.forward_basis_buffer`, random
encoder / LSTM weights, and an output layer calibrated (an affine map of the logit, fitted on a seeded sung-line clip with the
CPU oracle) so that loud windows land near sigmoid(+2.5) and quiet ones near sigmoid(-2.5): the hysteresis, the minimum
durations and the padding of `get_speech_timestamps` are all exercised.  The same dict drives the oracle and the HIP kernels.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def synth_silero_weights(seed: int = 0, calib: str = "voice") -> Dict[str, np.ndarray]:
    from audio_cut_amd.testing import signals
    from oracle import silero as OS
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    n = np.arange(256)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / 256.0)                       # periodic Hann
    k = np.arange(129)[:, None]
    basis = np.concatenate([np.cos(2 * np.pi * k * n / 256.0), -np.sin(2 * np.pi * k * n / 256.0)], axis=0) * win
    w["stft.forward_basis_buffer"] = basis[:, None, :].astype(np.float32)          # [258, 1, 256]
    for i, (co, ci) in enumerate(((128, 129), (64, 128), (64, 64), (128, 64))):
        w[f"encoder.{i}.reparam_conv.weight"] = (rng.standard_normal((co, ci, 3)) * np.sqrt(2.0 / (3 * ci))).astype(np.float32)
        w[f"encoder.{i}.reparam_conv.bias"] = (rng.standard_normal(co) * 0.05).astype(np.float32)
    w["decoder.rnn.weight_ih"] = (rng.standard_normal((512, 128)) * 0.5 / np.sqrt(128)).astype(np.float32)
    w["decoder.rnn.weight_hh"] = (rng.standard_normal((512, 128)) * 0.7 / np.sqrt(128)).astype(np.float32)
    b_ih = rng.standard_normal(512) * 0.05
    b_ih[128:256] -= 1.5                                                     # forget gate: a short memory, so that rests of a few hundred ms reset the trigger
    w["decoder.rnn.bias_ih"] = b_ih.astype(np.float32)
    w["decoder.rnn.bias_hh"] = (rng.standard_normal(512) * 0.05).astype(np.float32)
    w["decoder.decoder.2.weight"] = (rng.standard_normal((1, 128, 1)) / np.sqrt(128)).astype(np.float32)
    w["decoder.decoder.2.bias"] = np.zeros(1, np.float32)
    # calibrate the output layer: logit -> a * logit + b
    # "voice": a sung line at its natural level; "bursts": tone bursts / exact silences at the level a separated stem has
    clip44 = signals.voice_with_rests(12.0, seed=1000 + seed) if calib == "voice" else 0.35 * signals.c1_sine_silence(12.0, seed=1000 + seed)
    clip = OS.resample_to_16k(clip44, 44100)
    p = OS.silero_probs(w, clip).astype(np.float64)
    logit = np.log(np.clip(p, 1e-7, 1 - 1e-7) / np.clip(1 - p, 1e-7, 1))
    nw = len(p)
    energy = np.log10(np.mean(np.square(np.pad(clip, (0, nw * 512 - len(clip))).reshape(nw, 512)), axis=1) + 1e-12)
    lo, hi = np.quantile(logit, 0.2), np.quantile(logit, 0.8)
    sign = 1.0 if np.corrcoef(logit, energy)[0, 1] >= 0 else -1.0
    a = sign * 5.0 / max(hi - lo, 1e-3)
    b = -a * 0.5 * (hi + lo)
    w["decoder.decoder.2.weight"] = (w["decoder.decoder.2.weight"] * a).astype(np.float32)
    w["decoder.decoder.2.bias"] = np.asarray([b], dtype=np.float32)
    return w
