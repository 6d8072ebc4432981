"""Seeded synthetic weights of the Silero VAD v5 16 kHz architecture - TEST / BENCH HELPER (no oracle import: the calibration of the
output layer, which needs the CPU oracle, lives in tests/silero_synth.py and hands its two numbers over as `affine`).

The real weights (`silero_vad` package / torch.hub, `vocal_pause_detector.py:74-123`) cannot be fetched offline.  These have the
published shapes and state-dict names (`oracle/silero.py`), a true windowed Fourier basis in `stft.forward_basis_buffer`, random
encoder / LSTM weights, and an output layer calibrated (an affine map of the logit, fitted on a seeded sung-line clip with the
CPU oracle) so that loud windows land near sigmoid(+2.5) and quiet ones near sigmoid(-2.5): the hysteresis, the minimum
durations and the padding of `get_speech_timestamps` are all exercised.  The same dict drives the oracle and the HIP kernels.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def _base_weights(seed: int) -> Dict[str, np.ndarray]:
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
    return w


def synth_silero_weights(seed: int = 0, affine=(1.0, 0.0)) -> Dict[str, np.ndarray]:
    """`affine` = (a, b): the output layer's logit becomes a * logit + b (tests/silero_synth.calibration_affine fits the pair with the
    CPU oracle; the track fixtures store it, which is how bench.py's C4 leg gets its VAD weights without anything under oracle/)."""
    w = _base_weights(seed)
    a, b = float(affine[0]), float(affine[1])
    w["decoder.decoder.2.weight"] = (w["decoder.decoder.2.weight"].astype(np.float64) * np.float64(a)).astype(np.float32)   # float64 product, rounded once
    w["decoder.decoder.2.bias"] = np.asarray([b], dtype=np.float32)
    return w
