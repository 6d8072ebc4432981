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


def calibration_affine(seed: int = 0, calib: str = "voice"):
    """(a, b) of the output layer's calibration logit -> a * logit + b, fitted with the CPU oracle on a seeded clip.
    "voice": a sung line at its natural level; "bursts": tone bursts / exact silences at the level a separated stem has;
    "c2_stem": what the chunked VAD actually sees on the C2 / C4 track - the CPU oracle's vocal stem (synthetic U-Net weights
    seed 0) of the first 20 s of the C2 song (seed 2): a quiet, fairly even signal on which the other two calibrations find no
    speech at all (profiles/r02_parity_soak_h.log: vad_segments=0 on every song)."""
    from audio_cut_amd.testing import signals
    from oracle import silero as OS
    w = _base_weights(seed)
    if calib == "c2_stem":
        from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
        from oracle import separator as OSEP
        mix = signals.c2_song(20.0, seed=2)
        mix = np.mean(mix, axis=0).astype(np.float32) if mix.ndim == 2 else mix
        clip44 = OSEP.separate_track(mix, 44100, synth_weights(TfcTdfSpec(), seed=0))[0]
    else:
        clip44 = signals.voice_with_rests(12.0, seed=1000 + seed) if calib == "voice" else 0.35 * signals.c1_sine_silence(12.0, seed=1000 + seed)
    clip = OS.resample_to_16k(clip44, 44100)
    p = OS.silero_probs(w, clip).astype(np.float64)
    logit = np.log(np.clip(p, 1e-7, 1 - 1e-7) / np.clip(1 - p, 1e-7, 1))
    nw = len(p)
    energy = np.log10(np.mean(np.square(np.pad(clip, (0, nw * 512 - len(clip))).reshape(nw, 512)), axis=1) + 1e-12)
    # "c2_stem": the stem is quiet and even, its phrases are the loudest third of the windows: centre the logistic higher and
    # make it steeper (with seed 3: 11 speech segments in the first minute; 0.2 / 0.8 gives one segment spanning the track)
    lo, hi = (np.quantile(logit, 0.6), np.quantile(logit, 0.95)) if calib == "c2_stem" else (np.quantile(logit, 0.2), np.quantile(logit, 0.8))
    sign = 1.0 if np.corrcoef(logit, energy)[0, 1] >= 0 else -1.0
    a = sign * (8.0 if calib == "c2_stem" else 5.0) / max(hi - lo, 1e-3)
    b = -a * 0.5 * (hi + lo)
    return float(a), float(b)


def synth_silero_weights(seed: int = 0, calib: str = "voice", affine=None) -> Dict[str, np.ndarray]:
    """`affine` = (a, b) of an earlier `calibration_affine` (a fixture stores it): the output layer is set from it and the CPU
    oracle is not needed - this is how bench.py's C4 leg gets its VAD weights without importing anything under oracle/."""
    w = _base_weights(seed)
    a, b = (float(affine[0]), float(affine[1])) if affine is not None else calibration_affine(seed, calib)
    w["decoder.decoder.2.weight"] = (w["decoder.decoder.2.weight"].astype(np.float64) * np.float64(a)).astype(np.float32)   # float64 product, rounded once
    w["decoder.decoder.2.bias"] = np.asarray([b], dtype=np.float32)
    return w
