"""The oracle-calibrated half of the synthetic Silero weights - TEST HELPER (the oracle-free half, shapes / names / base weights, is
audio_cut_amd/testing/silero_synth.py).  The output layer is calibrated (an affine map of the logit, fitted on a seeded clip with the CPU
oracle) so that loud windows land near sigmoid(+2.5) and quiet ones near sigmoid(-2.5): the hysteresis, the minimum durations and the
padding of `get_speech_timestamps` are all exercised.  The same dict drives the oracle and the HIP kernels."""
from __future__ import annotations

from typing import Dict

import numpy as np

from audio_cut_amd.testing import silero_synth as _pkg
from audio_cut_amd.testing.silero_synth import _base_weights


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


def calibration_affine_on(clip44: np.ndarray, seed: int, sr: int = 44100, spread: float = 8.0, q=(0.6, 0.95)):
    """The same fit on a clip of the caller's choice (tools/parity_soak.py: the stem the chunked VAD will actually see)."""
    from oracle import silero as OS
    w = _base_weights(seed)
    clip = OS.resample_to_16k(np.asarray(clip44, dtype=np.float32), sr)
    p = OS.silero_probs(w, clip).astype(np.float64)
    logit = np.log(np.clip(p, 1e-7, 1 - 1e-7) / np.clip(1 - p, 1e-7, 1))
    nw = len(p)
    energy = np.log10(np.mean(np.square(np.pad(clip, (0, nw * 512 - len(clip))).reshape(nw, 512)), axis=1) + 1e-12)
    lo, hi = np.quantile(logit, q[0]), np.quantile(logit, q[1])
    sign = 1.0 if np.corrcoef(logit, energy)[0, 1] >= 0 else -1.0
    a = sign * spread / max(hi - lo, 1e-3)
    return float(a), float(-a * 0.5 * (hi + lo))


def synth_silero_weights(seed: int = 0, calib: str = "voice", affine=None) -> Dict[str, np.ndarray]:
    """`affine` = (a, b) of an earlier `calibration_affine` (a fixture stores it): then the CPU oracle is not needed."""
    return _pkg.synth_silero_weights(seed, affine if affine is not None else calibration_affine(seed, calib))
