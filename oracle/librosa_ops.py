"""numpy/scipy restatement of the librosa 0.10 ops the reference calls on the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  librosa is a third-party
dependency of the reference (``requirements.txt:5``, ``librosa>=0.10.0``,
unpinned) that is neither vendored under /root/reference nor installed here,
so this file restates the *published* librosa 0.10.x algorithms.  Parity of
these float series with a real librosa install is **unpinned**; the
closed-form known-answer tests in ``tests/test_oracle_librosa_kat.py`` are the
only anchor.  Call sites in the reference that fix the parameters:

* ``feature.rms``              features_cache.py:182, pure_vocal_pause_detector.py:1113,1397,
                               seamless_splitter.py:1714,1848, vocal_separator.py:483
* ``feature.spectral_flatness`` features_cache.py:183, pure_vocal_pause_detector.py:1117
* ``onset.onset_strength``     features_cache.py:184, adaptive_vad_enhancer.py:143
* ``onset.onset_detect``       features_cache.py:186
* ``beat.beat_track``          adaptive_vad_enhancer.py:61, features_cache.py:289
* ``feature.rhythm.tempo``     adaptive_vad_enhancer.py:151, features_cache.py:283
* ``frames_to_time``           features_cache.py:187

The module is also usable as a stand-in for the ``librosa`` import when the
reference's own *control logic* is executed to produce golden vectors
(tests/golden/make_golden.py); see ``install_as_librosa``.
"""
from __future__ import annotations

import sys
import types
from typing import Callable, Optional, Tuple

import numpy as np
import scipy.signal

__all__ = [
    "frame", "rms", "stft", "spectral_flatness", "mel_filters", "melspectrogram",
    "power_to_db", "onset_strength", "onset_detect", "peak_pick", "tempogram",
    "tempo", "beat_track", "frames_to_time", "time_to_frames", "install_as_librosa",
]


def tiny(x) -> float:
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.floating) or np.issubdtype(x.dtype, np.complexfloating):
        dtype = x.dtype
    else:
        dtype = np.dtype(np.float32)
    return float(np.finfo(dtype).tiny)


# ---------------------------------------------------------------------------
# framing / time helpers
# ---------------------------------------------------------------------------

def frame(x: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    """librosa.util.frame(axis=-1): view of shape (frame_length, n_frames)."""
    x = np.asarray(x)
    if x.shape[-1] < frame_length:
        raise ValueError(f"Input is too short (n={x.shape[-1]}) for frame_length={frame_length}")
    n_frames = 1 + (x.shape[-1] - frame_length) // hop_length
    strides = x.strides[-1]
    return np.lib.stride_tricks.as_strided(
        x, shape=(frame_length, n_frames), strides=(strides, strides * hop_length), writeable=False
    )


def frames_to_time(frames, sr: float = 22050, hop_length: int = 512, n_fft: Optional[int] = None):
    offset = int(n_fft // 2) if n_fft is not None else 0
    samples = (np.asanyarray(frames) * hop_length + offset).astype(int)
    return np.asanyarray(samples) / float(sr)


def time_to_frames(times, sr: float = 22050, hop_length: int = 512) -> np.ndarray:
    samples = (np.asanyarray(times) * sr).astype(int)
    return np.floor(np.asanyarray(samples) // hop_length).astype(int)


# ---------------------------------------------------------------------------
# rms / stft / flatness
# ---------------------------------------------------------------------------

def rms(y: np.ndarray, frame_length: int = 2048, hop_length: int = 512, center: bool = True,
        pad_mode: str = "constant") -> np.ndarray:
    """librosa.feature.rms(y=...) -> shape (1, n_frames), dtype of y."""
    y = np.asarray(y)
    if center:
        y = np.pad(y, int(frame_length // 2), mode=pad_mode)
    x = frame(y, frame_length, hop_length)
    power = np.mean(np.abs(x) ** 2, axis=-2, keepdims=True)
    return np.sqrt(power)


def hann_periodic(n: int) -> np.ndarray:
    return scipy.signal.get_window("hann", n, fftbins=True)


def stft(y: np.ndarray, n_fft: int = 2048, hop_length: Optional[int] = None, center: bool = True,
         pad_mode: str = "constant", max_block_frames: int = 4096) -> np.ndarray:
    """librosa.stft: periodic Hann (float64) * frames, rfft in float64, stored complex64."""
    if hop_length is None:
        hop_length = n_fft // 4
    y = np.asarray(y)
    win = hann_periodic(n_fft).reshape(-1, 1)
    if center:
        y = np.pad(y, int(n_fft // 2), mode=pad_mode)
    frames = frame(y, n_fft, hop_length)
    dtype = np.complex64 if y.dtype == np.float32 else np.complex128
    out = np.zeros((1 + n_fft // 2, frames.shape[1]), dtype=dtype, order="F")
    for s in range(0, frames.shape[1], max_block_frames):
        t = min(s + max_block_frames, frames.shape[1])
        out[:, s:t] = np.fft.rfft(win * frames[:, s:t], axis=0)
    return out


def _spectrogram(y: np.ndarray, n_fft: int, hop_length: int, power: float) -> np.ndarray:
    return np.abs(stft(y, n_fft=n_fft, hop_length=hop_length, center=True)) ** power


def spectral_flatness(y: np.ndarray, n_fft: int = 2048, hop_length: int = 512, amin: float = 1e-10,
                      power: float = 2.0) -> np.ndarray:
    S = _spectrogram(y, n_fft, hop_length, 1.0)
    S_thresh = np.maximum(amin, S ** power)
    gmean = np.exp(np.mean(np.log(S_thresh), axis=-2, keepdims=True))
    amean = np.mean(S_thresh, axis=-2, keepdims=True)
    return gmean / amean


# ---------------------------------------------------------------------------
# mel / onset strength
# ---------------------------------------------------------------------------

def hz_to_mel(f):
    f = np.asanyarray(f, dtype=float)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        log_t = f >= min_log_hz
        mels[log_t] = min_log_mel + np.log(f[log_t] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=float)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if m.ndim:
        log_t = m >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (m[log_t] - min_log_mel))
    elif m >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (m - min_log_mel))
    return freqs


def mel_filters(sr: float, n_fft: int, n_mels: int = 128, fmin: float = 0.0,
                fmax: Optional[float] = None) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm='slaney', dtype=float32)."""
    if fmax is None:
        fmax = float(sr) / 2
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def melspectrogram(y: np.ndarray, sr: float, n_fft: int = 2048, hop_length: int = 512,
                   n_mels: int = 128, fmax: Optional[float] = None) -> np.ndarray:
    S = _spectrogram(y, n_fft, hop_length, 2.0)
    basis = mel_filters(sr, n_fft, n_mels=n_mels, fmin=0.0, fmax=fmax)
    return np.einsum("ft,mf->mt", S, basis, optimize=True)


def power_to_db(S: np.ndarray, ref: float = 1.0, amin: float = 1e-10, top_db: Optional[float] = 80.0):
    S = np.asarray(S)
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def onset_strength(y: np.ndarray, sr: float = 22050, hop_length: int = 512, lag: int = 1,
                   n_fft: int = 2048, center: bool = True,
                   aggregate: Optional[Callable] = None) -> np.ndarray:
    """librosa.onset.onset_strength(y=..., max_size=1, detrend=False)."""
    if aggregate is None:
        aggregate = np.mean
    S = np.abs(melspectrogram(y, sr, n_fft=n_fft, hop_length=hop_length, fmax=0.5 * sr))
    S = power_to_db(S)
    onset_env = S[..., lag:] - S[..., :-lag]
    onset_env = np.maximum(0.0, onset_env)
    onset_env = aggregate(onset_env, axis=-2)
    pad_width = lag
    if center:
        pad_width += n_fft // (2 * hop_length)
    onset_env = np.pad(onset_env, (int(pad_width), 0), mode="constant")
    if center:
        onset_env = onset_env[: S.shape[-1]]
    return onset_env


def peak_pick(x: np.ndarray, pre_max, post_max, pre_avg, post_avg, delta: float, wait) -> np.ndarray:
    """librosa.util.peak_pick: x[n]==max(x[n-pre_max:n+post_max]) and
    x[n] >= mean(x[n-pre_avg:n+post_avg]) + delta and n - last > wait (windows truncated at edges)."""
    pre_max = int(np.ceil(pre_max)); post_max = int(np.ceil(post_max))
    pre_avg = int(np.ceil(pre_avg)); post_avg = int(np.ceil(post_avg))
    wait = int(np.ceil(wait))
    n = x.shape[0]
    peaks = []
    last = -np.inf
    for i in range(n):
        lo = max(0, i - pre_max)
        if x[i] != np.max(x[lo: i + post_max]):
            continue
        lo = max(0, i - pre_avg)
        if not (x[i] >= np.mean(x[lo: i + post_avg]) + delta):
            continue
        if x[i] == 0:
            # the 0.10.0 implementation masks by multiplication: zero-valued peaks vanish
            continue
        if i > last + wait:
            peaks.append(i)
            last = i
    return np.array(peaks, dtype=int)


def onset_detect(onset_envelope: np.ndarray, sr: float = 22050, hop_length: int = 512) -> np.ndarray:
    env = np.asarray(onset_envelope)
    if not env.any() or not np.all(np.isfinite(env)):
        return np.array([], dtype=int)
    env = env - np.min(env)
    env = env / (np.max(env) + tiny(env))
    return peak_pick(
        env,
        pre_max=0.03 * sr // hop_length,
        post_max=0.00 * sr // hop_length + 1,
        pre_avg=0.10 * sr // hop_length,
        post_avg=0.10 * sr // hop_length + 1,
        delta=0.07,
        wait=0.03 * sr // hop_length,
    )


# ---------------------------------------------------------------------------
# tempogram / tempo / beat tracking
# ---------------------------------------------------------------------------

def tempo_frequencies(n_bins: int, hop_length: int, sr: float) -> np.ndarray:
    bin_frequencies = np.zeros(int(n_bins), dtype=np.float64)
    bin_frequencies[0] = np.inf
    bin_frequencies[1:] = 60.0 * sr / (hop_length * np.arange(1.0, n_bins))
    return bin_frequencies


def tempogram(onset_envelope: np.ndarray, win_length: int, block_frames: int = 2048) -> np.ndarray:
    """librosa.feature.tempogram(center=True, window='hann', norm=inf) -> (win_length, n)."""
    onset_envelope = np.asarray(onset_envelope)
    n = onset_envelope.shape[-1]
    ac_window = hann_periodic(win_length).reshape(-1, 1)
    padded = np.pad(onset_envelope, (int(win_length // 2),) * 2, mode="linear_ramp", end_values=[0, 0])
    odf_frame = frame(padded, win_length, 1)[:, :n]
    out = np.empty((win_length, n), dtype=np.float64)
    n_pad = 2 * win_length - 1
    for s in range(0, n, block_frames):
        t = min(n, s + block_frames)
        blk = odf_frame[:, s:t] * ac_window
        spec = np.fft.rfft(blk, n=n_pad, axis=0)
        ac = np.fft.irfft(spec.real ** 2 + spec.imag ** 2, n=n_pad, axis=0)[:win_length]
        mag = np.abs(ac)
        length = np.max(mag, axis=0, keepdims=True)
        length[length < tiny(ac)] = 1.0
        out[:, s:t] = ac / length
    return out


def tempo(onset_envelope: np.ndarray, sr: float = 22050, hop_length: int = 512, start_bpm: float = 120.0,
          std_bpm: float = 1.0, ac_size: float = 8.0, max_tempo: float = 320.0,
          aggregate: Optional[Callable] = np.mean) -> np.ndarray:
    """librosa.feature.rhythm.tempo; aggregate=None -> per-frame curve."""
    win_length = int(time_to_frames(ac_size, sr=sr, hop_length=hop_length))
    tg = tempogram(onset_envelope, win_length)
    if aggregate is not None:
        tg = aggregate(tg, axis=-1, keepdims=True)
    bpms = tempo_frequencies(win_length, hop_length=hop_length, sr=sr)
    with np.errstate(divide="ignore"):
        logprior = -0.5 * ((np.log2(bpms) - np.log2(start_bpm)) / std_bpm) ** 2
    if max_tempo is not None:
        max_idx = int(np.argmax(bpms < max_tempo))
        logprior[:max_idx] = -np.inf
    logprior = logprior.reshape(-1, 1)
    best_period = np.argmax(np.log1p(1e6 * tg) + logprior, axis=-2)
    return np.take(bpms, best_period)


def _beat_local_score(onset_envelope: np.ndarray, period: float) -> np.ndarray:
    window = np.exp(-0.5 * (np.arange(-period, period + 1) * 32.0 / period) ** 2)
    norm = onset_envelope.std(ddof=1)
    return scipy.signal.convolve(onset_envelope / norm, window, "same")


def _beat_track_dp(localscore: np.ndarray, period: float, tightness: float):
    backlink = np.zeros_like(localscore, dtype=int)
    cumscore = np.zeros_like(localscore)
    window = np.arange(-2 * period, -np.round(period / 2) + 1, dtype=int)
    txwt = -tightness * (np.log(-window / period) ** 2)
    first_beat = True
    score_thresh = 0.01 * localscore.max()
    for i, score_i in enumerate(localscore):
        z_pad = np.maximum(0, min(-window[0], len(window)))
        candidates = txwt.copy()
        candidates[z_pad:] = candidates[z_pad:] + cumscore[window[z_pad:]]
        beat_location = np.argmax(candidates)
        cumscore[i] = score_i + candidates[beat_location]
        if first_beat and score_i < score_thresh:
            backlink[i] = -1
        else:
            backlink[i] = window[beat_location]
            first_beat = False
        window = window + 1
    return backlink, cumscore


def _localmax(x: np.ndarray) -> np.ndarray:
    x_pad = np.pad(x, (1, 1), mode="edge")
    return (x > x_pad[:-2]) & (x >= x_pad[2:])


def _last_beat(cumscore: np.ndarray) -> int:
    maxes = _localmax(cumscore)
    med_score = np.median(cumscore[np.argwhere(maxes)])
    return int(np.argwhere((cumscore * maxes * 2 > med_score)).max())


def _trim_beats(localscore: np.ndarray, beats: np.ndarray, trim: bool) -> np.ndarray:
    smooth_boe = scipy.signal.convolve(localscore[beats], scipy.signal.windows.hann(5), "same")
    threshold = 0.5 * ((smooth_boe ** 2).mean() ** 0.5) if trim else 0.0
    valid = np.argwhere(smooth_boe > threshold)
    return beats[valid.min(): valid.max()]


def beat_tracker(onset_envelope: np.ndarray, bpm: float, fft_res: float, tightness: float, trim: bool):
    period = round(60.0 * fft_res / bpm)
    localscore = _beat_local_score(onset_envelope, period)
    backlink, cumscore = _beat_track_dp(localscore, period, tightness)
    beats = [_last_beat(cumscore)]
    while backlink[beats[-1]] >= 0:
        beats.append(backlink[beats[-1]])
    beats = np.array(beats[::-1], dtype=int)
    return _trim_beats(localscore, beats, trim)


def beat_track(y: Optional[np.ndarray] = None, sr: float = 22050, onset_envelope: Optional[np.ndarray] = None,
               hop_length: int = 512, start_bpm: float = 120.0, tightness: float = 100,
               trim: bool = True) -> Tuple[float, np.ndarray]:
    """librosa.beat.beat_track(units='frames') following the 0.10.0 scalar code path."""
    if onset_envelope is None:
        onset_envelope = onset_strength(y, sr=sr, hop_length=hop_length, aggregate=np.median)
    if not onset_envelope.any():
        return 0.0, np.array([], dtype=int)
    bpm = float(tempo(onset_envelope, sr=sr, hop_length=hop_length, start_bpm=start_bpm)[0])
    beats = beat_tracker(onset_envelope, bpm, float(sr) / hop_length, tightness, trim)
    return bpm, beats


# ---------------------------------------------------------------------------
# stand-in module registration (golden generation only)
# ---------------------------------------------------------------------------

def install_as_librosa() -> types.ModuleType:
    """Register a module named ``librosa`` backed by this restatement.

    Used ONLY by tests/golden/make_golden.py in the build container so that the
    reference's own control logic (features_cache / pure_vocal_pause_detector /
    seamless_splitter helpers) can execute; the goldens it produces are
    conditional on the restated float ops above.
    """
    lib = types.ModuleType("librosa")
    feature = types.ModuleType("librosa.feature")
    rhythm = types.ModuleType("librosa.feature.rhythm")
    onset = types.ModuleType("librosa.onset")
    beat = types.ModuleType("librosa.beat")
    util = types.ModuleType("librosa.util")

    feature.rms = lambda y=None, S=None, frame_length=2048, hop_length=512, center=True, pad_mode="constant", **kw: rms(
        y, frame_length=frame_length, hop_length=hop_length, center=center, pad_mode=pad_mode)
    feature.spectral_flatness = lambda y=None, S=None, n_fft=2048, hop_length=512, amin=1e-10, power=2.0, **kw: (
        spectral_flatness(y, n_fft=n_fft, hop_length=hop_length, amin=amin, power=power))
    rhythm.tempo = lambda y=None, sr=22050, onset_envelope=None, hop_length=512, start_bpm=120.0, aggregate=np.mean, **kw: (
        tempo(onset_envelope, sr=sr, hop_length=hop_length, start_bpm=start_bpm, aggregate=aggregate))
    feature.rhythm = rhythm
    onset.onset_strength = lambda y=None, sr=22050, S=None, hop_length=512, aggregate=None, **kw: onset_strength(
        y, sr=sr, hop_length=hop_length, aggregate=aggregate)
    onset.onset_detect = lambda y=None, sr=22050, onset_envelope=None, hop_length=512, **kw: onset_detect(
        onset_envelope, sr=sr, hop_length=hop_length)
    beat.beat_track = lambda y=None, sr=22050, onset_envelope=None, hop_length=512, start_bpm=120.0, tightness=100, **kw: (
        beat_track(y=y, sr=sr, onset_envelope=onset_envelope, hop_length=hop_length, start_bpm=start_bpm, tightness=tightness))
    util.frame = frame
    feature.spectral_centroid = lambda y=None, sr=22050, S=None, n_fft=2048, hop_length=512, **kw: spectral_centroid(
        y, sr=sr, n_fft=n_fft, hop_length=hop_length)
    feature.zero_crossing_rate = lambda y, frame_length=2048, hop_length=512, center=True, **kw: zero_crossing_rate(
        y, frame_length=frame_length, hop_length=hop_length, center=center)
    lib.pyin = lambda y, fmin=None, fmax=None, sr=22050, frame_length=2048, hop_length=None, **kw: pyin(
        y, fmin, fmax, sr=sr, frame_length=frame_length, hop_length=hop_length)
    lib.lpc = lambda y, order=None, **kw: lpc(y, order)
    lib.note_to_hz = note_to_hz
    lib.amplitude_to_db = amplitude_to_db
    lib.feature = feature
    lib.onset = onset
    lib.beat = beat
    lib.util = util
    lib.frames_to_time = frames_to_time
    lib.time_to_frames = time_to_frames
    lib.stft = stft
    lib.__version__ = "0.10.0-oracle-restatement"
    for name, mod in (("librosa", lib), ("librosa.feature", feature), ("librosa.feature.rhythm", rhythm),
                      ("librosa.onset", onset), ("librosa.beat", beat), ("librosa.util", util)):
        sys.modules[name] = mod
    return lib


# ---------------------------------------------------------------------------
# YIN fundamental frequency (first stage of librosa.pyin, pure_vocal_pause_detector.py:422-428)
# ---------------------------------------------------------------------------

LEGACY_FFT_PRECISION = True
"""numpy < 2 (the reference pins it) computes every `np.fft` transform in double precision and returns complex128 /
float64 even for float32 input; numpy >= 2 keeps float32.  librosa's YIN autocorrelation goes through rfft/irfft, so
under the pinned environment the CMND series is float64 (float32 windowed energies + float64 autocorrelation).  True =
the pinned behaviour (what the product follows); False = this container's numpy 2.x behaviour."""


def cmnd_frames(y: np.ndarray, sr: float, fmin: float, fmax: float, frame_length: int = 2048, hop_length: int = 512,
                center: bool = True) -> Tuple[np.ndarray, int, int]:
    """librosa.core.pitch._cumulative_mean_normalized_difference over centred frames ->
    (cmnd [max_period-min_period+1, n_frames], min_period, max_period).  FFT autocorrelation (precision per
    LEGACY_FFT_PRECISION), windowed energies from a float32 cumulative sum, |acf| and |energy| below 1e-6 snapped to
    zero, exactly as librosa does."""
    win_length = frame_length // 2
    y = np.asarray(y)
    if center:
        y = np.pad(y, frame_length // 2, mode="constant")
    y_frames = frame(y, frame_length, hop_length)
    min_period = max(int(np.floor(sr / fmax)), 1)
    max_period = min(int(np.ceil(sr / fmin)), frame_length - win_length - 1)
    out_dtype = np.float64 if LEGACY_FFT_PRECISION else y.dtype
    out = np.empty((max_period - min_period + 1, y_frames.shape[1]), dtype=out_dtype)
    for s in range(0, y_frames.shape[1], 2048):
        blk = y_frames[:, s:s + 2048]
        fblk = blk.astype(np.float64) if LEGACY_FFT_PRECISION else blk
        a = np.fft.rfft(fblk, frame_length, axis=0)
        b = np.fft.rfft(fblk[win_length:0:-1, :], frame_length, axis=0)
        acf = np.fft.irfft(a * b, frame_length, axis=0)[win_length:, :]
        if not LEGACY_FFT_PRECISION:
            acf = acf.astype(y.dtype)
        acf[np.abs(acf) < 1e-6] = 0
        energy = np.cumsum(blk ** 2, axis=0)
        energy = energy[win_length:, :] - energy[:-win_length, :]
        energy[np.abs(energy) < 1e-6] = 0
        yin = energy[:1, :] + energy - 2 * acf
        num = yin[min_period: max_period + 1, :]
        tau = np.arange(1, max_period + 1).reshape(-1, 1)
        cum_mean = np.cumsum(yin[1: max_period + 1, :], axis=0) / tau
        den = cum_mean[min_period - 1: max_period, :]
        out[:, s:s + blk.shape[1]] = num / (den + tiny(den))
    return out, min_period, max_period


def yin(y: np.ndarray, fmin: float, fmax: float, sr: float = 22050, frame_length: int = 2048, hop_length: int = 512,
        trough_threshold: float = 0.1, center: bool = True) -> np.ndarray:
    """librosa.yin: first CMND trough below the threshold (else the global minimum) + parabolic refinement."""
    cm, min_period, _ = cmnd_frames(y, sr, fmin, fmax, frame_length, hop_length, center)
    shifts = np.zeros_like(cm)
    a = cm[2:] + cm[:-2] - 2 * cm[1:-1]
    b = (cm[2:] - cm[:-2]) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        shifts[1:-1] = np.where(np.abs(b) >= np.abs(a), 0, -b / a)
    pad = np.pad(cm, ((1, 1), (0, 0)), mode="edge")
    trough = (cm < pad[:-2]) & (cm <= pad[2:])
    trough[0, :] = cm[0, :] < cm[1, :]
    ok = trough & (cm < trough_threshold)
    period = np.argmax(ok, axis=0)
    none = np.all(~ok, axis=0)
    period[none] = np.argmin(cm, axis=0)[none]
    period = min_period + period + np.take_along_axis(shifts, period[None, :], axis=0)[0]
    return sr / period


# ---------------------------------------------------------------------------
# Dormant-branch feature ops (pure_vocal_pause_detector.py:410-459,937-1018)
# librosa 0.10 `pyin`, `lpc`, `feature.spectral_centroid`, `feature.zero_crossing_rate`, `amplitude_to_db`,
# `note_to_hz` restated from the published algorithms (parity unpinned: librosa is not installed here).
# ---------------------------------------------------------------------------

def note_to_hz(note: str) -> float:
    """librosa.note_to_hz for plain notes like 'C2', 'C7' (A4 = 440 Hz, equal temperament)."""
    pitch = {"C": 0, "D": 2, "E": 4, "F": 5, "G": 7, "A": 9, "B": 11}[note[0].upper()]
    rest = note[1:]
    while rest and rest[0] in "#b!":
        pitch += 1 if rest[0] == "#" else -1
        rest = rest[1:]
    midi = 12 * (int(rest) + 1) + pitch
    return float(440.0 * (2.0 ** ((midi - 69.0) / 12.0)))


def amplitude_to_db(S: np.ndarray, ref=1.0, amin: float = 1e-5, top_db: Optional[float] = 80.0) -> np.ndarray:
    S = np.asarray(S)
    magnitude = np.abs(S)
    ref_value = np.abs(ref(magnitude)) if callable(ref) else np.abs(ref)
    power = np.square(magnitude, out=magnitude.copy())
    log_spec = 10.0 * np.log10(np.maximum(amin ** 2, power))
    log_spec -= 10.0 * np.log10(np.maximum(amin ** 2, ref_value ** 2))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def spectral_centroid(y: np.ndarray, sr: float = 22050, n_fft: int = 2048, hop_length: int = 512) -> np.ndarray:
    """librosa.feature.spectral_centroid(y=...): sum(freq * util.normalize(S, norm=1, axis=-2)) over the magnitude
    spectrogram.  `normalize` measures the column lengths in float64 and stores S / length back in S's dtype."""
    S = _spectrogram(y, n_fft, hop_length, 1.0)                      # float32 magnitudes for float32 input
    freq = np.fft.rfftfreq(n_fft, 1.0 / sr).reshape(-1, 1)
    length = np.sum(np.abs(S).astype(float), axis=-2, keepdims=True)
    length[length < tiny(S)] = 1.0
    snorm = np.empty_like(S)
    snorm[:] = S / length
    return np.sum(freq * snorm, axis=-2, keepdims=True)


def zero_crossing_rate(y: np.ndarray, frame_length: int = 2048, hop_length: int = 512, center: bool = True) -> np.ndarray:
    """librosa.feature.zero_crossing_rate: edge-padded centred frames, sign changes with pad=False, mean per frame."""
    y = np.asarray(y)
    if center:
        y = np.pad(y, int(frame_length // 2), mode="edge")
    x = frame(y, frame_length, hop_length).copy()
    x[np.abs(x) <= 1e-10] = 0                                        # librosa.zero_crossings(threshold=1e-10)
    sign = np.signbit(x)                                             # zero_pos=True
    cross = np.empty(x.shape, dtype=bool)
    cross[0, :] = False                                              # pad=False
    cross[1:, :] = sign[1:, :] != sign[:-1, :]
    return np.mean(cross, axis=-2, keepdims=True)


def _localmin0(x: np.ndarray) -> np.ndarray:
    """librosa.util.localmin along axis 0: x[i] < x[i-1] and x[i] <= x[i+1] with edge padding."""
    pad = np.pad(x, 1, mode="edge")
    return (x < pad[:-2]) & (x <= pad[2:])


def _boltzmann_pmf(k: np.ndarray, lam: float, n: np.ndarray) -> np.ndarray:
    """scipy.stats.boltzmann.pmf(k, lam, N) evaluated as scipy does (`fact * exp(-lam * k)`,
    fact = (1 - e^-lam) / (1 - e^(-lam N))) for 0 <= k < N, else 0 (tests compare it with scipy's own)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        fact = (1.0 - np.exp(-lam)) / (1.0 - np.exp(-lam * n))
        p = fact * np.exp(-lam * k)
    return np.where((k >= 0) & (k < n), p, 0.0)


def pyin_tables(n_thresholds: int = 100, beta_parameters=(2, 18)) -> Tuple[np.ndarray, np.ndarray]:
    thresholds = np.linspace(0, 1, n_thresholds + 1)
    beta_cdf = scipy.stats.beta.cdf(thresholds, beta_parameters[0], beta_parameters[1])
    return thresholds, np.diff(beta_cdf)


def transition_local_triangle(n_states: int, width: int) -> np.ndarray:
    """librosa.sequence.transition_local(n_states, width, window='triangle', wrap=False)."""
    win = scipy.signal.get_window("triangle", width, fftbins=False)
    lpad = (n_states - width) // 2
    base = np.zeros(n_states)
    base[lpad: lpad + width] = win
    trans = np.zeros((n_states, n_states))
    for i in range(n_states):
        row = np.roll(base, n_states // 2 + i + 1)
        row[min(n_states, i + width // 2 + 1):] = 0
        row[: max(0, i - width // 2)] = 0
        trans[i] = row
    trans /= trans.sum(axis=1, keepdims=True)
    return trans


def viterbi(prob: np.ndarray, transition: np.ndarray, p_init: np.ndarray) -> np.ndarray:
    """librosa.sequence.viterbi (log domain, first-maximum ties). prob [n_states, n_steps]."""
    n_states, n_steps = prob.shape
    eps = tiny(prob)
    log_trans = np.log(transition + eps)
    log_prob = np.log(prob.T + eps)
    value = np.zeros((n_steps, n_states))
    ptr = np.zeros((n_steps, n_states), dtype=np.int64)
    value[0] = log_prob[0] + np.log(p_init + eps)
    lt = log_trans.T
    for t in range(1, n_steps):
        trans_out = value[t - 1] + lt                                   # [j, i]
        ptr[t] = np.argmax(trans_out, axis=1)
        value[t] = log_prob[t] + trans_out[np.arange(n_states), ptr[t]]
    states = np.zeros(n_steps, dtype=np.int64)
    states[-1] = np.argmax(value[-1])
    for t in range(n_steps - 2, -1, -1):
        states[t] = ptr[t + 1, states[t + 1]]
    return states


def pyin_observations(cm: np.ndarray, sr: float, fmin: float, fmax: float, min_period: int, *, n_thresholds: int = 100,
                      boltzmann_parameter: float = 2.0, resolution: float = 0.1, no_trough_prob: float = 0.01):
    """librosa.core.pitch.__pyin_helper on a CMND matrix [n_lags, n_frames] ->
    (observation_probs [2 * n_pitch_bins, n_frames] float64, voiced_prob [n_frames], n_pitch_bins, bins per semitone)."""
    thresholds, beta_probs = pyin_tables(n_thresholds)
    shifts = np.zeros_like(cm)
    a = cm[2:] + cm[:-2] - 2 * cm[1:-1]
    b = (cm[2:] - cm[:-2]) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        shifts[1:-1] = np.where(np.abs(b) >= np.abs(a), 0, -b / a)
    n_bins_per_semitone = int(np.ceil(1.0 / resolution))
    n_pitch_bins = int(np.floor(12 * n_bins_per_semitone * np.log2(fmax / fmin))) + 1
    yin_probs = np.zeros_like(cm)
    for i in range(cm.shape[1]):
        col = cm[:, i]
        is_trough = _localmin0(col)
        is_trough[0] = col[0] < col[1]
        (idx,) = np.nonzero(is_trough)
        if len(idx) == 0:
            continue
        heights = col[idx]
        below = np.less.outer(heights, thresholds[1:])
        positions = np.cumsum(below, axis=0) - 1
        n_troughs = np.count_nonzero(below, axis=0)
        prior = _boltzmann_pmf(positions, boltzmann_parameter, n_troughs)
        prior[~below] = 0
        probs = prior.dot(beta_probs)
        gmin = np.argmin(heights)
        n_below_min = np.count_nonzero(~below[gmin, :])
        probs[gmin] += no_trough_prob * np.sum(beta_probs[:n_below_min])
        yin_probs[idx, i] = probs
    yin_period, frame_index = np.nonzero(yin_probs)
    period = min_period + yin_period
    period = period + shifts[yin_period, frame_index]
    f0 = sr / period
    bin_index = 12 * n_bins_per_semitone * np.log2(f0 / fmin)
    bin_index = np.clip(np.round(bin_index), 0, n_pitch_bins).astype(int)
    obs = np.zeros((2 * n_pitch_bins, cm.shape[1]))
    obs[bin_index, frame_index] = yin_probs[yin_period, frame_index]
    voiced_prob = np.clip(np.sum(obs[:n_pitch_bins, :], axis=0, keepdims=True), 0, 1)
    obs[n_pitch_bins:, :] = (1 - voiced_prob) / n_pitch_bins
    return obs, voiced_prob[0], n_pitch_bins, n_bins_per_semitone


def pyin(y: np.ndarray, fmin: float, fmax: float, sr: float = 22050, frame_length: int = 2048, hop_length: Optional[int] = None,
         max_transition_rate: float = 35.92, switch_prob: float = 0.01, fill_na=np.nan, center: bool = True):
    """librosa.pyin -> (f0 [n_frames] with NaN where unvoiced, voiced_flag, voiced_prob)."""
    if hop_length is None:
        hop_length = frame_length // 4
    cm, min_period, _ = cmnd_frames(y, sr, fmin, fmax, frame_length, hop_length, center)
    obs, voiced_prob, n_pitch_bins, bps = pyin_observations(cm, sr, fmin, fmax, min_period)
    max_semitones_per_frame = round(max_transition_rate * 12 * hop_length / sr)
    width = max_semitones_per_frame * bps + 1
    transition = np.kron(np.array([[1 - switch_prob, switch_prob], [switch_prob, 1 - switch_prob]]),
                         transition_local_triangle(n_pitch_bins, width))
    p_init = np.zeros(2 * n_pitch_bins)
    p_init[n_pitch_bins:] = 1 / n_pitch_bins
    states = viterbi(obs, transition, p_init)
    freqs = fmin * 2 ** (np.arange(n_pitch_bins) / (12 * bps))
    f0 = freqs[states % n_pitch_bins]
    voiced_flag = states < n_pitch_bins
    if fill_na is not None:
        f0 = f0.copy()
        f0[~voiced_flag] = fill_na
    return f0, voiced_flag, voiced_prob


def lpc(y: np.ndarray, order: int) -> np.ndarray:
    """librosa.lpc (Burg's method) in the dtype of `y` -> [order + 1] coefficients, a[0] = 1."""
    y = np.asarray(y)
    dtype = y.dtype
    ar = np.zeros(order + 1, dtype=dtype); ar[0] = 1
    ar_prev = ar.copy()
    eps = dtype.type(tiny(y))
    fwd = y[1:]
    bwd = y[:-1]
    den = np.sum(fwd ** 2 + bwd ** 2, axis=0).astype(dtype)
    for i in range(order):
        rc = np.sum(bwd * fwd, axis=0).astype(dtype)
        rc = dtype.type(rc * dtype.type(-2))
        rc = dtype.type(rc / (den + eps))
        ar_prev, ar = ar, ar_prev
        for j in range(1, i + 2):
            ar[j] = ar_prev[j] + rc * ar_prev[i - j + 1]
        fwd_tmp = fwd
        fwd = fwd + rc * bwd
        bwd = bwd + rc * fwd_tmp
        q = dtype.type(1) - rc ** 2
        den = dtype.type(q * den - bwd[-1] ** 2 - fwd[0] ** 2)
        fwd = fwd[1:]
        bwd = bwd[:-1]
    return ar
