"""TEST INFRASTRUCTURE ONLY (never imported by the product): CPU restatement of the reference's resampling steps.

The reference resamples twice on the path: `librosa.load(path, sr=44100)` (`src/vocal_smart_splitter/utils/audio_processor.py:44-48`)
and `librosa.resample(audio, orig_sr=sr, target_sr=16000)` in front of Silero (`src/vocal_smart_splitter/core/vocal_pause_detector.py:189`);
both run librosa's default `res_type="soxr_hq"`, i.e. libsoxr at its HQ quality.  Neither librosa nor soxr exists offline and
libsoxr's own implementation (half-band stages + an interpolated poly-phase stage) cannot be reproduced bit for bit, so this row's
parity is "coefficients unpinned" (DESIGN.md 6 row 2).  What CAN be matched is the published specification of the HQ recipe, and this
module designs to it, independently of the product's designer (`audio_cut_amd/_native.py:Context._resample_filter`):

  soxr.h / soxr.c `soxr_quality_spec(SOXR_HQ, 0)`: precision 20 bits, linear phase, stop band begins at the Nyquist frequency of the
  lower of the two rates (stopband_begin = 1), pass band ends at 1 - 0.05 / TO_3dB(rej) of it with rej = 20 bits * 6.0206 dB and
  TO_3dB(a) = (1.6e-6 a - 7.5e-4) a + 0.646  ->  0.91363; rate.h designs its low-pass for (bits + 1) * 6.0206 dB = 126.43 dB.

One Kaiser-windowed sinc at the common rate up * fs_in to that specification (pass-band edge, stop-band edge, attenuation ->
length and beta by Kaiser's formulas), applied as a poly-phase FIR with zero padding, output sample 0 aligned with input sample 0 and
ceil(n * up / down) outputs (librosa's length rule) - scipy.signal.resample_poly's framing with this filter."""
from __future__ import annotations

import numpy as np
import scipy.signal
import scipy.special

SOXR_HQ_BITS = 20
_REJ_DB = SOXR_HQ_BITS * 20.0 * np.log10(2.0)                                  # 120.41 dB
SOXR_HQ_PASSBAND_END = 1.0 - 0.05 / ((1.6e-6 * _REJ_DB - 7.5e-4) * _REJ_DB + 0.646)        # 0.91363 of the lower Nyquist
SOXR_HQ_STOPBAND_BEGIN = 1.0
SOXR_HQ_ATT_DB = (SOXR_HQ_BITS + 1) * 20.0 * np.log10(2.0)                      # 126.43 dB


def soxr_hq_lowpass(up: int, down: int) -> np.ndarray:
    """float64 taps (odd length, symmetric, unit DC gain) of the anti-alias / anti-image low-pass at the rate up * fs_in."""
    g = int(np.gcd(int(up), int(down)))
    up, down = int(up) // g, int(down) // g
    rate = max(up, down)
    f_pass = SOXR_HQ_PASSBAND_END / rate                  # in units of the common rate's Nyquist frequency
    f_stop = SOXR_HQ_STOPBAND_BEGIN / rate
    d_omega = np.pi * (f_stop - f_pass)                   # transition width, rad / sample
    beta = 0.1102 * (SOXR_HQ_ATT_DB - 8.7)                # Kaiser (att > 50 dB)
    n_taps = int(np.ceil((SOXR_HQ_ATT_DB - 7.95) / (2.285 * d_omega))) + 1
    n_taps += 1 - (n_taps & 1)                            # odd: a zero-phase filter centred on a tap
    half = (n_taps - 1) // 2
    k = np.arange(-half, half + 1, dtype=np.float64)
    fc = 0.5 * (f_pass + f_stop)                          # -6 dB point in the middle of the transition band
    ideal = fc * np.sinc(fc * k)
    win = scipy.special.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - (k / half) ** 2))) / scipy.special.i0(beta)
    h = ideal * win
    return h / np.sum(h)


def resample(x: np.ndarray, up: int, down: int) -> np.ndarray:
    """x at fs_in -> fs_in * up / down, float32 out (the arithmetic runs in float64)."""
    g = int(np.gcd(int(up), int(down)))
    up, down = int(up) // g, int(down) // g
    x = np.asarray(x, dtype=np.float32)
    if up == down:
        return x.copy()
    return scipy.signal.resample_poly(x.astype(np.float64), up, down, window=soxr_hq_lowpass(up, down)).astype(np.float32)


def response_db(h: np.ndarray, freqs_in_common_nyquist: np.ndarray) -> np.ndarray:
    """|H| in dB of taps h at the given frequencies (in units of the common rate's Nyquist): a 2^22-point FFT of the taps, read at
    the nearest bin (resolution 4.8e-7 of Nyquist; the transition band of the longest filter here spans 200 bins)."""
    n_fft = 1 << 22
    mag = np.abs(np.fft.rfft(np.asarray(h, dtype=np.float64), n_fft))
    idx = np.clip(np.rint(np.asarray(freqs_in_common_nyquist, dtype=np.float64) * (n_fft // 2)).astype(np.int64), 0, n_fft // 2)
    return 20.0 * np.log10(np.maximum(mag[idx], 1e-300))
