"""Closed-form known-answer tests for the restated librosa ops (the only anchor for those float series:
librosa itself is not installed).  CPU only."""
import numpy as np

from oracle import librosa_ops as L

SR = 44100


def test_rms_of_sine_and_frame_count():
    t = np.arange(SR * 2) / SR
    x = (0.5 * np.sin(2 * np.pi * 441.0 * t)).astype(np.float32)        # 100 samples per period
    for frame, hop in ((4410, 2205), (1102, 441), (2048, 441), (2205, 882)):
        r = L.rms(x, frame_length=frame, hop_length=hop)[0]
        assert r.dtype == np.float32
        pad = frame // 2
        assert len(r) == 1 + (len(x) + 2 * pad - frame) // hop
        mid = r[len(r) // 4: -len(r) // 4]
        np.testing.assert_allclose(mid, 0.5 / np.sqrt(2), rtol=2e-3)     # A / sqrt(2)
    assert np.all(L.rms(np.zeros(10000, np.float32), 2048, 441) == 0)


def test_stft_matches_dft_definition():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(8192).astype(np.float32)
    S = L.stft(x, n_fft=2048, hop_length=441)
    assert S.shape == (1025, 1 + len(x) // 441) and S.dtype == np.complex64
    k = 7
    seg = np.pad(x, 1024)[k * 441: k * 441 + 2048].astype(np.float64)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(2048) / 2048)
    np.testing.assert_allclose(S[:, k], np.fft.rfft(seg * w).astype(np.complex64), rtol=1e-6, atol=1e-6)


def test_flatness_white_noise_and_tone():
    rng = np.random.default_rng(1)
    noise = rng.standard_normal(SR * 2).astype(np.float32)
    f = L.spectral_flatness(noise, hop_length=441)[0]
    # power spectrum of Gaussian noise is exponentially distributed per bin: E[gmean/amean] = exp(-gamma) = 0.5615
    assert abs(float(np.mean(f[5:-5])) - np.exp(-np.euler_gamma)) < 0.02
    t = np.arange(SR) / SR
    tone = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    assert float(np.max(L.spectral_flatness(tone, hop_length=441)[0][3:-3])) < 1e-5
    np.testing.assert_allclose(L.spectral_flatness(np.zeros(8192, np.float32), hop_length=441)[0], 1.0, rtol=1e-5)   # amin floor everywhere


def test_mel_filterbank_slaney_properties():
    M = L.mel_filters(SR, 2048, 128, 0.0, SR / 2)
    assert M.shape == (128, 1025) and M.dtype == np.float32
    assert np.all(M >= 0)
    peaks = np.argmax(M, axis=1)
    assert np.all(np.diff(peaks) >= 0)                                   # centre frequencies ascend
    # Slaney area normalisation: each triangle integrates to ~1 over frequency (bin width sr/n_fft)
    area = M.sum(axis=1) * (SR / 2048)
    assert np.all(np.abs(area - 1.0) < 0.15) and np.all(np.abs(area[60:] - 1.0) < 0.02)   # narrow low bands are 2-3 bins wide
    # below 1 kHz the scale is linear (200/3 Hz per mel)
    assert abs(L.mel_to_hz(15.0) - 1000.0) < 1e-9 and abs(L.hz_to_mel(1000.0) - 15.0) < 1e-9


def test_onset_strength_clicks_and_padding():
    x = np.zeros(SR * 4, np.float32)
    clicks = np.arange(0.5, 4.0, 0.5)
    for c in clicks:
        i = int(c * SR)
        x[i:i + 64] = 0.9
    for hop in (512, 2205):
        env = L.onset_strength(x, sr=SR, hop_length=hop)
        assert len(env) == 1 + len(x) // hop and env.dtype == np.float32
        assert env[0] == 0.0
        pk = L.onset_detect(env, sr=SR, hop_length=hop)
        t = pk * hop / SR
        assert len(pk) == len(clicks)
        assert np.max(np.abs(t - clicks)) < 2.5 * hop / SR
    assert len(L.onset_detect(np.zeros(100, np.float32), SR, 512)) == 0


def test_tempo_and_beats_of_click_track():
    bpm = 100.0
    x = np.zeros(SR * 20, np.float32)
    period = 60.0 / bpm
    for c in np.arange(0.3, 20.0, period):
        i = int(c * SR)
        x[i:i + 128] = np.hanning(128) * 0.9
    tempo, beats = L.beat_track(y=x, sr=SR, hop_length=512)
    assert abs(tempo - bpm) / bpm < 0.03
    iv = np.diff(beats) * 512 / SR
    assert abs(float(np.median(iv)) - period) < 0.03
    curve = L.tempo(L.onset_strength(x, sr=SR, hop_length=512, aggregate=np.median), sr=SR, hop_length=512, aggregate=None)
    assert abs(float(np.median(curve)) - bpm) / bpm < 0.03
    assert L.beat_track(y=np.zeros(SR * 3, np.float32), sr=SR)[0] == 0.0


def test_tempogram_lag0_is_one_and_linear_ramp_padding():
    rng = np.random.default_rng(2)
    env = np.abs(rng.standard_normal(500)).astype(np.float32)
    tg = L.tempogram(env, 160)
    assert tg.shape == (160, 500)
    np.testing.assert_allclose(tg[0], 1.0, rtol=0, atol=1e-12)          # max-normalised autocorrelation
    assert np.all(np.abs(tg) <= 1.0 + 1e-12)
