"""Seeded synthetic inputs of SURVEY.md §8d (C1 sine+silence, C2 "song"), shared by tests and bench.py.

All signals are float32, peak <= 0.9, generated with numpy.random.default_rng(seed).
"""
from __future__ import annotations

import numpy as np

SR = 44100


def c1_sine_silence(duration_s: float = 60.0, seed: int = 1, sr: int = SR) -> np.ndarray:
    """Alternating 440 Hz bursts (amp 0.5, 2-4 s, 10 ms raised-cosine edges) and exact-zero silences (0.6-1.5 s)."""
    rng = np.random.default_rng(seed)
    n = int(round(duration_s * sr))
    out = np.zeros(n, dtype=np.float32)
    edge = int(round(0.010 * sr))
    ramp = (0.5 - 0.5 * np.cos(np.pi * np.arange(edge) / edge)).astype(np.float64)
    pos = int(round(rng.uniform(0.3, 0.8) * sr))
    while pos < n:
        length = int(round(rng.uniform(2.0, 4.0) * sr))
        length = min(length, n - pos)
        if length <= 2 * edge:
            break
        t = np.arange(length) / float(sr)
        burst = 0.5 * np.sin(2 * np.pi * 440.0 * t)
        burst[:edge] *= ramp
        burst[-edge:] *= ramp[::-1]
        out[pos:pos + length] = burst.astype(np.float32)
        pos += length + int(round(rng.uniform(0.6, 1.5) * sr))
    return out


def c2_song(duration_s: float = 240.0, seed: int = 2, sr: int = SR, stereo: bool = False) -> np.ndarray:
    """Vocal-like harmonic stack (f0 110-330 Hz, vibrato, 2-6 s phrases, 0.3-1.2 s rests) over a
    120 BPM kick / noise-hat pattern and a sustained chord; L/R decorrelated by 5 %.
    Returns the mono down-mix (channel mean, as `librosa.load(mono=True)` does,
    `audio_processor.py:45-49`) unless `stereo=True`."""
    rng = np.random.default_rng(seed)
    n = int(round(duration_s * sr))
    t = np.arange(n) / float(sr)
    # --- vocal-like part, gated into phrases
    gate = np.zeros(n, dtype=np.float64)
    f0 = np.zeros(n, dtype=np.float64)
    pos = int(round(rng.uniform(0.5, 1.5) * sr))
    edge = int(round(0.030 * sr))
    ramp = 0.5 - 0.5 * np.cos(np.pi * np.arange(edge) / edge)
    while pos < n:
        length = min(int(round(rng.uniform(2.0, 6.0) * sr)), n - pos)
        if length <= 2 * edge:
            break
        g = np.ones(length)
        g[:edge] = ramp
        g[-edge:] = ramp[::-1]
        gate[pos:pos + length] = g
        note = rng.uniform(110.0, 330.0)
        f0[pos:pos + length] = note * (1.0 + 0.012 * np.sin(2 * np.pi * rng.uniform(4.5, 6.5) * t[:length]))
        pos += length + int(round(rng.uniform(0.3, 1.2) * sr))
    f0[f0 == 0] = 220.0
    phase = 2 * np.pi * np.cumsum(f0) / float(sr)
    voice = np.zeros(n)
    for h in range(1, 9):
        voice += (0.6 / h) * np.sin(h * phase + rng.uniform(0, 2 * np.pi))
    voice *= gate * 0.30
    # --- backing: 120 BPM kick + hats + chord
    beat = 0.5
    kick_t = (t % beat)
    kick = np.sin(2 * np.pi * (55.0 + 60.0 * np.exp(-kick_t * 30.0)) * kick_t) * np.exp(-kick_t * 14.0)
    hat_t = ((t + beat / 2) % beat)
    hats = rng.standard_normal(n) * np.exp(-hat_t * 60.0)
    chord = sum(np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi)) for f in (130.81, 164.81, 196.0)) / 3.0
    backing = 0.22 * kick + 0.05 * hats + 0.08 * chord
    decor = rng.standard_normal(n) * 0.05
    left = voice + backing * (1.0 + 0.05) + 0.02 * decor
    right = voice + backing * (1.0 - 0.05) - 0.02 * decor
    st = np.stack([left, right])
    peak = np.max(np.abs(st))
    if peak > 0.9:
        st *= 0.9 / peak
    st = st.astype(np.float32)
    if stereo:
        return st
    return np.mean(st, axis=0).astype(np.float32)


def vocal_like(duration_s: float, seed: int, sr: int = SR) -> np.ndarray:
    """A stand-in 'separated vocal' with clean phrase gaps (for detector-only tests): the voice part of c2_song
    plus a -70 dB noise bed so silences are quiet but not exact zeros."""
    rng = np.random.default_rng(seed + 7919)
    song = c2_song(duration_s, seed=seed, sr=sr)
    n = len(song)
    bed = (rng.standard_normal(n) * 3e-4).astype(np.float32)
    # crude "separation": high-pass-ish residual of the mix by differencing removes most of the kick/chord
    voc = np.empty_like(song)
    voc[0] = song[0]
    voc[1:] = song[1:] - 0.97 * song[:-1]
    return (voc + bed).astype(np.float32)


def voice_with_rests(duration_s: float, seed: int, sr: int = SR) -> np.ndarray:
    """A clean sung line for the multi-feature (pyin / LPC) detector branch: harmonic stacks with vibrato and a
    formant-like spectral tilt in 1.2-3 s phrases, separated by 0.15-0.9 s rests that hold only a -75 dB noise bed
    (so rests are > 40 dB below the peak RMS but never exact zeros), plus short breath-like noise puffs in some rests."""
    rng = np.random.default_rng(seed + 104729)
    n = int(round(duration_s * sr))
    t = np.arange(n) / float(sr)
    out = rng.standard_normal(n) * 1.2e-4
    pos = int(round(rng.uniform(0.3, 0.8) * sr))
    edge = int(round(0.020 * sr))
    ramp = 0.5 - 0.5 * np.cos(np.pi * np.arange(edge) / edge)
    while pos < n:
        length = min(int(round(rng.uniform(1.2, 3.0) * sr)), n - pos)
        if length <= 2 * edge:
            break
        note = rng.uniform(130.0, 390.0)
        f0 = note * (1.0 + 0.010 * np.sin(2 * np.pi * rng.uniform(4.5, 6.5) * t[:length]))
        phase = 2 * np.pi * np.cumsum(f0) / float(sr)
        seg = np.zeros(length)
        for h in range(1, 13):
            fh = note * h
            tilt = 1.0 / (1.0 + ((fh - 700.0) / 600.0) ** 2) + 0.4 / (1.0 + ((fh - 1800.0) / 500.0) ** 2) + 0.05
            seg += tilt * np.sin(h * phase + rng.uniform(0, 2 * np.pi)) / np.sqrt(h)
        g = np.ones(length)
        g[:edge] = ramp
        g[-edge:] = ramp[::-1]
        out[pos:pos + length] += 0.25 * seg * g / np.max(np.abs(seg))
        rest = int(round(rng.uniform(0.15, 0.9) * sr))
        if rest > int(0.5 * sr) and rng.uniform() < 0.5:      # a breath puff inside a long rest
            b0 = pos + length + int(0.1 * sr)
            bl = int(0.12 * sr)
            if b0 + bl < n:
                out[b0:b0 + bl] += rng.standard_normal(bl) * 0.004 * np.hanning(bl)
        pos += length + rest
    return out.astype(np.float32)


def c5_long_form(duration_s: float = 1800.0, seed: int = 5, sr: int = SR, section_s: float = 60.0, stereo: bool = False) -> np.ndarray:
    """BASELINE configs[4]: the C2 generator looped with per-section seeds (`seed * 1000 + i`),
    each section with its own level so no two minutes are alike.  `sr=48000, stereo=True` gives the 48 kHz stereo source the
    loader leg resamples; `sr=44100` (mono channel mean) is the post-resample parity input."""
    n = int(round(duration_s * sr))

    def section(i: int, length: float) -> np.ndarray:
        return c2_song(length, seed=seed * 1000 + i, sr=sr, stereo=stereo) * np.float32(0.55 + 0.45 * ((i * 7) % 10) / 9.0)

    # the sections are independent: plan them first (a section of `length` seconds is round(length * sr) samples), then generate eight at
    # a time on a thread pool (the generator is numpy-bound); a plan that the generator does not confirm falls back to the serial loop
    plan, got = [], 0
    while got < n:
        length = min(section_s, (n - got) / float(sr))
        plan.append(length)
        got += int(round(length * sr))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as pool:
        parts = list(pool.map(lambda it: section(*it), enumerate(plan)))
    if any(p.shape[-1] != int(round(length * sr)) for p, length in zip(parts, plan)):
        parts, i, got = [], 0, 0
        while got < n:
            sec = section(i, min(section_s, (n - got) / float(sr)))
            parts.append(sec)
            got += sec.shape[-1]
            i += 1
    out = np.concatenate(parts, axis=-1)[..., :n]
    return np.ascontiguousarray(out, dtype=np.float32)
