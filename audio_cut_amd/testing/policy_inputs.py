"""Deterministic inputs for the post-path boundary policy tests: shared by
tests/golden/make_golden.py (which runs the reference on them) and the CPU / GPU tests."""
from __future__ import annotations

import numpy as np

from . import signals

SR = 44100


def policy_case(seed: int):
    """A 60 s 'vocal stem' with phrases, rests, a long busy stretch and an instrumental break, a cut list with fragments,
    a 50 ms-hop RMS series, a beat grid and a few guard-suppressed points -> (vocal, cuts, rms, hop_s, beats, suppressed)."""
    rng = np.random.default_rng(seed)
    voc = signals.voice_with_rests(60.0, seed=seed)
    n = len(voc)
    a, b = int(20.0 * SR), int(33.5 * SR)
    voc[a:b] += signals.voice_with_rests(13.5, seed=seed + 1)[: b - a] * 0.5
    voc[int(40.0 * SR): int(47.0 * SR)] *= 0.002
    cuts = sorted(set([0, n] + [int(t * SR) for t in np.sort(rng.uniform(1.0, 59.0, 14))] + [int(40.9 * SR), int(41.0 * SR) + 200]))
    hop_s = 0.05
    frame, hop = int(SR * 0.1), int(SR * hop_s)
    pad = np.pad(voc, frame // 2)                       # librosa.feature.rms(center=True) framing
    nf = 1 + (len(pad) - frame) // hop
    idx = np.arange(frame)[None, :] + hop * np.arange(nf)[:, None]
    rms = np.sqrt(np.mean(np.abs(pad[idx]) ** 2, axis=1, dtype=np.float32)).astype(np.float32)
    beats = np.arange(0.25, 60.0, 0.5)
    supp = [(float(t), float(sc)) for t, sc in zip(rng.uniform(2.0, 58.0, 6), rng.uniform(0.2, 0.9, 6))]
    return voc, cuts, rms, hop_s, beats, supp


def random_layout_case(rng: np.random.Generator, case: int):
    """One random segmentation for the layout refiner -> (edges, kinds, rms, hop_s, beats, suppressed, config kwargs, midpoint)."""
    k = int(rng.integers(2, 14))
    durs = rng.choice([0.4, 0.9, 1.5, 2.5, 4.0, 6.5, 9.0, 13.0, 19.0, 26.0], size=k) * rng.uniform(0.85, 1.15, size=k)
    edges = np.concatenate(([0.0], np.cumsum(durs)))
    kinds = ["human" if rng.uniform() < 0.6 else "music" for _ in range(k)]
    total = float(edges[-1])
    hop_s = 0.05
    frames = int(total / hop_s) + 2
    rms = (0.2 + 0.15 * np.sin(np.arange(frames) * 0.07) + 0.1 * rng.uniform(size=frames)).astype(np.float32)
    for _ in range(int(total / 6) + 1):
        c = int(rng.integers(2, frames - 2))
        rms[max(0, c - 3): c + 3] *= 0.05
    beats = np.arange(0.3, total, 0.5)
    supp = [(float(t), float(sc)) for t, sc in zip(rng.uniform(0.5, total - 0.5, 4), rng.uniform(0.1, 0.9, 4))]
    cfg = dict(enable=True, micro_merge_s=2.0, soft_min_s=5.0, soft_max_s=12.0, min_gap_s=1.0, beat_snap_ms=50.0)
    if case % 5 == 4:
        cfg.update(soft_max_s=18.0, beat_snap_ms=0.0, min_gap_s=1.2)
    return edges, kinds, rms, hop_s, beats, supp, cfg, (case % 7 == 0)
