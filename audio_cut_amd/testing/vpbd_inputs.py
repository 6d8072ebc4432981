"""Seeded inputs for the VPBD host logic (tests + golden generation): fake pauses (as the reference's own tests inject,
`tests/integration/test_pipeline_vpbd_acoustic_fallback.py:86-96`), a synthetic cache and a gated-noise vocal."""
import types

import numpy as np

SR = 44100


def vpbd_case(seed: int, duration: float = 60.0):
    """cache: beats every 0.5 s, a loud middle section, an MDD series with valleys; 40 pauses; gated-noise vocal."""
    rng = np.random.default_rng(seed)
    n_frames = int(duration / 0.05) + 1
    rms = np.full(n_frames, 0.08, np.float32)
    rms[int(0.3 * n_frames): int(0.75 * n_frames)] = 0.6 + 0.1 * rng.random(int(0.75 * n_frames) - int(0.3 * n_frames)).astype(np.float32)
    mdd = (0.5 + 0.4 * np.sin(np.arange(n_frames) * 0.21) * rng.uniform(0.5, 1.0, n_frames)).astype(np.float32)
    cache = types.SimpleNamespace(beat_times=np.arange(0.25, duration, 0.5, dtype=np.float32), rms_series=rms, mdd_series=mdd,
                                  hop_s=0.05, duration_s=duration)
    times = np.sort(rng.uniform(1.0, duration - 1.0, 40))
    pauses = [types.SimpleNamespace(cut_point=float(t), confidence=float(c), start_time=float(t) - 0.2, end_time=float(t) + 0.25,
                                    duration=0.45, pause_type="energy_valley_mdd")
              for t, c in zip(times, rng.uniform(0.2, 0.99, 40))]
    n = int(duration * SR)
    env = (np.sin(2 * np.pi * 0.07 * np.arange(n) / SR) > -0.2).astype(np.float32)
    vocal = (rng.standard_normal(n).astype(np.float32) * 0.2 * env)
    return cache, pauses, vocal


class FixedPauses:
    def __init__(self, pauses):
        self.pauses = pauses

    def detect_pure_vocal_pauses(self, *a, **k):
        return list(self.pauses)
