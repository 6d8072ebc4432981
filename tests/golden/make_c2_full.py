"""Writes tests/golden/c2_full_oracle.npz: the CPU oracle's result on the full BASELINE configs[1] track (4-min C2 song,
seed 2, seeded synthetic weights seed 0); with arguments `SECONDS SEED NAME`, another track of the same generator into
tests/golden/NAME.npz (c2_150s_seed11_oracle.npz was written with `150 11 c2_150s_seed11_oracle`); two more optional
arguments pick the signal generator and the weight seed (c1_170s_seed64_w17_oracle.npz: `170 64 c1_170s_seed64_w17_oracle
c1_sine_silence 17`, the soak track whose guard boundary #8 sits on a threshold crossing - DESIGN.md 7).  The oracle takes minutes on the CPU, so the GPU test compares against this
fixture instead of re-running it (the same comparison, run live on the GPU box, is scratch/c2_full.py).
Run from the repo root:  python tests/golden/make_c2_full.py"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights  # noqa: E402
from audio_cut_amd.testing import signals  # noqa: E402
from oracle import e2e as OE, refine as OR  # noqa: E402

if __name__ == "__main__":
    OR.LEGACY_PROMOTION = True
    seconds, seed, name = (float(sys.argv[1]), int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else (240.0, 2, "c2_full_oracle")
    generator, weight_seed = (sys.argv[4] if len(sys.argv) > 4 else "c2_song"), (int(sys.argv[5]) if len(sys.argv) > 5 else 0)
    mix = getattr(signals, generator)(seconds, seed=seed)
    mix = np.mean(mix, axis=0).astype(np.float32) if mix.ndim == 2 else mix
    w = synth_weights(TfcTdfSpec(), seed=weight_seed)
    t0 = time.time()
    ref = OE.run_track(mix, 44100, w)
    print(f"oracle: {time.time() - t0:.1f} s, {len(ref.sample_boundaries)} boundaries, {len(ref.policy.cuts)} manifest cuts")
    sec = 44100
    nsec = len(mix) // sec
    np.savez_compressed(ROOT / "tests" / "golden" / f"{name}.npz", seconds=np.float64(seconds), seed=np.int64(seed),
                        sample_boundaries=np.asarray(ref.sample_boundaries, dtype=np.int64),
                        cuts=np.asarray(ref.policy.cuts, dtype=np.int64), flags=np.asarray(ref.policy.flags, dtype=np.int8),
                        pieces=np.asarray(ref.policy.pieces, dtype=np.int64),
                        pause_cut_points=np.asarray([p.cut_point for p in ref.pauses], dtype=np.float64),
                        vocal_rms_per_second=np.sqrt(np.mean(ref.vocal[: nsec * sec].reshape(nsec, sec).astype(np.float64) ** 2, axis=1)),
                        vocal_head=ref.vocal[: 4 * sec: 7].astype(np.float32), vocal_peak=np.float64(np.max(np.abs(ref.vocal))),
                        vad_segments=np.asarray([[s["start"], s["end"]] for s in ref.vad_segments], dtype=np.float64),
                        cache_rms=np.asarray(ref.cache.rms_series, dtype=np.float32), beat_times=np.asarray(ref.cache.beat_times, dtype=np.float64))
    print(f"wrote tests/golden/{name}.npz")
