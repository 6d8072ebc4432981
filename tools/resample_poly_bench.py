"""ac_resample_poly / ac_resample_poly_segments with the soxr-HQ-specification filters: time per call for the two conversions of the
path (30-min 48 kHz -> 44.1 kHz load leg; every chunk of a 4-min track 44.1 kHz -> 16 kHz in front of Silero)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
hip = _native.Context()
rng = np.random.default_rng(0)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


x48 = hip.to_device(rng.standard_normal(1800 * 48000).astype(np.float32))
ms = timed(lambda: hip.resample_poly(x48, 147, 160), 3)
print(f"48k -> 44.1k, 30 min ({x48.numel() / 1e6:.1f} M in): {ms:8.2f} ms  ({(x48.numel() * 4 + x48.numel() * 147 // 160 * 4) / ms / 1e6:.1f} GB/s algorithmic)", flush=True)
del x48
n_chunk, chunk = 32, int(10.0 * 44100)
x = hip.to_device(rng.standard_normal(n_chunk * chunk).astype(np.float32))
offs = [i * chunk for i in range(n_chunk)]; lens = [chunk] * n_chunk
ms = timed(lambda: hip.resample_poly_segments(x, offs, lens, 160, 441, bucket=4096))
print(f"44.1k -> 16k, 32 chunks of 10 s: {ms:8.2f} ms", flush=True)
