"""ac_stft2048_features on a 4-min track (every launch on a different copy: HBM rates), time and output hash.
usage: [AC_LIB=libaudiocut_hip_<tag>.so] python tools/stft_bench.py"""
import os, sys, hashlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
if os.environ.get("AC_LIB"):
    os.environ["AUDIOCUT_HIP_LIBNAME"] = os.environ["AC_LIB"]
from audio_cut_amd.testing import signals
hip = _native.Context()
mix = hip.to_device(signals.c2_song(240.0, seed=2))
copies = [mix] + [mix.clone() for _ in range(7)]
n = mix.numel()
for name, kw, hop, nbytes in (("hop 441 flatness", dict(want_flat=True, want_mel=False), 441, 4 * n + 4 * (1 + n // 441)),
                              ("hop 512 mel-128", dict(want_flat=False, want_mel=True), 512, 4 * n + 512 * (1 + n // 512)),
                              ("hop 441, first 10 s (one chunk)", dict(want_flat=True, want_mel=False), 441, None)):
    srcs = [c[:441000] for c in copies] if nbytes is None else copies
    nb = nbytes if nbytes is not None else 4 * 441000 + 4 * 1001
    out = hip.stft2048_features(srcs[0], hop, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(16):
        hip.stft2048_features(srcs[i % 8], hop, **kw)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 16
    o = out[0] if kw["want_flat"] else out[1]
    print(f"{name}: {ms:7.4f} ms  {nb / ms / 1e6:7.1f} GB/s  sha1 {hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
