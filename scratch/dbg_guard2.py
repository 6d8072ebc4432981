import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.testing import signals
ctx = _native.Context()
song = signals.c2_song(20.0, seed=3); gated = signals.c1_sine_silence(20.0, seed=5)
rng = np.random.default_rng(0)
for x in (song, gated):
    xd = ctx.to_device(x)
    idx = np.sort(rng.integers(1, len(x) - 1, 64))
    span, win = 19845, 3528
    garg, gval = ctx.quiet_guard_slow(xd, idx, span, win)
    for q, i in enumerate(idx):
        seg = x[i:min(len(x), i+span)]
        if seg.size <= win: continue
        padded = np.pad(seg, (0, win-1), mode='edge')
        lvl = np.sqrt(np.convolve(padded*padded, np.ones(win)/win, mode='valid') + 1e-12)
        rdb = 20*np.log10(lvl+1e-12)
        k = int(np.argmin(rdb))
        if k != garg[q]:
            print("MISMATCH", i, len(seg), garg[q], k, gval[q], rdb[0], rdb[k], rdb[garg[q]] if garg[q] < len(rdb) else None)
print("done")
