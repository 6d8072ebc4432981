import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.testing import signals
ctx = _native.Context()
x = signals.c2_song(20.0, seed=3)
xd = ctx.to_device(x)
idx = np.array([1000, 50000, 300000, 700000])
for span, win in [(5000, 3528), (12000, 3528), (19845, 3528), (19845, 441), (9000, 1000)]:
    garg, gval = ctx.quiet_guard_slow(xd, idx, span, win)
    for q, i in enumerate(idx):
        seg = x[i:i+span]
        padded = np.pad(seg, (0, win-1), mode='edge')
        lvl = np.sqrt(np.convolve(padded*padded, np.ones(win)/win, mode='valid') + 1e-12)
        rdb = 20*np.log10(lvl+1e-12)
        print(span, win, i, garg[q], int(np.argmin(rdb)), gval[q], rdb[0], rdb.min(), rdb[garg[q]])
