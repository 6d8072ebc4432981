"""Kernel-level parity: every C-ABI entry point against the oracle on seeded inputs (GPU box only)."""
import numpy as np
import pytest
import torch

from audio_cut_amd.testing import signals
from oracle import chunking as OC, detector as OD, librosa_ops as L, refine as OR, separator as OS

pytestmark = pytest.mark.gpu
SR = 44100


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))


@pytest.fixture(scope="module")
def song():
    return signals.c2_song(20.0, seed=3)


@pytest.fixture(scope="module")
def gated():
    return signals.c1_sine_silence(20.0, seed=5)


@pytest.mark.parametrize("frame,hop", [(4410, 2205), (1102, 441), (2048, 441), (2205, 882)])
def test_frame_rms(hip_ctx, song, gated, frame, hop):
    for x in (song, gated):
        got = hip_ctx.frame_rms(hip_ctx.to_device(x), frame, hop).cpu().numpy()
        ref = L.rms(x, frame_length=frame, hop_length=hop)[0]
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-9)
        assert np.array_equal(got == 0.0, ref == 0.0)      # exact-zero frames stay exact zeros


def test_frame_rms_multi_is_the_single_kernel_bit_for_bit(hip_ctx, song, gated):
    """ac_frame_rms_multi: the stem's three RMS series (markers 2205/882, detector 1102/441, no-vocal runs 2048/441) in one pass over
    the wave - every frame summed in ac_frame_rms's order, so the series are the SAME bits; ragged lengths (a frame centre exactly on
    the last sample, a wave shorter than one span) included."""
    cfgs = [(2205, 882), (1102, 441), (2048, 441), (4410, 2205)]
    rng = np.random.default_rng(5)
    for x in (song, gated, song[: 8192 * 3 + 1], song[: 441 * 70], rng.standard_normal(5000).astype(np.float32)):
        xd = hip_ctx.to_device(x)
        for k in (1, 3, 4):
            got = hip_ctx.frame_rms_multi(xd, cfgs[:k])
            for (frame, hop), g in zip(cfgs[:k], got):
                want = hip_ctx.frame_rms(xd, frame, hop)
                assert g.shape == want.shape and torch.equal(g, want), (len(x), frame, hop)
    # the prefetch form: one launch, each series handed to the frame_rms call that asks for it
    xd = hip_ctx.to_device(song)
    hip_ctx.prefetch_begin()
    hip_ctx.prefetch_frame_rms_multi(xd, cfgs[:3])
    a = hip_ctx.frame_rms(xd, 1102, 441)
    st = hip_ctx.prefetch_stats()
    assert st["queued"] == 3 and st["hits"] == 1 and len(st["unused"]) == 2
    hip_ctx.prefetch_begin()
    assert torch.equal(a, hip_ctx.frame_rms(xd, 1102, 441)) and hip_ctx.prefetch_stats()["hits"] == 0


@pytest.mark.parametrize("hop", [441, 2205, 512])
def test_stft_flatness_and_mel(hip_ctx, song, hop):
    flat, mel = hip_ctx.stft2048_features(hip_ctx.to_device(song), hop, want_flat=True, want_mel=True)
    ref_flat = L.spectral_flatness(song, hop_length=hop)[0]
    ref_mel = L.melspectrogram(song, SR, hop_length=hop, fmax=0.5 * SR).T
    np.testing.assert_allclose(flat.cpu().numpy(), ref_flat, rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(mel.cpu().numpy(), ref_mel, rtol=1e-4, atol=1e-12)


def test_stft_flatness_tonal_and_silent(hip_ctx, gated):
    flat, _ = hip_ctx.stft2048_features(hip_ctx.to_device(gated), 441)
    ref = L.spectral_flatness(gated, hop_length=441)[0]
    # tonal frames have flatness ~1e-8: the float64 FFT keeps the leakage floor like librosa's does
    np.testing.assert_allclose(flat.cpu().numpy(), ref, rtol=1e-4, atol=0)


@pytest.mark.parametrize("hop,agg", [(2205, "mean"), (512, "median"), (512, "mean")])
def test_onset_strength(hip_ctx, song, hop, agg):
    _, mel = hip_ctx.stft2048_features(hip_ctx.to_device(song), hop, want_flat=False, want_mel=True)
    env = hip_ctx.onset_strength(mel, hop, agg).cpu().numpy()
    ref = L.onset_strength(song, sr=SR, hop_length=hop, aggregate=np.mean if agg == "mean" else np.median)
    assert env.shape == ref.shape
    np.testing.assert_allclose(env, ref, rtol=1e-4, atol=2e-5)


def test_onset_strength_groups_match_per_chunk_calls(hip_ctx, song):
    """Per-chunk librosa calls (features_cache.py:181-187) == one grouped launch with chunk-local padding."""
    hop = 2205
    plans = OC.chunk_plan(len(song) / SR)
    centers, los, his, gs, refs = [], [], [], [0], []
    for (cs, ce, es, ee) in OC.plan_sample_ranges(plans, SR, len(song)):
        nf = 1 + (ce - cs) // hop
        centers += [cs + k * hop for k in range(nf)]
        los += [cs] * nf; his += [ce] * nf
        gs.append(gs[-1] + nf)
        refs.append(L.onset_strength(song[cs:ce], sr=SR, hop_length=hop))
    x = hip_ctx.to_device(song)
    fc = hip_ctx.to_device(np.array(centers, dtype=np.int64))
    lo = hip_ctx.to_device(np.array(los, dtype=np.int64)); hi = hip_ctx.to_device(np.array(his, dtype=np.int64))
    flat, mel = hip_ctx.stft2048_features(x, hop, want_flat=True, want_mel=True, frame_center=fc, frame_lo=lo, frame_hi=hi)
    env = hip_ctx.onset_strength(mel, hop, "mean", group_start=gs).cpu().numpy()
    np.testing.assert_allclose(env, np.concatenate(refs), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("hop,win", [(512, 689), (2205, 160)])
def test_tempogram_reduce(hip_ctx, song, hop, win):
    env = L.onset_strength(song, sr=SR, hop_length=hop, aggregate=np.median if hop == 512 else np.mean).astype(np.float32)
    bpms = L.tempo_frequencies(win, hop_length=hop, sr=SR)
    with np.errstate(divide="ignore"):
        logprior = -0.5 * ((np.log2(bpms) - np.log2(120.0)) / 1.0) ** 2
    logprior[: int(np.argmax(bpms < 320.0))] = -np.inf
    mean, arg = hip_ctx.tempogram_reduce(hip_ctx.to_device(env), win, logprior)
    tg = L.tempogram(env, win)
    np.testing.assert_allclose(mean.cpu().numpy(), tg.mean(axis=1), rtol=1e-9, atol=1e-12)
    ref_arg = np.argmax(np.log1p(1e6 * tg) + logprior[:, None], axis=0)
    assert np.array_equal(arg.cpu().numpy(), ref_arg)


@pytest.mark.parametrize("win", [3528, 1102, 441, 7])
def test_moving_meansq_db_and_next_leq(hip_ctx, gated, win):
    x = gated[: 6 * SR]
    db = hip_ctx.moving_meansq_db(hip_ctx.to_device(x), win)
    ref = OR.moving_meansq_db(x, win)
    got = db.cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    # windows of exact zeros give bit-identical dB (so np.argmin ties resolve the same way)
    silent = ref == ref.min()
    assert np.array_equal(got[silent], ref[silent])
    floor = float(np.percentile(ref, 40))
    nq = hip_ctx.next_leq_scan(db, floor).cpu().numpy()
    assert np.array_equal(nq, OR.next_leq_scan(got, floor))


def test_window_argmin_zero_cross_and_slow_guard(hip_ctx, song, gated):
    rng = np.random.default_rng(0)
    for x in (song, gated):
        xd = hip_ctx.to_device(x)
        db = hip_ctx.moving_meansq_db(xd, 3528)
        dbh = db.cpu().numpy()
        idx = np.sort(rng.integers(1, len(x) - 1, 64))
        span = 19845
        arg, val = hip_ctx.window_argmin(db, idx, np.full(64, span))
        for q, i in enumerate(idx):
            e = min(len(x), i + span)
            assert arg[q] == i + int(np.argmin(dbh[i:e]))
            assert val[q, 0] == dbh[i] and val[q, 1] == dbh[arg[q]]
        pos = hip_ctx.zero_cross_nearest(xd, idx, 353)
        for q, i in enumerate(idx):
            t = i / SR
            ref_t = OR.zero_cross_snap(x, SR, t, 8.0, legacy_promotion=True)
            if int(round(t * SR)) != i:
                continue
            if np.isnan(pos[q]):
                assert ref_t == t
            else:
                assert float(pos[q]) / SR == ref_t
        garg, gval = hip_ctx.quiet_guard_slow(xd, idx, span, 3528)
        for q, i in enumerate(idx):
            seg = x[i: min(len(x), i + span)]
            if seg.size <= 3528:
                continue
            padded = np.pad(seg, (0, 3527), mode="edge")
            lvl = np.sqrt(np.convolve(padded * padded, np.ones(3528) / 3528.0, mode="valid") + 1e-12)
            rdb = 20.0 * np.log10(lvl + 1e-12)
            k = int(np.argmin(rdb))
            # a periodic signal (the 440 Hz bursts repeat every 2205 samples) has window sums that tie to
            # the last ulp; the summation order then picks among them.  Equal minima are equivalent for
            # refine.py:152 (only db[0] - db[argmin] and db[argmin] enter the decision).
            assert garg[q] == k or abs(rdb[garg[q]] - rdb[k]) < 1e-9, (q, garg[q], k)
            np.testing.assert_allclose(gval[q], [rdb[0], rdb[k]], rtol=0, atol=1e-9)


def test_pause_cut_points(hip_ctx):
    voc = signals.vocal_like(20.0, seed=9)
    quiet = signals.c1_sine_silence(20.0, seed=9)
    rng = np.random.default_rng(1)
    for x in (voc, quiet):
        a = np.sort(rng.integers(0, len(x) - 50000, 24))
        b = a + rng.integers(300, 40000, 24)
        b[0] = a[0] + 600          # shorter than the 1102-sample kernel: numpy swaps the operands
        cut, aux = hip_ctx.pause_cut_points(hip_ctx.to_device(x), a, b, 1102, 5292)
        for q in range(len(a)):
            seg = x[a[q]:b[q]]
            c = a[q] + int(np.argmin(OD._local_rms(seg, 1102)))
            g_end = min(len(x), c + 5292)
            c = min(g_end - 1, c + int(np.argmin(OD._local_rms(x[c:g_end], 1102))))
            assert cut[q] == c, (q, cut[q], c)
            assert aux[q, 0] == int(np.sum(seg == 0)) and aux[q, 1] == int(x[c] != 0)


def test_mdx_stft_istft_assemble(hip_ctx, song):
    x = song[: int(12.3 * SR)]
    plans = OC.chunk_plan(len(x) / SR)
    ranges = OC.plan_sample_ranges(plans, SR, len(x))
    cs_l, cl_l, wi_l, base = [], [], [], []
    ref_specs = []
    for (cs, ce, es, ee) in ranges:
        batch, stereo, orig = OC.mdx_windows(x[cs:ce])
        base.append(len(cs_l))
        for k in range(batch.shape[0]):
            cs_l.append(cs); cl_l.append(ce - cs); wi_l.append(k)
        ref_specs.append(OS.mdx_stft(batch))
    ref_spec = torch.cat(ref_specs)                              # [items, 4, F, T]
    xd = hip_ctx.to_device(x)
    spec = hip_ctx.mdx_stft(xd, hip_ctx.to_device(np.array(cs_l, np.int64)), hip_ctx.to_device(np.array(cl_l, np.int64)),
                            hip_ctx.to_device(np.array(wi_l, np.int32)))
    got = spec.permute(0, 1, 3, 2).cpu()                         # -> [items, 4, F, T]
    assert _rel(got.numpy(), ref_spec.numpy()) < 2e-6
    # iSTFT of a *different* (non-STFT-consistent) spectrogram, like the U-Net output
    g = torch.Generator().manual_seed(0)
    fake = ref_spec * (1.0 + 0.3 * torch.randn(ref_spec.shape, generator=g))
    ref_wave = OS.mdx_istft(fake)
    wave = hip_ctx.mdx_istft(fake.permute(0, 1, 3, 2).contiguous().to(hip_ctx.device))
    assert _rel(wave.cpu().numpy(), ref_wave) < 2e-6
    # assemble + OLA
    outs = []
    k0 = 0
    for (cs, ce, es, ee) in ranges:
        batch, stereo, orig = OC.mdx_windows(x[cs:ce])
        nb = batch.shape[0]
        outs.append(OC.mdx_assemble(ref_wave[k0:k0 + nb], stereo, orig))
        k0 += nb
    ref_v, ref_i = OC.overlap_add(len(x), ranges, outs)
    v, i = hip_ctx.mdx_assemble_ola(
        xd, hip_ctx.to_device(ref_wave), hip_ctx.to_device(np.array([r[0] for r in ranges], np.int64)),
        hip_ctx.to_device(np.array([r[1] - r[0] for r in ranges], np.int64)),
        hip_ctx.to_device(np.array([r[2] for r in ranges], np.int64)),
        hip_ctx.to_device(np.array([r[3] for r in ranges], np.int64)), hip_ctx.to_device(np.array(base, np.int32)))
    assert np.array_equal(v.cpu().numpy(), ref_v)
    assert np.array_equal(i.cpu().numpy(), ref_i)


def test_yin_f0_autocorrelation_kernel(hip_ctx):
    """`ac_yin_f0` (librosa.yin = the deterministic stage of the pyin call at pure_vocal_pause_detector.py:422-428)."""
    t = np.arange(SR * 3) / SR
    rng = np.random.default_rng(3)
    f_inst = 180.0 * (1.0 + 0.3 * np.sin(2 * np.pi * 0.4 * t))
    phase = 2 * np.pi * np.cumsum(f_inst) / SR
    voiced = sum((0.4 / h) * np.sin(h * phase) for h in range(1, 6))
    x = (voiced * (np.sin(2 * np.pi * 0.5 * t) > -0.5) + 0.002 * rng.standard_normal(len(t))).astype(np.float32)
    fmin, fmax = 65.40639132514966, 2093.004522404789          # librosa.note_to_hz('C2'), ('C7')
    f0, cmnd = hip_ctx.yin_f0(hip_ctx.to_device(x), SR, fmin, fmax, frame_length=2048, hop=441, want_cmnd=True)
    ref_cm, mn, mx = L.cmnd_frames(x, SR, fmin, fmax, 2048, 441)
    ref_f0 = L.yin(x, fmin, fmax, sr=SR, frame_length=2048, hop_length=441)
    assert f0.shape == ref_f0.shape and cmnd.shape == ref_cm.T.shape
    np.testing.assert_allclose(cmnd.cpu().numpy(), ref_cm.T, rtol=1e-6, atol=1e-7)   # float64 series (pinned numpy < 2 semantics); FFT vs direct autocorrelation
    strong = ref_cm.min(axis=0) < 0.05                     # clearly periodic frames: the trough is well defined
    assert strong.sum() > 100
    np.testing.assert_allclose(f0[strong], ref_f0[strong], rtol=1e-4)
    assert np.mean(np.abs(f0 - ref_f0) / ref_f0 < 1e-3) > 0.97      # elsewhere the pick can hop between near-equal troughs
    sil, _ = hip_ctx.yin_f0(hip_ctx.to_device(np.zeros(8192, np.float32)), SR, fmin, fmax, 2048, 441)
    assert np.all(sil == SR / 21.0)                        # silent frames: cmnd == 0 everywhere -> first lag, as librosa


def test_kernels_stay_exact_beside_the_conv_on_another_stream(hip_ctx):
    """Regression guard for round 2's "two processes on one GPU" corruption, root-caused in round 3
    (profiles/r03_gpu_sharing_rootcause.log): built WITH hipcc's packed-float32 instructions the FFT kernels computed wrong frames
    whenever workgroups of the MFMA conv kernels shared their compute unit - from another stream of the same process just as from
    another process (1637 of 1640 launches).  The library is built without those instructions (csrc/Makefile NOPK); here the STFT,
    the iSTFT, the 1x1 conv and the Silero front end (the kernels that carried the most packed ops) run on one stream while the
    3x3 conv loops on another, and every launch must equal its solo result bit for bit."""
    import time
    from audio_cut_amd._native import _ptr, _stream
    from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96
    dev = hip_ctx.device
    g = torch.Generator().manual_seed(5)
    x48 = torch.randn(4, 48, 256, 3072, generator=g).to(dev)
    w = torch.randn(48, 48, 3, 3, generator=g) / 20
    pk, un = pack_conv3x3_w96(w.numpy(), 48)
    wp = torch.from_numpy(pk.view(np.int16)).to(dev); b48 = torch.zeros(48, device=dev); y48 = torch.empty_like(x48)
    conv = lambda: hip_ctx.lib.ac_conv3x3_f16x3_s8(hip_ctx._h, _ptr(x48), _ptr(wp), _ptr(b48), _ptr(y48), 4, 48, 48, 256, 3072, float(un), 1,
                                                   None, None, _stream())
    trk = (torch.randn(441000 * 6, generator=g) * 0.3).to(dev)
    cs = hip_ctx.to_device(np.repeat(np.arange(4) * 330750, 2).astype(np.int64)); cl = hip_ctx.to_device(np.full(8, 441000, np.int64))
    wi = hip_ctx.to_device(np.tile([0, 1], 4).astype(np.int32))
    spec = hip_ctx.mdx_stft(trk, cs, cl, wi)
    w11 = (torch.randn(4, 48, generator=g) / 7).to(dev); b11 = torch.zeros(4, device=dev)
    victims = {"mdx_stft": lambda: hip_ctx.mdx_stft(trk, cs, cl, wi), "mdx_istft": lambda: hip_ctx.mdx_istft(spec),
               "conv1x1_small": lambda: hip_ctx.conv1x1_small(x48, w11, b11, relu=False)}
    refs = {k: f().clone() for k, f in victims.items()}
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    wrong = {k: 0 for k in victims}
    count = {k: 0 for k in victims}
    t_end = time.time() + 2.0
    while time.time() < t_end:
        with torch.cuda.stream(sa):
            for _ in range(6):
                assert conv() == 0
        with torch.cuda.stream(sb):
            res = {k: [(f() != refs[k]).sum() for _ in range(2)] for k, f in victims.items()}
        torch.cuda.synchronize()
        for k, rs in res.items():
            count[k] += len(rs); wrong[k] += sum(1 for r in rs if int(r))
    print("launches beside the conv (wrong / total):", {k: f"{wrong[k]} / {count[k]}" for k in victims})
    assert all(v == 0 for v in wrong.values()), wrong
    assert min(count.values()) >= 20


@pytest.mark.gpu
def test_window_mean_squares_is_the_per_window_mean_square_bit_for_bit(hip_ctx):
    """ABI 6 (`ac_window_sum_squares`): the VPBD beat candidates' vocal-risk windows (`beat_candidates.py:97-109`) in one launch - every value
    bit-identical to `mean_square` of the slice (the per-candidate launch + download it replaces), empty and clipped windows included; and
    `_VocalRisk.many` == the per-candidate calls on a track."""
    from audio_cut_amd.cutting.beat_candidates import _VocalRisk
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(3 * SR) * np.exp(rng.uniform(-9, 0, 3 * SR))).astype(np.float32)
    dev = hip_ctx.to_device(x)
    n = len(x)
    starts = np.concatenate([rng.integers(0, n - 8191, 200), [0, n - 5, n - 8191, 17, n]])
    lens = np.concatenate([rng.integers(1, 8192, 200), [8191, 5, 8191, 0, 0]])
    ends = np.minimum(starts + lens, n)
    got = hip_ctx.window_mean_squares(dev, starts, ends)
    want = np.array([hip_ctx.mean_square(dev[a:b]) if b > a else 0.0 for a, b in zip(starts, ends)])
    assert got.dtype == np.float64 and np.array_equal(got, want)
    assert np.allclose(got, [np.mean(np.square(x[a:b], dtype=np.float64)) if b > a else 0.0 for a, b in zip(starts, ends)], rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        hip_ctx.window_mean_squares(dev, [0], [8192])
    risk = _VocalRisk(x, SR, 80.0, hip=hip_ctx, vocal_dev=dev)
    times = list(rng.uniform(-0.05, 3.05, 60)) + [0.0, 3.0]
    assert risk.many(times) == [risk(t) for t in times]
    host = _VocalRisk(x, SR, 80.0)
    assert host.many(times) == [host(t) for t in times]
