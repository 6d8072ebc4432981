"""Two processes on ONE GPU: which kernel, of which process, corrupts the iSTFT frames?  (one run, not a loop.)

    python tools/sharing_probe_rootcause.py victim    &
    python tools/sharing_probe_rootcause.py aggressor &

Round A: the VICTIM runs nothing but the iSTFT (k_mdx_istft_frames + k_mdx_istft_ola) on a fixed spectrogram, into a frame scratch
and an output that are re-poisoned with NaN before every call, and compares both with a reference taken while it was alone;
meanwhile the AGGRESSOR walks a wall-clock schedule of phases, ONE kernel type per phase.  A victim failure is attributed to the
phase it happened in: if the FFT-only victim fails only while the other process runs kernel X, the corruption crosses a process
boundary (it is not a stream-ordering bug of the product) and X is named; NaN in a failed output = lost stores / foreign
overwrite with garbage is told from "finite but wrong" = wrong arithmetic inputs (LDS / register state).
Round B: the victim runs the in-process chain U-Net -> iSTFT with every hand-off tensor kept (net output, frame scratch, wave)
and names the first tensor that differs from its solo reference, while the aggressor is idle / runs the U-Net / runs FFTs."""
import os, sys, time, json, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
from audio_cut_amd._native import _ptr, _stream, _check
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96, pack_linear, pack_tdf_small
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights

role = sys.argv[1]
SYNC = "/tmp/ac_share_probe"
os.makedirs(SYNC, exist_ok=True)
PHASE_S = 5.0
PHASES_A = ["idle", "conv_first", "conv_s8", "conv_w96", "tdf_l1", "tdf_l2", "tdf_small", "down2x", "up2x", "conv1x1", "torch_stream", "net", "fft"]
PHASES_B = ["idle", "net", "fft", "torch_stream"]
PHASE_B_S = 8.0

hip = _native.Context("cuda:0"); dev = hip.device
backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip, max_items_per_forward=32); backend.load_model()
net = backend.net
g = torch.Generator().manual_seed(0 if role == "victim" else 1)
track = (torch.randn(441000 * 20, generator=g) * 0.3).to(dev)
NI = 32
cs = hip.to_device(np.repeat(np.arange(16) * 330750, 2).astype(np.int64)); cl = hip.to_device(np.full(NI, 441000, np.int64)); wi = hip.to_device(np.tile([0, 1], 16).astype(np.int32))
spec_fixed = hip.mdx_stft(track, cs, cl, wi); torch.cuda.synchronize()


def istft_into(spec, scratch, wave):
    _check(hip.lib.ac_mdx_istft(hip._h, _ptr(spec), spec.shape[0], _ptr(wave), _ptr(scratch), _stream()))


def rendezvous():
    open(f"{SYNC}/ready_{role}", "w").write("1")
    other = "aggressor" if role == "victim" else "victim"
    while not os.path.exists(f"{SYNC}/ready_{other}"):
        time.sleep(0.05)
    if role == "victim":
        t0 = time.time() + 3.0
        open(f"{SYNC}/t0.tmp", "w").write(repr(t0)); os.replace(f"{SYNC}/t0.tmp", f"{SYNC}/t0")
    else:
        while not os.path.exists(f"{SYNC}/t0"):
            time.sleep(0.05)
        t0 = float(open(f"{SYNC}/t0").read())
    return t0


def phase_at(t0, now):
    """-> (round, phase name) or None when the schedule is over."""
    dt = now - t0
    if dt < 0:
        return ("wait", "idle")
    ia = int(dt // PHASE_S)
    if ia < len(PHASES_A):
        return ("A", PHASES_A[ia])
    dt -= len(PHASES_A) * PHASE_S
    ib = int(dt // PHASE_B_S)
    if ib < len(PHASES_B):
        return ("B", PHASES_B[ib])
    return None


if role == "aggressor":
    B = 8
    gg = torch.Generator().manual_seed(5)
    work = {}
    x48 = torch.randn(B, 48, 256, 3072, generator=gg).to(dev)
    x96 = torch.randn(B, 96, 128, 1536, generator=gg).to(dev)
    w = torch.randn(48, 48, 3, 3, generator=gg) / 20; pk, un_s8 = pack_conv3x3_w96(w.numpy(), 48); wp_s8 = torch.from_numpy(pk.view(np.int16)).to(dev); b48 = torch.zeros(48, device=dev)
    work["conv_s8"] = lambda: hip.conv3x3_f16x3_s8(x48, wp_s8, b48, 48, un_s8, relu=True)
    w = torch.randn(96, 96, 3, 3, generator=gg) / 30; pk, un_w96 = pack_conv3x3_w96(w.numpy(), 96); wp_w96 = torch.from_numpy(pk.view(np.int16)).to(dev); b96 = torch.zeros(96, device=dev)
    work["conv_w96"] = lambda: hip.conv3x3_f16x3_w96(x96, wp_w96, b96, 96, un_w96, relu=True)
    sp8 = spec_fixed[:B].contiguous(); am8 = sp8.abs().amax(dim=(1, 3)).contiguous()
    e0 = net.enc[0]
    work["conv_first"] = lambda: hip.conv3x3_f16x3_first(sp8, net.first_w, net.first_b, e0.cws0, e0.cb0, e0.cw0.shape[0], e0._w_unscale[0], relu=True,
                                                         spec_amax=am8, amax_gain=net._first_gain, amax_offs=net._first_offs)
    sc = torch.ones(48, device=dev); sh = torch.zeros(48, device=dev)
    wt = torch.randn(384, 3072, generator=gg) / 55; pk, un_l1 = pack_linear(wt.numpy()); wp_l1 = torch.from_numpy(pk.view(np.int16)).to(dev)
    work["tdf_l1"] = lambda: hip.tdf_linear_f16x3(x48, wp_l1, 384, sc, sh, un_l1)
    h384 = torch.randn(B, 48, 256, 384, generator=gg).to(dev)
    wt = torch.randn(3072, 384, generator=gg) / 20; pk, un_l2 = pack_linear(wt.numpy()); wp_l2 = torch.from_numpy(pk.view(np.int16)).to(dev)
    work["tdf_l2"] = lambda: hip.tdf_linear_f16x3(h384, wp_l2, 3072, sc, sh, un_l2, resid=x48)
    xs = torch.randn(B, 192, 32, 384, generator=gg).to(dev); w1 = torch.randn(48, 384, generator=gg) / 20; w2 = torch.randn(384, 48, generator=gg) / 7
    p1, p2 = pack_tdf_small(w1.numpy(), w2.numpy()); p1 = torch.from_numpy(p1).to(dev); p2 = torch.from_numpy(p2).to(dev); o192 = torch.ones(192, device=dev); z192 = torch.zeros(192, device=dev)
    work["tdf_small"] = lambda: hip.tdf_small_fused(xs, p1, p2, 48, o192, z192, o192, z192)
    wd = torch.randn(96, 48, 2, 2, generator=gg) / 14; pk, un_d = pack_linear(wd.numpy().reshape(96, -1), bn=96); wpd = torch.from_numpy(pk.view(np.int16)).to(dev)
    work["down2x"] = lambda: hip.down2x_f16x3(x48, wpd, b96, 96, un_d)
    wu = torch.randn(96, 48, 2, 2, generator=gg) / 10; pk, un_u = pack_linear(wu.numpy().transpose(1, 2, 3, 0).reshape(-1, 96), bn=96); wpu = torch.from_numpy(pk.view(np.int16)).to(dev)
    work["up2x"] = lambda: hip.up2x_f16x3(x96, wpu, b48, 48, un_u)
    work["conv1x1"] = lambda: hip.conv1x1_small(x48, net.final_w, net.final_b, relu=False)
    big = torch.randn(1 << 27, device=dev)
    work["torch_stream"] = lambda: big * 1.0001
    work["net"] = lambda: net.forward_tf(spec_fixed)
    scr = torch.empty((NI * 2 * 256 * 6144,), dtype=torch.float32, device=dev); wv = torch.empty((NI, 2, 261120), dtype=torch.float32, device=dev)

    def fft_work():
        hip.mdx_stft(track, cs, cl, wi)
        istft_into(spec_fixed, scr, wv)
    work["fft"] = fft_work
    for k, f in work.items():          # warm every kernel once before the schedule starts
        f(); torch.cuda.synchronize()
    t0 = rendezvous()
    counts = {}
    while True:
        ph = phase_at(t0, time.time())
        if ph is None:
            break
        if ph[1] == "idle":
            time.sleep(0.01); continue
        f = work[ph[1]]
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        counts[ph] = counts.get(ph, 0) + 4
    print("aggressor launches per phase:", {f"{r}:{p}": n for (r, p), n in counts.items()}, flush=True)
    sys.exit(0)

# ---------------------------------------------------------------------------------------------------------------- victim
scratch = torch.empty((NI * 2 * 256 * 6144,), dtype=torch.float32, device=dev)
wave = torch.empty((NI, 2, 261120), dtype=torch.float32, device=dev)
istft_into(spec_fixed, scratch, wave); torch.cuda.synchronize()
ref_frames = scratch.clone(); ref_wave = wave.clone()
y_ref = net.forward_tf(spec_fixed); torch.cuda.synchronize()
scr2 = torch.empty_like(scratch); wv2 = torch.empty_like(wave)
istft_into(y_ref, scr2, wv2); torch.cuda.synchronize()
ref2_frames = scr2.clone(); ref2_wave = wv2.clone()
# solo sanity: the same calls again, alone, must reproduce
scratch.fill_(float("nan")); wave.fill_(float("nan")); istft_into(spec_fixed, scratch, wave); torch.cuda.synchronize()
assert torch.equal(scratch, ref_frames) and torch.equal(wave, ref_wave), "not reproducible even alone"


def describe(bad_frames: torch.Tensor, got: torch.Tensor, ref: torch.Tensor) -> str:
    """bad_frames: flat indices into the frame scratch that differ."""
    idx = bad_frames.cpu().numpy()
    gv = got[bad_frames].cpu().numpy(); rv = ref[bad_frames].cpu().numpy()
    fr = idx // 6144; off = idx % 6144
    frames = np.unique(fr)
    out = [f"{len(idx)} scratch samples differ in {len(frames)} frame(s) {frames[:6].tolist()}; NaN (poison still there): {int(np.isnan(gv).sum())}; "
           f"max |diff| {np.nanmax(np.abs(gv - rv)):.3e} (ref peak {float(ref.abs().max()):.2f})"]
    f0 = frames[0]; o = np.sort(off[fr == f0]); m = o // 2
    runs = np.split(o, np.flatnonzero(np.diff(o) > 1) + 1)
    out.append(f"   frame {int(f0)}: {len(o)} samples, {len(runs)} runs, first runs (start, len) {[(int(r[0]), len(r)) for r in runs[:6]]}, "
               f"m mod 4 classes {np.unique(m % 4).tolist()}, m mod 256 classes {np.unique(m % 256).tolist()[:20]}")
    return "\n".join(out)


t0 = rendezvous()
stats = {}
reported = {}
while True:
    ph = phase_at(t0, time.time())
    if ph is None:
        break
    if ph[0] == "wait":
        time.sleep(0.01); continue
    if ph[0] == "A":
        # four asynchronous poisoned iSTFTs, then one look at the counters
        bad = []
        for _ in range(4):
            scratch.fill_(float("nan")); wave.fill_(float("nan"))
            istft_into(spec_fixed, scratch, wave)
            nf = (scratch != ref_frames).sum(); nw = (wave != ref_wave).sum()      # NaN != x is True: poison counts as different
            bad.append((nf, nw, None))
        # the buffers still hold the last iteration: a closer look at it if it failed
        torch.cuda.synchronize()
        s = stats.setdefault(ph, [0, 0, 0])
        for nf, nw, _ in bad:
            s[0] += 1
            if int(nf) or int(nw):
                s[1] += 1
        if (int(bad[-1][0]) or int(bad[-1][1])) and reported.get(ph, 0) < 3:
            reported[ph] = reported.get(ph, 0) + 1
            d = (scratch != ref_frames).nonzero().flatten()
            print(f"[A:{ph[1]}] t={time.time() - t0:.1f}s frames bad {int(bad[-1][0])}, wave bad {int(bad[-1][1])}", flush=True)
            if d.numel():
                print(describe(d, scratch, ref_frames), flush=True)
    else:
        # in-process chain, every hand-off tensor kept: net output -> frames -> wave, two chains back to back
        ys, scs, wvs = [], [], []
        for _ in range(2):
            y = net.forward_tf(spec_fixed)
            sc_ = torch.full_like(scratch, float("nan")); wv_ = torch.full_like(wave, float("nan"))
            istft_into(y, sc_, wv_)
            ys.append(y); scs.append(sc_); wvs.append(wv_)
        torch.cuda.synchronize()
        s = stats.setdefault(ph, [0, 0, 0])
        for y, sc_, wv_ in zip(ys, scs, wvs):
            s[0] += 1
            ny = int((y != y_ref).sum()); nf = int((sc_ != ref2_frames).sum()); nw = int((wv_ != ref2_wave).sum())
            if ny or nf or nw:
                s[1] += 1
                first = "net output" if ny else ("frame scratch" if nf else "wave (OLA)")
                if reported.get(ph, 0) < 3:
                    reported[ph] = reported.get(ph, 0) + 1
                    print(f"[B:{ph[1]}] t={time.time() - t0:.1f}s first differing tensor: {first}; net output {ny}, frames {nf}, wave {nw} elements differ", flush=True)
                    if nf and not ny:
                        print(describe((sc_ != ref2_frames).nonzero().flatten(), sc_, ref2_frames), flush=True)
                    if ny:
                        d = (y != y_ref)
                        it = d.flatten(1).any(1).nonzero().flatten().tolist()
                        per_t = d.any(dim=3)            # [item, ch, T]
                        print(f"   net output: items {it[:8]}, channels {d.any(dim=3).any(dim=2).any(dim=0).nonzero().flatten().tolist()}, "
                              f"time rows touched {int(per_t.any(1).sum())}, max |diff| {float((y - y_ref).abs().max()):.3e} (peak {float(y_ref.abs().max()):.2f})", flush=True)
        del ys, scs, wvs
print("victim: iterations / failures per phase:")
for (r, p), (n, b, _) in stats.items():
    print(f"   {r}:{p:13s} {b:4d} / {n:5d}")
print("victim done", flush=True)
