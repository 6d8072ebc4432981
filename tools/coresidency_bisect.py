"""ONE process, two streams: which ingredient of the 3x3 conv kernel, and which ingredient of the iSTFT kernel, makes the iSTFT
compute wrong frames when workgroups of the two share compute units?  (tools/sharing_probe_victims.py part 0 showed that no second
process is needed: 1637 of 1640 iSTFT launches beside a conv loop on another stream of the SAME process were wrong.)

Aggressors: the product conv (48- and 96-channel workgroups) and ablated builds of it (tools/probes/build_probes.sh, W9_PROBE bits:
4 no MFMA, 8 no activation staging, 0x10 no shared tap-8 step, 0x20 no epilogue, 0x40 no K loop, 0x80 no activation loads), the TDF
GEMM as the known-harmless control.  Victims: the product iSTFT, builds of it without twiddle factors / without SLP-packed float
math / with L1-bypassing loads, and the FFT's LDS traffic as pure data movement (canary.hip).  Every cell = wrong launches / launches
in ~1.2 s, each launch checked bit for bit against the victim's own solo result."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_cut_amd import _native
from audio_cut_amd._native import _ptr, _stream
from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96, pack_linear

CELL_S = float(os.environ.get("CELL_S", "1.2"))
hip = _native.Context("cuda:0"); dev = hip.device
PB = f"{ROOT}/tools/probes/build"


def probe_lib(name):
    lib = C.CDLL(f"{PB}/lib{name}.so")
    for fn in ("ac_conv3x3_f16x3_s8", "ac_conv3x3_f16x3_w96", "ac_mdx_istft"):
        if hasattr(lib, fn):
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = _native.SIGNATURES[fn]
    return lib


# ---- aggressors ------------------------------------------------------------------------------------------------------------
gg = torch.Generator().manual_seed(5)
B = 8
x48 = torch.randn(B, 48, 256, 3072, generator=gg).to(dev); x96 = torch.randn(B, 96, 128, 1536, generator=gg).to(dev)
w = torch.randn(48, 48, 3, 3, generator=gg) / 20; pk, un_s8 = pack_conv3x3_w96(w.numpy(), 48); wp_s8 = torch.from_numpy(pk.view(np.int16)).to(dev); b48 = torch.zeros(48, device=dev)
w = torch.randn(96, 96, 3, 3, generator=gg) / 30; pk, un_w96 = pack_conv3x3_w96(w.numpy(), 96); wp_w96 = torch.from_numpy(pk.view(np.int16)).to(dev); b96 = torch.zeros(96, device=dev)
y48 = torch.empty_like(x48); y96 = torch.empty_like(x96)


def conv_call(lib, wide=False):
    if wide:
        return lambda: lib.ac_conv3x3_f16x3_w96(hip._h, _ptr(x96), _ptr(wp_w96), _ptr(b96), _ptr(y96), B, 96, 96, 128, 1536, float(un_w96), 1, None, None, _stream())
    return lambda: lib.ac_conv3x3_f16x3_s8(hip._h, _ptr(x48), _ptr(wp_s8), _ptr(b48), _ptr(y48), B, 48, 48, 256, 3072, float(un_s8), 1, None, None, _stream())


wt = torch.randn(384, 3072, generator=gg) / 55; pk, un_l1 = pack_linear(wt.numpy()); wp_l1 = torch.from_numpy(pk.view(np.int16)).to(dev)
sc = torch.ones(48, device=dev); sh = torch.zeros(48, device=dev)
aggressors = {"none": None, "tdf_l1 (control)": lambda: hip.tdf_linear_f16x3(x48, wp_l1, 384, sc, sh, un_l1),
              "conv48 product": conv_call(hip.lib), "conv96 product": conv_call(hip.lib, True)}
for bits, what in (("0x04", "no MFMA"), ("0x08", "no activation staging"), ("0x10", "no shared tap-8 step"), ("0x20", "no epilogue"),
                   ("0x40", "no K loop (prologue + epilogue)"), ("0x80", "no activation loads"), ("0x60", "prologue only"),
                   ("0x0c", "no MFMA, no staging"), ("0x8c", "no MFMA, no staging, no loads")):
    aggressors[f"conv48 {what}"] = conv_call(probe_lib(f"conv_p{bits}"))

# ---- victims ---------------------------------------------------------------------------------------------------------------
NI = 16
g = torch.Generator().manual_seed(0)
spec = (torch.randn(NI, 4, 256, 3072, generator=g) * 0.3).to(dev)
scratch = torch.empty((NI * 2 * 256 * 6144,), dtype=torch.float32, device=dev); wave = torch.empty((NI, 2, 261120), dtype=torch.float32, device=dev)
can = C.CDLL(f"{PB}/libcanary.so")
can.canary_launch.restype = C.c_int; can.canary_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
errors = torch.zeros(4, dtype=torch.int32, device=dev)
table = torch.empty(3072 * 2, dtype=torch.int32, device=dev)
assert can.canary_launch(4, None, table.data_ptr(), 0, 0, _stream(), None) == 0
# where does word y of the shuffled array come from (host simulation of fft3072_f32's index moves)
a = np.arange(3072); Ns = 1
for _ in range(5):
    b = np.empty_like(a)
    j = np.arange(768); k = j & (Ns - 1); base = ((j - k) << 2) + k
    for i in range(4):
        b[base + i * Ns] = a[j + i * 768]
    a = b; Ns <<= 2
src_of = hip.to_device(a.astype(np.int32))


def istft_with(lib):
    def run():
        scratch.fill_(float("nan")); wave.fill_(float("nan"))
        assert lib.ac_mdx_istft(hip._h, _ptr(spec), NI, _ptr(wave), _ptr(scratch), _stream()) == 0
        return wave
    return run


def canary(which, rounds):
    def run():
        errors.zero_()
        assert can.canary_launch(which, errors.data_ptr(), table.data_ptr(), 8192, rounds, _stream(), src_of.data_ptr()) == 0
        return errors.clone()
    return run


victims = {}
for name, fn in (("iSTFT product", istft_with(hip.lib)), ("iSTFT without twiddles", istft_with(probe_lib("mdx_notw"))),
                 ("iSTFT -fno-slp-vectorize", istft_with(probe_lib("mdx_noslp"))), ("iSTFT nt loads", istft_with(probe_lib("mdx_nt")))):
    ref = fn().clone(); torch.cuda.synchronize()
    assert torch.equal(fn(), ref), name
    victims[name] = (lambda fn=fn, ref=ref: (fn() != ref).sum())
victims["FFT LDS traffic only (8-byte words)"] = lambda f=canary(5, 2): f().sum()
victims["FFT LDS traffic + table loads"] = lambda f=canary(6, 2): f().sum()
victims["packed float32 math only (registers)"] = lambda f=canary(7, 64): f()[:3].sum()
for k, f in victims.items():
    v = int(f()); torch.cuda.synchronize(); assert v == 0, (k, v)
mf_src = (torch.rand(4096 * 8, device=dev) * 0.02 - 0.01).to(torch.float16); mf_sink = torch.zeros(4, dtype=torch.float32, device=dev)
aggressors["register-only MFMA loop (no LDS, no memory)"] = lambda: can.canary_launch(8, mf_sink.data_ptr(), mf_src.data_ptr(), 2048, 512, _stream(), None)
for k, f in aggressors.items():
    if f is not None:
        f(); torch.cuda.synchronize()

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def cell(aggr, victim):
    n = bad = 0
    t_end = time.time() + CELL_S
    while time.time() < t_end:
        if aggr is not None:
            with torch.cuda.stream(sa):
                for _ in range(6):
                    aggr()
        with torch.cuda.stream(sb):
            res = [victim() for _ in range(4)]
        torch.cuda.synchronize()
        n += 4; bad += sum(1 for r in res if int(r))
    return bad, n


only = os.environ.get("ONLY_AGGR", "").split(",") if os.environ.get("ONLY_AGGR") else None
print("== which ingredient of the conv kernel?  victim = the product iSTFT", flush=True)
for an, af in aggressors.items():
    if only and not any(o in an for o in only):
        continue
    bad, n = cell(af, victims["iSTFT product"])
    print(f"   {an:42s} {bad:5d} / {n:5d} iSTFT launches wrong", flush=True)
print("== which ingredient of the iSTFT?  aggressor = the product conv (48-channel workgroups)", flush=True)
for vn, vf in victims.items():
    bad, n = cell(aggressors["conv48 product"], vf)
    b0, n0 = cell(None, vf)
    print(f"   {vn:42s} {bad:5d} / {n:5d} wrong beside the conv;  {b0} / {n0} alone", flush=True)
# the packed operand forms of k_mdx_istft_frames, one counter each
err8 = torch.zeros(8, dtype=torch.int32, device=dev)
forms = ["pk_mul op_sel_hi:[1,0]", "pk_mul op_sel:[1,1] op_sel_hi:[0,1]", "pk_add neg_lo/neg_hi:[0,1]", "pk_add op_sel:[0,1] op_sel_hi:[1,0]",
         "pk_add v, 0 neg:[1,1]", "pk_mul v, s[n:n+1]", "pk_mov op_sel:[1,0]", "pk_mul v, -0.5 op_sel_hi:[1,0]"]
for an in ("none", "conv48 product"):
    err8.zero_(); n = 0
    t_end = time.time() + 2 * CELL_S
    while time.time() < t_end:
        if aggressors[an] is not None:
            with torch.cuda.stream(sa):
                for _ in range(6):
                    aggressors[an]()
        with torch.cuda.stream(sb):
            for _ in range(4):
                assert can.canary_launch(9, err8.data_ptr(), None, 8192, 32, _stream(), None) == 0
        torch.cuda.synchronize(); n += 4
    print(f"== packed operand forms beside '{an}' ({n} launches x 8192 workgroups x 256 lanes x 32 iterations):", flush=True)
    for f, e in zip(forms, err8.cpu().tolist()):
        print(f"   {f:40s} {e} mismatches", flush=True)
# the STFT kernel (same FFT, forward direction) as a victim too
trk = (torch.randn(441000 * 10, generator=torch.Generator().manual_seed(3)) * 0.3).to(dev)
cs = hip.to_device(np.repeat(np.arange(8) * 330750, 2).astype(np.int64)); cl = hip.to_device(np.full(16, 441000, np.int64)); wi = hip.to_device(np.tile([0, 1], 8).astype(np.int32))
sref = hip.mdx_stft(trk, cs, cl, wi).clone(); torch.cuda.synchronize()
stft_v = lambda: (hip.mdx_stft(trk, cs, cl, wi) != sref).sum()
bad, n = cell(aggressors["conv48 product"], stft_v)
print(f"== STFT (product library) beside the conv: {bad} / {n} launches wrong", flush=True)
print("done", flush=True)
