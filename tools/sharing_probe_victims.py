"""Third two-process run: WHICH state of the victim does a co-resident 3x3-conv workgroup disturb, and is a second PROCESS needed?

Known from the first two runs (profiles/r03_gpu_sharing_rootcause.log): an iSTFT-only process computes wrong frames in ~100 % of its
launches while ANOTHER process runs any k_conv3x3_f16x3_w96 kernel - with or without LDS-DMA - and never while it runs the other
U-Net kernels; a TDF-GEMM victim is never disturbed.
Part 0 (one process, two streams): the conv loop on stream A, the checked iSTFT on stream B - the same co-residency without a
second process.  Part 1 (two processes): the victim walks
    istft        the product kernel                                  istft_nt   the same with every global load bypassing L1 (`nt`)
    vgpr         32 live registers per lane re-checked for ~30 us    lds        48 KiB of LDS written once, re-checked
    lds_rmw      48 KiB of LDS read-modify-written through barriers  l1         a 24 KiB global table re-read with the FFT's strides
each against an idle and a conv-running aggressor."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_cut_amd import _native
from audio_cut_amd._native import _ptr, _stream, _check
from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96

role = sys.argv[1]
SYNC = "/tmp/ac_share_probe3"
os.makedirs(SYNC, exist_ok=True)
PHASE_S = 3.0
VICTIMS = ["istft", "istft_nt", "vgpr", "lds", "lds_rmw", "l1"]
PHASES = [(v, a) for v in VICTIMS for a in ("idle", "s8_base")]

hip = _native.Context("cuda:0"); dev = hip.device


def wait_for(name):
    while not os.path.exists(f"{SYNC}/{name}"):
        time.sleep(0.05)


def phase_at(t0, now):
    dt = now - t0
    if dt < 0:
        return ("wait", "idle")
    i = int(dt // PHASE_S)
    return PHASES[i] if i < len(PHASES) else None


def conv_work(B=8):
    gg = torch.Generator().manual_seed(5)
    x48 = torch.randn(B, 48, 256, 3072, generator=gg).to(dev)
    w = torch.randn(48, 48, 3, 3, generator=gg) / 20; pk, un = pack_conv3x3_w96(w.numpy(), 48); wp = torch.from_numpy(pk.view(np.int16)).to(dev); b48 = torch.zeros(48, device=dev)
    return lambda: hip.conv3x3_f16x3_s8(x48, wp, b48, 48, un, relu=True)


if role == "aggressor":
    wait_for("part0_done")                      # nothing of this process touches the GPU while the victim runs its one-process control
    f = conv_work(); f(); torch.cuda.synchronize()
    open(f"{SYNC}/ready_aggressor", "w").write("1")
    wait_for("t0")
    t0 = float(open(f"{SYNC}/t0").read())
    n = 0
    while True:
        ph = phase_at(t0, time.time())
        if ph is None:
            break
        if ph[1] == "idle" or ph[0] == "wait":
            time.sleep(0.01); continue
        for _ in range(4):
            f()
        torch.cuda.synchronize(); n += 4
    print("aggressor conv launches:", n, flush=True)
    sys.exit(0)

# ------------------------------------------------------------------------------------------------------------------ victim
NI = 16
g = torch.Generator().manual_seed(0)
spec = (torch.randn(NI, 4, 256, 3072, generator=g) * 0.3).to(dev)
scratch = torch.empty((NI * 2 * 256 * 6144,), dtype=torch.float32, device=dev); wave = torch.empty((NI, 2, 261120), dtype=torch.float32, device=dev)
lib_nt = C.CDLL(f"{ROOT}/tools/probes/build/libaudiocut_hip_v3.so"); _native._declare(lib_nt)
h_nt = C.c_void_p(); assert lib_nt.ac_ctx_create(hip.index, C.byref(h_nt)) == 0
can = C.CDLL(f"{ROOT}/tools/probes/build/libcanary.so")
can.canary_launch.restype = C.c_int; can.canary_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
errors = torch.zeros(4, dtype=torch.int32, device=dev)
table = torch.empty(3072 * 2, dtype=torch.int32, device=dev)
assert can.canary_launch(4, None, table.data_ptr(), 0, 0, _stream()) == 0


def istft(lib=hip.lib, h=hip._h):
    scratch.fill_(float("nan")); wave.fill_(float("nan"))
    rc = lib.ac_mdx_istft(h, _ptr(spec), NI, _ptr(wave), _ptr(scratch), _stream())
    assert rc == 0
    return wave


ref = istft().clone(); torch.cuda.synchronize()
assert torch.equal(istft(), ref), "not reproducible even alone"
assert torch.equal(istft(lib_nt, h_nt), ref), "the nt build differs from the product kernel"


def canary(which):
    errors.zero_()
    assert can.canary_launch(which, errors.data_ptr(), table.data_ptr(), 8192, 24, _stream()) == 0
    return errors.clone()


fns = {"istft": lambda: (istft() != ref).sum(), "istft_nt": lambda: (istft(lib_nt, h_nt) != ref).sum(),
       "vgpr": lambda: canary(0).sum(), "lds": lambda: canary(1).sum(), "lds_rmw": lambda: canary(2).sum(), "l1": lambda: canary(3).sum()}
for k, f in fns.items():
    v = int(f()); torch.cuda.synchronize()
    assert v == 0, (k, v)
# how long does one canary launch take (the state must live about as long as an iSTFT workgroup's)
for k in ("istft", "vgpr", "lds", "lds_rmw", "l1"):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        fns[k]()
    torch.cuda.synchronize()
    print(f"solo: {k:8s} {(time.perf_counter() - t) / 5 * 1e3:.2f} ms per launch (8192 workgroups)", flush=True)

# ---- Part 0: ONE process, two streams --------------------------------------------------------------------------------------
conv = conv_work()
conv(); torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for label, aggr in (("conv_s8 on a second stream of the SAME process", conv), ("no second stream (control)", None)):
    n = bad = 0
    t_end = time.time() + 3.0
    while time.time() < t_end:
        if aggr is not None:
            with torch.cuda.stream(sa):
                for _ in range(6):
                    aggr()
        with torch.cuda.stream(sb):
            res = [fns["istft"]() for _ in range(4)]
        torch.cuda.synchronize()
        n += 4; bad += sum(1 for r in res if int(r))
    print(f"part 0, one process: iSTFT beside {label}: {bad} / {n} launches wrong", flush=True)
open(f"{SYNC}/part0_done", "w").write("1")
wait_for("ready_aggressor")
t0 = time.time() + 2.0
open(f"{SYNC}/t0.tmp", "w").write(repr(t0)); os.replace(f"{SYNC}/t0.tmp", f"{SYNC}/t0")
stats = {}
while True:
    ph = phase_at(t0, time.time())
    if ph is None:
        break
    if ph[0] == "wait":
        time.sleep(0.01); continue
    res = [fns[ph[0]]() for _ in range(4)]
    torch.cuda.synchronize()
    s = stats.setdefault(ph, [0, 0, 0])
    for r in res:
        s[0] += 1; s[1] += 1 if int(r) else 0; s[2] += int(r)
print("part 1, two processes:  victim : aggressor      failed / launches   (mismatching words)")
for (v, a), (n, b, w) in stats.items():
    print(f"   {v:9s} : {a:8s} {b:5d} / {n:5d}   ({w})")
print("victim done", flush=True)
