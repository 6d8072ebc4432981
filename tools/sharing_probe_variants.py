"""Second (and last) two-process run: WHAT in the 3x3 conv kernel disturbs another process's workgroups?

tools/sharing_probe_rootcause.py named the aggressor: an FFT-only victim process fails in ~99 % of its iterations while the
OTHER process runs any kernel of the k_conv3x3_f16x3_w96 family and never while it runs the TDF GEMMs, the 2x2 resampling,
k_tdf_small, the 1x1 conv or PyTorch streaming kernels.  Here the aggressor walks probe builds of that one kernel
(tools/probes/build/libaudiocut_hip_v{1,2}.so: ac_conv96.hip compiled with -DW9_VARIANT):
    base  the product kernel: LDS-DMA weight fragments land behind the 16,000-byte activation patch (128-byte aligned only)
    v1    weight buffers first: every 1 KiB LDS-DMA wave-instruction lands 1 KiB aligned
    v2    no LDS-DMA at all: the fragments go global -> registers -> ds_write
and the victim is, in turn, the iSTFT (plain LDS + barriers) and a TDF GEMM (LDS-DMA + MFMA) checked bit for bit against
their solo results.  Also prints the amdgpu / kfd module parameters that decide how two processes share the device.

    python tools/sharing_probe_variants.py victim & python tools/sharing_probe_variants.py aggressor"""
import ctypes as C, glob, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_cut_amd import _native
from audio_cut_amd._native import _ptr, _stream, _check
from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96, pack_linear

role = sys.argv[1]
SYNC = "/tmp/ac_share_probe2"
os.makedirs(SYNC, exist_ok=True)
PHASE_S = 4.0
AGG = ["idle", "s8_base", "s8_v1", "s8_v2", "w96_base", "w96_v1", "w96_v2", "s8_base_B2"]
PHASES = [("istft", a) for a in AGG] + [("tdf", a) for a in ("idle", "s8_base", "s8_v1", "s8_v2", "w96_base")]

hip = _native.Context("cuda:0"); dev = hip.device


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
    dt = now - t0
    if dt < 0:
        return ("wait", "idle")
    i = int(dt // PHASE_S)
    return PHASES[i] if i < len(PHASES) else None


if role == "aggressor":
    for f in sorted(glob.glob("/sys/module/amdgpu/parameters/*")):
        if os.path.basename(f) in ("cwsr_enable", "sched_policy", "hws_max_conc_proc", "max_num_of_queues_per_device", "noretry", "mes", "mes_kiq",
                                   "halt_if_hws_hang", "queue_preemption_timeout_ms", "hws_gws_support", "ignore_crat", "mcbp", "user_partt_mode"):
            try:
                print("amdgpu." + os.path.basename(f), "=", open(f).read().strip(), flush=True)
            except Exception as e:
                print("amdgpu." + os.path.basename(f), "unreadable", e, flush=True)
    for f in ("/sys/class/kfd/kfd/topology/nodes/1/properties",):
        try:
            keep = [l.strip() for l in open(f) if any(k in l for k in ("simd_count", "lds_size", "cu_per_simd", "num_xcc", "max_waves", "gfx_target", "num_cp_queues", "num_sdma"))]
            print("kfd node 1:", "; ".join(keep), flush=True)
        except Exception as e:
            print("kfd topology unreadable", e, flush=True)

    def variant_ctx(path):
        c = object.__new__(_native.Context)
        c.device, c.index = hip.device, hip.index
        c.lib = C.CDLL(path); _native._declare(c.lib)
        h = C.c_void_p()
        assert c.lib.ac_ctx_create(c.index, C.byref(h)) == 0
        c._h = h
        return c
    ctxs = {"base": hip, "v1": variant_ctx(f"{ROOT}/tools/probes/build/libaudiocut_hip_v1.so"), "v2": variant_ctx(f"{ROOT}/tools/probes/build/libaudiocut_hip_v2.so")}
    gg = torch.Generator().manual_seed(5)
    work = {}
    for B in (8, 2):
        x48 = torch.randn(B, 48, 256, 3072, generator=gg).to(dev); x96 = torch.randn(B, 96, 128, 1536, generator=gg).to(dev)
        w = torch.randn(48, 48, 3, 3, generator=gg) / 20; pk, un_s8 = pack_conv3x3_w96(w.numpy(), 48); wp_s8 = torch.from_numpy(pk.view(np.int16)).to(dev); b48 = torch.zeros(48, device=dev)
        w = torch.randn(96, 96, 3, 3, generator=gg) / 30; pk, un_w96 = pack_conv3x3_w96(w.numpy(), 96); wp_w96 = torch.from_numpy(pk.view(np.int16)).to(dev); b96 = torch.zeros(96, device=dev)
        for v, c in ctxs.items():
            sfx = "" if B == 8 else "_B2"
            work[f"s8_{v}{sfx}"] = (lambda c=c, x=x48, wp=wp_s8, b=b48, un=un_s8: c.conv3x3_f16x3_s8(x, wp, b, 48, un, relu=True))
            work[f"w96_{v}{sfx}"] = (lambda c=c, x=x96, wp=wp_w96, b=b96, un=un_w96: c.conv3x3_f16x3_w96(x, wp, b, 96, un, relu=True))
    ref = {}
    for k, f in work.items():            # the probe builds must still compute the product's result
        base = k.replace("_v1", "_base").replace("_v2", "_base")
        if base == k:
            ref[k] = f(); torch.cuda.synchronize()
    for k, f in work.items():
        base = k.replace("_v1", "_base").replace("_v2", "_base")
        if base != k:
            y = f(); torch.cuda.synchronize()
            print(f"probe build {k}: bit-identical to the product kernel: {bool(torch.equal(y, ref[base]))}", flush=True)
    del ref
    t0 = rendezvous()
    counts = {}
    while True:
        ph = phase_at(t0, time.time())
        if ph is None:
            break
        if ph[1] == "idle" or ph[0] == "wait":
            time.sleep(0.01); continue
        f = work[ph[1]]
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        counts[ph] = counts.get(ph, 0) + 4
    print("aggressor launches per phase:", {f"{r}:{p}": n for (r, p), n in counts.items()}, flush=True)
    sys.exit(0)

# ------------------------------------------------------------------------------------------------------------------ victim
NI = 16
g = torch.Generator().manual_seed(0)
spec = (torch.randn(NI, 4, 256, 3072, generator=g) * 0.3).to(dev)
scratch = torch.empty((NI * 2 * 256 * 6144,), dtype=torch.float32, device=dev); wave = torch.empty((NI, 2, 261120), dtype=torch.float32, device=dev)


def istft():
    scratch.fill_(float("nan")); wave.fill_(float("nan"))
    _check(hip.lib.ac_mdx_istft(hip._h, _ptr(spec), NI, _ptr(wave), _ptr(scratch), _stream()))
    return wave


xt = torch.randn(4, 48, 256, 3072, generator=g).to(dev)
wt = torch.randn(384, 3072, generator=g) / 55; pk, un_l1 = pack_linear(wt.numpy()); wp_l1 = torch.from_numpy(pk.view(np.int16)).to(dev)
sc = torch.ones(48, device=dev); sh = torch.zeros(48, device=dev)
tdf = lambda: hip.tdf_linear_f16x3(xt, wp_l1, 384, sc, sh, un_l1)
refs = {"istft": istft().clone(), "tdf": tdf().clone()}
torch.cuda.synchronize()
assert torch.equal(istft(), refs["istft"]) and torch.equal(tdf(), refs["tdf"]), "not reproducible even alone"
fns = {"istft": istft, "tdf": tdf}
t0 = rendezvous()
stats = {}
while True:
    ph = phase_at(t0, time.time())
    if ph is None:
        break
    if ph[0] == "wait":
        time.sleep(0.01); continue
    fn, ref = fns[ph[0]], refs[ph[0]]
    bad = [(fn() != ref).sum() for _ in range(4)]
    torch.cuda.synchronize()
    s = stats.setdefault(ph, [0, 0])
    for b in bad:
        s[0] += 1
        s[1] += 1 if int(b) else 0
print("victim kernel : aggressor kernel     failed / iterations")
for (v, a), (n, b) in stats.items():
    print(f"   {v:6s} : {a:12s} {b:5d} / {n:5d}")
print("victim done", flush=True)
