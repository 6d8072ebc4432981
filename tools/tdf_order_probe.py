"""Work order of ac_tdf_linear_f16x3 (layer 2 of the TDF pairs: K = F / 8 -> N = F with the residual): G column blocks x R row tiles per
super-group of the walk.  Needs the probe build (make OUT=../libaudiocut_hip_probe.so EXTRA=-DAC_PROBES=1; AC_LIB=libaudiocut_hip_probe.so).
Timed alone (20 launches) or, with AC_PROBE_PMC=1, two launches per variant for a rocprofv3 --pmc pass (tools/pmc_by_dispatch.py --seq).
usage: python tools/tdf_order_probe.py [batch]"""
import os, sys, json, hashlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("AC_LIB"):
    os.environ["AUDIOCUT_HIP_LIBNAME"] = os.environ["AC_LIB"]
from audio_cut_amd import _native
from audio_cut_amd._native import _ptr, _stream, _check
from audio_cut_amd.separation.conv_pack import pack_linear
hip = _native.Context(); dev = hip.device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PMC = os.environ.get("AC_PROBE_PMC") == "1"
g = torch.Generator().manual_seed(0)
seq = []
LEVELS = ((48, 256, 3072, [(16, 1), (8, 8), (8, 16), (4, 16), (4, 32), (2, 32), (2, 64), (1, 64), (1, 128)]),
          (96, 128, 1536, [(8, 1), (4, 16), (2, 32), (1, 64)]), (144, 64, 768, [(4, 1), (2, 32), (1, 64)]))
for c, t, f, variants in LEVELS:
    k, n = f // 8, f
    x = (torch.randn(B, c, t, k, generator=g) * 2).to(dev)
    wt = torch.randn(n, k, generator=g) / np.sqrt(k)
    sc = (torch.rand(c, generator=g) + 0.5).to(dev); sh = (torch.randn(c, generator=g) * 0.1).to(dev)
    resid = torch.randn(B, c, t, n, generator=g).to(dev)
    packed, un = pack_linear(wt.numpy())
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    ia = x.abs().amax(dim=(1, 3)).contiguous(); oa = torch.zeros((B, t), device=dev)
    out = torch.empty((B, c, t, n), device=dev)
    run = lambda: _check(hip.lib.ac_tdf_linear_f16x3(hip._h, _ptr(x), _ptr(wp), _ptr(sc), _ptr(sh), _ptr(resid), _ptr(out), B * c * t, n, k, t, c, float(un), _ptr(ia), _ptr(oa), _stream()))
    alg_r = (x.numel() + resid.numel()) * 4.0; alg_w = out.numel() * 4.0
    for G, R in variants:
        os.environ["AC_PROBE_TDF_ORDER"] = f"{G},{R}"
        run(); torch.cuda.synchronize()
        if PMC:
            run(); torch.cuda.synchronize()
            seq.append({"C": c, "G": G, "R": R, "launches": 2, "alg_read": alg_r, "alg_write": alg_w})
            continue
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"C={c:3d} T={t} {k}->{n} G={G:2d} R={R:3d}: {ms:7.3f} ms  {(alg_r + alg_w) / ms / 1e9:6.2f} TB/s alg  sha1 {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
    del x, out, resid
os.environ.pop("AC_PROBE_TDF_ORDER", None)
if PMC:
    print("SEQ " + json.dumps(seq))
