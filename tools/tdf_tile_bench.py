"""TDF layers per U-Net level: time per launch of ac_tdf_linear_f16x3 with 192- and 96-column workgroup tiles.
usage: python tools/tdf_tile_bench.py [batch]     (AC_TDF_NARROW=1 in a scratch build selects the 96-column kernel)"""
import os, sys, hashlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
if os.environ.get("AC_LIB"):
    os.environ["AUDIOCUT_HIP_LIBNAME"] = os.environ["AC_LIB"]
from audio_cut_amd._native import _ptr, _stream, _check
from audio_cut_amd.separation.conv_pack import pack_linear
hip = _native.Context()
dev = hip.device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
narrow = bool(os.environ.get("AC_TDF_NARROW"))
g = torch.Generator().manual_seed(0)
for c, t, f in ((48, 256, 3072), (96, 128, 1536), (144, 64, 768)):
    hd = f // 8
    for layer, (k, n) in (("1", (f, hd)), ("2", (hd, f))):
        x = (torch.randn(B, c, t, k, generator=g) * 2).to(dev)
        wt = torch.randn(n, k, generator=g) / np.sqrt(k)
        sc = (torch.rand(c, generator=g) + 0.5).to(dev); sh = (torch.randn(c, generator=g) * 0.1).to(dev)
        resid = (torch.randn(B, c, t, n, generator=g)).to(dev) if layer == "2" else None
        packed, un = pack_linear(wt.numpy(), bn=96 if (narrow or n % 192) else 0)
        wp = torch.from_numpy(packed.view(np.int16)).to(dev)
        ia = x.abs().amax(dim=(1, 3)).contiguous()
        oa = torch.zeros((B, t), device=dev)
        out = torch.empty((B, c, t, n), device=dev)        # allocated once: a 4.8 GB torch.empty inside the timed loop is not free
        run = lambda: _check(hip.lib.ac_tdf_linear_f16x3(hip._h, _ptr(x), _ptr(wp), _ptr(sc), _ptr(sh), _ptr(resid), _ptr(out), B * c * t, n, k,
                                                         t, c, float(un), _ptr(ia), _ptr(oa), _stream()))
        run(); run(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = (x.numel() + out.numel() * (2 if resid is not None else 1)) * 4 / 1e9
        print(f"C={c:3d} T={t} layer {layer} ({k}->{n}): {ms:7.3f} ms  {2.0 * B * c * t * k * n / ms / 1e9:7.1f} TFLOP/s alg  {gb / ms:6.2f} TB/s  "
              f"sha1 {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
        del x, out, resid
