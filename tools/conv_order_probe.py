"""Work order of the 3x3 conv kernels: band width (tile columns walked per tile row before the order moves down a row) per level.
Timed alone (20 launches, HIP events) or, with AC_PROBE_PMC=1, two launches per variant for a rocprofv3 --pmc pass whose
per-dispatch rows tools/pmc_by_dispatch.py maps back to the variants (the sequence is printed as JSON on the last line).
usage: python tools/conv_order_probe.py [batch]"""
import os, sys, json, hashlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("AC_LIB"):            # the probe build: make OUT=../libaudiocut_hip_probe.so EXTRA=-DAC_PROBES=1
    os.environ["AUDIOCUT_HIP_LIBNAME"] = os.environ["AC_LIB"]
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96
hip = _native.Context(); dev = hip.device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PMC = os.environ.get("AC_PROBE_PMC") == "1"
g = torch.Generator().manual_seed(0)
seq = []
# (C, H, W, [(tile-width code GX of the 48-channel tile: 2 = 32 px, 3 = 48, 4 = 64; band width in tiles), ...])
if os.environ.get("AC_PROBE_SET", "order") == "order":
    LEVELS = ((48, 256, 3072, [(2, v) for v in (1, 2, 4, 8, 12, 16, 24, 32, 48, 96)]), (96, 128, 1536, [(2, v) for v in (1, 2, 4, 8, 12, 16, 24, 48)]),
              (144, 64, 768, [(2, v) for v in (1, 2, 4, 6, 8, 12, 24)]), (192, 32, 384, [(2, v) for v in (1, 2, 4, 6, 12)]), (240, 16, 192, [(2, v) for v in (1, 2, 3, 6)]))
else:   # "tile": the wider 48-channel tiles
    LEVELS = ((48, 256, 3072, [(2, 4), (2, 24), (3, 8), (3, 16), (3, 32), (4, 6), (4, 12), (4, 24)]), (144, 64, 768, [(2, 4), (2, 24), (3, 8), (3, 16), (4, 6), (4, 12)]),
              (240, 16, 192, [(2, 6), (3, 4), (4, 3)]))
for c, h, w_, variants in LEVELS:
    cob = 96 if c % 96 == 0 else 48
    x = (torch.randn(B, c, h, w_, generator=g) * 2).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)
    b = (torch.randn(c, generator=g) * 0.1).to(dev)
    packed, un = pack_conv3x3_w96(wt.numpy(), cob)
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    fn = hip.conv3x3_f16x3_w96 if cob == 96 else hip.conv3x3_f16x3_s8
    ia = x.abs().amax(dim=(1, 3)).contiguous(); out = torch.empty_like(x); oa = torch.zeros((B, h), device=dev)
    alg = 4.0 * B * c * h * w_
    for gx, bw in variants:
        os.environ["AC_PROBE_CONV_BW"] = str(bw); os.environ["AC_PROBE_CONV_GX"] = str(gx)
        fn(x, wp, b, c, un, relu=True, out=out, in_amax=ia, out_amax=oa)
        torch.cuda.synchronize()
        if PMC:
            fn(x, wp, b, c, un, relu=True, out=out, in_amax=ia, out_amax=oa); torch.cuda.synchronize()
            seq.append({"C": c, "gx": gx, "bw": bw, "launches": 2, "alg_read": alg, "alg_write": alg})
            continue
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(x, wp, b, c, un, relu=True, out=out, in_amax=ia, out_amax=oa)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"C={c:3d} {h}x{w_} gx={gx} bw={bw:3d}: {ms:7.3f} ms  {2.0 * B * c * c * 9 * h * w_ / ms / 1e9:7.1f} TFLOP/s alg  sha1 {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
    del x, out
os.environ.pop("AC_PROBE_CONV_BW", None); os.environ.pop("AC_PROBE_CONV_GX", None)
if PMC:
    print("SEQ " + json.dumps(seq))
