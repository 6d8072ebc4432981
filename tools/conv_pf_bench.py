"""3x3 conv kernels per U-Net level: time per launch (HIP events, 20 launches) and bit-equality across the prefetch variants.
usage: python tools/conv_pf_bench.py [batch]"""
import os, sys, numpy as np, torch, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
if os.environ.get("AC_LIB"):            # A/B runs: another build of the library (make OUT=../libaudiocut_hip_<tag>.so BUILD=build_<tag> EXTRA=-D...)
    os.environ["AUDIOCUT_HIP_LIBNAME"] = os.environ["AC_LIB"]
from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96
hip = _native.Context()
dev = hip.device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(0)
for c, h, w_ in ((48, 256, 3072), (96, 128, 1536), (144, 64, 768), (192, 32, 384), (240, 16, 192), (288, 8, 96)):
    cob = 96 if c % 96 == 0 else 48
    x = (torch.randn(B, c, h, w_, generator=g) * 2).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)
    b = (torch.randn(c, generator=g) * 0.1).to(dev)
    packed, un = pack_conv3x3_w96(wt.numpy(), cob)
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    fn = hip.conv3x3_f16x3_w96 if cob == 96 else hip.conv3x3_f16x3_s8
    ia = x.abs().amax(dim=(1, 3)).contiguous()
    out = torch.empty_like(x)
    oa = torch.zeros((B, h), device=dev)
    fn(x, wp, b, c, un, relu=True, out=out, in_amax=ia, out_amax=oa)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(x, wp, b, c, un, relu=True, out=out, in_amax=ia, out_amax=oa)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * B * c * c * 9 * h * w_
    print(f"C={c:3d} {h}x{w_}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s alg  sha1 {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
    del x, out
