"""Where a 3x3 conv launch spends its time: the same launch on ablation builds of the library (W9_PROBE bits of ac_conv96.hip: 4 no MFMA,
8 no activation staging, 0x20 no epilogue, 0x40 no K loop, 0x80 no activation loads, 0x100 weights fetched for stage 0 only).  Outputs of
ablation builds are wrong by construction; only the time is read.
usage: AC_LIB=libaudiocut_hip_<tag>.so python tools/conv_ablation.py [batch]   (tools/run_gpu_batch.sh <tag> conv_ablation:<tags>)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("AC_LIB"):
    os.environ["AUDIOCUT_HIP_LIBNAME"] = os.environ["AC_LIB"]
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96
hip = _native.Context()
dev = hip.device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
res = []
for c, h, w_ in ((48, 256, 3072), (96, 128, 1536), (144, 64, 768)):
    cob = 96 if c % 96 == 0 else 48
    x = torch.randn(B, c, h, w_, device=dev) * 2
    wt = torch.randn(c, c, 3, 3) / np.sqrt(9 * c)
    b = (torch.randn(c) * 0.1).to(dev)
    packed, un = pack_conv3x3_w96(wt.numpy(), cob)
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    fn = hip.conv3x3_f16x3_w96 if cob == 96 else hip.conv3x3_f16x3_s8
    ia = x.abs().amax(dim=(1, 3)).contiguous()
    out = torch.empty_like(x)
    oa = torch.zeros((B, h), device=dev)
    for _ in range(3): fn(x, wp, b, c, un, relu=True, out=out, in_amax=ia, out_amax=oa)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn(x, wp, b, c, un, relu=True, out=out, in_amax=ia, out_amax=oa)
    e1.record(); e1.synchronize()
    res.append(f"C={c} {e0.elapsed_time(e1) / 20:6.3f} ms")
    del x, out
print(f"{os.environ.get('AC_LIB', 'product'):40s} " + "   ".join(res), flush=True)
