"""2x2 down- / up-sampling kernels per U-Net level: time per launch, algorithmic TB/s (input once + output once (+ skip once)) and an
output hash (bit-identity across kernel variants).  usage: [AC_LIB=libaudiocut_hip_<tag>.so] python tools/resample_bench.py [batch]"""
import os, sys, hashlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
if os.environ.get("AC_LIB"):
    os.environ["AUDIOCUT_HIP_LIBNAME"] = os.environ["AC_LIB"]
from audio_cut_amd.separation.conv_pack import pack_linear
hip = _native.Context()
dev = hip.device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(0)
levels = [(48 * (i + 1), 256 >> i, 3072 >> i) for i in range(6)]


def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for i in range(3):
    (c, h, w), (c2, h2, w2) = levels[i], levels[i + 1]
    # down: level i -> i + 1
    x = (torch.randn(B, c, h, w, generator=g) * 2).to(dev)
    dsw = (torch.randn(c2, c, 2, 2, generator=g) / np.sqrt(4 * c)).numpy()
    packed, un = pack_linear(dsw.reshape(c2, -1), bn=96)
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    bias = (torch.randn(c2, generator=g) * 0.1).to(dev)
    ia = x.abs().amax(dim=(1, 3)).contiguous()
    oa = torch.zeros((B, h2), device=dev)
    out = [None]
    def run_dn():
        out[0] = hip.down2x_f16x3(x, wp, bias, c2, un, in_amax=ia, out_amax=oa)
    ms = timed(run_dn)
    gb = (x.numel() + out[0].numel()) * 4 / 1e9
    print(f"down L{i}->{i + 1} ({c}x{h}x{w} -> {c2}): {ms:7.3f} ms  {gb / ms:6.2f} TB/s  sha1 {hashlib.sha1(out[0].cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
    # up: level i + 1 -> i
    y = out[0]
    usw = (torch.randn(c2, c, 2, 2, generator=g) / np.sqrt(c2)).numpy()
    packed, un2 = pack_linear(usw.transpose(1, 2, 3, 0).reshape(-1, c2), bn=96)
    wp2 = torch.from_numpy(packed.view(np.int16)).to(dev)
    bias2 = (torch.randn(c, generator=g) * 0.1).to(dev)
    ia2 = y.abs().amax(dim=(1, 3)).contiguous()
    oa2 = torch.zeros((B, h), device=dev)
    up = [None]
    def run_up():
        up[0] = hip.up2x_f16x3(y, wp2, bias2, c, un2, skip=x, in_amax=ia2, out_amax=oa2)
    ms = timed(run_up)
    gb = (y.numel() + 2 * up[0].numel()) * 4 / 1e9
    print(f"up   L{i + 1}->{i} ({c2}x{h2}x{w2} -> {c}): {ms:7.3f} ms  {gb / ms:6.2f} TB/s  sha1 {hashlib.sha1(up[0].cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
    del x, y, out, up
