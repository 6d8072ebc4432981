"""The U-Net's big kernels, three launches each at the shapes that carry the time (B = 32): the target of rocprofv3 --pmc passes
(tools/run_gpu_batch.sh <tag> sq_probe) that ask where their waves spend their cycles."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
from audio_cut_amd._native import _ptr, _stream, _check
from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96, pack_linear
hip = _native.Context()
dev = hip.device
B = 32
g = torch.Generator().manual_seed(0)
N_LAUNCH = 3
# 3x3 convs: level 0 (48-channel tile), level 1 (96-channel tile)
for c, h, w_ in ((48, 256, 3072), (96, 128, 1536)):
    cob = 96 if c % 96 == 0 else 48
    x = (torch.randn(B, c, h, w_, generator=g) * 2).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)
    b = (torch.randn(c, generator=g) * 0.1).to(dev)
    packed, un = pack_conv3x3_w96(wt.numpy(), cob)
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    fn = hip.conv3x3_f16x3_w96 if cob == 96 else hip.conv3x3_f16x3_s8
    ia = x.abs().amax(dim=(1, 3)).contiguous(); out = torch.empty_like(x); oa = torch.zeros((B, h), device=dev)
    for _ in range(N_LAUNCH):
        fn(x, wp, b, c, un, relu=True, out=out, in_amax=ia, out_amax=oa)
    torch.cuda.synchronize(); del x, out
# TDF layers at level 0
c, t, f = 48, 256, 3072
for layer, (k, n) in (("1", (f, f // 8)), ("2", (f // 8, f))):
    x = (torch.randn(B, c, t, k, generator=g) * 2).to(dev)
    wt = torch.randn(n, k, generator=g) / np.sqrt(k)
    sc = (torch.rand(c, generator=g) + 0.5).to(dev); sh = (torch.randn(c, generator=g) * 0.1).to(dev)
    resid = torch.randn(B, c, t, n, generator=g).to(dev) if layer == "2" else None
    packed, un = pack_linear(wt.numpy())
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    ia = x.abs().amax(dim=(1, 3)).contiguous(); oa = torch.zeros((B, t), device=dev)
    out = torch.empty((B, c, t, n), device=dev)
    for _ in range(N_LAUNCH):
        _check(hip.lib.ac_tdf_linear_f16x3(hip._h, _ptr(x), _ptr(wp), _ptr(sc), _ptr(sh), _ptr(resid), _ptr(out), B * c * t, n, k, t, c, float(un), _ptr(ia), _ptr(oa), _stream()))
    torch.cuda.synchronize(); del x, out, resid
# 2x2 resampling between levels 0 and 1
c, h, w_, c2 = 48, 256, 3072, 96
x = (torch.randn(B, c, h, w_, generator=g) * 2).to(dev)
dsw = (torch.randn(c2, c, 2, 2, generator=g) / np.sqrt(4 * c)).numpy()
packed, un = pack_linear(dsw.reshape(c2, -1), bn=96); wp = torch.from_numpy(packed.view(np.int16)).to(dev)
bias = (torch.randn(c2, generator=g) * 0.1).to(dev)
ia = x.abs().amax(dim=(1, 3)).contiguous(); oa = torch.zeros((B, h // 2), device=dev)
for _ in range(N_LAUNCH):
    y = hip.down2x_f16x3(x, wp, bias, c2, un, in_amax=ia, out_amax=oa)
usw = (torch.randn(c2, c, 2, 2, generator=g) / np.sqrt(c2)).numpy()
packed, un2 = pack_linear(usw.transpose(1, 2, 3, 0).reshape(-1, c2), bn=96); wp2 = torch.from_numpy(packed.view(np.int16)).to(dev)
bias2 = (torch.randn(c, generator=g) * 0.1).to(dev)
ia2 = y.abs().amax(dim=(1, 3)).contiguous(); oa2 = torch.zeros((B, h), device=dev)
for _ in range(N_LAUNCH):
    u = hip.up2x_f16x3(y, wp2, bias2, c, un2, skip=x, in_amax=ia2, out_amax=oa2)
torch.cuda.synchronize()
print("done")
