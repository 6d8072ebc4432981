import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_linear
import torch.nn.functional as F
hip = _native.Context()
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
torch.manual_seed(0)
for (B, C, T, K, N, resid) in [(1, 48, 8, 96, 96, False), (2, 48, 16, 384, 192, True), (16, 48, 256, 3072, 384, False), (16, 48, 256, 384, 3072, True),
                               (16, 96, 128, 1536, 192, False), (16, 96, 128, 192, 1536, True), (16, 144, 64, 768, 96, False), (16, 144, 64, 96, 768, True)]:
    x = torch.randn(B, C, T, K, device='cuda') * 3
    w = torch.randn(N, K) / np.sqrt(K)
    s = (torch.rand(C) + 0.5).cuda(); sh = torch.randn(C).cuda() * 0.3
    r = torch.randn(B, C, T, N, device='cuda') if resid else None
    pk, un = pack_linear(w.numpy()); wp = torch.from_numpy(pk.view(np.int16)).cuda()
    wd = w.cuda()
    y = hip.tdf_linear_f16x3(x, wp, N, s, sh, un, resid=r)
    def ref32():
        z = F.linear(x, wd)
        return hip.affine_relu_add(z, s, sh, r) if resid else hip.affine_relu_(z, s, sh)
    y32 = ref32()
    # float64 reference on a slice of rows (full for small)
    nb = min(B, 2)
    z64 = F.linear(x[:nb].double(), wd.double()) * s.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    z64 = torch.relu(z64)
    if resid: z64 = z64 + r[:nb].double()
    den = z64.abs().max().item()
    e_mine = (y[:nb].double() - z64).abs().max().item() / den
    e_32 = (y32[:nb].double() - z64).abs().max().item() / den
    fl = 2.0 * B * C * T * K * N
    t1 = bench(lambda: hip.tdf_linear_f16x3(x, wp, N, s, sh, un, resid=r))
    t2 = bench(ref32)
    print(f"B{B} C{C} T{T} K{K} N{N} resid={resid}: err f16x3 {e_mine:.2e} (rocblas f32 {e_32:.2e}) | mine {t1:.3f} ms {fl/t1/1e9:.1f} TF/s | rocblas+epi {t2:.3f} ms {fl/t2/1e9:.1f} TF/s")
