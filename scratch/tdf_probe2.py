import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_linear
hip = _native.Context()
torch.manual_seed(0)
def run(B, C, T, K, N, resid, reps=6):
    x = torch.randn(B, C, T, K, device='cuda')
    w = torch.randn(N, K) / np.sqrt(K)
    s = (torch.rand(C) + 0.5).cuda(); sh = torch.randn(C).cuda() * 0.3
    r = torch.randn(B, C, T, N, device='cuda') if resid else None
    pk, un = pack_linear(w.numpy()); wp = torch.from_numpy(pk.view(np.int16)).cuda()
    for _ in range(2): hip.tdf_linear_f16x3(x, wp, N, s, sh, un, resid=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): hip.tdf_linear_f16x3(x, wp, N, s, sh, un, resid=r)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * C * T * K * N
    gb = (x.numel() + B * C * T * N * (2 if resid else 1)) * 4 / 1e9
    print(f"B{B} C{C} T{T} K{K} N{N} resid={int(resid)}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s alg  {3 * fl / ms / 1e9:7.1f} issued  min-traffic {gb:5.2f} GB = {gb / ms:5.2f} TB/s", flush=True)
for args in [(32, 48, 256, 384, 3072, True), (32, 48, 256, 384, 3168, True), (32, 96, 128, 192, 1536, True), (32, 96, 128, 192, 1632, True),
             (32, 48, 256, 3072, 384, False), (32, 48, 256, 3072, 480, False)]:
    run(*args)
