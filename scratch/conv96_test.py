import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3, pack_conv3x3_w96
hip = _native.Context()
torch.manual_seed(0)
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (B, ci, co, H, W) in ((1, 32, 96, 8, 32), (2, 96, 96, 16, 64), (1, 64, 192, 24, 96), (32, 96, 96, 1536, 128), (32, 192, 192, 384, 32), (32, 288, 288, 96, 8)):
    if W % 32: W = 32 * max(1, W // 32)
    x = torch.randn(B, ci, H, W, device='cuda') * 2
    w = torch.randn(co, ci, 3, 3) / np.sqrt(9 * ci); b = torch.randn(co, device='cuda') * 0.1
    pk, un = pack_conv3x3(w.numpy()); wp = torch.from_numpy(pk.view(np.int16)).cuda()
    pk9, un9 = pack_conv3x3_w96(w.numpy()); wp9 = torch.from_numpy(pk9.view(np.int16)).cuda()
    y48 = hip.conv3x3_f16x3(x, wp, b, co, un, relu=True)
    y96 = hip.conv3x3_f16x3_w96(x, wp9, b, co, un9, relu=True)
    nb = min(B, 1)
    ref = torch.relu(torch.nn.functional.conv2d(x[:nb].double(), w.cuda().double(), b.double(), padding=1))
    den = ref.abs().max().item()
    e48 = (y48[:nb].double() - ref).abs().max().item() / den
    e96 = (y96[:nb].double() - ref).abs().max().item() / den
    d = (y96 - y48).abs().max().item() / den
    line = f"B{B} {ci}->{co} {H}x{W}: err48 {e48:.2e} err96 {e96:.2e} |96-48| {d:.2e}"
    if B >= 8:
        x2 = y48.clone()          # post-ReLU data like the network's
        t48 = timeit(lambda: hip.conv3x3_f16x3(x2, wp, b, co, un, relu=True, out=y48), 12)
        t96 = timeit(lambda: hip.conv3x3_f16x3_w96(x2, wp9, b, co, un9, relu=True, out=y96), 12)
        fl = 2.0 * B * ci * co * 9 * H * W
        line += f" | 48: {t48:.3f} ms ({fl / t48 / 1e9:.0f} TF/s)  96: {t96:.3f} ms ({fl / t96 / 1e9:.0f} TF/s)  ratio {t96 / t48:.3f}"
    print(line, flush=True)
