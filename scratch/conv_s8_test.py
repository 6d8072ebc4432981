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
for (B, ci, co, H, W) in ((1, 48, 48, 8, 32), (2, 16, 96, 16, 64), (1, 48, 96, 24, 96), (32, 48, 48, 3072, 256), (32, 144, 144, 768, 64), (32, 240, 240, 192, 32)):
    x = torch.randn(B, ci, H, W, device='cuda') * 2
    w = torch.randn(co, ci, 3, 3) / np.sqrt(9 * ci); b = torch.randn(co, device='cuda') * 0.1
    ref = torch.relu(torch.nn.functional.conv2d(x[:1].double(), w.cuda().double(), b.double(), padding=1)); den = ref.abs().max().item()
    line = f"B{B} {ci}->{co} {H}x{W}:"
    outs = {}
    if co % 48 == 0 and ci % 16 == 0:
        pk, un = pack_conv3x3(w.numpy()); wp = torch.from_numpy(pk.view(np.int16)).cuda()
        outs["48"] = (lambda o=None: hip.conv3x3_f16x3(x, wp, b, co, un, relu=True, out=o))
        pk8, un8 = pack_conv3x3_w96(w.numpy(), 48); wp8 = torch.from_numpy(pk8.view(np.int16)).cuda()
        outs["s8"] = (lambda o=None: hip.conv3x3_f16x3_s8(x, wp8, b, co, un8, relu=True, out=o))
    if co % 96 == 0:
        pk9, un9 = pack_conv3x3_w96(w.numpy(), 96); wp9 = torch.from_numpy(pk9.view(np.int16)).cuda()
        outs["w96"] = (lambda o=None: hip.conv3x3_f16x3_w96(x, wp9, b, co, un9, relu=True, out=o))
    for name, fn in outs.items():
        y = fn()
        line += f" err_{name} {(y[:1].double() - ref).abs().max().item() / den:.2e}"
    if B >= 8:
        x.copy_(torch.relu(x))
        buf = torch.empty(B, co, H, W, device='cuda')
        fl = 2.0 * B * ci * co * 9 * H * W
        for name, fn in outs.items():
            t = timeit(lambda: fn(buf), 10)
            line += f" | {name}: {t:.3f} ms ({fl / t / 1e9:.0f} TF/s)"
    print(line, flush=True)
