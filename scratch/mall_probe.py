"""Does a small-batch chain of level-0 layers run out of the 256 MB Infinity Cache?  Per-item time of the conv / TDF / resample
kernels at B = 1, 2, 4, 8, 32 when consecutive launches chain through three buffers (x -> y -> z -> x ...)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3
hip = _native.Context()
torch.manual_seed(0)

def chain_ms(B, c, H, W, reps):
    bufs = [torch.randn(B, c, H, W, device='cuda') for _ in range(3)]
    w = torch.randn(c, c, 3, 3) / np.sqrt(9 * c); b = torch.randn(c, device='cuda') * 0.1
    pk, un = pack_conv3x3(w.numpy()); wp = torch.from_numpy(pk.view(np.int16)).cuda()
    def run(n):
        for i in range(n):
            hip.conv3x3_f16x3(bufs[i % 3], wp, b, c, un, relu=True, out=bufs[(i + 1) % 3])
    run(6); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for (c, H, W) in ((48, 3072, 256), (96, 1536, 128)):
    for B in (1, 2, 4, 8, 32):
        ms = chain_ms(B, c, H, W, max(12, 96 // B))
        mb = B * c * H * W * 4 / 1e6
        print(f"C={c} {H}x{W} B={B:2d}: {ms:8.3f} ms/launch = {ms / B * 1e3:7.1f} us/item  (tensor {mb:7.1f} MB, in+out {2 * mb / ms / 1e3:5.2f} TB/s)", flush=True)
