"""Where do the irreproducible outputs differ?  net -> istft chain under GPU sharing; differing results compared element by element."""
import sys, os, hashlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_cut_amd import _native
from audio_cut_amd.separation.backends import MDX23HipBackend
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
rank = int(sys.argv[1])
hip = _native.Context("cuda:0"); dev = hip.device
backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip, max_items_per_forward=32); backend.load_model()
net = backend.net
g = torch.Generator().manual_seed(rank)
track = torch.randn(441000 * 20, generator=g).to(dev) * 0.3
cs = hip.to_device(np.repeat(np.arange(16) * 330750, 2).astype(np.int64)); cl = hip.to_device(np.full(32, 441000, np.int64)); wi = hip.to_device(np.tile([0, 1], 16).astype(np.int32))
spec_fixed = hip.mdx_stft(track, cs, cl, wi); torch.cuda.synchronize()
ref = hip.mdx_istft(net.forward_tf(spec_fixed)); torch.cuda.synchronize()
refh = ref.cpu().numpy()
for rep in range(4):
    outs = [hip.mdx_istft(net.forward_tf(spec_fixed)) for _ in range(4)]
    for k, o in enumerate(outs):
        oh = o.cpu().numpy()
        d = np.abs(oh - refh)
        if d.max() > 0:
            idx = np.argwhere(d > 0)
            items = np.unique(idx[:, 0]); chans = np.unique(idx[:, 1])
            pos = idx[:, 2]
            print(f"rank {rank} rep {rep} out {k}: {len(idx)} samples differ, max {d.max():.3e} (peak {np.abs(refh).max():.2f}); items {items.tolist()[:12]} (n={len(items)}), channels {chans.tolist()}, "
                  f"positions {pos.min()}..{pos.max()}, per-item counts {[int((idx[:,0]==i).sum()) for i in items[:8]]}", flush=True)
            # run structure inside the first differing item
            i0 = items[0]; p = np.flatnonzero(d[i0, chans[0]] > 0)
            runs = np.split(p, np.flatnonzero(np.diff(p) > 1) + 1)
            print("   first item runs (start,len):", [(int(r[0]), len(r)) for r in runs[:10]], "n_runs", len(runs), flush=True)
print(rank, "done", flush=True)
