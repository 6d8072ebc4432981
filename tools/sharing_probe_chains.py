"""Asynchronous chains of single stages under GPU sharing: which chain is not reproducible?"""
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
h = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:8]
def chain_stft(n=6):
    outs = [hip.mdx_stft(track, cs, cl, wi) for _ in range(n)]
    return [h(o) for o in outs]
spec_fixed = hip.mdx_stft(track, cs, cl, wi); torch.cuda.synchronize()
def chain_istft(n=6):
    outs = [hip.mdx_istft(spec_fixed) for _ in range(n)]
    return [h(o) for o in outs]
def chain_stft_istft(n=6):
    outs = [hip.mdx_istft(hip.mdx_stft(track, cs, cl, wi)) for _ in range(n)]
    return [h(o) for o in outs]
def chain_net(n=4):
    outs = [net.forward_tf(spec_fixed) for _ in range(n)]
    return [h(o) for o in outs]
def chain_net_istft(n=4):
    outs = [hip.mdx_istft(net.forward_tf(spec_fixed)) for _ in range(n)]
    return [h(o) for o in outs]
def chain_stft_net(n=4):
    outs = [net.forward_tf(hip.mdx_stft(track, cs, cl, wi)) for _ in range(n)]
    return [h(o) for o in outs]
import contextlib
ctxm = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("AC_EXPLICIT_STREAM") else contextlib.nullcontext()
ctxm.__enter__()
if os.environ.get("AC_EXPLICIT_STREAM"):
    spec_fixed = hip.mdx_stft(track, cs, cl, wi); torch.cuda.synchronize()
for name, fn in (("stft", chain_stft), ("istft", chain_istft), ("stft->istft", chain_stft_istft), ("net", chain_net), ("net->istft", chain_net_istft), ("stft->net", chain_stft_net)):
    for rep in range(2):
        o = fn()
        print(rank, f"{name:12s}", "distinct outputs:", len(set(o)), o, flush=True)
