"""TFC-TDF U-Net (PyTorch-ROCm, folded BN, T-major layout) against a float64 CPU evaluation, block by block."""
import copy

import numpy as np
import pytest
import torch

from audio_cut_amd.separation.tfc_tdf import TfcTdfNet, TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
from oracle import chunking as OC
from oracle.separator import mdx_stft, unet_forward

pytestmark = pytest.mark.gpu


def test_unet_blocks_and_output_vs_float64(hip_ctx):
    spec = TfcTdfSpec()
    w = synth_weights(spec, seed=0)
    mix = signals.c2_song(10.0, seed=4)
    batch, _, _ = OC.mdx_windows(mix)
    x = mdx_stft(batch[:1])[..., :64].contiguous()          # [1, 4, 3072, 64]: 64 frames keep the float64 CPU pass short
    net = TfcTdfNet(w, spec).eval()
    net64 = copy.deepcopy(net).double()
    netg = copy.deepcopy(net).to(hip_ctx.device)
    got, ref = {}, {}

    def hook(store, name):
        return lambda m, i, o: store.__setitem__(name, o.detach().double().cpu())

    for n_, st in ((netg, got), (net64, ref)):
        for i, b in enumerate(n_.enc):
            b.register_forward_hook(hook(st, f"enc{i}"))
        n_.bottleneck.register_forward_hook(hook(st, "bottleneck"))
        for i, b in enumerate(n_.dec):
            b.register_forward_hook(hook(st, f"dec{i}"))
    yg = netg(x.to(hip_ctx.device)).double().cpu()
    y64 = net64(x.double())
    for name, r in ref.items():
        err = float((got[name] - r).abs().max() / r.abs().max())
        assert err < 2e-4, (name, err)                      # every level, relative to that block's own peak
    assert float((yg - y64).abs().max() / y64.abs().max()) < 1e-5
    # the un-fused oracle graph (conv -> BN -> ReLU as separate float32 ops) agrees with the folded net
    yo = unet_forward(x, w).double()
    assert float((yo - y64).abs().max() / y64.abs().max()) < 1e-5
    # ONNX-shaped entry point == T-major entry point
    ytf = netg.forward_tf(x.transpose(-1, -2).contiguous().to(hip_ctx.device)).transpose(-1, -2).double().cpu()
    assert torch.equal(ytf, yg)
