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


def test_fused_epilogues_match_the_unfused_torch_ops(hip_ctx):
    """TfcTdfNet with the HIP epilogues (bias+ReLU, affine+ReLU(+residual), bias+ReLU*skip) == plain PyTorch elementwise ops."""
    spec = TfcTdfSpec()
    w = synth_weights(spec, seed=0)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(2, 4, 32, 3072, generator=g) * 3.0).to(hip_ctx.device)
    plain = TfcTdfNet(w, spec).to(hip_ctx.device).eval()
    fused = TfcTdfNet(w, spec, hip=hip_ctx).to(hip_ctx.device).eval()
    a = plain.forward_tf(x); b = fused.forward_tf(x)
    assert float((a - b).abs().max() / a.abs().max()) < 2e-6
    # the kernels themselves, on odd row counts
    t = torch.randn(3, 5, 7, 12, device=hip_ctx.device); bias = torch.randn(5, device=hip_ctx.device)
    sk = torch.randn_like(t); sc = torch.randn(5, device=hip_ctx.device)
    assert torch.equal(hip_ctx.bias_relu_(t.clone(), bias), torch.relu(t + bias.view(1, -1, 1, 1)))
    assert torch.equal(hip_ctx.bias_relu_mul_(t.clone(), bias, sk), torch.relu(t + bias.view(1, -1, 1, 1)) * sk)
    ref = torch.relu(t * sc.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    assert torch.equal(hip_ctx.affine_relu_(t.clone(), sc, bias), ref)
    assert torch.equal(hip_ctx.affine_relu_add(t.clone(), sc, bias, sk), sk + ref)
