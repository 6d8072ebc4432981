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
    netg = TfcTdfNet(w, spec, hip=hip_ctx).to(hip_ctx.device).eval()    # product configuration: f16x3 MFMA convs + fused epilogues
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
    errs = {name: float((got[name] - r).abs().max() / r.abs().max()) for name, r in ref.items()}
    print("block errors vs float64 (relative to block peak):", {k: f"{v:.1e}" for k, v in errs.items()})
    for name, err in errs.items():
        assert err < 1e-3, (name, err)                      # every level, relative to that block's own peak
    out_err = float((yg - y64).abs().max() / y64.abs().max())
    print(f"output error vs float64: {out_err:.2e}")
    assert out_err < 1e-5           # float32-class: the split-f16 convs keep 22 mantissa bits
    hip_ctx.conv_impl = "miopen"                                            # float32 MIOpen convs + rocBLAS TDF: plain float32 accuracy
    hip_ctx.tdf_impl = "rocblas"
    hip_ctx.resample_impl = "gemm"
    try:
        y32 = netg(x.to(hip_ctx.device)).double().cpu()
    finally:
        hip_ctx.conv_impl = "f16x3"
        hip_ctx.tdf_impl = "f16x3"
        hip_ctx.resample_impl = "f16x3"
    assert float((y32 - y64).abs().max() / y64.abs().max()) < 1e-5
    # the un-fused oracle graph (conv -> BN -> ReLU as separate float32 ops) agrees with the folded net
    yo = unet_forward(x, w).double()
    assert float((yo - y64).abs().max() / y64.abs().max()) < 1e-5
    # ONNX-shaped entry point == T-major entry point
    ytf = netg.forward_tf(x.transpose(-1, -2).contiguous().to(hip_ctx.device)).transpose(-1, -2).double().cpu()
    assert torch.equal(ytf, yg)


def test_conv3x3_f16x3_kernel(hip_ctx):
    """ac_conv3x3_f16x3 against a float64 convolution on every U-Net level shape (batch 2)."""
    import torch.nn.functional as F
    from audio_cut_amd.separation.conv_pack import pack_conv3x3
    g = torch.Generator().manual_seed(0)
    for c, h, w_ in ((48, 256, 3072), (96, 128, 1536), (144, 64, 768), (192, 32, 384), (240, 16, 192), (288, 8, 96)):
        x = (torch.randn(2, c, h, w_, generator=g) * 2).to(hip_ctx.device)
        wt = torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)
        b = torch.randn(c, generator=g) * 0.1
        packed, unscale = pack_conv3x3(wt.numpy())
        wp = torch.from_numpy(packed.view(np.int16)).to(hip_ctx.device)
        for relu in (True, False):
            y = hip_ctx.conv3x3_f16x3(x, wp, b.to(hip_ctx.device), c, unscale, relu=relu).double().cpu()
            ref = F.conv2d(x.double().cpu(), wt.double(), b.double(), padding=1)
            ref = F.relu(ref) if relu else ref
            assert float((y - ref).abs().max() / ref.abs().max()) < 2e-6, (c, relu)
        # borders: zero padding on all four sides, asymmetric weights catch a transposed tap or row/col swap
        edge = F.conv2d(x.double().cpu(), wt.double(), b.double(), padding=1)[:, :, [0, -1], :]
        got = hip_ctx.conv3x3_f16x3(x, wp, b.to(hip_ctx.device), c, unscale, relu=False).double().cpu()[:, :, [0, -1], :]
        assert float((got - edge).abs().max() / edge.abs().max()) < 2e-6


def test_conv3x3_f16x3_w96_kernel(hip_ctx):
    """ac_conv3x3_f16x3_w96 (96 output channels per workgroup, 8-channel stages, tap 8 shared by four stages) against a
    float64 convolution: the U-Net level shapes it serves, a rectangular C_in != C_out case, one tile, and the borders."""
    import torch.nn.functional as F
    from audio_cut_amd.separation.conv_pack import conv3x3_wide_tileable, pack_conv3x3_w96
    g = torch.Generator().manual_seed(5)
    dev = hip_ctx.device
    for ci, co, h, w_ in ((96, 96, 128, 1536), (192, 192, 32, 384), (288, 288, 8, 96), (32, 96, 8, 32), (64, 192, 24, 96)):
        assert conv3x3_wide_tileable(co, ci)
        x = (torch.randn(2, ci, h, w_, generator=g) * 2).to(dev)
        wt = torch.randn(co, ci, 3, 3, generator=g) / np.sqrt(9 * ci)
        b = torch.randn(co, generator=g) * 0.1
        packed, unscale = pack_conv3x3_w96(wt.numpy())
        wp = torch.from_numpy(packed.view(np.int16)).to(dev)
        full = F.conv2d(x.double().cpu(), wt.double(), b.double(), padding=1)
        for relu in (True, False):
            y = hip_ctx.conv3x3_f16x3_w96(x, wp, b.to(dev), co, unscale, relu=relu).double().cpu()
            ref = F.relu(full) if relu else full
            assert float((y - ref).abs().max() / ref.abs().max()) < 2e-6, (ci, co, relu)
            if not relu:
                edges = [y[:, :, [0, -1], :] - full[:, :, [0, -1], :], y[:, :, :, [0, -1]] - full[:, :, :, [0, -1]]]
                assert max(float(e.abs().max()) for e in edges) / float(full.abs().max()) < 2e-6
    assert not conv3x3_wide_tileable(144, 144) and conv3x3_wide_tileable(96, 48) and not conv3x3_wide_tileable(96, 40)
    with pytest.raises(Exception):
        hip_ctx.conv3x3_f16x3_w96(torch.zeros(1, 40, 8, 32, device=dev), wp, b.to(dev), 96, 1.0)
    # trailing group of two stages (C_in % 32 == 16) and the 48-channel / three-workgroups-per-CU variant on the other levels
    for ci, co, h, w_, cob in ((48, 96, 16, 64, 96), (16, 96, 8, 32, 96), (48, 48, 256, 3072, 48), (144, 144, 64, 768, 48), (240, 240, 16, 192, 48)):
        x = (torch.randn(2, ci, h, w_, generator=g) * 2).to(dev)
        wt = torch.randn(co, ci, 3, 3, generator=g) / np.sqrt(9 * ci)
        b = torch.randn(co, generator=g) * 0.1
        packed, unscale = pack_conv3x3_w96(wt.numpy(), cob)
        wp = torch.from_numpy(packed.view(np.int16)).to(dev)
        fn = hip_ctx.conv3x3_f16x3_w96 if cob == 96 else hip_ctx.conv3x3_f16x3_s8
        full = F.conv2d(x.double().cpu(), wt.double(), b.double(), padding=1)
        y = fn(x, wp, b.to(dev), co, unscale, relu=False).double().cpu()
        assert float((y - full).abs().max() / full.abs().max()) < 2e-6, (ci, co, cob)
        edges = [y[:, :, [0, -1], :] - full[:, :, [0, -1], :], y[:, :, :, [0, -1]] - full[:, :, :, [0, -1]]]
        assert max(float(e.abs().max()) for e in edges) / float(full.abs().max()) < 2e-6
        y = fn(x, wp, b.to(dev), co, unscale, relu=True).double().cpu()
        assert float((y - F.relu(full)).abs().max() / full.abs().max()) < 2e-6


def test_tdf_linear_f16x3_kernel(hip_ctx):
    """ac_tdf_linear_f16x3 (GEMM + per-channel affine + ReLU (+ residual)) against float64 on the U-Net's TDF shapes."""
    import torch.nn.functional as F
    from audio_cut_amd.separation.conv_pack import pack_linear
    g = torch.Generator().manual_seed(3)
    dev = hip_ctx.device
    for (b, c, t, k, n) in ((2, 48, 256, 3072, 384), (2, 48, 256, 384, 3072), (2, 96, 128, 1536, 192), (2, 96, 128, 192, 1536),
                            (2, 144, 64, 768, 96), (2, 144, 64, 96, 768), (1, 48, 8, 96, 96), (3, 48, 8, 32, 288)):
        x = (torch.randn(b, c, t, k, generator=g) * 3).to(dev)
        wt = torch.randn(n, k, generator=g) / np.sqrt(k)           # asymmetric: catches a transposed operand
        sc = (torch.rand(c, generator=g) + 0.5).to(dev); sh = (torch.randn(c, generator=g) * 0.3).to(dev)
        packed, unscale = pack_linear(wt.numpy())
        wp = torch.from_numpy(packed.view(np.int16)).to(dev)
        for with_resid in (False, True):
            r = torch.randn(b, c, t, n, generator=g).to(dev) if with_resid else None
            y = hip_ctx.tdf_linear_f16x3(x, wp, n, sc, sh, unscale, resid=r).double().cpu()
            ref = torch.relu(F.linear(x.double().cpu(), wt.double()) * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1))
            if with_resid:
                ref = ref + r.double().cpu()
            assert float((y - ref).abs().max() / ref.abs().max()) < 3e-6, (b, c, t, k, n, with_resid)
    with pytest.raises(Exception):
        hip_ctx.tdf_linear_f16x3(torch.zeros(1, 3, 5, 32, device=dev), wp, 96, sc, sh, 1.0)     # rows % 128 != 0: refused, not mis-tiled


def test_fused_epilogues_match_the_unfused_torch_ops(hip_ctx):
    """TfcTdfNet with the HIP epilogues (bias+ReLU, affine+ReLU(+residual), bias+ReLU*skip) == plain PyTorch elementwise ops."""
    spec = TfcTdfSpec()
    w = synth_weights(spec, seed=0)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(2, 4, 32, 3072, generator=g) * 3.0).to(hip_ctx.device)
    plain = TfcTdfNet(w, spec).to(hip_ctx.device).eval()
    fused = TfcTdfNet(w, spec, hip=hip_ctx).to(hip_ctx.device).eval()
    hip_ctx.conv_impl = "miopen"        # same float32 MIOpen convs / rocBLAS GEMMs on both sides: only the epilogues differ
    hip_ctx.tdf_impl = "rocblas"
    hip_ctx.resample_impl = "miopen"
    try:
        a = plain.forward_tf(x); b = fused.forward_tf(x)
    finally:
        hip_ctx.conv_impl = "f16x3"
        hip_ctx.tdf_impl = "f16x3"
        hip_ctx.resample_impl = "f16x3"
    assert float((a - b).abs().max() / a.abs().max()) < 2e-6
    # the kernels themselves, on odd row counts
    t = torch.randn(3, 5, 7, 12, device=hip_ctx.device); bias = torch.randn(5, device=hip_ctx.device)
    sk = torch.randn_like(t); sc = torch.randn(5, device=hip_ctx.device)
    assert torch.equal(hip_ctx.bias_relu_(t.clone(), bias), torch.relu(t + bias.view(1, -1, 1, 1)))
    assert torch.equal(hip_ctx.bias_relu_mul_(t.clone(), bias, sk), torch.relu(t + bias.view(1, -1, 1, 1)) * sk)
    ref = torch.relu(t * sc.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    assert torch.equal(hip_ctx.affine_relu_(t.clone(), sc, bias), ref)
    assert torch.equal(hip_ctx.affine_relu_add(t.clone(), sc, bias, sk), sk + ref)


def test_gemm_form_resampling_matches_strided_convs(hip_ctx):
    """space-to-depth + GEMM (down) and GEMM + depth-to-space (up) against conv2d(stride 2) / conv_transpose2d(stride 2)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(2)
    dev = hip_ctx.device
    for c, h, w_ in ((48, 32, 64), (96, 16, 32), (240, 8, 12)):
        x = torch.randn(3, c, h, w_, generator=g).to(dev)
        wd = (torch.randn(c + 48, c, 2, 2, generator=g) / np.sqrt(4 * c)).to(dev)
        bd = torch.randn(c + 48, generator=g).to(dev)
        ref = F.relu(F.conv2d(x, wd, bd, stride=2))
        x2 = hip_ctx.space_to_depth2x(x).view(3, 4 * c, (h // 2) * (w_ // 2))
        got = hip_ctx.bias_relu_(torch.matmul(wd.permute(0, 2, 3, 1).reshape(c + 48, -1), x2).view(3, c + 48, h // 2, w_ // 2), bd)
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
        co = max(48, c - 48)
        wu = (torch.randn(c, co, 2, 2, generator=g) / np.sqrt(c)).to(dev)
        bu = torch.randn(co, generator=g).to(dev)
        skip = torch.randn(3, co, 2 * h, 2 * w_, generator=g).to(dev)
        ref = F.relu(F.conv_transpose2d(x, wu, bu, stride=2)) * skip
        y4 = torch.matmul(wu.permute(2, 3, 1, 0).reshape(-1, c), x.view(3, c, h * w_)).view(3, 4 * co, h, w_)
        got = hip_ctx.depth_to_space2x_bias_relu_mul(y4, bu, skip)
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6


def test_fused_resampling_kernels_vs_float64(hip_ctx):
    """ac_down2x_f16x3 / ac_up2x_f16x3 (gather/scatter fused around the split-f16 MFMA GEMM) against float64 strided /
    transposed convolutions, on U-Net level shapes incl. the channel counts that need N / K zero padding (144, 240)."""
    import torch.nn.functional as F
    from audio_cut_amd.separation.conv_pack import pack_linear
    g = torch.Generator().manual_seed(5)
    dev = hip_ctx.device
    for c, h, w_ in ((48, 32, 3072), (96, 16, 1536), (144, 64, 768), (192, 32, 384), (240, 16, 192)):
        x = (torch.randn(2, c, h, w_, generator=g) * 2)
        co = c + 48
        wd = torch.randn(co, c, 2, 2, generator=g) / np.sqrt(4 * c)
        bd = torch.randn(co, generator=g) * 0.2
        packed, un = pack_linear(wd.numpy().reshape(co, -1), bn=96)
        got = hip_ctx.down2x_f16x3(x.to(dev), torch.from_numpy(packed.view(np.int16)).to(dev), bd.to(dev), co, un).double().cpu()
        ref = F.relu(F.conv2d(x.double(), wd.double(), bd.double(), stride=2))
        assert got.shape == ref.shape
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6, ("down", c)
    for c, h, w_ in ((288, 8, 96), (240, 16, 192), (192, 32, 384), (144, 64, 768), (96, 16, 1536)):
        x = (torch.randn(2, c, h, w_, generator=g) * 2)
        co = c - 48
        wu = torch.randn(c, co, 2, 2, generator=g) / np.sqrt(c)
        bu = torch.randn(co, generator=g) * 0.2
        skip = torch.randn(2, co, 2 * h, 2 * w_, generator=g)
        packed, un = pack_linear(wu.numpy().transpose(1, 2, 3, 0).reshape(-1, c), bn=96)
        wp = torch.from_numpy(packed.view(np.int16)).to(dev)
        ref = F.relu(F.conv_transpose2d(x.double(), wu.double(), bu.double(), stride=2))
        got = hip_ctx.up2x_f16x3(x.to(dev), wp, bu.to(dev), co, un, skip=None).double().cpu()
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6, ("up", c)
        got = hip_ctx.up2x_f16x3(x.to(dev), wp, bu.to(dev), co, un, skip=skip.to(dev)).double().cpu()
        ref = ref * skip.double()
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6, ("up*skip", c)


def test_conv1x1_small_kernel(hip_ctx):
    """ac_conv1x1_small (the graph's first 4 -> g and last g -> 4 convolutions) against float64."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(6)
    dev = hip_ctx.device
    for ci, co, relu in ((4, 48, True), (48, 4, False), (3, 5, True), (7, 2, False)):
        x = torch.randn(3, ci, 8, 100, generator=g) * 2
        wt = torch.randn(co, ci, 1, 1, generator=g) / np.sqrt(ci)
        b = torch.randn(co, generator=g)
        got = hip_ctx.conv1x1_small(x.to(dev), wt.to(dev), b.to(dev), relu=relu).double().cpu()
        ref = F.conv2d(x.double(), wt.double(), b.double())
        ref = F.relu(ref) if relu else ref
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-6, (ci, co)


def test_first_conv_fused_into_the_3x3_loader_is_bit_identical(hip_ctx):
    """ac_conv3x3_f16x3_first == ac_conv1x1_small followed by ac_conv3x3_f16x3, bit for bit (same float32 FMA order for the
    generated channels, same zero padding of the conv input), and the whole net is unchanged by the fusion."""
    from audio_cut_amd.separation.conv_pack import pack_conv3x3
    g = torch.Generator().manual_seed(8)
    dev = hip_ctx.device
    spec = (torch.randn(2, 4, 16, 64, generator=g) * 3).to(dev)
    w1 = (torch.randn(48, 4, 1, 1, generator=g) * 0.5).to(dev); b1 = (torch.randn(48, generator=g) * 0.3).to(dev)
    w3 = torch.randn(48, 48, 3, 3, generator=g) / np.sqrt(9 * 48); b3 = (torch.randn(48, generator=g) * 0.1).to(dev)
    packed, un = pack_conv3x3(w3.numpy())
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    mid = hip_ctx.conv1x1_small(spec, w1, b1, relu=True)
    ref = hip_ctx.conv3x3_f16x3(mid, wp, b3, 48, un, relu=True)
    got = hip_ctx.conv3x3_f16x3_first(spec, w1, b1, wp, b3, 48, un, relu=True)
    assert torch.equal(got, ref)
    spec_net = TfcTdfSpec()
    w = synth_weights(spec_net, seed=0)
    net = TfcTdfNet(w, spec_net, hip=hip_ctx).to(dev).eval()
    x = (torch.randn(1, 4, 32, 3072, generator=g) * 2.0).to(dev)
    a = net.forward_tf(x)
    hip_ctx.fuse_first_conv = False
    try:
        b = net.forward_tf(x)            # the unfused first 3x3 conv now runs the 8-channel-stage kernel: other summation order
        hip_ctx.conv_wide = False
        c = net.forward_tf(x)            # every conv on the 16-channel-stage kernel, the fused loader's order
        hip_ctx.fuse_first_conv = True
        d = net.forward_tf(x)
    finally:
        hip_ctx.fuse_first_conv, hip_ctx.conv_wide = True, True
    assert torch.equal(c, d)
    peak = float(a.abs().max())
    assert float((a - b).abs().max()) / peak < 2e-6 and float((a - c).abs().max()) / peak < 2e-6
