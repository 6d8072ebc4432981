"""TFC-TDF U-Net (PyTorch-ROCm, folded BN, T-major layout) against a float64 CPU evaluation, block by block."""
import numpy as np
import pytest
import torch

from audio_cut_amd.separation.tfc_tdf import TfcTdfNet, TfcTdfSpec, synth_weights
from audio_cut_amd.testing import signals
from oracle import chunking as OC
from oracle.separator import mdx_stft, unet_forward

import unet_torch

pytestmark = pytest.mark.gpu


def test_unet_blocks_and_output_vs_float64(hip_ctx):
    spec = TfcTdfSpec()
    w = synth_weights(spec, seed=0)
    mix = signals.c2_song(10.0, seed=4)
    batch, _, _ = OC.mdx_windows(mix)
    x = mdx_stft(batch[:1]).contiguous()                    # [1, 4, 3072, 256]: a full item (the deep levels need T % 256 == 0 to tile)
    netg = TfcTdfNet(w, spec, hip=hip_ctx).to(hip_ctx.device).eval()    # product: one HIP kernel per layer
    got, ref = {}, {}
    netg.block_tap = lambda name, t: got.__setitem__(name, t.detach().double().cpu())
    try:
        yg = netg(x.to(hip_ctx.device)).double().cpu()
    finally:
        netg.block_tap = None
    y64 = unet_torch.forward(netg, x.double(), tap=lambda name, t: ref.__setitem__(name, t.detach().clone()))
    errs = {name: float((got[name] - r).abs().max() / r.abs().max()) for name, r in ref.items()}
    print("block errors vs float64 (relative to block peak):", {k: f"{v:.1e}" for k, v in errs.items()})
    assert set(errs) == {f"enc{i}" for i in range(5)} | {f"dec{i}" for i in range(5)} | {"bottleneck"}
    for name, err in errs.items():
        assert err < 1e-3, (name, err)                      # every level, relative to that block's own peak
    out_err = float((yg - y64).abs().max() / y64.abs().max())
    print(f"output error vs float64: {out_err:.2e}")
    assert out_err < 1e-5           # float32-class: the split-f16 convs keep 22 mantissa bits
    # the same folded graph in plain float32 through MIOpen / rocBLAS (tests/unet_torch.py): what "float32" itself delivers
    y32 = unet_torch.forward(netg, x.to(hip_ctx.device)).double().cpu()
    lib_err = float((y32 - y64).abs().max() / y64.abs().max())
    print(f"MIOpen/rocBLAS float32 error vs float64: {lib_err:.2e}")
    assert lib_err < 1e-5 and out_err < 4 * lib_err + 1e-6
    # the un-fused oracle graph (conv -> BN -> ReLU as separate float32 ops) agrees with the folded net
    yo = unet_forward(x, w).double()
    assert float((yo - y64).abs().max() / y64.abs().max()) < 1e-5
    # ONNX-shaped entry point == T-major entry point
    ytf = netg.forward_tf(x.transpose(-1, -2).contiguous().to(hip_ctx.device)).transpose(-1, -2).double().cpu()
    assert torch.equal(ytf, yg)
    # no library path: a net without the HIP context, or a shape the kernels cannot tile, raises
    from audio_cut_amd._native import NativeError
    with pytest.raises(NativeError):
        TfcTdfNet(w, spec).forward_tf(x.transpose(-1, -2).contiguous())
    with pytest.raises(NativeError):
        netg.forward_tf(torch.zeros(1, 4, 36, 3072, device=hip_ctx.device))


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
        hip_ctx.tdf_linear_f16x3(torch.zeros(1, 3, 5, 32, device=dev), wp, 96, sc, sh, 1.0)     # C % 16, T % 8: refused, not mis-tiled


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
    """ac_conv3x3_f16x3_first == ac_conv1x1_small followed by ac_conv3x3_f16x3_s8, bit for bit (same float32 FMA order for the
    generated channels, same zero padding of the conv input, the same kernel behind both), and the whole net is unchanged by the
    fusion."""
    from audio_cut_amd.separation.conv_pack import pack_conv3x3_w96
    g = torch.Generator().manual_seed(8)
    dev = hip_ctx.device
    spec = (torch.randn(2, 4, 16, 64, generator=g) * 3).to(dev)
    w1 = (torch.randn(48, 4, 1, 1, generator=g) * 0.5).to(dev); b1 = (torch.randn(48, generator=g) * 0.3).to(dev)
    w3 = torch.randn(48, 48, 3, 3, generator=g) / np.sqrt(9 * 48); b3 = (torch.randn(48, generator=g) * 0.1).to(dev)
    packed, un = pack_conv3x3_w96(w3.numpy(), 48)
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    mid = hip_ctx.conv1x1_small(spec, w1, b1, relu=True)
    ref = hip_ctx.conv3x3_f16x3_s8(mid, wp, b3, 48, un, relu=True)
    got = hip_ctx.conv3x3_f16x3_first(spec, w1, b1, wp, b3, 48, un, relu=True)
    assert torch.equal(got, ref)
    # with the per-row activation scale: the fused kernel scales by the BOUND of the generated tensor, the unfused pair by its
    # measured maximum - the same values up to the float16 low parts' last bit
    amax_spec = _blk_amax(spec)
    gain = float(w1.abs().sum(dim=(1, 2, 3)).max()); offs = float(b1.abs().max())
    oa = torch.zeros((2, 16), device=dev)
    got2 = hip_ctx.conv3x3_f16x3_first(spec, w1, b1, wp, b3, 48, un, relu=True, spec_amax=amax_spec, amax_gain=gain, amax_offs=offs, out_amax=oa)
    assert bool((_blk_amax(mid) <= amax_spec * gain + offs).all())
    ref2 = hip_ctx.conv3x3_f16x3_s8(mid, wp, b3, 48, un, relu=True, in_amax=_blk_amax(mid))
    assert float((got2 - ref2).abs().max() / ref2.abs().max()) < 1e-6
    assert torch.equal(oa, _blk_amax(got2))
    # a burst's leakage into digital silence (rows falling by 1e-3 each, then exact zeros), bias-free like the synthetic net: the
    # fused kernel's row-exact path holds every element to float32-class accuracy relative to its OWN sum |x||w|
    import torch.nn.functional as F
    ramp = torch.zeros(16, dtype=torch.float64); ramp[:3] = 1.0; ramp[3:12] = 1e-3 ** torch.arange(1, 10, dtype=torch.float64)
    spec_c = (spec.cpu() * ramp.float().view(1, 1, -1, 1)).to(dev)
    zb1 = torch.zeros(48, device=dev); zb3 = torch.zeros(48, device=dev)
    got3 = hip_ctx.conv3x3_f16x3_first(spec_c, w1, zb1, wp, zb3, 48, un, relu=True, spec_amax=_blk_amax(spec_c), amax_gain=gain, amax_offs=0.0)
    mid64 = F.relu(F.conv2d(spec_c.double().cpu(), w1.double().cpu()))
    ref64 = F.relu(F.conv2d(mid64, w3.double(), padding=1)); sc64 = F.conv2d(mid64.abs(), w3.abs().double(), padding=1)
    mid32 = F.relu(F.conv2d(spec_c.cpu(), w1.cpu()))
    e32 = _elementwise_error(F.relu(F.conv2d(mid32, w3, padding=1)), ref64, sc64)
    e16 = _elementwise_error(got3.cpu(), ref64, sc64)
    assert e16 <= 4.0 * e32 + 2.0 ** -24, (e16, e32)


def _blk_amax(t: torch.Tensor) -> torch.Tensor:
    """max |t| per item and row of the time axis (dim 2): the "amax" tensors of include/audiocut_hip.h."""
    return t.abs().amax(dim=(1, 3)).contiguous()


def _elementwise_error(y: torch.Tensor, ref64: torch.Tensor, scale64: torch.Tensor) -> float:
    """max over elements of |y - ref| / (sum |x| |w| + |b|): the backward-error scale of each output element itself, so a
    quiet region is held to the same relative accuracy as a loud one (unlike an error relative to the tensor's peak)."""
    return float(((y.double() - ref64).abs() / scale64.clamp_min(1e-300)).max())


def test_split_f16_kernels_are_float32_class_at_every_magnitude(hip_ctx):
    """The per-item activation scale (include/audiocut_hip.h, "amax"): every split-float16 kernel is fed inputs from 1e-6 to 1e6
    (past the float16 range: nothing saturates), uniform and with a 1e-8 decay along the time axis inside one item (the scale is
    local in time: the rows one accumulation reads; the two "cliff" cases - rows falling by 1e-2 resp. 1e-9 EACH into exact zeros,
    the leakage next to a burst in digital silence - take the conv kernels' row-exact path, ac_common.h), and its
    ELEMENT-WISE error against float64 must stay within 4x of what a true float32 evaluation (PyTorch CPU float32) of the same
    layer delivers.  Without the scale the 1e-6 and decaying cases are 2-4 orders of magnitude worse (the low float16 part is
    a subnormal) and the 1e6 case clips.  out_amax must be the exact per-item maximum of the result."""
    import torch.nn.functional as F
    from audio_cut_amd.separation.conv_pack import pack_conv3x3, pack_conv3x3_w96, pack_linear
    g = torch.Generator().manual_seed(11)
    dev = hip_ctx.device
    mags = (1e-6, 1e-3, 1.0, 3.0e2, 6.0e4, 1.0e6)
    worst = {}
    failures = []

    def check(name, run, ref_fn, x, has_bias_scale):
        """run(x_dev, in_amax, out_amax) -> y ; ref_fn(x, dtype) -> (y, scale) on the CPU in `dtype`."""
        for mag in mags:
            for decay in (False, True, "cliff", "cliff9"):
                if decay in ("cliff", "cliff9") and not (1e-3 <= mag <= 3.0e2):
                    continue            # the cliffs already span 28 decades: keep their products clear of float32 subnormals
                xm = x * mag
                if decay is True:       # item 0 keeps its level, item 1 falls by 1e-8 along H (time): a decay into silence inside one item
                    ramp = torch.logspace(0, -8, x.shape[2], dtype=torch.float32).view(1, 1, -1, 1)
                    xm = torch.cat([xm[:1], xm[1:] * ramp], dim=0)
                elif decay:             # the leakage of a burst into digital silence: rows fall by 1e-2 (1e-9) each, then exact zeros
                    ramp = torch.zeros(x.shape[2], dtype=torch.float64)
                    ramp[:5] = 1.0
                    n, step = (14, 1e-2) if decay == "cliff" else (3, 1e-9)
                    ramp[5:5 + n] = step ** torch.arange(1, n + 1, dtype=torch.float64)
                    xm = torch.cat([xm[:1], xm[1:] * ramp.float().view(1, 1, -1, 1)], dim=0)
                ia = _blk_amax(xm).to(dev)
                y = run(xm.to(dev), ia, None)
                oa = torch.zeros((x.shape[0], y.shape[2]), device=dev)
                assert torch.equal(run(xm.to(dev), ia, oa), y) and torch.equal(oa, _blk_amax(y)), (name, mag)
                ref64, scale64 = ref_fn(xm.double(), torch.float64)
                y32, _ = ref_fn(xm, torch.float32)
                e16 = _elementwise_error(y.cpu(), ref64, scale64)
                e32 = _elementwise_error(y32, ref64, scale64)
                worst[name] = max(worst.get(name, 0.0), e16 / max(e32, 1e-12))
                if not e16 <= 4.0 * e32 + 2.0 ** -24:
                    failures.append((name, mag, decay, e16, e32))

    # --- 3x3 convs: 16-channel-stage kernel, 96-channel and 48-channel 8-channel-stage kernels
    for ci, co, h, w_, kind in ((48, 48, 64, 32, "plain"), (96, 96, 64, 32, "w96"), (48, 48, 64, 32, "s8"), (144, 144, 64, 32, "s8")):
        x = torch.randn(2, ci, h, w_, generator=g)
        wt = torch.randn(co, ci, 3, 3, generator=g) / np.sqrt(9 * ci)
        bias = torch.zeros(co)              # bias-free like the synthetic net's convs: the output scales with the input
        if kind == "plain":
            packed, un = pack_conv3x3(wt.numpy()); fn = hip_ctx.conv3x3_f16x3
        else:
            packed, un = pack_conv3x3_w96(wt.numpy(), 96 if kind == "w96" else 48)
            fn = hip_ctx.conv3x3_f16x3_w96 if kind == "w96" else hip_ctx.conv3x3_f16x3_s8
        wp = torch.from_numpy(packed.view(np.int16)).to(dev); bd = bias.to(dev)

        def ref(xx, dt, wt=wt):
            y = F.relu(F.conv2d(xx, wt.to(dt), None, padding=1))
            return y.double(), F.conv2d(xx.abs().double(), wt.abs().double(), None, padding=1)
        check(f"conv_{kind}_{ci}", lambda xd, ia, oa, fn=fn, wp=wp, bd=bd, co=co, un=un: fn(xd, wp, bd, co, un, relu=True, in_amax=ia, out_amax=oa),
              ref, x, False)
    # --- TDF layer (GEMM over the last axis + affine + ReLU + residual)
    for (c, t, k, n) in ((48, 32, 384, 96), (48, 32, 96, 384)):
        x = torch.randn(2, c, t, k, generator=g)
        wt = torch.randn(n, k, generator=g) / np.sqrt(k)
        sc = torch.rand(c, generator=g) + 0.5; sh = torch.zeros(c)
        packed, un = pack_linear(wt.numpy())
        wp = torch.from_numpy(packed.view(np.int16)).to(dev)

        def ref(xx, dt, wt=wt, sc=sc):
            y = F.relu(F.linear(xx, wt.to(dt)) * sc.to(dt).view(1, -1, 1, 1))
            return y.double(), F.linear(xx.abs().double(), wt.abs().double()) * sc.double().view(1, -1, 1, 1)
        check(f"tdf_{k}x{n}", lambda xd, ia, oa, wp=wp, n=n, sc=sc, sh=sh, un=un: hip_ctx.tdf_linear_f16x3(xd, wp, n, sc.to(dev), sh.to(dev), un, in_amax=ia, out_amax=oa),
              ref, x, False)
    # --- 2x2 down / up sampling
    c, h, w_ = 48, 64, 32
    x = torch.randn(2, c, h, w_, generator=g)
    wd = torch.randn(c + 48, c, 2, 2, generator=g) / np.sqrt(4 * c)
    packed, un = pack_linear(wd.numpy().reshape(c + 48, -1), bn=96)
    wpd = torch.from_numpy(packed.view(np.int16)).to(dev); zb = torch.zeros(c + 48, device=dev)

    def ref_dn(xx, dt):
        return F.relu(F.conv2d(xx, wd.to(dt), None, stride=2)).double(), F.conv2d(xx.abs().double(), wd.abs().double(), None, stride=2)
    check("down", lambda xd, ia, oa: hip_ctx.down2x_f16x3(xd, wpd, zb, c + 48, un, in_amax=ia, out_amax=oa), ref_dn, x, False)
    c2 = 96
    x = torch.randn(2, c2, 32, 64, generator=g)          # W >= 64: a wave's 64 pixels lie in at most two rows (the net's narrowest is 96)
    wu = torch.randn(c2, c2 - 48, 2, 2, generator=g) / np.sqrt(c2)
    packed, un2 = pack_linear(wu.numpy().transpose(1, 2, 3, 0).reshape(-1, c2), bn=96)
    wpu = torch.from_numpy(packed.view(np.int16)).to(dev); zb2 = torch.zeros(c2 - 48, device=dev)

    def ref_up(xx, dt):
        return F.relu(F.conv_transpose2d(xx, wu.to(dt), None, stride=2)).double(), F.conv_transpose2d(xx.abs().double(), wu.abs().double(), None, stride=2)
    check("up", lambda xd, ia, oa: hip_ctx.up2x_f16x3(xd, wpu, zb2, c2 - 48, un2, in_amax=ia, out_amax=oa), ref_up, x, False)
    print("worst element-wise error relative to a float32 evaluation:", {k: f"{v:.2f}x" for k, v in worst.items()})
    assert not failures, failures
    # and what the scale buys: the same 1e-6 input WITHOUT it is orders of magnitude off element-wise
    x = torch.randn(2, 48, 16, 64, generator=g) * 1e-6
    wt = torch.randn(48, 48, 3, 3, generator=g) / np.sqrt(9 * 48)
    packed, un = pack_conv3x3_w96(wt.numpy(), 48)
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    ref64 = F.relu(F.conv2d(x.double(), wt.double(), None, padding=1)); sc64 = F.conv2d(x.abs().double(), wt.abs().double(), None, padding=1)
    y0 = hip_ctx.conv3x3_f16x3_s8(x.to(dev), wp, torch.zeros(48, device=dev), 48, un, relu=True).cpu()
    y1 = hip_ctx.conv3x3_f16x3_s8(x.to(dev), wp, torch.zeros(48, device=dev), 48, un, relu=True, in_amax=_blk_amax(x).to(dev)).cpu()
    assert _elementwise_error(y0, ref64, sc64) > 100 * _elementwise_error(y1, ref64, sc64)


def test_tdf_small_fused_kernel(hip_ctx):
    """ac_tdf_small_fused (both narrow TDF layers + residual of the deep levels, exact float32 on v_mfma_f32_16x16x4_f32)
    against float64 on the three shapes the U-Net has (F = 384 / 192 / 96, bottleneck 48 / 24 / 12), asymmetric weights, and
    the per-item output maximum."""
    import torch.nn.functional as F
    from audio_cut_amd.separation.conv_pack import pack_tdf_small
    g = torch.Generator().manual_seed(9)
    dev = hip_ctx.device
    for (b, c, t, f) in ((2, 192, 32, 384), (2, 240, 16, 192), (3, 288, 8, 96), (1, 4, 8, 32)):
        hd = f // 8
        x = torch.randn(b, c, t, f, generator=g) * 2
        w1 = torch.randn(hd, f, generator=g) / np.sqrt(f); w2 = torch.randn(f, hd, generator=g) / np.sqrt(hd)
        s1 = torch.rand(c, generator=g) + 0.5; b1 = torch.randn(c, generator=g) * 0.3
        s2 = torch.rand(c, generator=g) + 0.5; b2 = torch.randn(c, generator=g) * 0.3
        p1, p2 = pack_tdf_small(w1.numpy(), w2.numpy())
        oa = torch.zeros((b, t), device=dev)
        y = hip_ctx.tdf_small_fused(x.to(dev), torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev), hd, s1.to(dev), b1.to(dev),
                                    s2.to(dev), b2.to(dev), out_amax=oa)
        v = lambda a: a.double().view(1, -1, 1, 1)
        xd = x.double()
        h = F.relu(F.linear(xd, w1.double()) * v(s1) + v(b1))
        ref = xd + F.relu(F.linear(h, w2.double()) * v(s2) + v(b2))
        assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 1e-6, (b, c, t, f)
        assert torch.equal(oa, _blk_amax(y))
    with pytest.raises(Exception):
        hip_ctx.tdf_small_fused(torch.zeros(1, 3, 5, 96, device=dev), torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev), 12,
                                s1.to(dev), b1.to(dev), s2.to(dev), b2.to(dev))           # rows % 32 != 0: refused


def test_product_forward_through_the_onnx_reader(hip_ctx, tmp_path):
    """SURVEY 8(f) row 3 as far as it can go offline: the full-size network written as an ONNX file the way an exporter writes it
    (anonymous `onnx::w_N` initializer names, a mix of raw_data / float_data payloads, initializers in shuffled order, MatMul TDFs)
    and loaded the way the reference's deployment loads `Kim_Vocal_1.onnx` - `MDX23HipBackend(model_dir=...)`, `backends.py:222-255`
    -> `separation/onnx_weights.py` - must drive the HIP kernels to the SAME bits as the dict-weights network, and to float32-class
    accuracy against the float64 evaluation."""
    from pathlib import Path
    from tests.onnx_writer import write_tfc_tdf_onnx
    from audio_cut_amd.separation.backends import MDX23HipBackend
    spec = TfcTdfSpec()
    w = synth_weights(spec, seed=3)
    rng = np.random.default_rng(7)
    for k in list(w):                                       # informative biases / running means (synthetic ones are zeros)
        if k.endswith("running_mean") or (k.endswith(".bias") and "bn" not in k):
            w[k] = (rng.standard_normal(w[k].shape) * 0.05).astype(np.float32)
    model_dir = Path(tmp_path)
    write_tfc_tdf_onnx(model_dir / "Kim_Vocal_1.onnx", w, spec, fold_conv_bn=False, raw=False, shuffle_seed=11)
    assert (model_dir / "Kim_Vocal_1.onnx").stat().st_size > 60e6            # the real file is 66.8 MB
    via_file = MDX23HipBackend(model_dir=model_dir, ctx=hip_ctx); via_file.load_model()
    via_dict = MDX23HipBackend(weights=w, ctx=hip_ctx); via_dict.load_model()
    mix = signals.c2_song(10.0, seed=5)
    batch, _, _ = OC.mdx_windows(mix)
    x = mdx_stft(batch[:1]).contiguous().to(hip_ctx.device)
    y_file = via_file.net(x); y_dict = via_dict.net(x)
    assert torch.equal(y_file, y_dict)                      # same bits: the reader hands the kernels exactly the writer's tensors
    y64 = unet_torch.forward(via_dict.net, x.double().cpu())
    err = float((y_file.double().cpu() - y64).abs().max() / y64.abs().max())
    print(f"forward through the ONNX reader vs float64: {err:.2e}")
    assert err < 1e-5
