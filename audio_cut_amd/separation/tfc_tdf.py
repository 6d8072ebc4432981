"""MDX23 TFC-TDF v2 U-Net as PyTorch-ROCm conv stacks.

The reference runs `Kim_Vocal_1.onnx` through ONNX Runtime (`src/audio_cut/separation/backends.py:358`,
`config/expert.yaml:22`); in/out `[B,4,3072,256]` f32 (`tests/sanity/ort_mdx23_cuda_sanity.py:38`).
The graph is the KUIELab TFC-TDF v2 net (first 1x1 conv, 5 encoder blocks with stride-2 2x2
down-sampling, bottleneck, 5 decoder blocks with 2x2 transposed-conv up-sampling and
*multiplicative* skips, final 1x1 conv; every block = 3 x [3x3 conv, BN, ReLU] + a bias-free
frequency-axis bottleneck MLP (f -> f/8 -> f) with BN+ReLU, residual-added).  With growth g=48
this is 16.7 M parameters = the 66.8 MB ONNX file.

Here batch-norm (inference mode) is folded into the preceding conv / linear and every layer is ONE
hand-written MFMA kernel of libaudiocut_hip.so with bias / affine / ReLU / residual / skip fused into its
epilogue; activations are kept in float32 (the reference's precision).  All sub-windows of a track are
batched through one forward (`MDX23HipBackend.separate_track`) instead of the reference's one-chunk-at-a-time loop.

Weights: a plain name -> ndarray dict, read from `Kim_Vocal_1.onnx` by `separation/onnx_weights.py` or, offline,
made by `testing/synth_unet.py` (seeded tensors of exactly this architecture); the same dict drives the CPU oracle
in the tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

Weights = Dict[str, np.ndarray]


@dataclass(frozen=True)
class TfcTdfSpec:
    dim_c: int = 4
    dim_f: int = 3072
    dim_t: int = 256
    n_levels: int = 5          # L // 2 with L = 11 (backends.py:263)
    l: int = 3
    g: int = 48
    k: int = 3
    bn: int = 8
    bn_eps: float = 1e-5

    def channels(self, level: int) -> int:
        return self.g * (level + 1)

    def flops_per_item(self) -> float:
        """2*MACs of every Conv/ConvTranspose/Linear for one [4, dim_f, dim_t] item."""
        t, f = self.dim_t, self.dim_f
        total = 2.0 * self.dim_c * self.g * t * f * 2          # first + final 1x1
        def block(c, tt, ff):
            conv = 2.0 * c * c * self.k * self.k * tt * ff * self.l
            tdf = 2.0 * (c * tt) * ff * (ff // self.bn) * 2
            return conv + tdf
        for i in range(self.n_levels):
            c = self.channels(i)
            total += 2 * block(c, t, f)                          # encoder + decoder block at this level
            total += 2 * (2.0 * c * (c + self.g) * 4 * (t // 2) * (f // 2))   # ds + us
            t //= 2; f //= 2
        total += block(self.channels(self.n_levels), t, f)
        return total

    def param_count(self) -> int:
        n = self.dim_c * self.g * 2 + self.g + self.dim_c
        f = self.dim_f
        def block(c, ff):
            return self.l * (c * c * self.k * self.k + c) + 2 * ff * (ff // self.bn)
        for i in range(self.n_levels):
            c = self.channels(i)
            n += 2 * block(c, f) + 2 * (c * (c + self.g) * 4) + (c + self.g) + c
            f //= 2
        n += block(self.channels(self.n_levels), f)
        return n


def synth_weights(spec: TfcTdfSpec = TfcTdfSpec(), seed: int = 0, calib_t: int = 128) -> Weights:
    """Seeded synthetic weights of this architecture (`audio_cut_amd/testing/synth_unet.py`: test / bench data; the real
    `Kim_Vocal_1.onnx` cannot be fetched offline)."""
    from ..testing.synth_unet import synth_weights as _make
    return _make(spec, seed, calib_t)


# ---------------------------------------------------------------------------
# inference module with folded batch-norm
# ---------------------------------------------------------------------------

def _fold(weight: np.ndarray, bias: Optional[np.ndarray], w: Weights, bn_name: str, eps: float, out_axis: int):
    """conv/linear followed by eval-mode BN -> (scaled weight, shift); float64 algebra, rounded once."""
    gamma = w[bn_name + ".weight"].astype(np.float64)
    beta = w[bn_name + ".bias"].astype(np.float64)
    mean = w[bn_name + ".running_mean"].astype(np.float64)
    var = w[bn_name + ".running_var"].astype(np.float64)
    s = gamma / np.sqrt(var + eps)
    shape = [1] * weight.ndim
    shape[out_axis] = -1
    wf = weight.astype(np.float64) * s.reshape(shape)
    b0 = bias.astype(np.float64) if bias is not None else 0.0
    bf = (b0 - mean) * s + beta
    return wf.astype(np.float32), bf.astype(np.float32)


AMAX_ROWS = 1       # AC_AMAX_ROWS (csrc/ac_common.h): one activation maximum per item and row of the time axis


class _AmaxTape:
    """Per-forward scratch of activation maxima (include/audiocut_hip.h, "amax"): one zeroed float32 [B, T] array per tensor
    that a split-float16 kernel will read; the producing kernel reduces max |x| per item and time row into it, the consumer
    derives its time-local power-of-two activation scale from exactly the rows one accumulation reads.  One allocation + one
    memset per forward."""

    def __init__(self, batch: int, t_full: int, n_tensors: int, device: torch.device):
        self._batch = batch
        self._buf = torch.zeros(n_tensors * batch * (t_full // AMAX_ROWS), dtype=torch.float32, device=device)
        self._next = 0

    def new(self, t: int) -> torch.Tensor:
        n = self._batch * (t // AMAX_ROWS)
        if t % AMAX_ROWS or self._next + n > self._buf.numel():
            raise RuntimeError("amax tape exhausted or T % 8 != 0")
        out = self._buf[self._next:self._next + n].view(self._batch, t // AMAX_ROWS)
        self._next += n
        return out


class _Block(nn.Module):
    """One TFC-TDF block: l x [3x3 conv + folded BN + ReLU], then x + TDF(x) (two bias-free Linears over F with BN + ReLU)."""

    def __init__(self, w: Weights, prefix: str, spec: TfcTdfSpec):
        super().__init__()
        self.l = spec.l
        self.pad = spec.k // 2
        for j in range(spec.l):
            wf, bf = _fold(w[f"{prefix}.tfc.{j}.conv.weight"], w[f"{prefix}.tfc.{j}.conv.bias"], w,
                           f"{prefix}.tfc.{j}.bn", spec.bn_eps, 0)
            self.register_buffer(f"cw{j}", torch.from_numpy(wf))
            self.register_buffer(f"cb{j}", torch.from_numpy(bf))
        # TDF linears act on the last (frequency) axis, BN on the channel axis: only the BN *scale*
        # commutes with the bias-free linear, so TDF keeps a per-channel affine after the matmul.
        for j in range(2):
            self.register_buffer(f"lw{j}", torch.from_numpy(np.ascontiguousarray(w[f"{prefix}.tdf.{j}.weight"])))
            name = f"{prefix}.tdf.{j}.bn"
            s = w[name + ".weight"].astype(np.float64) / np.sqrt(w[name + ".running_var"].astype(np.float64) + spec.bn_eps)
            sh = w[name + ".bias"].astype(np.float64) - w[name + ".running_mean"].astype(np.float64) * s
            self.register_buffer(f"ls{j}", torch.from_numpy(s.astype(np.float32)).view(1, -1, 1, 1))
            self.register_buffer(f"lb{j}", torch.from_numpy(sh.astype(np.float32)).view(1, -1, 1, 1))

    def pack_for_hip(self) -> None:
        """Kernel-ready copies of the folded weights: f16 hi/lo MFMA fragments for the 3x3 convs (ac_conv3x3_f16x3_w96 / _s8,
        ac_conv3x3_f16x3_first for the fused first one) and for the wide TDF layers (ac_tdf_linear_f16x3); float32 fragments
        for the narrow TDF pairs of the deep levels (ac_tdf_small_fused)."""
        from .conv_pack import conv3x3_wide_tileable, pack_conv3x3_w96, pack_linear, pack_tdf_small
        w0 = self.lw0.detach().cpu().numpy(); w1 = self.lw1.detach().cpu().numpy()
        dev = self.lw0.device
        self._l_unscale = [None, None]
        if w0.shape[0] % 96 == 0 and w0.shape[1] % 32 == 0 and w1.shape[0] % 96 == 0 and w1.shape[1] % 32 == 0:
            for j, w in enumerate((w0, w1)):
                packed, unscale = pack_linear(w)
                self._l_unscale[j] = unscale
                self.register_buffer(f"lwp{j}", torch.from_numpy(packed.view(np.int16)).to(dev))
        elif w0.shape[1] % 16 == 0 and w0.shape[0] <= 48:
            p1, p2 = pack_tdf_small(w0, w1)
            self.register_buffer("lws0", torch.from_numpy(p1).to(dev))
            self.register_buffer("lws1", torch.from_numpy(p2).to(dev))
        self._w_unscale = []
        for j in range(self.l):
            w = getattr(self, f"cw{j}").detach().cpu().numpy()
            if w.shape[0] % 48 or w.shape[1] % 16:
                self._w_unscale.append(None)
                continue
            # 96 output channels per workgroup where the shape allows, else 48 with three workgroups per CU (the graph's first 3x3
            # conv, fused with the 1x1 in front of it, takes the 48-channel layout of its block's conv 0)
            if conv3x3_wide_tileable(w.shape[0], w.shape[1]):
                wide, unscale = pack_conv3x3_w96(w, 96)
                self.register_buffer(f"cwq{j}", torch.from_numpy(wide.view(np.int16)).to(dev))
            else:
                narrow, unscale = pack_conv3x3_w96(w, 48)
                self.register_buffer(f"cws{j}", torch.from_numpy(narrow.view(np.int16)).to(dev))
            self._w_unscale.append(unscale)

    def _conv(self, x: torch.Tensor, ax: torch.Tensor, j: int, hip, tape: _AmaxTape, probe):
        """3x3 conv + bias + ReLU: one fused kernel on the f16 matrix cores (3-term hi/lo split, float32-class accuracy)."""
        from .._native import NativeError
        if x.shape[2] % 8 or x.shape[3] % 32 or not (hasattr(self, f"cwq{j}") or hasattr(self, f"cws{j}")):
            raise NativeError(f"3x3 conv of shape {tuple(x.shape)} is not tileable by the HIP kernels (C % 48, C_in % 16, H % 8, W % 32)")
        if probe is not None:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        ay = tape.new(x.shape[2])
        if hasattr(self, f"cwq{j}"):
            y = hip.conv3x3_f16x3_w96(x, getattr(self, f"cwq{j}"), getattr(self, f"cb{j}"), x.shape[1], self._w_unscale[j], relu=True,
                                      in_amax=ax, out_amax=ay)
        else:
            y = hip.conv3x3_f16x3_s8(x, getattr(self, f"cws{j}"), getattr(self, f"cb{j}"), x.shape[1], self._w_unscale[j], relu=True,
                                     in_amax=ax, out_amax=ay)
        if probe is not None:
            e1.record()
            probe.append((e0, e1, 2.0 * x.shape[0] * x.shape[1] * x.shape[1] * 9 * x.shape[2] * x.shape[3]))
        return y, ay

    def _tdf(self, x: torch.Tensor, ax: torch.Tensor, hip, tape: _AmaxTape):
        """x + relu(bn(linear(relu(bn(linear(x)))))) over the frequency axis: two fused GEMM kernels on the f16 matrix cores
        (ac_tdf_linear_f16x3: + per-channel affine + ReLU (+ residual)), or one exact-float32 kernel for the narrow pairs of the
        deep levels (ac_tdf_small_fused)."""
        from .._native import NativeError
        rows = x.shape[0] * x.shape[1] * x.shape[2]
        ay = tape.new(x.shape[2])
        if hasattr(self, "lwp0") and x.shape[1] % 16 == 0 and x.shape[2] % 8 == 0:
            ah = tape.new(x.shape[2])
            h = hip.tdf_linear_f16x3(x, self.lwp0, self.lw0.shape[0], self.ls0.view(-1), self.lb0.view(-1), self._l_unscale[0],
                                     in_amax=ax, out_amax=ah)
            y = hip.tdf_linear_f16x3(h, self.lwp1, self.lw1.shape[0], self.ls1.view(-1), self.lb1.view(-1), self._l_unscale[1], resid=x,
                                     in_amax=ah, out_amax=ay)
            return y, ay
        if hasattr(self, "lws0") and rows % 32 == 0 and (x.shape[1] * x.shape[2]) % 32 == 0:
            y = hip.tdf_small_fused(x, self.lws0, self.lws1, self.lw0.shape[0], self.ls0.view(-1), self.lb0.view(-1),
                                    self.ls1.view(-1), self.lb1.view(-1), out_amax=ay)
            return y, ay
        raise NativeError(f"TDF of shape {tuple(x.shape)} -> {self.lw0.shape[0]} is not tileable by the HIP kernels")

    def forward_hip(self, x: torch.Tensor, ax: torch.Tensor, hip, tape: _AmaxTape, probe=None, first=None):
        """`first` = (w1, b1, gain, offs) of the graph's first 1x1 convolution: fused into this block's first 3x3 conv (x is the
        spectrogram, ax its per-item max; gain / offs bound the generated tensor)."""
        start = 0
        if first is not None:
            if probe is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            spec = x
            ay = tape.new(x.shape[2])
            if not hasattr(self, "cws0"):
                from .._native import NativeError
                raise NativeError("the fused first conv needs a 48-output-channel-tileable 3x3 conv behind the 1x1 (ac_conv3x3_f16x3_first)")
            x = hip.conv3x3_f16x3_first(spec, first[0], first[1], self.cws0, self.cb0, self.cw0.shape[0], self._w_unscale[0], relu=True,
                                        spec_amax=ax, amax_gain=first[2], amax_offs=first[3], out_amax=ay)
            ax = ay
            if probe is not None:
                e1.record()
                c = self.cw0.shape[0]
                probe.append((e0, e1, 2.0 * spec.shape[0] * c * c * 9 * spec.shape[2] * spec.shape[3]))
            start = 1
        for j in range(start, self.l):
            x, ax = self._conv(x, ax, j, hip, tape, probe)
        return self._tdf(x, ax, hip, tape)


class TfcTdfNet(nn.Module):
    """Inference-only TFC-TDF v2 with folded BN on the HIP kernels.  Input/output `[B, 4, dim_f, T]` float32.

    Every layer is a kernel of libaudiocut_hip.so; a shape the kernels cannot tile raises `NativeError` - there is no MIOpen /
    rocBLAS / CPU path (the float32 / float64 PyTorch evaluation of the same folded weights used as the reference in the tests
    lives in `tests/unet_torch.py`)."""

    def __init__(self, weights: Weights, spec: TfcTdfSpec = TfcTdfSpec(), hip=None):
        super().__init__()
        self.spec = spec
        self.hip = hip          # audio_cut_amd._native.Context
        self.conv_probe = None  # set to a list to collect (start event, end event, flops) per 3x3 conv launch (bench.py)
        self.block_tap = None   # callable(name, tensor) invoked with every block's output (tests compare block by block)
        w = weights
        wf, bf = _fold(w["first_conv.weight"], w["first_conv.bias"], w, "first_bn", spec.bn_eps, 0)
        self.register_buffer("first_w", torch.from_numpy(wf))
        self.register_buffer("first_b", torch.from_numpy(bf))
        # bound of the generated first-layer tensor: |relu(w1 x + b1)| <= max|x| * max_c sum_j |w1[c][j]| + max_c |b1[c]|
        self._first_gain = float(np.max(np.sum(np.abs(wf.reshape(wf.shape[0], -1).astype(np.float64)), axis=1)))
        self._first_offs = float(np.max(np.abs(bf.astype(np.float64)))) if bf.size else 0.0
        self.enc = nn.ModuleList(_Block(w, f"enc.{i}", spec) for i in range(spec.n_levels))
        self.dec = nn.ModuleList(_Block(w, f"dec.{i}", spec) for i in range(spec.n_levels))
        self.bottleneck = _Block(w, "bottleneck", spec)
        for i in range(spec.n_levels):
            wf, bf = _fold(w[f"ds.{i}.conv.weight"], w[f"ds.{i}.conv.bias"], w, f"ds.{i}.bn", spec.bn_eps, 0)
            self.register_buffer(f"ds_w{i}", torch.from_numpy(wf))
            self.register_buffer(f"ds_b{i}", torch.from_numpy(bf))
            wf, bf = _fold(w[f"us.{i}.conv.weight"], w[f"us.{i}.conv.bias"], w, f"us.{i}.bn", spec.bn_eps, 1)
            self.register_buffer(f"us_w{i}", torch.from_numpy(wf))
            self.register_buffer(f"us_b{i}", torch.from_numpy(bf))
        if hip is not None:
            from .conv_pack import pack_linear
            for blk in [*self.enc, *self.dec, self.bottleneck]:
                blk.pack_for_hip()
            self._rs_unscale = {}
            for i in range(spec.n_levels):
                dsw = getattr(self, f"ds_w{i}").numpy()                        # [co, ci, 2, 2] -> W[co][(ci, dy, dx)]
                packed, un = pack_linear(dsw.reshape(dsw.shape[0], -1), bn=96)
                self.register_buffer(f"ds_p{i}", torch.from_numpy(packed.view(np.int16)))
                self._rs_unscale[f"ds{i}"] = un
                usw = getattr(self, f"us_w{i}").numpy()                        # [ci, co, 2, 2] -> W[(co, dy, dx)][ci]
                packed, un = pack_linear(usw.transpose(1, 2, 3, 0).reshape(-1, usw.shape[0]), bn=96)
                self.register_buffer(f"us_p{i}", torch.from_numpy(packed.view(np.int16)))
                self._rs_unscale[f"us{i}"] = un
        self.register_buffer("final_w", torch.from_numpy(np.ascontiguousarray(w["final_conv.weight"])))
        self.register_buffer("final_b", torch.from_numpy(np.ascontiguousarray(w["final_conv.bias"])))

    def _down(self, x: torch.Tensor, ax: torch.Tensor, i: int, hip, tape: _AmaxTape):
        """2x2 / stride-2 conv + bias + ReLU: one fused MFMA kernel (space-to-depth gather in the loader)."""
        from .._native import NativeError
        b, c, h, w = x.shape
        if h % 2 or w % 4 or ((h // 2) * (w // 2)) % 128 or c % 8:
            raise NativeError(f"down-sampling of shape {tuple(x.shape)} is not tileable by ac_down2x_f16x3")
        ay = tape.new(h // 2)
        y = hip.down2x_f16x3(x, getattr(self, f"ds_p{i}"), getattr(self, f"ds_b{i}"), getattr(self, f"ds_w{i}").shape[0],
                             self._rs_unscale[f"ds{i}"], in_amax=ax, out_amax=ay)
        return y, ay

    def _up(self, x: torch.Tensor, ax: torch.Tensor, i: int, hip, skip: torch.Tensor, tape: _AmaxTape):
        """2x2 / stride-2 transposed conv + bias + ReLU + multiplicative skip: one fused MFMA kernel (depth-to-space in the epilogue)."""
        from .._native import NativeError
        b, c, h, w = x.shape
        c_out = getattr(self, f"us_w{i}").shape[1]
        if w % 4 or (h * w) % 128 or (4 * c_out) % 96:
            raise NativeError(f"up-sampling of shape {tuple(x.shape)} is not tileable by ac_up2x_f16x3")
        ay = tape.new(2 * h)
        y = hip.up2x_f16x3(x, getattr(self, f"us_p{i}"), getattr(self, f"us_b{i}"), c_out, self._rs_unscale[f"us{i}"], skip=skip,
                           in_amax=ax, out_amax=ay)
        return y, ay

    @torch.no_grad()
    def forward(self, spec_in: torch.Tensor) -> torch.Tensor:
        """ONNX-shaped call: `[B, 4, F, T]` in and out (what `session.run` takes at backends.py:358)."""
        return self.forward_tf(spec_in.transpose(-1, -2).contiguous()).transpose(-1, -2)

    @torch.no_grad()
    def forward_tf(self, spec_tf: torch.Tensor, spec_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """T-major call `[B, 4, T, F]` -> `[B, 4, T, F]`: the graph transposes right after its first 1x1
        conv and right before its last one (1x1 convs commute with the transpose), so the HIP STFT writes
        and the HIP iSTFT reads this layout directly and no transpose is ever materialised.
        `spec_amax` [B, T] = max |spec| per item and frame as ac_mdx_stft reduces it (computed here when absent)."""
        from .._native import NativeError
        hip = self.hip
        if hip is None or not spec_tf.is_cuda:
            raise NativeError("TfcTdfNet runs on the HIP kernels only: construct it with hip=Context and feed device tensors "
                              "(tests/unet_torch.py holds the PyTorch reference evaluation)")
        n = self.spec.n_levels
        b, c0, t, f = spec_tf.shape
        if not spec_tf.is_contiguous() or c0 > 4 or self.first_w.shape[0] > 64 or t % 8 or f % 32 or not hasattr(self.enc[0], "cws0"):
            raise NativeError(f"spectrogram of shape {tuple(spec_tf.shape)} is not tileable by ac_conv3x3_f16x3_first")
        if spec_amax is None:
            spec_amax = spec_tf.abs().amax(dim=(1, 3)).view(b, t // AMAX_ROWS, AMAX_ROWS).amax(dim=2).contiguous()
        tape = _AmaxTape(b, t, 6 * (2 * n + 1) + 2 * n, spec_tf.device)
        skips = []
        x, ax = spec_tf, spec_amax
        for i in range(n):
            first = (self.first_w, self.first_b, self._first_gain, self._first_offs) if i == 0 else None
            x, ax = self.enc[i].forward_hip(x, ax, hip, tape, self.conv_probe, first=first)
            if self.block_tap is not None: self.block_tap(f"enc{i}", x)
            skips.append(x)
            x, ax = self._down(x, ax, i, hip, tape)
        x, ax = self.bottleneck.forward_hip(x, ax, hip, tape, self.conv_probe)
        if self.block_tap is not None: self.block_tap("bottleneck", x)
        for i in range(n):
            x, ax = self._up(x, ax, i, hip, skips.pop(), tape)
            x, ax = self.dec[i].forward_hip(x, ax, hip, tape, self.conv_probe)
            if self.block_tap is not None: self.block_tap(f"dec{i}", x)
        return hip.conv1x1_small(x, self.final_w, self.final_b, relu=False)
