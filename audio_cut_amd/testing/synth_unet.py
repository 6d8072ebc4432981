"""Seeded synthetic weights of the MDX23 TFC-TDF architecture - TEST / BENCH DATA, not part of the inference path.

`Kim_Vocal_1.onnx` cannot be fetched offline, so tests, `bench.py`, `__graft_entry__.smoke()` and an unconfigured
`MDX23HipBackend` use seeded tensors of exactly that architecture (16.67 M parameters), with the batch-norm statistics
calibrated on a seeded pseudo-spectrogram so that activations stay O(1) like a trained net.  The calibration pass is plain
PyTorch on the CPU (run once, on the host, to MAKE weights); inference never comes through here.  The same name -> ndarray dict
drives the HIP net (`separation/tfc_tdf.py:TfcTdfNet`) and the CPU oracle in the tests.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from ..separation.tfc_tdf import TfcTdfSpec

Weights = Dict[str, np.ndarray]


def _block_names(prefix: str, spec: TfcTdfSpec) -> List[str]:
    names = []
    for j in range(spec.l):
        names += [f"{prefix}.tfc.{j}"]
    return names


def synth_weights(spec: TfcTdfSpec = TfcTdfSpec(), seed: int = 0, calib_t: int = 128) -> Weights:
    """Seeded synthetic weights of the TFC-TDF architecture with calibrated batch-norm statistics."""
    rng = np.random.default_rng(seed)
    w: Weights = {}

    def conv(name, cout, cin, kh, kw, transpose=False):
        fan_in = cin * kh * kw
        shape = (cin, cout, kh, kw) if transpose else (cout, cin, kh, kw)
        w[name + ".weight"] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        # Only the up-sampling path carries additive terms (see `bn`): with every other bias / BN shift
        # at zero the synthetic net maps silence to silence and is ~degree-1 in the input level, which a
        # random net with multiplicative skips otherwise is not (it would be degree 6 and explode).
        amp = 0.02 if transpose else 0.0
        w[name + ".bias"] = (rng.standard_normal(cout) * amp).astype(np.float32)

    def bn(name, c):
        gate = name.startswith("us.")
        if gate:   # gate ~ 1 +- 0.004 at the calibration level; spectral peaks and loud passages sit 10-100 sigma out: x * skip stays close to the skip tensor
            w[name + ".weight"] = rng.uniform(0.002, 0.006, c).astype(np.float32)
            w[name + ".bias"] = (1.0 + rng.standard_normal(c) * 0.05).astype(np.float32)
        else:
            w[name + ".weight"] = rng.uniform(0.8, 1.2, c).astype(np.float32)
            w[name + ".bias"] = np.zeros(c, np.float32)
        w[name + ".running_mean"] = np.zeros(c, np.float32)
        w[name + ".running_var"] = np.ones(c, np.float32)

    def block(prefix, c, f):
        for j in range(spec.l):
            conv(f"{prefix}.tfc.{j}.conv", c, c, spec.k, spec.k)
            bn(f"{prefix}.tfc.{j}.bn", c)
        h = f // spec.bn
        w[f"{prefix}.tdf.0.weight"] = (rng.standard_normal((h, f)) * np.sqrt(2.0 / f)).astype(np.float32)
        bn(f"{prefix}.tdf.0.bn", c)
        w[f"{prefix}.tdf.1.weight"] = (rng.standard_normal((f, h)) * np.sqrt(2.0 / h)).astype(np.float32)
        bn(f"{prefix}.tdf.1.bn", c)

    conv("first_conv", spec.g, spec.dim_c, 1, 1)
    bn("first_bn", spec.g)
    f = spec.dim_f
    for i in range(spec.n_levels):
        c = spec.channels(i)
        block(f"enc.{i}", c, f)
        conv(f"ds.{i}.conv", c + spec.g, c, 2, 2)
        bn(f"ds.{i}.bn", c + spec.g)
        f //= 2
    block("bottleneck", spec.channels(spec.n_levels), f)
    for i in range(spec.n_levels):
        c = spec.channels(spec.n_levels - i)
        conv(f"us.{i}.conv", c - spec.g, c, 2, 2, transpose=True)
        bn(f"us.{i}.bn", c - spec.g)
        f *= 2
        block(f"dec.{i}", c - spec.g, f)
    conv("final_conv", spec.dim_c, spec.g, 1, 1)

    _calibrate(w, spec, rng, calib_t)
    return w


def _calibration_spectrogram(spec: TfcTdfSpec, rng: np.random.Generator, frames: int) -> torch.Tensor:
    """STFT (n_fft = 2*dim_f, hop = n_fft/6, periodic Hann, reflect-centred) of a seeded song-like clip:
    harmonic stack with vibrato + decaying noise bursts, laid out [1, 4, dim_f, frames] like the MDX23 input."""
    n_fft = 2 * spec.dim_f
    hop = max(1, n_fft // 6)
    n = hop * (frames - 1)
    t = np.arange(n) / 44100.0
    f0 = 180.0 * (1.0 + 0.4 * np.sin(2 * np.pi * 0.23 * t)) * (1.0 + 0.01 * np.sin(2 * np.pi * 5.0 * t))
    phase = 2 * np.pi * np.cumsum(f0) / 44100.0
    gate = (np.sin(2 * np.pi * 0.31 * t) > -0.3).astype(np.float64)         # sung phrases with rests
    voice = gate * sum((0.5 / h) * np.sin(h * phase) for h in range(1, 12))
    kick_t = t % 0.5
    kick = np.sin(2 * np.pi * (55.0 + 60.0 * np.exp(-kick_t * 30.0)) * kick_t) * np.exp(-kick_t * 14.0)
    burst = rng.standard_normal(n) * np.exp(-((t * 4.0) % 1.0) * 12.0)
    left = 0.30 * voice + 0.22 * kick + 0.08 * burst
    right = 0.29 * voice + 0.20 * kick + 0.08 * np.roll(burst, 17)
    wave = torch.from_numpy(np.stack([left, right]).astype(np.float32))
    st = torch.stft(wave, n_fft=n_fft, hop_length=hop, window=torch.hann_window(n_fft, periodic=True),
                    center=True, return_complex=True)
    st = torch.view_as_real(st).permute(0, 3, 1, 2).reshape(1, 4, n_fft // 2 + 1, -1)
    return st[:, :, : spec.dim_f, :frames].contiguous()


def _calibrate(w: Weights, spec: TfcTdfSpec, rng: np.random.Generator, calib_t: int) -> None:
    """One training-mode-like pass on a seeded pseudo-spectrogram: every BN takes the batch statistics
    of its input as running statistics, and the final conv is scaled to return the input's scale."""
    calib_t = max(calib_t, 2 ** spec.n_levels)
    x = _calibration_spectrogram(spec, rng, calib_t)
    in_std = float(x.std())

    def t(name):
        return torch.from_numpy(w[name])

    def bn_relu(y, name):
        dims = (0, 2, 3)
        if name.startswith("us."):
            mean = y.mean(dim=dims)
            var = y.var(dim=dims, unbiased=False)
        else:                       # shift-free layers: normalise the second moment only
            mean = torch.zeros(y.shape[1])
            var = (y * y).mean(dim=dims)
        w[name + ".running_mean"] = mean.numpy().astype(np.float32).copy()
        w[name + ".running_var"] = np.maximum(var.numpy(), 1e-6).astype(np.float32)
        y = F.batch_norm(y, t(name + ".running_mean"), t(name + ".running_var"), t(name + ".weight"),
                         t(name + ".bias"), training=False, eps=spec.bn_eps)
        return F.relu(y)

    def block(y, prefix):
        for j in range(spec.l):
            y = F.conv2d(y, t(f"{prefix}.tfc.{j}.conv.weight"), t(f"{prefix}.tfc.{j}.conv.bias"), padding=spec.k // 2)
            y = bn_relu(y, f"{prefix}.tfc.{j}.bn")
        z = bn_relu(F.linear(y, t(f"{prefix}.tdf.0.weight")), f"{prefix}.tdf.0.bn")
        z = bn_relu(F.linear(z, t(f"{prefix}.tdf.1.weight")), f"{prefix}.tdf.1.bn")
        return y + z

    with torch.no_grad():
        y = bn_relu(F.conv2d(x, t("first_conv.weight"), t("first_conv.bias")), "first_bn").transpose(-1, -2)
        skips = []
        for i in range(spec.n_levels):
            y = block(y, f"enc.{i}")
            skips.append(y)
            y = bn_relu(F.conv2d(y, t(f"ds.{i}.conv.weight"), t(f"ds.{i}.conv.bias"), stride=2), f"ds.{i}.bn")
        y = block(y, "bottleneck")
        for i in range(spec.n_levels):
            y = bn_relu(F.conv_transpose2d(y, t(f"us.{i}.conv.weight"), t(f"us.{i}.conv.bias"), stride=2), f"us.{i}.bn")
            y = y * skips[-i - 1]
            y = block(y, f"dec.{i}")
        y = y.transpose(-1, -2)
        out = F.conv2d(y, t("final_conv.weight"), t("final_conv.bias"))
        scale = 0.5 * in_std / max(float(out.std()), 1e-12)
    w["final_conv.weight"] = (w["final_conv.weight"] * scale).astype(np.float32)
    w["final_conv.bias"] = np.zeros_like(w["final_conv.bias"])
