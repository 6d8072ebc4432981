"""Oracle for SURVEY.md §8 rows a3, a4 and the a1-a6 loop: MDX23 STFT -> TFC-TDF U-Net -> iSTFT on the CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference's CPU path is
`torch.stft` / ONNX Runtime CPU EP / `torch.istft` (`src/audio_cut/separation/backends.py:338-388`)
driven per chunk by `enhanced_vocal_separator.py:300-488`.  Neither onnxruntime, the
`Kim_Vocal_1.onnx` weights nor the external `MVSEP-MDX23-music-separation-model/inference.py`
are available, so:

* `mdx_stft` / `mdx_istft` restate the external `Conv_TDF_net_trim_model.stft/.istft`
  (constructor args at backends.py:260-265: L=11, n_fft=6144, hop 1024, dim_f 3072, dim_t 256)
  with the very library calls it makes (`torch.stft(center=True)` -> reflect padding, periodic Hann;
  `torch.istft`), on the CPU.
* `unet_forward` restates the published KUIELab TFC-TDF v2 graph that `Kim_Vocal_1.onnx` holds
  (g=48, l=3, k=3, bn=8, bias-free TDF; 16.7 M parameters = the 66.8 MB file) as un-fused
  conv -> batchnorm(eval) -> relu `torch.nn.functional` calls on the CPU.  Weights are an argument
  (seeded synthetic tensors; **parity unpinned**: no reference tensor exists to compare with).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import chunking as C
from .chunking import DIM_F, DIM_T, HOP, ITEM_LEN, N_FFT

Weights = Dict[str, np.ndarray]


def mdx_stft(batch: np.ndarray) -> torch.Tensor:
    """[B,2,261120] f32 -> [B,4,3072,256] (channels L.re, L.im, R.re, R.im; top bin dropped)."""
    x = torch.from_numpy(np.ascontiguousarray(batch)).reshape(-1, ITEM_LEN)
    win = torch.hann_window(N_FFT, periodic=True)
    s = torch.stft(x, n_fft=N_FFT, hop_length=HOP, window=win, center=True, return_complex=True)
    s = torch.view_as_real(s).permute(0, 3, 1, 2)
    s = s.reshape(-1, 2, 2, N_FFT // 2 + 1, DIM_T).reshape(-1, 4, N_FFT // 2 + 1, DIM_T)
    return s[:, :, :DIM_F].contiguous()


def mdx_istft(spec: torch.Tensor) -> np.ndarray:
    """[B,4,3072,256] -> [B,2,261120] f32 (dropped bin re-inserted as zeros)."""
    b = spec.shape[0]
    pad = torch.zeros(b, 4, N_FFT // 2 + 1 - DIM_F, DIM_T)
    x = torch.cat([spec, pad], dim=-2)
    x = x.reshape(-1, 2, 2, N_FFT // 2 + 1, DIM_T).reshape(-1, 2, N_FFT // 2 + 1, DIM_T)
    x = torch.view_as_complex(x.permute(0, 2, 3, 1).contiguous())
    win = torch.hann_window(N_FFT, periodic=True)
    y = torch.istft(x, n_fft=N_FFT, hop_length=HOP, window=win, center=True)
    return y.reshape(-1, 2, ITEM_LEN).numpy()


def _t(w: Weights, name: str) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(w[name]))


def _bn_relu(x: torch.Tensor, w: Weights, prefix: str, eps: float = 1e-5) -> torch.Tensor:
    y = F.batch_norm(x, _t(w, prefix + ".running_mean"), _t(w, prefix + ".running_var"),
                     _t(w, prefix + ".weight"), _t(w, prefix + ".bias"), training=False, eps=eps)
    return F.relu(y)


def _tfc_tdf(x: torch.Tensor, w: Weights, prefix: str, l: int) -> torch.Tensor:
    for j in range(l):
        x = F.conv2d(x, _t(w, f"{prefix}.tfc.{j}.conv.weight"), _t(w, f"{prefix}.tfc.{j}.conv.bias"), padding=1)
        x = _bn_relu(x, w, f"{prefix}.tfc.{j}.bn")
    y = F.linear(x, _t(w, f"{prefix}.tdf.0.weight"))
    y = _bn_relu(y, w, f"{prefix}.tdf.0.bn")
    y = F.linear(y, _t(w, f"{prefix}.tdf.1.weight"))
    y = _bn_relu(y, w, f"{prefix}.tdf.1.bn")
    return x + y


def unet_forward(spec: torch.Tensor, w: Weights, n_levels: int = 5, l: int = 3) -> torch.Tensor:
    """TFC-TDF v2 forward, [B,4,F,T] -> [B,4,F,T] (internally [B,c,T,F])."""
    with torch.no_grad():
        x = F.conv2d(spec, _t(w, "first_conv.weight"), _t(w, "first_conv.bias"))
        x = _bn_relu(x, w, "first_bn")
        x = x.transpose(-1, -2)
        skips: List[torch.Tensor] = []
        for i in range(n_levels):
            x = _tfc_tdf(x, w, f"enc.{i}", l)
            skips.append(x)
            x = F.conv2d(x, _t(w, f"ds.{i}.conv.weight"), _t(w, f"ds.{i}.conv.bias"), stride=2)
            x = _bn_relu(x, w, f"ds.{i}.bn")
        x = _tfc_tdf(x, w, "bottleneck", l)
        for i in range(n_levels):
            x = F.conv_transpose2d(x, _t(w, f"us.{i}.conv.weight"), _t(w, f"us.{i}.conv.bias"), stride=2)
            x = _bn_relu(x, w, f"us.{i}.bn")
            x = x * skips[-i - 1]
            x = _tfc_tdf(x, w, f"dec.{i}", l)
        x = x.transpose(-1, -2)
        return F.conv2d(x, _t(w, "final_conv.weight"), _t(w, "final_conv.bias"))


def infer_chunk(mix_chunk: np.ndarray, w: Weights, align_hop: int = 4096, output_type: str = "vocal",
                n_levels: int = 5, l: int = 3, net_fn: Optional[Callable] = None) -> Tuple[np.ndarray, np.ndarray]:
    """backends.py:299-406 on the CPU.  `net_fn` replaces the U-Net (e.g. identity for round-trip property tests)."""
    batch, stereo, orig = C.mdx_windows(mix_chunk, align_hop)
    spec = mdx_stft(batch)
    out = net_fn(spec) if net_fn is not None else unet_forward(spec, w, n_levels=n_levels, l=l)
    wave = mdx_istft(out)
    return C.mdx_assemble(wave, stereo, orig, output_type)


def separate_track(audio: np.ndarray, sr: int, w: Weights, *, chunk_s: float = 10.0, overlap_s: float = 2.5,
                   halo_s: float = 0.5, align_hop: int = 4096,
                   on_chunk: Optional[Callable] = None, n_levels: int = 5, l: int = 3, net_fn: Optional[Callable] = None):
    """enhanced_vocal_separator.py:300-488 without the feature/VAD side channels (those are
    `on_chunk(plan, mix_chunk, vocal_chunk)` callbacks so tests can wire oracle.features / oracle.vad)."""
    total = len(audio)
    plans = C.chunk_plan(total / float(sr), chunk_s, overlap_s, halo_s)
    ranges = C.plan_sample_ranges(plans, sr, total)
    outs = []
    kept_ranges = []
    for plan, (cs, ce, es, ee) in zip(plans, ranges):
        chunk = np.ascontiguousarray(audio[cs:ce], dtype=np.float32)
        if chunk.size == 0:
            continue
        voc, inst = infer_chunk(chunk, w, align_hop, n_levels=n_levels, l=l, net_fn=net_fn)
        if on_chunk is not None:
            on_chunk(plan, chunk, voc, (ee - es) > 0)
        outs.append((voc, inst))
        kept_ranges.append((cs, ce, es, ee))
    vocal, inst = C.overlap_add(total, kept_ranges, outs)
    return vocal, inst, plans
