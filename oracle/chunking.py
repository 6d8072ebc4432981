"""Oracle for SURVEY.md §8 rows a1, a2, a5, a6: chunk plan, MDX23 windowing, stem algebra, OLA.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
`src/audio_cut/utils/gpu_pipeline.py:54-84,333-375` (plan),
`src/audio_cut/separation/backends.py:268-281,306-330,389-406` (windowing, crop,
mix-minus-stem, mono mean) and
`src/vocal_smart_splitter/core/enhanced_vocal_separator.py:322-324,366-373,423-437,456-458`
(effective-region uniform overlap-add).  `chunk_plan` is pinned against the
reference's own `chunk_schedule` (importable here) by tests/golden/make_golden.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import numpy as np


@dataclass(frozen=True)
class Plan:
    index: int
    start_s: float
    end_s: float
    halo_left_s: float
    halo_right_s: float

    @property
    def effective_start_s(self) -> float:
        return self.start_s + self.halo_left_s

    @property
    def effective_end_s(self) -> float:
        return self.end_s - self.halo_right_s


def chunk_plan(total_s: float, chunk_s: float = 10.0, overlap_s: float = 2.5, halo_s: float = 0.5) -> List[Plan]:
    """gpu_pipeline.py:333-375."""
    total_s = max(0.0, float(total_s))
    chunk_s = max(0.1, float(chunk_s))
    overlap_s = max(0.0, min(float(overlap_s), chunk_s * 0.9))
    halo_s = max(0.0, min(float(halo_s), chunk_s * 0.5))
    if total_s <= chunk_s:
        return [Plan(0, 0.0, total_s, 0.0, 0.0)]
    stride = chunk_s - overlap_s
    if stride <= 0:
        stride = chunk_s
    out: List[Plan] = []
    start = 0.0
    while start < total_s - 1e-6:
        end = min(total_s, start + chunk_s)
        more = end < total_s - 1e-6
        out.append(Plan(len(out), start, end, halo_s if out else 0.0, halo_s if more else 0.0))
        if not more:
            break
        start += stride
    return out


def plan_sample_ranges(plans: List[Plan], sr: int, total: int) -> List[Tuple[int, int, int, int]]:
    """(chunk_start, chunk_end, eff_start, eff_end) in samples, enhanced_vocal_separator.py:367-368,423-425."""
    out = []
    for p in plans:
        cs = max(0, int(round(p.start_s * sr)))
        ce = min(total, int(round(p.end_s * sr)))
        es = cs + int(round(p.halo_left_s * sr))
        ee = ce - int(round(p.halo_right_s * sr))
        ee = max(es, min(total, ee))
        out.append((cs, ce, es, ee))
    return out


# MDX23 constants (external Conv_TDF_net_trim_model instantiated at backends.py:260-265)
N_FFT = 6144
HOP = 1024
DIM_T = 256
DIM_F = 3072
ITEM_LEN = HOP * (DIM_T - 1)     # 261120  ("chunk_size" of the external model)
TRIM = N_FFT // 2                # 3072
GEN = ITEM_LEN - 2 * TRIM        # 254976


def mdx_windows(mix_chunk: np.ndarray, align_hop: int = 4096) -> Tuple[np.ndarray, np.ndarray, int]:
    """backends.py:268-281 + 306-330 -> (batch [B,2,261120] f32, aligned stereo mix [2,L], original_len)."""
    if mix_chunk.ndim == 1:
        stereo = np.stack([mix_chunk, mix_chunk], axis=0)
    else:
        stereo = mix_chunk
    stereo = np.ascontiguousarray(stereo.astype(np.float32, copy=False))
    original_len = stereo.shape[-1]
    a_pad = (-original_len) % max(1, align_hop)
    if a_pad:
        stereo = np.pad(stereo, ((0, 0), (0, a_pad)))
    L = stereo.shape[-1]
    pad = (GEN - L % GEN) % GEN
    padded = np.concatenate([np.zeros((2, TRIM), np.float32), stereo, np.zeros((2, pad + TRIM), np.float32)], axis=1)
    items = [padded[:, i: i + ITEM_LEN] for i in range(0, L + pad, GEN)]
    return np.stack(items).astype(np.float32), stereo, original_len


def mdx_assemble(wave_items: np.ndarray, stereo_aligned: np.ndarray, original_len: int,
                 output_type: str = "vocal") -> Tuple[np.ndarray, np.ndarray]:
    """backends.py:377,389-406: trim margins, concatenate, crop, mix-minus-stem, channel mean."""
    L = stereo_aligned.shape[-1]
    wave = wave_items[:, :, TRIM:-TRIM].transpose(1, 0, 2).reshape(2, -1)[:, :L]
    mix = stereo_aligned[:, :L]
    if original_len != L:
        wave = wave[:, :original_len]
        mix = mix[:, :original_len]
    if output_type == "vocal":
        vocal, inst = wave, mix - wave
    else:
        inst, vocal = wave, mix - wave
    return vocal.mean(axis=0).astype(np.float32), inst.mean(axis=0).astype(np.float32)


def overlap_add(total: int, ranges: List[Tuple[int, int, int, int]],
                chunk_outputs: List[Tuple[np.ndarray, Optional[np.ndarray]]]) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """enhanced_vocal_separator.py:322-324,423-437,456-458 (uniform mean over effective regions)."""
    v_acc = np.zeros(total, np.float32)
    i_acc = np.zeros(total, np.float32)
    w_acc = np.zeros(total, np.float32)
    for (cs, ce, es, ee), (voc, inst) in zip(ranges, chunk_outputs):
        ls = es - cs
        le = ls + (ee - es)
        ev = voc[ls:le]
        if ev.size == 0:
            continue
        v_acc[es:ee] += ev
        w_acc[es:ee] += 1.0
        if inst is not None:
            i_acc[es:ee] += inst[ls:le]
    w_acc[w_acc == 0.0] = 1.0
    vocal = (v_acc / w_acc).astype(np.float32)
    inst = (i_acc / w_acc).astype(np.float32) if np.any(i_acc) else None
    return vocal, inst
