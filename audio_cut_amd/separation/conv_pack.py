"""Host-side packing of 3x3 conv weights for `ac_conv3x3_f16x3` (csrc/ac_conv.hip).

float32 weights [C_out, C_in, 3, 3] (BatchNorm already folded) are scaled by a power of two (so that the low
halves stay in the normal float16 range; the kernel multiplies the accumulator by the exact inverse), split
into float16 hi/lo parts (hi = f16(w), lo = f16(w - hi), round-to-nearest-even) and laid out in the A-operand
fragment order of `v_mfma_f32_16x16x32_f16`: lane l holds A[row = l & 15][k = 8 (l >> 4) + j], j = 0..7.
K walks blocks of 16 input channels; inside a block taps 0..7 are paired into 4 k-steps (lane groups 0-1: tap 2p,
groups 2-3: tap 2p+1).  The fifth k-step of an ODD block carries tap 8 of the block before it in lane groups 0-1 and
its own tap 8 in groups 2-3 (the kernel keeps the even block's activation fragments in registers), so an even block
with a partner has an all-zero, never-loaded fifth step; a trailing unpaired block keeps its tap 8 in groups 0-1 against
zeros in groups 2-3.
Result: (uint16 array [C_out/48][C_in/16][5][2 (hi, lo)][3 (row tiles)][64 lanes][8], w_unscale).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def split_hi_lo(w: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    w = np.ascontiguousarray(w, dtype=np.float32)
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def weight_scale(weight: np.ndarray, target: float = 1024.0) -> float:
    """power of two s with max|w| * s <= target (keeps hi far from the f16 limit and lo out of the subnormals)."""
    m = float(np.max(np.abs(weight))) if weight.size else 0.0
    if not np.isfinite(m) or m <= 0.0:
        return 1.0
    return float(2.0 ** np.floor(np.log2(target / m)))


def pack_conv3x3(weight: np.ndarray) -> Tuple[np.ndarray, float]:
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or co % 48 or ci % 16:
        raise ValueError("pack_conv3x3 needs [C_out % 48 == 0, C_in % 16 == 0, 3, 3]")
    scale = weight_scale(weight)
    hi, lo = split_hi_lo(np.asarray(weight, dtype=np.float32) * np.float32(scale))
    taps = np.zeros((2, co, ci, 10), dtype=np.uint16)               # tap 9 = zero padding slot
    taps[0, :, :, :9] = hi.view(np.uint16).reshape(co, ci, 9)
    taps[1, :, :, :9] = lo.view(np.uint16).reshape(co, ci, 9)
    lane = np.arange(64)
    r = lane & 15
    g = lane >> 4
    j = np.arange(8)
    out = np.empty((co // 48, ci // 16, 5, 2, 3, 64, 8), dtype=np.uint16)
    for cob in range(co // 48):
        for mt in range(3):
            rows = cob * 48 + mt * 16 + r                                        # [64]
            for cb in range(ci // 16):
                cols = cb * 16 + 8 * (g & 1)[:, None] + j[None, :]               # [64, 8]
                for pair in range(4):
                    tap = pair * 2 + (g >> 1)                                    # [64]
                    for part in range(2):
                        out[cob, cb, pair, part, mt] = taps[part, rows[:, None], cols, tap[:, None]]
                n_cb = ci // 16
                if cb & 1:                     # shared step: k 0..15 = tap 8 of block cb - 1, k 16..31 = tap 8 of block cb
                    cols8 = (cb - 1 + (g >> 1))[:, None] * 16 + 8 * (g & 1)[:, None] + j[None, :]
                    tap8 = np.full(64, 8)
                elif cb + 1 < n_cb:            # even block with a partner: nothing of its own in the fifth step
                    cols8, tap8 = cols, np.full(64, 9)
                else:                          # trailing unpaired block: groups 0-1 tap 8, groups 2-3 the zero slot
                    cols8, tap8 = cols, 8 + (g >> 1)
                for part in range(2):
                    out[cob, cb, 4, part, mt] = taps[part, rows[:, None], cols8, tap8[:, None]]
    return out, 1.0 / scale


def conv3x3_wide_tileable(c_out: int, c_in: int) -> bool:
    """shapes `ac_conv3x3_f16x3_w96` takes (96 output channels per workgroup, 8-channel stages, tap 8 shared by up to 4 stages)."""
    return c_out % 96 == 0 and c_in % 16 == 0


def pack_conv3x3_w96(weight: np.ndarray, cob: int = 96) -> Tuple[np.ndarray, float]:
    """Weights for `ac_conv3x3_f16x3_w96` (cob = 96) / `ac_conv3x3_f16x3_s8` (cob = 48) (csrc/ac_conv96.hip): K walks stages
    of 8 input channels; a k-step of 32 is 4 taps x 8 channels (lane group g: tap 4 ks + g, ks = 0, 1); the third k-step of a
    stage with cb % 4 == 3 carries tap 8 of the stages cb - 3 + g, and the last stage of a trailing group of two (C_in % 32 == 16)
    carries tap 8 of the stages cb - 1 + g in lane groups 0-1 against zeros in 2-3 (all-zero and never loaded elsewhere).
    Result: (uint16 [C_out/cob][C_in/8][3][2 (hi, lo)][cob/16 (row tiles)][64 lanes][8], w_unscale)."""
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or cob not in (48, 96) or co % cob or ci % 16:
        raise ValueError("pack_conv3x3_w96 needs [C_out % cob == 0, C_in % 16 == 0, 3, 3], cob 96 or 48")
    scale = weight_scale(weight)
    hi, lo = split_hi_lo(np.asarray(weight, dtype=np.float32) * np.float32(scale))
    taps = np.zeros((2, co, ci + 8, 9), dtype=np.uint16)                                                 # channels ci.. = zeros
    taps[0, :, :ci] = hi.view(np.uint16).reshape(co, ci, 9)
    taps[1, :, :ci] = lo.view(np.uint16).reshape(co, ci, 9)
    lane = np.arange(64)
    r, g, j = lane & 15, lane >> 4, np.arange(8)
    n_cb, mt_n = ci // 8, cob // 16
    out = np.zeros((co // cob, n_cb, 3, 2, mt_n, 64, 8), dtype=np.uint16)
    for blk in range(co // cob):
        for mt in range(mt_n):
            rows = (blk * cob + mt * 16 + r)[:, None]
            for cb in range(n_cb):
                own = cb * 8 + j[None, :] + 0 * g[:, None]                                                # [64, 8]
                for ks in range(2):
                    out[blk, cb, ks, :, mt] = taps[:, rows, own, (4 * ks + g)[:, None]]
                if cb % 4 == 3 or cb == n_cb - 1:
                    stage = np.where(g <= cb % 4, cb - cb % 4 + g, n_cb)                                  # n_cb -> the zero channels
                    out[blk, cb, 2, :, mt] = taps[:, rows, stage[:, None] * 8 + j[None, :], 8]
    return out, 1.0 / scale


def linear_tileable(n_out: int, k_in: int, channels: int, t: int) -> bool:
    """shapes `ac_tdf_linear_f16x3` tiles: C % 16 == 0, T % 8 == 0 (a tile is 16 channels x 8 time rows), K % 32 == 0, N % 96 == 0."""
    return channels % 16 == 0 and t % 8 == 0 and k_in % 32 == 0 and n_out % 96 == 0


def pack_linear(weight: np.ndarray, bn: int = 0) -> Tuple[np.ndarray, float]:
    """Linear weight [N, K] (y = x @ W^T) -> B-operand fragments of `v_mfma_f32_16x16x32_f16` for ac_tdf_linear_f16x3 /
    ac_down2x_f16x3 / ac_up2x_f16x3: lane l of a fragment holds B[k = 8 (l >> 4) + j][n = l & 15]
    = W[n0 + (l & 15)][k0 + 8 (l >> 4) + j], j = 0..7.
    `bn` = columns per workgroup: 0 picks 192 when N % 192 == 0 else 96 and needs exact tiling (the TDF kernel);
    an explicit `bn` zero-pads N to a multiple of it and K to a multiple of 32 (the resampling kernels).
    Result: (uint16 [N'/bn][K'/32][2 (hi, lo)][bn/16][64][8], w_unscale)."""
    weight = np.asarray(weight, dtype=np.float32)
    n, k = weight.shape
    if bn == 0:
        if n % 96 or k % 32:
            raise ValueError("pack_linear needs N % 96 == 0 and K % 32 == 0")
        bn = 192 if n % 192 == 0 else 96
    else:
        n2, k2 = -(-n // bn) * bn, -(-k // 32) * 32
        if (n2, k2) != (n, k):
            padded = np.zeros((n2, k2), dtype=np.float32)
            padded[:n, :k] = weight
            weight, n, k = padded, n2, k2
    scale = weight_scale(weight)
    hi, lo = split_hi_lo(weight * np.float32(scale))
    parts = np.stack([hi.view(np.uint16), lo.view(np.uint16)])            # [2, N, K]
    # [2, N/BN, BN/16, 16 (n in tile), K/32, 4 (lane group), 8 (j)]
    p = parts.reshape(2, n // bn, bn // 16, 16, k // 32, 4, 8)
    # -> [N/BN, K/32, 2, BN/16, lane = group * 16 + n_in_tile, 8]
    out = p.transpose(1, 4, 0, 2, 5, 3, 6).reshape(n // bn, k // 32, 2, bn // 16, 64, 8)
    return np.ascontiguousarray(out), 1.0 / scale


def tdf_small_tileable(f: int, hidden: int, rows: int) -> bool:
    """shapes `ac_tdf_small_fused` takes: F % 32 == 0, bottleneck <= 48, rows % 32 == 0."""
    return f % 32 == 0 and 0 < hidden <= 48 and rows % 32 == 0


def pack_tdf_small(w1: np.ndarray, w2: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """The two bias-free Linear weights of a narrow TDF pair (w1 [Hd, F], w2 [F, Hd]) as float32 operand fragments of
    `v_mfma_f32_16x16x4_f32` for ac_tdf_small_fused (csrc/ac_tdf_small.hip): lane l holds B[k = l >> 4][n = l & 15].
      w1p [ceil(Hd / 16)][F / 16][64][4]: element i of lane l = w1[16 nt + (l & 15)][16 kk + 4 (l >> 4) + i] - the kernel feeds
          the four elements of one float4 of x to four k-steps, so k-step i of group kk is column 16 kk + 4 g + i for lane group g
      w2p [F / 16][ceil(Hd / 4)][64]:     lane l = w2[16 nt + (l & 15)][4 ks + (l >> 4)]
    Rows / columns beyond Hd are zero.  No scaling: float32 products."""
    hd, f = w1.shape
    if w2.shape != (f, hd) or f % 16 or not (0 < hd <= 48):
        raise ValueError("pack_tdf_small needs w1 [Hd <= 48, F % 16 == 0] and w2 [F, Hd]")
    lane = np.arange(64)
    col, grp = lane & 15, lane >> 4
    nt1, n_kk, n_ks = (hd + 15) // 16, f // 16, (hd + 3) // 4
    w1z = np.zeros((nt1 * 16, f), np.float32); w1z[:hd] = w1
    w2z = np.zeros((f, n_ks * 4), np.float32); w2z[:, :hd] = w2
    rows = (16 * np.arange(nt1)[:, None, None, None] + col[None, None, :, None])                     # [nt][1][64][1]
    cols = (16 * np.arange(n_kk)[None, :, None, None] + 4 * grp[None, None, :, None] + np.arange(4)[None, None, None, :])
    w1p = w1z[rows, cols]                                                                              # [nt1][n_kk][64][4]
    rows2 = 16 * np.arange(f // 16)[:, None, None] + col[None, None, :]
    cols2 = 4 * np.arange(n_ks)[None, :, None] + grp[None, None, :]
    w2p = w2z[rows2, cols2]                                                                            # [F/16][n_ks][64]
    return np.ascontiguousarray(w1p, np.float32), np.ascontiguousarray(w2p, np.float32)
