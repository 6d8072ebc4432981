"""Test infrastructure: the equivalence class of a quiet-guard boundary that the reference decides on numerical dust.

The guard (reference `src/audio_cut/cutting/refine.py:172-174,184-214`) takes the FIRST minimum of
`rms_db = 20*log10(sqrt(moving mean square + 1e-12) + 1e-12)` over an 80 ms window.  Inside digital silence that series is one
flat run of bit-equal values (mean square < what float64 resolves against the 1e-12: "the epsilon plateau", -119.999991 dB) and
the first minimum is the sample at which the last non-zero stem sample leaves the window - a sample whose value is the float32
inverse FFT's rounding noise.  The reference's own answer there depends on its FFT library build.

The predicate used by the tests and by tools/parity_soak.py - NOT a sample tolerance:
  * a boundary that is not on the plateau in the oracle must be equal;
  * a boundary on the plateau may differ only if the ORACLE's own vocal and mix dB series are bit-equal at the two indices
    (the reference could not tell them apart either) and the GPU stem agrees with the oracle's stem around it to `stem_atol`.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

HALF = 4096
GUARD_WIN_MS = 80.0


def plateau_db() -> np.float64:
    from oracle import refine as OR
    return np.float64(20.0 * np.log10(np.sqrt(np.float64(0.0) + OR.EPS) + OR.EPS))     # the series where the mean square is 0


def boundary_context(vocal: np.ndarray, mix: np.ndarray, bounds: Sequence[int], sr: int, half: int = HALF) -> Dict[str, np.ndarray]:
    """The oracle's own dB series (vocal, mix) and stem over [b - half, b + half] for every boundary b."""
    from oracle import refine as OR
    n = len(mix)
    win = max(1, int(round(GUARD_WIN_MS / 1000.0 * sr)))
    plat = plateau_db()
    on_plateau, dbv, dbm, stem = [], [], [], []
    for b in bounds:
        b = int(b)
        lo, hi = max(0, b - half - win), min(n, b + half + win + 1)
        dv = OR.moving_meansq_db(vocal[lo:hi], win)
        dm = OR.moving_meansq_db(mix[lo:hi], win)
        idx = np.arange(b - half, b + half + 1)
        ok = (idx >= 0) & (idx < n)
        seg_v = np.full(idx.shape, np.nan); seg_m = np.full(idx.shape, np.nan); seg_s = np.zeros(idx.shape, np.float32)
        seg_v[ok] = dv[idx[ok] - lo]; seg_m[ok] = dm[idx[ok] - lo]; seg_s[ok] = vocal[idx[ok]]
        at = min(max(b, 0), n - 1) - lo
        on_plateau.append(bool(0 < b < n and dv[at] == plat and dm[at] == plat))
        dbv.append(seg_v); dbm.append(seg_m); stem.append(seg_s)
    return {"on_plateau": np.asarray(on_plateau, dtype=bool), "db_vocal": np.asarray(dbv, dtype=np.float64).reshape(len(bounds), -1),
            "db_mix": np.asarray(dbm, dtype=np.float64).reshape(len(bounds), -1),
            "stem_window": np.asarray(stem, dtype=np.float32).reshape(len(bounds), -1), "guard_half_window": np.int64(half)}


def classify_boundaries(got: Sequence[int], want: Sequence[int], ctx: Dict[str, np.ndarray], gpu_vocal: np.ndarray,
                        stem_atol: float) -> Tuple[List[int], List[Tuple[int, int]], List[str]]:
    """-> (exact boundaries, [(gpu index, oracle index)] plateau-equivalent ones, failure messages)."""
    fails: List[str] = []
    if len(got) != len(want):
        return [], [], [f"boundary count {len(got)} != {len(want)}: {list(got)} vs {list(want)}"]
    half = int(ctx["guard_half_window"])
    exact: List[int] = []
    equiv: List[Tuple[int, int]] = []
    n = len(gpu_vocal)
    for k, (g, o) in enumerate(zip(got, want)):
        g, o = int(g), int(o)
        if g == o:
            exact.append(g)
            continue
        if not bool(ctx["on_plateau"][k]):
            fails.append(f"boundary #{k}: {g} != {o} and the oracle's dB series is NOT on the epsilon plateau there")
            continue
        if abs(g - o) > half:
            fails.append(f"boundary #{k}: {g} vs {o}: further apart than the stored window")
            continue
        dv, dm = ctx["db_vocal"][k], ctx["db_mix"][k]
        if not (dv[half + (g - o)] == dv[half] and dm[half + (g - o)] == dm[half]):
            fails.append(f"boundary #{k}: {g} vs {o}: the oracle's dB series differ between the two indices "
                         f"(vocal {dv[half + (g - o)]!r} vs {dv[half]!r}, mix {dm[half + (g - o)]!r} vs {dm[half]!r})")
            continue
        idx = np.arange(o - half, o + half + 1)
        ok = (idx >= 0) & (idx < n)
        err = float(np.max(np.abs(gpu_vocal[idx[ok]].astype(np.float64) - ctx["stem_window"][k][ok].astype(np.float64))))
        if err > stem_atol:
            fails.append(f"boundary #{k}: {g} vs {o}: stem differs by {err:.3e} > {stem_atol:.3e} around the boundary")
            continue
        equiv.append((g, o))
    return exact, equiv, fails


def map_cuts(cuts: Sequence[int], equiv: Sequence[Tuple[int, int]]) -> List[int]:
    """Manifest cuts with every plateau-equivalent GPU boundary replaced by the oracle's index."""
    m = {g: o for g, o in equiv}
    return [m.get(int(c), int(c)) for c in cuts]
