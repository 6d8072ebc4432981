"""Track-global kernels queued ahead of the host logic that consumes them (single-track latency; `_native.Context.prefetch`).

After a track's U-Net the reference path is a chain of host decisions, each preceded by a small kernel and a download: marker RMS,
detector RMS + flatness, no-vocal-run RMS, the two quiet-guard lookups, three mean squares.  None of those launches depends on a host
decision - only on the stems and on configuration - so they are queued as soon as the stems' producers are: the mix-only ones on the
side stream while the U-Net runs, the vocal ones right behind the separation.  The consumers are unchanged: their wrapper calls find
the result in the context's prefetch cache (a parameter derived differently here than there merely misses; `prefetch_stats()` reports
unused entries, `tests/test_pipeline_gpu.py::test_full_path_separate_detect_against_oracle` keeps the two sides in step).

Every parameter below is derived exactly where its consumer derives it (cited per line)."""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..config import get_config


def guard_window_samples(sr: int) -> int:
    """`cutting/refine.py:_Wave.prepare_lookup` (reference `refine.py:161-181`) with `quality_control.enforce_quiet_cut.win_ms`."""
    win_ms = float(get_config("quality_control.enforce_quiet_cut.win_ms", 80))
    return max(1, int(round(win_ms / 1000.0 * sr)))


def guard_floor_db(rms2048_db: np.ndarray) -> float:
    """`seamless_splitter.py:1829-1850` (quirk Q1: floor_percentile 0.5 is read as a fraction): the quiet floor of the guards from the
    MIX's RMS(2048, 10 ms) in dB.  `SeamlessSplitter._finalize_and_filter_cuts_v2` calls this too."""
    override = get_config("quality_control.enforce_quiet_cut.floor_db_override", None)
    if override is not None:
        return float(override)
    try:
        cfg = get_config("quality_control.enforce_quiet_cut.floor_percentile", 5)
        pct = float(cfg) / 100.0 if float(cfg) > 1 else float(cfg)
    except Exception:
        pct = 0.05
    return float(np.percentile(rms2048_db, max(0.0, min(100.0, pct * 100.0))))


def queue_mix_globals(hip, mix_dev, sr: int) -> Optional[float]:
    """On the CURRENT (side) stream, while the U-Net runs: the mix's guard lookup and mean square.  Returns the guards' floor (needs one
    small download from this stream - the U-Net runs on another one) or None when the guards are off."""
    if mix_dev is None or int(mix_dev.numel()) == 0:
        return None
    hip.prefetch("sum_squares_parts", mix_dev)                                           # enhanced_vocal_separator.py:490-501 (confidence)
    if not bool(get_config("quality_control.enforce_quiet_cut.enable", False)):
        return None
    hop = max(1, int(0.01 * sr))
    rms = hip.prefetch("frame_rms", mix_dev, 2048, hop, True).cpu().numpy()              # SeamlessSplitter._rms2048_db (asked again by _finalize_and_filter_cuts_v2)
    floor_db = guard_floor_db(20.0 * np.log10(rms + 1e-12))
    db = hip.prefetch("moving_meansq_db", mix_dev, guard_window_samples(sr))
    hip.prefetch("next_leq_scan", db, float(floor_db))
    return floor_db


def queue_vocal_globals(hip, vocal_dev, inst_dev, sr: int, floor_db: Optional[float]) -> None:
    """On the CURRENT stream, right behind the separation (the stems' producers are queued, nothing has to have run yet)."""
    if vocal_dev is None or int(vocal_dev.numel()) == 0:
        return
    for t in (inst_dev, vocal_dev):                                                      # confidence: `:490-501`, in the consumer's order
        if t is not None and int(t.numel()):
            hip.prefetch("sum_squares_parts", t)
    n = int(vocal_dev.numel())
    rms_cfgs = []                                                                        # the stem's RMS series: ONE fused pass (ac_frame_rms_multi)
    hop = max(1, int(0.02 * sr)); frame = max(hop * 2, int(0.05 * sr))                   # compute_vocal_presence_markers (2205 / 882)
    rms_cfgs.append((frame, hop))
    det_frame, det_hop = int(sr * 0.025), int(sr * 0.01)                                 # PureVocalPauseDetector._detect_energy_valleys (1102 / 441)
    relative = det_hop > 0 and det_frame > 0 and bool(get_config("pure_vocal_detection.enable_relative_energy_mode", False))
    if relative:
        rms_cfgs.append((det_frame, det_hop))
    if float(get_config("quality_control.pure_music_min_duration", 0.0)) > 0.0:          # SeamlessSplitter._find_no_vocal_runs (2048 / 441)
        rms_cfgs.append((2048, max(1, int(0.01 * sr))))
    rms_cfgs = [(f, h) for f, h in dict.fromkeys(rms_cfgs) if n + 2 * (f // 2) >= f and f <= 8192]
    if rms_cfgs:
        hip.prefetch_frame_rms_multi(vocal_dev, rms_cfgs)
    if relative:
        hip.prefetch("stft2048_flatness", vocal_dev, det_hop)
    if floor_db is not None:                                                             # finalize_cut_points: the vocal guard's lookup
        db = hip.prefetch("moving_meansq_db", vocal_dev, guard_window_samples(sr))
        hip.prefetch("next_leq_scan", db, float(floor_db))
