"""Oracle for SURVEY.md §8 rows a7, a14, a15, a16: pause detection on the vocal stem.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates the live (energy-valley) branch of
`src/vocal_smart_splitter/core/pure_vocal_pause_detector.py:131-290` with its helpers
(`:293-408` focus windows, `:461-547` cap/merge, `:1020-1094` precise cut, `:1096-1235` valleys,
`:1237-1368` MDD boost, `:1389-1532` VPP multiplier), `src/audio_cut/config/derive.py:149-150,287-336`,
`src/vocal_smart_splitter/core/vocal_separator.py:460-529` (presence markers) and
`src/vocal_smart_splitter/core/seamless_splitter.py:1706-1790` (no-vocal runs).
Pinned by running the reference's own methods (constructed with `object.__new__`, no
Silero/Demucs download) over the restated librosa ops: tests/golden/make_golden.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import librosa_ops as L
from .config import get_config
from .features import FeatureCache


@dataclass
class Pause:
    start_time: float
    end_time: float
    duration: float
    pause_type: str
    confidence: float
    features: Dict = field(default_factory=dict)
    cut_point: float = 0.0
    quality_grade: str = "B"


# ----------------------------- derive.py ---------------------------------

def _clamp(v, lo, hi):
    return max(lo, min(hi, v))


@dataclass(frozen=True)
class Thresholds:
    peak_ratio: float
    rms_ratio: float
    slow_multiplier: float
    fast_multiplier: float
    clamp_min: float
    clamp_max: float


def resolve_threshold(base_ratio: float, adapt_cfg: Dict, bpm: Optional[float], global_mdd: Optional[float]) -> Thresholds:
    """derive.py:287-326."""
    adapt_cfg = adapt_cfg or {}
    bpm_cfg = adapt_cfg.get("bpm", {})
    cmin = float(adapt_cfg.get("clamp_min", 0.85)); cmax = float(adapt_cfg.get("clamp_max", 1.15))
    slow = float(bpm_cfg.get("slow_multiplier", 1.08)); fast = float(bpm_cfg.get("fast_multiplier", 0.92))
    peak = base_ratio
    rms = _clamp(base_ratio + 0.06, 0.05, 0.7)
    if bpm and bpm > 0:
        if bpm < 90.0:
            peak *= _clamp(slow, cmin, cmax)
        elif bpm > 140.0:
            peak *= _clamp(fast, cmin, cmax)
        peak = _clamp(peak, base_ratio * cmin, base_ratio * cmax)
    mdd_cfg = adapt_cfg.get("mdd", {})
    if global_mdd is not None:
        peak *= _clamp(float(mdd_cfg.get("base", 1.0)) + float(mdd_cfg.get("gain", 0.2)) * global_mdd, cmin, cmax)
    peak = _clamp(peak, 0.05, 0.6)
    rms = _clamp(rms, peak + 0.02, 0.72)
    return Thresholds(peak, rms, slow, fast, cmin, cmax)


def resolve_min_pause(base_pause: float, strength: float, bpm: Optional[float]) -> float:
    """derive.py:329-336."""
    if not bpm or bpm <= 0:
        return base_pause
    strength = _clamp(strength, 0.0, 1.5)
    return max(0.3, base_pause + -0.18 * strength * _clamp((bpm - 110.0) / 110.0, -1.0, 1.0))


# ----------------------------- run-length helpers ---------------------------------

def _runs(mask: np.ndarray) -> List[Tuple[int, int, bool]]:
    """[(start, stop, value)] maximal constant runs of a bool vector."""
    m = np.asarray(mask, dtype=bool)
    if m.size == 0:
        return []
    edges = np.flatnonzero(m[1:] != m[:-1]) + 1
    starts = np.concatenate(([0], edges))
    stops = np.concatenate((edges, [m.size]))
    return [(int(a), int(b), bool(m[a])) for a, b in zip(starts, stops)]


def fill_false_runs(mask: np.ndarray, max_len: int) -> np.ndarray:
    """pure_vocal_pause_detector.py:1429-1443 (closing)."""
    out = np.asarray(mask, dtype=bool).copy()
    for a, b, v in _runs(out):
        if not v and (b - a) <= max_len:
            out[a:b] = True
    return out


def remove_true_runs(mask: np.ndarray, max_len: int) -> np.ndarray:
    """pure_vocal_pause_detector.py:1445-1459 (opening).  Runs are taken on the input mask."""
    out = np.asarray(mask, dtype=bool).copy()
    for a, b, v in _runs(out):
        if v and (b - a) <= max_len:
            out[a:b] = False
    return out


# ----------------------------- focus windows ---------------------------------

def merge_windows(windows: Sequence[Tuple[float, float]], min_width: float = 0.0) -> List[Tuple[float, float]]:
    """pure_vocal_pause_detector.py:394-408."""
    merged: List[Tuple[float, float]] = []
    for s, e in sorted(windows, key=lambda w: w[0]):
        if e <= s:
            continue
        if merged and s <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(merged[-1][1], e))
        else:
            merged.append((s, e))
    if min_width > 0.0:
        merged = [(s, e) for s, e in merged if (e - s) >= min_width]
    return merged


def focus_windows_from_vad(segments: Sequence[Dict[str, float]], pad_s: float = 0.2, min_width_s: float = 0.0) -> List[Tuple[float, float]]:
    """pure_vocal_pause_detector.py:293-366 — returns the *gap* windows (what list(FocusWindowList) yields)."""
    if not segments:
        return []
    pad = max(0.0, float(pad_s)); min_w = max(0.0, float(min_width_s))
    merge_gap = float(get_config("advanced_vad.focus_merge_gap_s", 0.12))
    spans = []
    for seg in segments:
        s = float(seg.get("start", seg.get("start_time", 0.0)))
        e = float(seg.get("end", seg.get("end_time", s)))
        if e > s:
            spans.append((s, e))
    if not spans:
        return []
    spans.sort(key=lambda it: it[0])
    track_end = max(e for _, e in spans)

    def _gap_merge(ws):
        ws = sorted(ws, key=lambda it: it[0])
        out: List[Tuple[float, float]] = []
        for s, e in ws:
            if not out or s - out[-1][1] > merge_gap:
                out.append((s, e))
            else:
                out[-1] = (out[-1][0], max(out[-1][1], e))
        return out

    gaps = []
    prev_end = 0.0
    for s, e in spans:
        if s > prev_end:
            gl = max(0.0, prev_end - pad); gr = min(track_end + pad, s + pad)
            if gr > gl:
                gaps.append((gl, gr))
        prev_end = max(prev_end, e)
    if track_end > prev_end:
        tl = max(0.0, prev_end - pad); tr = max(tl, track_end + pad)
        if tr > tl:
            gaps.append((tl, tr))
    if not gaps:
        gaps.append((0.0, track_end + pad))
    out = merge_windows(_gap_merge(gaps), min_width=0.0)
    if min_w > 0.0:
        out = [(s, e) for s, e in out if (e - s) >= min_w]
    return out


# ----------------------------- VPP multiplier ---------------------------------

def vpp_multiplier(vocal: np.ndarray, sr: int, hop: int, focus: Optional[Sequence[Tuple[float, float]]],
                   rms2048: Optional[np.ndarray] = None) -> Tuple[float, str]:
    """pure_vocal_pause_detector.py:1389-1532."""
    rms = L.rms(vocal, hop_length=hop)[0] if rms2048 is None else rms2048
    db = 20.0 * np.log10(rms + 1e-12)
    delta_db = get_config("pure_vocal_detection.pause_stats_adaptation.delta_db", 3.0)
    floor_pct = float(get_config("quality_control.enforce_quiet_cut.floor_percentile", 5))
    thr_db = np.percentile(db, floor_pct) + float(delta_db)
    mask = db > thr_db
    frame_sec = hop / float(sr)
    if focus:
        axis = np.arange(len(mask), dtype=np.float32) * frame_sec
        fm = np.zeros_like(mask, dtype=bool)
        for a, b in focus:
            if b <= a:
                continue
            fm |= (axis >= float(a)) & (axis <= float(b))
        if not np.any(fm):
            return 1.0, "VPP{focus_empty}"
        mask &= fm
    if not np.any(mask):
        return 1.0, "VPP{no_active_frames}"
    close_k = max(1, int(get_config("pure_vocal_detection.pause_stats_adaptation.morph_close_ms", 150) / 1000.0 / frame_sec))
    open_k = max(1, int(get_config("pure_vocal_detection.pause_stats_adaptation.morph_open_ms", 50) / 1000.0 / frame_sec))
    mask = remove_true_runs(fill_false_runs(mask, close_k), open_k)
    min_block = max(1, int(get_config("pure_vocal_detection.pause_stats_adaptation.sing_block_min_s", 2.0) / frame_sec))
    blocks = [(a, b) for a, b, v in _runs(mask) if v and (b - a) >= min_block]
    if not blocks:
        return 1.0, "VPP{no_singing_blocks}"
    # blocks are maximal True runs of the mask, so no False frame lies inside one: the rest
    # statistics of pure_vocal_pause_detector.py:1488-1506 are structurally empty.
    rests: List[float] = []
    total = sum(b - a for a, b in blocks)
    for a, b in blocks:
        for x, y, v in _runs(mask[a:b]):
            if v:
                continue
            span = y - x
            if span >= int(get_config("pure_vocal_detection.pause_stats_adaptation.interlude_min_s", 4.0) / frame_sec):
                continue
            rests.append(span * frame_sec)
    if not rests or total == 0:
        return 1.0, "VPP{no_rests}"
    mpd = float(np.median(rests)); p95 = float(np.percentile(rests, 95))
    pr = float(len(rests) / (total * frame_sec / 60.0)); rr = float(sum(rests) / (total * frame_sec))
    th = get_config("pure_vocal_detection.pause_stats_adaptation.classify_thresholds", {})
    slow = th.get("slow", {"mpd": 0.60, "p95": 1.20, "rr": 0.35}); fast = th.get("fast", {"mpd": 0.25, "pr": 18, "rr": 0.15})
    if (mpd >= slow.get("mpd", 0.6)) or (p95 >= slow.get("p95", 1.2)) or (rr >= slow.get("rr", 0.35)):
        cls = "slow"
    elif (mpd <= fast.get("mpd", 0.25)) and (pr >= fast.get("pr", 18)) and (rr <= fast.get("rr", 0.15)):
        cls = "fast"
    else:
        cls = "medium"
    mults = get_config("pure_vocal_detection.relative_threshold_adaptation.pause_stats_multipliers", {}) or {}
    return float(mults.get(cls, {"slow": 1.08, "medium": 1.0, "fast": 0.92}[cls])), f"VPP{{cls={cls}}}"


# ----------------------------- energy valleys ---------------------------------

def energy_valleys(vocal: np.ndarray, sr: int, peak_ratio: float, rms_ratio: float,
                   focus: Optional[Sequence[Tuple[float, float]]] = None,
                   rms_series: Optional[np.ndarray] = None, flat_series: Optional[np.ndarray] = None) -> List[Pause]:
    """pure_vocal_pause_detector.py:1096-1235."""
    frame_length = int(sr * 0.025); hop = int(sr * 0.01)
    rms = L.rms(vocal, frame_length=frame_length, hop_length=hop)[0] if rms_series is None else rms_series
    flat = L.spectral_flatness(vocal, hop_length=hop)[0] if flat_series is None else flat_series
    peak_thr = np.max(rms) * peak_ratio
    rms_thr = np.mean(rms) * rms_ratio
    thr = min(peak_thr, rms_thr)
    low = rms < thr
    times = L.frames_to_time(np.arange(len(rms)), sr=sr, hop_length=hop)
    if focus:
        valid = np.zeros_like(low, dtype=bool)
        for a, b in focus:
            if b <= a:
                continue
            valid |= (times >= a) & (times <= b)
        if np.any(valid):
            low &= valid
    w_len = get_config("pure_vocal_detection.valley_scoring.w_len", 0.6)
    w_quiet = get_config("pure_vocal_detection.valley_scoring.w_quiet", 0.4)
    w_flat = get_config("pure_vocal_detection.valley_scoring.w_flat", 0.1)
    out: List[Pause] = []
    runs = _runs(low)
    for a, b, v in runs:
        if not v:
            continue
        if b == len(low):           # run reaches the last frame: tail rule (:1218-1232)
            ps, pe = times[a], times[-1]
            if pe - ps >= 0.2:
                out.append(Pause(ps, pe, pe - ps, "energy_valley", 0.8, {"energy": 0.0, "threshold": thr}, (ps + pe) / 2))
            continue
        ps, pe = times[a], times[b]
        dur = pe - ps
        if dur < 0.2:
            continue
        fa = max(0, int(ps * sr / hop)); fb = min(len(rms), int(pe * sr / hop))
        if fa >= fb:
            continue
        energy = np.mean(rms[fa:fb])
        len_score = float(np.clip((dur - 0.20) / (1.50 - 0.20), 0.0, 1.0))
        quiet = float(np.clip(1.0 - float(energy / max(1e-12, thr)), 0.0, 1.0))
        hint = 0.5
        if flat is not None:
            sa = max(0, int(ps * sr / hop)); sb = min(len(flat), int(pe * sr / hop))
            if sb > sa:
                hint = float(np.clip(1.0 - float(np.mean(flat[sa:sb])), 0.0, 1.0))
        conf = max(0.1, min(0.99, (w_len * len_score) + (w_quiet * quiet) + (w_flat * hint)))
        out.append(Pause(ps, pe, dur, "energy_valley", conf, {"energy": energy, "threshold": thr}, (ps + pe) / 2))
    return out


def compress_pauses(pauses: List[Pause]) -> List[Pause]:
    """pure_vocal_pause_detector.py:503-547."""
    if not pauses:
        return pauses
    gap_s = float(get_config("pure_vocal_detection.valley_scoring.merge_close_ms", 80)) / 1000.0
    if gap_s > 0 and len(pauses) > 1:
        pauses = sorted(pauses, key=lambda p: p.start_time)
        merged: List[Pause] = []
        cur = pauses[0]
        for nxt in pauses[1:]:
            if nxt.start_time - cur.end_time <= gap_s:
                end = max(cur.end_time, nxt.end_time)
                cur = Pause(cur.start_time, end, end - cur.start_time, cur.pause_type,
                            max(cur.confidence, nxt.confidence), cur.features, 0.0, cur.quality_grade)
            else:
                merged.append(cur)
                cur = nxt
        merged.append(cur)
        pauses = merged
    cap = int(get_config("pure_vocal_detection.valley_scoring.max_raw_candidates", 1200))
    if len(pauses) > cap:
        pauses = sorted(pauses, key=lambda p: p.confidence, reverse=True)[:cap]
    return pauses


def total_valley_cap(pauses: List[Pause], duration_s: float) -> List[Pause]:
    """pure_vocal_pause_detector.py:461-501."""
    if not pauses:
        return pauses
    seg_min = float(get_config("quality_control.segment_min_duration", 4.0))
    if seg_min <= 0:
        seg_min = 4.0
    limit = max(1, int(math.floor(duration_s / seg_min)))
    if len(pauses) <= limit:
        return pauses

    def key(p: Pause):
        q = float(p.features.get("threshold", 0.0)) - float(p.features.get("energy", 0.0))
        if not np.isfinite(q):
            q = 0.0
        return (q, float(p.confidence))

    return sorted(sorted(pauses, key=key, reverse=True)[:limit], key=lambda p: p.start_time)


def mdd_enhance(pauses: List[Pause], cache: FeatureCache, focus: Optional[Sequence[Tuple[float, float]]],
                times: Optional[np.ndarray] = None) -> List[Pause]:
    """pure_vocal_pause_detector.py:1237-1368.  `times=None` is the cache branch (float32 arange*hop_s,
    `:1263`); the no-cache branch passes librosa.frames_to_time (float64, `:1274`)."""
    if not pauses:
        return pauses
    hop_s = float(cache.hop_s)
    rms = np.asarray(cache.rms_series, dtype=np.float32)
    flat = np.asarray(cache.spectral_flatness, dtype=np.float32)
    onset_frames = np.asarray(cache.onset_frames, dtype=np.int64)
    if times is None:
        times = np.arange(cache.frame_count(), dtype=np.float32) * hop_s
    rms_max = float(cache.rms_max)
    if rms_max <= 0:
        rms_max = 1.0
    fmask = None
    if focus:
        fmask = np.zeros_like(times, dtype=bool)
        for a, b in focus:
            if b <= a:
                continue
            fmask |= (times >= float(a)) & (times <= float(b))
        if not np.any(fmask):
            return pauses
    we = get_config("musical_dynamic_density.energy_weight", 0.7)
    ws = get_config("musical_dynamic_density.spectral_weight", 0.3)
    wo = get_config("musical_dynamic_density.onset_weight", 0.2)
    tm = get_config("musical_dynamic_density.threshold_multiplier", 0.3)
    hi = get_config("musical_dynamic_density.max_multiplier", 1.4)
    lo = get_config("musical_dynamic_density.min_multiplier", 0.6)
    out: List[Pause] = []
    for p in pauses:
        sf = int(np.argmin(np.abs(times - p.start_time))) if len(times) else 0
        ef = int(np.argmin(np.abs(times - p.end_time))) if len(times) else 0
        a = max(0, sf - 10); b = min(len(rms), ef + 10)
        if b <= a:
            out.append(p); continue
        if fmask is not None:
            rel = np.where(fmask[a:b])[0]
            if rel.size == 0:
                out.append(p); continue
            idx = rel + a
        else:
            idx = np.arange(a, b)
        e_score = float(np.mean(rms[idx])) / rms_max
        s_score = 1.0 - float(np.mean(flat[idx]))
        if onset_frames.size:
            om = (onset_frames >= idx[0]) & (onset_frames <= idx[-1])
            if fmask is not None:
                om &= fmask[np.clip(onset_frames, 0, len(fmask) - 1)]
            cnt = int(np.sum(om))
        else:
            cnt = 0
        o_score = min(1.0, cnt / 5.0) if cnt > 0 else 0.0
        score = (e_score * we) + (s_score * ws) + (o_score * wo)
        mult = max(lo, min(hi, 1.0 + (score * tm)))
        out.append(Pause(p.start_time, p.end_time, p.duration, f"{p.pause_type}_mdd", p.confidence * mult,
                         {**p.features, "mdd_score": score, "confidence_multiplier": mult}, p.cut_point, p.quality_grade))
    return out


def _local_rms(data: np.ndarray, win: int) -> np.ndarray:
    """pure_vocal_pause_detector.py:1047-1054 (float32 'same' convolution; numpy swaps operands when len<win)."""
    if data.size == 0:
        return np.empty(0, dtype=np.float32)
    if win <= 1:
        return np.abs(data.astype(np.float32))
    kernel = np.ones(win, dtype=np.float32) / float(win)
    return np.sqrt(np.maximum(np.convolve(data.astype(np.float32) ** 2, kernel, mode="same"), 1e-12))


def precise_cut_points(pauses: List[Pause], vocal: np.ndarray, sr: int) -> List[Pause]:
    """pure_vocal_pause_detector.py:1020-1094."""
    win = max(1, int(float(get_config("vocal_pause_splitting.local_rms_window_ms", 25)) / 1000.0 * sr))
    guard = max(0, int(float(get_config("vocal_pause_splitting.lookahead_guard_ms", 120)) / 1000.0 * sr))
    pct = float(get_config("vocal_pause_splitting.silence_floor_percentile", 5))
    allow = float(get_config("vocal_pause_splitting.silence_floor_allowance", 1.5))
    for p in pauses:
        a = max(0, int(round(p.start_time * sr))); b = min(len(vocal), int(round(p.end_time * sr)))
        if b - a <= 1:
            continue
        seg = vocal[a:b]
        cut = a + int(np.argmin(_local_rms(seg, win)))
        fallback = False
        if guard > 0:
            g_end = min(len(vocal), cut + guard)
            g = vocal[cut:g_end]
            if g.size > 0:
                cut = min(g_end - 1, cut + int(np.argmin(_local_rms(g, win))))
        floor_val = np.percentile(np.abs(seg), pct) if seg.size else 0.0
        if floor_val > 0.0 and np.abs(vocal[cut]) > floor_val * allow:
            cut = a + (b - a) // 2
            fallback = True
        p.cut_point = cut / float(sr)
        p.quality_grade = "B" if fallback else "A"
    return pauses


def simple_mdd_score(x: np.ndarray, sr: int) -> float:
    """pure_vocal_pause_detector.py:175-195."""
    try:
        r = L.rms(x, hop_length=512)[0]
        fl = L.spectral_flatness(x)[0]
        env = L.onset_strength(x, sr=sr, hop_length=512)
        onsets = L.onset_detect(env, sr=sr, hop_length=512)
        rate = len(onsets) / max(0.1, len(x) / sr)

        def nz(v):
            q10, q90 = np.quantile(v, 0.1), np.quantile(v, 0.9)
            if q90 - q10 < 1e-9:
                return 0.0
            return float(np.clip((np.mean(v) - q10) / (q90 - q10), 0, 1))

        return float(np.clip(0.5 * nz(r) + 0.3 * nz(fl) + 0.2 * float(np.clip(rate / 10.0, 0, 1)), 0, 1))
    except Exception:
        return 0.5


@dataclass
class DetectTrace:
    tempo: Optional[float] = None
    mdd: float = 0.0
    peak_ratio: float = 0.0
    rms_ratio: float = 0.0
    vpp_mul: float = 1.0
    n_raw: int = 0


def detect_pure_vocal_pauses(vocal: np.ndarray, sr: int, *, enable_mdd_enhancement: bool = False,
                             original_audio: Optional[np.ndarray] = None, feature_cache: Optional[FeatureCache] = None,
                             vad_segments: Optional[List[Dict[str, float]]] = None,
                             trace: Optional[DetectTrace] = None) -> List[Pause]:
    """pure_vocal_pause_detector.py:131-290, `enable_relative_energy_mode: true` branch (expert.yaml:38).

    The cache-without-VAD branch (`:166-167`) needs Silero on the full vocal; with Silero unavailable the
    reference's `_compute_focus_windows` returns [] (`:376-380`), which is what is restated here.
    """
    hop = int(sr * 0.01)
    cache = feature_cache if (feature_cache is not None and feature_cache.sr == sr and feature_cache.frame_count() > 0) else None
    focus: Optional[List[Tuple[float, float]]] = None
    if vad_segments:
        focus = focus_windows_from_vad(vad_segments, float(get_config("advanced_vad.focus_window_pad_s", 0.2)),
                                       float(get_config("advanced_vad.focus_window_min_width_s", 0.0)))
    elif cache is not None:
        focus = []
    ref = original_audio if original_audio is not None else vocal
    tempo: Optional[float]
    if cache is not None and cache.bpm_features is not None:
        tempo = float(getattr(cache.bpm_features, "main_bpm", 0.0) or 0.0)
    else:
        try:
            tempo = float(np.squeeze(np.asarray(L.beat_track(y=ref, sr=sr)[0])))
        except Exception:
            tempo = None
    if tempo is not None and tempo <= 0:
        tempo = None
    mdd = float(np.clip(cache.global_mdd, 0.0, 1.0)) if (cache is not None and cache.global_mdd is not None) else simple_mdd_score(ref, sr)
    th = resolve_threshold(get_config("pure_vocal_detection.peak_relative_threshold_ratio", 0.1),
                           get_config("pure_vocal_detection.relative_threshold_adaptation", {}), tempo, mdd)
    peak_ratio, rms_ratio = th.peak_ratio, th.rms_ratio
    mul = 1.0
    if get_config("pure_vocal_detection.pause_stats_adaptation.enable", True):
        try:
            mul, _ = vpp_multiplier(vocal, sr, hop, focus)
            mul = float(np.clip(mul, th.clamp_min, th.clamp_max))
            peak_ratio *= mul; rms_ratio *= mul
        except Exception:
            pass
    pauses = energy_valleys(vocal, sr, peak_ratio, rms_ratio, focus)
    n_raw = len(pauses)
    pauses = compress_pauses(pauses)
    pauses = total_valley_cap(pauses, float(len(vocal)) / float(sr))
    if enable_mdd_enhancement and (original_audio is not None or cache is not None):
        if cache is not None:
            pauses = mdd_enhance(pauses, cache, focus)
        else:
            adhoc = _adhoc_cache(original_audio, sr)
            pauses = mdd_enhance(pauses, adhoc, focus,
                                 times=L.frames_to_time(np.arange(adhoc.frame_count()), sr=sr, hop_length=adhoc.hop_length))
    if pauses:
        pauses = precise_cut_points(pauses, vocal, sr)
    if trace is not None:
        trace.tempo, trace.mdd, trace.peak_ratio, trace.rms_ratio, trace.vpp_mul, trace.n_raw = tempo, mdd, peak_ratio, rms_ratio, mul, n_raw
    return pauses


def _adhoc_cache(audio: np.ndarray, sr: int) -> FeatureCache:
    """pure_vocal_pause_detector.py:1266-1276 (no cache: 100 ms / 50 ms features of the mix)."""
    frame_length = int(sr * 0.1); hop = int(sr * 0.05)
    r = L.rms(audio, frame_length=frame_length, hop_length=hop)[0]
    fl = L.spectral_flatness(audio, hop_length=hop)[0]
    env = L.onset_strength(audio, sr=sr, hop_length=hop)
    of = L.onset_detect(env, sr=sr, hop_length=hop)
    # the no-cache branch takes times from librosa.frames_to_time (float64), the cache branch uses float32 arange*hop_s
    c = FeatureCache(sr, hop, float(hop) / float(sr), len(audio) / float(sr), r, fl, env, env, np.asarray(of),
                     float(np.max(r)) if r.size else 0.0, float(np.max(env)) if env.size else 0.0, None, None,
                     np.array([]), 0.0, np.array([]))
    return c


# ----------------------------- markers / no-vocal runs ---------------------------------

def vocal_presence_markers(vocal: np.ndarray, sr: int) -> Dict:
    """vocal_separator.py:460-529."""
    empty = {"vocal_presence_cut_points_sec": [], "vocal_presence_cut_points_samples": [],
             "vocal_presence_segments": [], "pure_music_segments": []}
    if sr <= 0 or vocal is None or len(vocal) == 0:
        return empty
    duration = float(len(vocal)) / sr
    thr_db = float(get_config("quality_control.segment_vocal_threshold_db", -50.0))
    music_min = float(get_config("quality_control.pure_music_min_duration", 0.0))
    hop = max(1, int(0.02 * sr)); frame_length = max(hop * 2, int(0.05 * sr))
    r = L.rms(vocal, frame_length=frame_length, hop_length=hop)[0]
    mask = (20.0 * np.log10(r + 1e-12)) > thr_db
    if mask.size == 0:
        return empty
    times = L.frames_to_time(np.arange(len(mask)), sr=sr, hop_length=hop)
    segs: List[Dict] = []
    start = 0.0
    runs = _runs(mask)
    for k, (a, b, v) in enumerate(runs):
        end = float(times[b]) if k + 1 < len(runs) else duration
        segs.append({"start": start, "end": end, "is_vocal": v})
        start = end
    clamp = lambda x: float(min(max(x, 0.0), duration))
    cuts = set()
    first = next((s for s in segs if s["is_vocal"] and s["end"] > s["start"]), None)
    if first is not None:
        cuts.add(clamp(first["start"] - 1.0))
    for prev, nxt in zip(segs, segs[1:]):
        if (not prev["is_vocal"]) and nxt["is_vocal"] and (prev["end"] - prev["start"]) >= music_min:
            c = clamp(nxt["start"] - 1.0)
            if c >= prev["start"]:
                cuts.add(c)
    last = next((s for s in reversed(segs) if s["is_vocal"] and s["end"] > s["start"]), None)
    if last is not None:
        cuts.add(clamp(last["end"] + 1.0))
    secs = sorted(c for c in cuts if 0.0 <= c <= duration)
    return {"vocal_presence_cut_points_sec": secs, "vocal_presence_cut_points_samples": [int(round(c * sr)) for c in secs],
            "vocal_presence_segments": segs, "pure_music_segments": [s for s in segs if not s["is_vocal"] and s["end"] > s["start"]]}


def no_vocal_runs(vocal: np.ndarray, sr: int, min_duration: float, rms2048: Optional[np.ndarray] = None) -> List[Tuple[float, float]]:
    """seamless_splitter.py:1706-1790."""
    hop = max(1, int(0.01 * sr))
    r = L.rms(vocal, hop_length=hop)[0] if rms2048 is None else rms2048
    db = 20.0 * np.log10(r + 1e-12)
    noise_pct = float(get_config("quality_control.enforce_quiet_cut.floor_percentile", 10))
    voice_pct = float(get_config("pure_vocal_detection.pause_stats_adaptation.voice_percentile_hint", 90))
    noise_db = float(np.percentile(db, np.clip(noise_pct, 0, 50)))
    voice_db = float(np.percentile(db, np.clip(voice_pct, 50, 100)))
    delta_db = float(get_config("pure_vocal_detection.pause_stats_adaptation.delta_db", 3.0))
    thr_db = max(noise_db + delta_db, 0.5 * (noise_db + voice_db))
    active = db > thr_db
    frame_sec = hop / float(sr)
    close_k = max(1, int(int(get_config("pure_vocal_detection.pause_stats_adaptation.morph_close_ms", 150)) / 1000.0 / frame_sec))
    open_k = max(1, int(int(get_config("pure_vocal_detection.pause_stats_adaptation.morph_open_ms", 50)) / 1000.0 / frame_sec))
    inactive = ~remove_true_runs(fill_false_runs(active, close_k), open_k)
    times = L.frames_to_time(np.arange(len(r)), sr=sr, hop_length=hop)
    spans: List[Tuple[float, float]] = []
    for a, b, v in _runs(inactive):
        if not v:
            continue
        st = float(times[a])
        en = float(times[b]) if b < len(inactive) else float(len(vocal) / float(sr))
        if en - st >= float(min_duration):
            spans.append((st, en))
    return spans


def estimate_confidence(vocal: np.ndarray, inst: Optional[np.ndarray], mix: np.ndarray) -> float:
    """enhanced_vocal_separator.py:490-501."""
    ve = float(np.mean(np.square(vocal))) if vocal.size else 0.0
    me = float(np.mean(np.square(mix))) if mix.size else 1e-8
    ratio = float(np.clip(ve / (me + 1e-8), 0.0, 1.0))
    if inst is not None and inst.size:
        bal = ve / (float(np.mean(np.square(inst))) + 1e-8)
        return float(np.clip(0.5 * ratio + 0.5 * np.clip(bal / (1.0 + bal), 0.0, 1.0), 0.0, 1.0))
    return float(np.clip(ratio, 0.0, 1.0))


# --------------------- dormant multi-feature branch ---------------------
# pure_vocal_pause_detector.py:268-281 when `enable_relative_energy_mode` is false: pyin F0, LPC-12 formant peak
# magnitudes, spectral centroid, low-third-bin magnitude ratio, ZCR, RMS -> candidate runs -> scored pauses.

@dataclass
class VocalFeatures:
    f0_contour: np.ndarray
    f0_confidence: np.ndarray
    formant_energies: List[np.ndarray]
    spectral_centroid: np.ndarray
    harmonic_ratio: np.ndarray
    zero_crossing_rate: np.ndarray
    rms_energy: np.ndarray


def extract_formants(audio: np.ndarray, sr: int, hop: int) -> List[np.ndarray]:
    """`_extract_formants` (`:959-1018`): 25 ms frames, 0.95 pre-emphasis, Burg LPC-12, |1/A| on 512 points,
    peaks >= 10 % of the maximum, the three lowest-frequency peaks' magnitudes."""
    import scipy.signal as signal
    frame_length = int(0.025 * sr)
    tracks: List[List[float]] = [[], [], []]
    for i in range(0, len(audio) - frame_length, hop):
        fr = audio[i:i + frame_length]
        fr = np.append(fr[0], fr[1:] - 0.95 * fr[:-1])
        try:
            a = L.lpc(fr, 12)
            w, h = signal.freqz(1, a, worN=512, fs=sr)
            mag = np.abs(h)
            peaks, _ = signal.find_peaks(mag, height=np.max(mag) * 0.1)
            if len(peaks) > 0:
                pm = mag[peaks]                                      # find_peaks returns ascending positions = ascending frequency
                for j in range(min(3, len(peaks))):
                    tracks[j].append(float(pm[j]))
            else:
                for j in range(3):
                    tracks[j].append(0.0)
        except Exception:
            for j in range(3):
                tracks[j].append(0.0)
    return [np.array(t) for t in tracks]


def harmonic_ratio_direct(audio: np.ndarray, hop: int) -> np.ndarray:
    """`_calculate_harmonic_ratio_direct` (`:936-957`): magnitude in the lowest third of the 1025 bins over the total."""
    mag = np.abs(L.stft(audio, hop_length=hop))
    nb = mag.shape[0]
    low = np.sum(mag[:nb // 3, :], axis=0)
    high = np.sum(mag[nb // 3:, :], axis=0)
    return low / ((low + high) + 1e-10)


def extract_vocal_features(audio: np.ndarray, sr: int) -> VocalFeatures:
    """`_extract_vocal_features` (`:410-459`)."""
    hop = int(sr * 0.01)
    f0, _, vp = L.pyin(audio, L.note_to_hz("C2"), L.note_to_hz("C7"), sr=sr, hop_length=hop)
    return VocalFeatures(
        f0_contour=f0, f0_confidence=vp, formant_energies=extract_formants(audio, sr, hop),
        spectral_centroid=L.spectral_centroid(audio, sr=sr, hop_length=hop)[0],
        harmonic_ratio=harmonic_ratio_direct(audio, hop),
        zero_crossing_rate=L.zero_crossing_rate(audio, hop_length=hop)[0],
        rms_energy=L.rms(audio, hop_length=hop)[0])


def detect_candidate_pauses(ft: VocalFeatures, sr: int) -> List[Tuple[int, int]]:
    """`_detect_candidate_pauses` (`:618-682`) in the absolute-dB mode the dormant branch runs with."""
    from scipy.ndimage import gaussian_filter1d
    hop = int(sr * 0.01)
    if get_config("pure_vocal_detection.enable_relative_energy_mode", False):
        peak = np.max(ft.rms_energy); avg = np.mean(ft.rms_energy)
        thr = min(peak * get_config("pure_vocal_detection.peak_relative_threshold_ratio", 0.1),
                  avg * get_config("pure_vocal_detection.rms_relative_threshold_ratio", 0.2))
        low_energy = ft.rms_energy < thr
    else:
        energy_db = L.amplitude_to_db(ft.rms_energy, ref=np.max)
        low_energy = energy_db < get_config("pure_vocal_detection.energy_threshold_db", -40)
    f0_missing = ft.f0_confidence < get_config("pure_vocal_detection.f0_drop_threshold", 0.7)
    pause_frames = gaussian_filter1d((low_energy & f0_missing).astype(float), sigma=3) > 0.5
    min_dur = get_config("pure_vocal_detection.breath_duration_range", [0.1, 0.3])[0]
    out: List[Tuple[int, int]] = []
    in_pause = False; start = 0
    for i, is_pause in enumerate(pause_frames):
        if is_pause and not in_pause:
            start = i; in_pause = True
        elif not is_pause and in_pause:
            if (i - start) * hop / sr >= min_dur:
                out.append((start, i))
            in_pause = False
    if in_pause and (len(pause_frames) - start) * hop / sr >= min_dur:
        out.append((start, len(pause_frames)))
    return out


def pause_interval_features(ft: VocalFeatures, s: int, e: int, sr: int) -> Dict:
    """`_extract_pause_interval_features` (`:733-806`)."""
    hop = int(sr * 0.01)
    ctx = int(0.5 * sr / hop)
    pre = max(0, s - ctx); post = min(len(ft.rms_energy), e + ctx)
    f0_drop = 0.0
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if pre < s:
                pre_f0 = np.nanmean(ft.f0_contour[pre:s]); pause_f0 = np.nanmean(ft.f0_contour[s:e])
                if not np.isnan(pre_f0) and not np.isnan(pause_f0) and pre_f0 > 0:
                    f0_drop = 1.0 - (pause_f0 / pre_f0)
    pre_e = np.mean(ft.rms_energy[pre:s]) if pre < s else 0
    pause_e = np.mean(ft.rms_energy[s:e])
    post_e = np.mean(ft.rms_energy[e:post]) if e < post else 0
    energy_drop = (pre_e - pause_e) / (pre_e + 1e-10)
    energy_rise = (post_e - pause_e) / (pause_e + 1e-10)
    centroid_shift = 0.0; harmonic_drop = 0.0
    if pre < s:
        pc = np.mean(ft.spectral_centroid[pre:s]); qc = np.mean(ft.spectral_centroid[s:e])
        centroid_shift = abs(pc - qc) / (pc + 1e-10)
        ph = np.mean(ft.harmonic_ratio[pre:s]); qh = np.mean(ft.harmonic_ratio[s:e])
        harmonic_drop = (ph - qh) / (ph + 1e-10)
    stab = []
    for tr in ft.formant_energies:
        if len(tr) > e:
            seg = tr[s:e]
            stab.append(1.0 - (np.std(seg) / (np.mean(seg) + 1e-10)))
    return {"f0_drop_rate": f0_drop, "energy_drop": energy_drop, "energy_rise": energy_rise, "centroid_shift": centroid_shift,
            "harmonic_drop": harmonic_drop, "formant_stability": np.mean(stab) if stab else 0.5,
            "pre_energy": pre_e, "pause_energy": pause_e, "post_energy": post_e}


def pause_confidence(f: Dict, duration: float, min_pause: float) -> float:
    """`_calculate_pause_confidence` (`:808-848`)."""
    breath = get_config("pure_vocal_detection.breath_duration_range", [0.1, 0.3])
    f0_score = min(1.0, f["f0_drop_rate"] / 0.5)
    energy_score = min(1.0, f["energy_drop"] / 0.7)
    spectral_score = min(1.0, f["centroid_shift"] / 0.3)
    if duration < breath[1]:
        dur_score = 0.3
    elif duration >= min_pause:
        dur_score = min(1.0, duration / 1.0)
    else:
        dur_score = 0.5
    conf = (get_config("pure_vocal_detection.f0_weight", 0.3) * f0_score
            + get_config("pure_vocal_detection.formant_weight", 0.25) * (1.0 - f.get("formant_stability", 0.5))
            + get_config("pure_vocal_detection.spectral_weight", 0.25) * spectral_score
            + get_config("pure_vocal_detection.duration_weight", 0.2) * dur_score)
    conf = conf * (0.7 + 0.3 * energy_score)
    return min(1.0, conf)


def merge_adjacent_pauses(pauses: List[Pause], merge_threshold: float = 0.3) -> List[Pause]:
    """`_merge_adjacent_pauses` (`:896-934`)."""
    if not pauses:
        return pauses
    pauses = sorted(pauses, key=lambda p: p.start_time)
    merged = []; cur = pauses[0]
    for nxt in pauses[1:]:
        if nxt.start_time - cur.end_time <= merge_threshold:
            cur = Pause(cur.start_time, nxt.end_time, nxt.end_time - cur.start_time, "true_pause",
                        max(cur.confidence, nxt.confidence), {**cur.features, **nxt.features})
        else:
            merged.append(cur); cur = nxt
    merged.append(cur)
    return merged


def detect_multifeature_pauses(vocal: np.ndarray, sr: int, *, include_breath_candidates: bool = False,
                               features: Optional[VocalFeatures] = None) -> List[Pause]:
    """The dormant branch end to end (`:268-281` + `_analyze_pause_features` `:684-731` + `_classify_and_filter`
    `:850-894`), followed by the shared precise cut points (`:286-287`)."""
    hop = int(sr * 0.01)
    min_pause = get_config("pure_vocal_detection.min_pause_duration", 0.5)
    breath = get_config("pure_vocal_detection.breath_duration_range", [0.1, 0.3])
    ft = features if features is not None else extract_vocal_features(vocal, sr)
    analyzed: List[Pause] = []
    for s, e in detect_candidate_pauses(ft, sr):
        st = s * hop / sr; et = e * hop / sr; dur = et - st
        pf = pause_interval_features(ft, s, e, sr)
        conf = pause_confidence(pf, dur, min_pause)
        kind = "breath" if dur <= breath[1] else ("true_pause" if dur >= min_pause else "uncertain")
        analyzed.append(Pause(st, et, dur, kind, conf, pf))
    hi = get_config("pure_vocal_detection.pause_confidence_threshold", 0.7)
    lo = get_config("pure_vocal_detection.breath_filter_threshold", 0.3)
    kept: List[Pause] = []
    for p in analyzed:
        if p.confidence >= hi:
            p.pause_type = "true_pause"; kept.append(p)
        elif p.confidence <= lo:
            p.pause_type = "breath"
            if include_breath_candidates:
                kept.append(p)
        elif p.duration >= min_pause:
            p.pause_type = "true_pause"; kept.append(p)
    kept = merge_adjacent_pauses(kept)
    if kept:
        kept = precise_cut_points(kept, vocal, sr)
    return kept
