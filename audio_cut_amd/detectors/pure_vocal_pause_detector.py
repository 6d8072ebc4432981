"""PureVocalPauseDetector on HIP kernels — drop-in for the live (energy-valley) branch of the
reference's `src/vocal_smart_splitter/core/pure_vocal_pause_detector.py`
(`detect_pure_vocal_pauses(vocal_audio, enable_mdd_enhancement=False, original_audio=None,
feature_cache=None, vad_segments=None, include_breath_candidates=False) -> List[PureVocalPause]`,
`:131-136`; dataclass `:50-61`).

Device work per track (vocal stem resident in HBM):
  * RMS 1102/441 and STFT-2048 flatness at hop 441 (`:1113-1119`)      -> ac_frame_rms, ac_stft2048_features
  * per-pause 25 ms moving-RMS argmin + 120 ms look-ahead (`:1047-1078`) -> ac_pause_cut_points (one launch)
  * no-cache fallbacks (`:175-213`, `:1266-1276`)                        -> the same kernels on the mix
Host work: threshold derivation (`derive.py:287-336`), run-length scan over 24 k frames, merge /
cap / MDD weighting over a few hundred pauses.

`_estimate_vpp_multiplier` (`:1389-1532`): its singing blocks are maximal True-runs of the mask, so
the "rests inside a block" it counts never exist and every path returns 1.0 (`VPP{no_rests}` at
`:1508-1509` or an earlier `return 1.0`).  The multiplier is therefore the constant 1.0 here and the
RMS-2048 pass + percentile that could not influence it are not computed.
The dormant multi-feature branch (`enable_relative_energy_mode: false`) is not built in this round.
"""
from __future__ import annotations

import logging
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _native
from ..analysis.features_cache import TrackFeatureCache
from ..analysis.rhythm import BPMAnalyzer, onset_detect
from ..config import get_config

logger = logging.getLogger(__name__)


@dataclass
class PureVocalPause:
    start_time: float
    end_time: float
    duration: float
    pause_type: str
    confidence: float
    features: Dict
    cut_point: float = 0.0
    quality_grade: str = "B"
    is_valid: bool = True


class FocusWindowList(list):
    """List of gap windows that also compares equal to the speech windows (reference `:63-82`)."""

    def __init__(self, gap_windows, speech_windows):
        super().__init__(gap_windows)
        self._gap_windows = list(gap_windows)
        self._speech_windows = list(speech_windows) if speech_windows else list(gap_windows)

    def __eq__(self, other):  # pragma: no cover
        try:
            if other == self._speech_windows or other == self._gap_windows:
                return True
        except Exception:
            pass
        return super().__eq__(other)


# ---------------------------------------------------------------------------- derive.py:287-336
def _clamp(v, lo, hi):
    return max(lo, min(hi, v))


@dataclass
class VocalFeatures:
    """`pure_vocal_pause_detector.py:39-48` (same field names and order)."""
    f0_contour: np.ndarray
    f0_confidence: np.ndarray
    formant_energies: List[np.ndarray]
    spectral_centroid: np.ndarray
    harmonic_ratio: np.ndarray
    zero_crossing_rate: np.ndarray
    rms_energy: np.ndarray


@dataclass(frozen=True)
class AdaptStats:
    bpm: Optional[float] = None
    global_mdd: Optional[float] = None


@dataclass(frozen=True)
class DerivedThresholds:
    peak_ratio: float
    rms_ratio: float
    slow_multiplier: float
    fast_multiplier: float
    clamp_min: float
    clamp_max: float


def resolve_threshold(base_ratio: float, adapt_cfg: Dict, stats: AdaptStats) -> DerivedThresholds:
    adapt_cfg = adapt_cfg or {}
    bpm_cfg = adapt_cfg.get("bpm", {})
    cmin = float(adapt_cfg.get("clamp_min", 0.85)); cmax = float(adapt_cfg.get("clamp_max", 1.15))
    slow = float(bpm_cfg.get("slow_multiplier", 1.08)); fast = float(bpm_cfg.get("fast_multiplier", 0.92))
    peak = base_ratio
    rms = _clamp(base_ratio + 0.06, 0.05, 0.7)
    if stats.bpm and stats.bpm > 0:
        if stats.bpm < 90.0:
            peak *= _clamp(slow, cmin, cmax)
        elif stats.bpm > 140.0:
            peak *= _clamp(fast, cmin, cmax)
        peak = _clamp(peak, base_ratio * cmin, base_ratio * cmax)
    mdd_cfg = adapt_cfg.get("mdd", {})
    if stats.global_mdd is not None:
        peak *= _clamp(float(mdd_cfg.get("base", 1.0)) + float(mdd_cfg.get("gain", 0.2)) * stats.global_mdd, cmin, cmax)
    peak = _clamp(peak, 0.05, 0.6)
    rms = _clamp(rms, peak + 0.02, 0.72)
    return DerivedThresholds(peak, rms, slow, fast, cmin, cmax)


def resolve_min_pause(base_pause: float, adapt_strength: float, stats: AdaptStats) -> float:
    if not stats.bpm or stats.bpm <= 0:
        return base_pause
    adapt_strength = _clamp(adapt_strength, 0.0, 1.5)
    return max(0.3, base_pause + (-0.18 * adapt_strength * _clamp((stats.bpm - 110.0) / 110.0, -1.0, 1.0)))


def _bool_runs(mask: np.ndarray):
    m = np.asarray(mask, dtype=bool)
    if m.size == 0:
        return []
    cuts = np.flatnonzero(m[1:] != m[:-1]) + 1
    a = np.concatenate(([0], cuts)); b = np.concatenate((cuts, [m.size]))
    return [(int(s), int(e), bool(m[s])) for s, e in zip(a, b)]


class PureVocalPauseDetector:
    def __init__(self, sample_rate: int = 44100, ctx: Optional["_native.Context"] = None):
        self.sample_rate = sample_rate
        self.min_pause_duration = get_config("pure_vocal_detection.min_pause_duration", 0.5)
        self.hop_length = int(sample_rate * 0.01)
        self.frame_length = int(sample_rate * 0.025)
        self.n_fft = 2048
        self._ctx = ctx
        self._last_feature_cache: Optional[TrackFeatureCache] = None
        self._last_focus_windows: List[Tuple[float, float]] = []
        self.last_decision_margin: Optional[float] = None     # min |rms - thr| / thr over frames (parity diagnostics)

    def _context(self) -> "_native.Context":
        if self._ctx is None:
            self._ctx = _native.Context()
        return self._ctx

    def _dev(self, audio, dev):
        if dev is not None:
            return dev
        return self._context().to_device(np.ascontiguousarray(audio, dtype=np.float32))

    # ------------------------------------------------------------------------------------------
    def detect_pure_vocal_pauses(self, vocal_audio: np.ndarray, enable_mdd_enhancement: bool = False,
                                 original_audio: Optional[np.ndarray] = None,
                                 feature_cache: Optional[TrackFeatureCache] = None,
                                 vad_segments: Optional[List[Dict[str, float]]] = None,
                                 include_breath_candidates: bool = False, *, vocal_dev=None, original_dev=None) -> List[PureVocalPause]:
        sr = self.sample_rate
        cache = feature_cache if (isinstance(feature_cache, TrackFeatureCache) and feature_cache.sr == sr
                                  and feature_cache.frame_count() > 0) else None
        if cache is not None:
            self._last_feature_cache = cache
        focus: Optional[List[Tuple[float, float]]] = None
        if vad_segments:
            focus = self._focus_windows_from_vad_segments(
                vad_segments, pad_s=float(get_config("advanced_vad.focus_window_pad_s", 0.2)),
                min_width_s=float(get_config("advanced_vad.focus_window_min_width_s", 0.0)))
        elif cache is not None:
            # reference `:166-167,368-392`: Silero on the whole vocal; without Silero weights that call
            # yields no speech segments and therefore no focus restriction
            focus = []
        self._last_focus_windows = list(focus or [])
        vocal_dev = self._dev(vocal_audio, vocal_dev)
        n = int(vocal_dev.numel())
        if not get_config("pure_vocal_detection.enable_relative_energy_mode", False):
            # the multi-feature branch (`:268-281`; dormant under config/expert.yaml:38)
            pauses = self._detect_multifeature(vocal_dev, include_breath_candidates)
            if enable_mdd_enhancement and (original_audio is not None or original_dev is not None or cache is not None):
                pauses = self._apply_mdd_enhancement(pauses, original_audio, cache, focus, original_dev=original_dev)
            if pauses:
                pauses = self._calculate_precise_cut_points(pauses, vocal_dev)
            return pauses

        tempo: Optional[float]
        if cache is not None and cache.bpm_features is not None:
            tempo = float(getattr(cache.bpm_features, "main_bpm", 0.0) or 0.0)
        else:
            ref_dev = self._dev(original_audio, original_dev) if original_audio is not None or original_dev is not None else vocal_dev
            try:
                tempo = self._tempo_of(ref_dev)
            except _native.NativeError:
                raise
            except Exception:
                tempo = None
        if tempo is not None and tempo <= 0:
            tempo = None
        if cache is not None and cache.global_mdd is not None:
            mdd_value = float(np.clip(cache.global_mdd, 0.0, 1.0))
        else:
            ref_dev = self._dev(original_audio, original_dev) if original_audio is not None or original_dev is not None else vocal_dev
            mdd_value = self._mdd_score_simple(ref_dev)

        stats = AdaptStats(bpm=tempo, global_mdd=mdd_value)
        th = resolve_threshold(get_config("pure_vocal_detection.peak_relative_threshold_ratio", 0.1),
                               get_config("pure_vocal_detection.relative_threshold_adaptation", {}), stats)
        peak_ratio, rms_ratio = th.peak_ratio, th.rms_ratio
        strength = (th.slow_multiplier - th.fast_multiplier) / 0.16
        self.min_pause_duration = resolve_min_pause(float(get_config("pure_vocal_detection.min_pause_duration", self.min_pause_duration)),
                                                    strength, stats)
        if get_config("pure_vocal_detection.pause_stats_adaptation.enable", True):
            mul = float(np.clip(1.0, th.clamp_min, th.clamp_max))      # VPP multiplier is identically 1.0, see module doc
            peak_ratio *= mul; rms_ratio *= mul

        pauses = self._detect_energy_valleys(vocal_dev, peak_ratio, rms_ratio, focus)
        pauses = self._compress_pauses(pauses)
        pauses = self._apply_total_valley_cap(pauses, float(n) / float(sr))
        if enable_mdd_enhancement and (original_audio is not None or original_dev is not None or cache is not None):
            pauses = self._apply_mdd_enhancement(pauses, original_audio, cache, focus, original_dev=original_dev)
        if pauses:
            pauses = self._calculate_precise_cut_points(pauses, vocal_dev)
        return pauses

    # ---- multi-feature branch ------------------------------------------------
    def _extract_vocal_features(self, vocal_dev) -> "VocalFeatures":
        """`_extract_vocal_features` (`:410-459`): every series comes from a HIP kernel on the resident vocal
        (ac_yin_f0 -> ac_pyin_observe -> ac_pyin_viterbi, ac_lpc_formants, ac_stft2048_spectral, ac_zero_crossing_rate,
        ac_frame_rms); the host only assembles the formant tracks with the reference's append rule (`:1003-1012`)."""
        ctx = self._context()
        sr, hop = self.sample_rate, self.hop_length
        c2 = 440.0 * 2.0 ** ((36 - 69) / 12.0)        # librosa.note_to_hz('C2'), 'C7' (`:422-425`)
        c7 = 440.0 * 2.0 ** ((96 - 69) / 12.0)
        f0, _, voiced_prob = ctx.pyin(vocal_dev, sr, c2, c7, frame_length=2048, hop=hop)
        cnt, mag = ctx.lpc_formants(vocal_dev, int(0.025 * sr), hop, order=12, preemph=0.95)
        tracks = []
        for j in range(3):
            # a frame with peaks appends to tracks 0..min(3, peaks)-1 only; a frame without peaks appends 0.0 to all three
            keep = (cnt == 0) | (cnt > j)
            tracks.append(np.where(cnt[keep] == 0, 0.0, mag[keep, j]))
        centroid, ratio = ctx.stft2048_spectral(vocal_dev, sr, hop)
        zcr = ctx.zero_crossing_rate(vocal_dev, 2048, hop)
        rms = ctx.frame_rms(vocal_dev, 2048, hop).cpu().numpy()
        return VocalFeatures(f0_contour=f0, f0_confidence=voiced_prob, formant_energies=tracks, spectral_centroid=centroid,
                             harmonic_ratio=ratio, zero_crossing_rate=zcr, rms_energy=rms)

    def _detect_candidate_pauses(self, ft: "VocalFeatures") -> List[Tuple[int, int]]:
        """`:618-682`."""
        from scipy.ndimage import gaussian_filter1d
        sr, hop = self.sample_rate, self.hop_length
        if get_config("pure_vocal_detection.enable_relative_energy_mode", False):
            peak = np.max(ft.rms_energy); avg = np.mean(ft.rms_energy)
            thr = min(peak * get_config("pure_vocal_detection.peak_relative_threshold_ratio", 0.1),
                      avg * get_config("pure_vocal_detection.rms_relative_threshold_ratio", 0.2))
            low_energy = ft.rms_energy < thr
        else:
            r = np.abs(ft.rms_energy)                                  # librosa.amplitude_to_db(rms, ref=np.max), top_db 80
            power = np.square(r)
            db = 10.0 * np.log10(np.maximum(1e-10, power))
            db -= 10.0 * np.log10(np.maximum(1e-10, np.max(r) ** 2))
            db = np.maximum(db, db.max() - 80.0)
            low_energy = db < get_config("pure_vocal_detection.energy_threshold_db", -40)
        f0_missing = ft.f0_confidence < get_config("pure_vocal_detection.f0_drop_threshold", 0.7)
        frames = gaussian_filter1d((low_energy & f0_missing).astype(float), sigma=3) > 0.5
        min_dur = get_config("pure_vocal_detection.breath_duration_range", [0.1, 0.3])[0]
        edges = np.flatnonzero(np.diff(np.concatenate(([False], frames, [False])).astype(np.int8)))
        return [(int(a), int(b)) for a, b in zip(edges[0::2], edges[1::2]) if (b - a) * hop / sr >= min_dur]

    def _pause_interval_features(self, ft: "VocalFeatures", s: int, e: int) -> Dict:
        """`:733-806`."""
        import warnings
        ctx_frames = int(0.5 * self.sample_rate / self.hop_length)
        pre = max(0, s - ctx_frames); post = min(len(ft.rms_energy), e + ctx_frames)
        f0_drop = 0.0
        with warnings.catch_warnings(), np.errstate(all="ignore"):
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

    def _pause_confidence(self, f: Dict, duration: float) -> float:
        """`:808-848`."""
        breath = get_config("pure_vocal_detection.breath_duration_range", [0.1, 0.3])
        f0_score = min(1.0, f["f0_drop_rate"] / 0.5)
        energy_score = min(1.0, f["energy_drop"] / 0.7)
        spectral_score = min(1.0, f["centroid_shift"] / 0.3)
        if duration < breath[1]:
            dur_score = 0.3
        elif duration >= self.min_pause_duration:
            dur_score = min(1.0, duration / 1.0)
        else:
            dur_score = 0.5
        conf = (get_config("pure_vocal_detection.f0_weight", 0.3) * f0_score
                + get_config("pure_vocal_detection.formant_weight", 0.25) * (1.0 - f.get("formant_stability", 0.5))
                + get_config("pure_vocal_detection.spectral_weight", 0.25) * spectral_score
                + get_config("pure_vocal_detection.duration_weight", 0.2) * dur_score)
        conf = conf * (0.7 + 0.3 * energy_score)
        return min(1.0, conf)

    def _detect_multifeature(self, vocal_dev, include_breath_candidates: bool) -> List[PureVocalPause]:
        """`:268-281`: features -> candidate runs -> per-run scores (`:684-731`) -> classify / filter / merge (`:850-934`)."""
        sr, hop = self.sample_rate, self.hop_length
        ft = self._extract_vocal_features(vocal_dev)
        self._last_vocal_features = ft
        breath = get_config("pure_vocal_detection.breath_duration_range", [0.1, 0.3])
        analyzed: List[PureVocalPause] = []
        for s, e in self._detect_candidate_pauses(ft):
            st = s * hop / sr; et = e * hop / sr; dur = et - st
            pf = self._pause_interval_features(ft, s, e)
            conf = self._pause_confidence(pf, dur)
            kind = "breath" if dur <= breath[1] else ("true_pause" if dur >= self.min_pause_duration else "uncertain")
            analyzed.append(PureVocalPause(start_time=st, end_time=et, duration=dur, pause_type=kind, confidence=conf, features=pf))
        hi = get_config("pure_vocal_detection.pause_confidence_threshold", 0.7)
        lo = get_config("pure_vocal_detection.breath_filter_threshold", 0.3)
        kept: List[PureVocalPause] = []
        for p in analyzed:
            if p.confidence >= hi:
                p.pause_type = "true_pause"; kept.append(p)
            elif p.confidence <= lo:
                p.pause_type = "breath"
                if include_breath_candidates:
                    kept.append(p)
            elif p.duration >= self.min_pause_duration:
                p.pause_type = "true_pause"; kept.append(p)
        if not kept:
            return kept
        kept = sorted(kept, key=lambda p: p.start_time)              # `_merge_adjacent_pauses` (`:896-934`), threshold 0.3 s
        merged: List[PureVocalPause] = []
        cur = kept[0]
        for nxt in kept[1:]:
            if nxt.start_time - cur.end_time <= 0.3:
                cur = PureVocalPause(start_time=cur.start_time, end_time=nxt.end_time, duration=nxt.end_time - cur.start_time,
                                     pause_type="true_pause", confidence=max(cur.confidence, nxt.confidence),
                                     features={**cur.features, **nxt.features})
            else:
                merged.append(cur); cur = nxt
        merged.append(cur)
        return merged

    # ------------------------------------------------------------------------------------------
    def _tempo_of(self, wave_dev) -> float:
        """`librosa.beat.beat_track(y=ref_audio, sr=sr)` tempo only (`:207-208`)."""
        from ..analysis.rhythm import tempo_from_device
        ctx = self._context()
        _, mel = ctx.stft2048_features(wave_dev, 512, want_flat=False, want_mel=True)
        env = ctx.onset_strength(mel, 512, "median")
        if not bool(env.any().item()):
            return 0.0
        return tempo_from_device(ctx, env, self.sample_rate, 512)[0]

    def _mdd_score_simple(self, wave_dev) -> float:
        """`:175-195` — rms(2048/512), flatness(2048/512), onset rate at hop 512."""
        ctx = self._context()
        try:
            rms = ctx.frame_rms(wave_dev, 2048, 512).cpu().numpy()
            flat, mel = ctx.stft2048_features(wave_dev, 512, want_flat=True, want_mel=True)
            env = ctx.onset_strength(mel, 512, "mean").cpu().numpy()
            onsets = onset_detect(env, self.sample_rate, 512)
            rate = len(onsets) / max(0.1, wave_dev.numel() / self.sample_rate)

            def nz(v):
                q10, q90 = np.quantile(v, 0.1), np.quantile(v, 0.9)
                if q90 - q10 < 1e-9:
                    return 0.0
                return float(np.clip((np.mean(v) - q10) / (q90 - q10), 0, 1))

            return float(np.clip(0.5 * nz(rms) + 0.3 * nz(flat.cpu().numpy()) + 0.2 * float(np.clip(rate / 10.0, 0, 1)), 0, 1))
        except _native.NativeError:
            raise
        except Exception:
            return 0.5

    def _focus_windows_from_vad_segments(self, segments: Sequence[Dict[str, float]], *, pad_s: float = 0.2,
                                         min_width_s: float = 0.0):
        """`:293-366`."""
        if not segments:
            return []
        pad = max(0.0, float(pad_s)); min_w = max(0.0, float(min_width_s))
        merge_gap = float(get_config("advanced_vad.focus_merge_gap_s", 0.12))
        spans = []
        for seg in segments:
            try:
                s = float(seg.get("start", seg.get("start_time", 0.0)))
                e = float(seg.get("end", seg.get("end_time", s)))
            except Exception:
                continue
            if e > s:
                spans.append((s, e))
        if not spans:
            return []
        spans.sort(key=lambda it: it[0])
        track_end = max(e for _, e in spans)

        def gap_merge(ws):
            out = []
            for s, e in sorted(ws, key=lambda it: it[0]):
                if not out or s - out[-1][1] > merge_gap:
                    out.append((s, e))
                else:
                    out[-1] = (out[-1][0], max(out[-1][1], e))
            return out

        def min_width(ws):
            return ws if min_w <= 0.0 else [(s, e) for s, e in ws if (e - s) >= min_w]

        speech = []
        for s, e in spans:
            l, r = max(0.0, s - pad), min(track_end + pad, e + pad)
            if r > l:
                speech.append((l, r))
        speech_w = min_width(self._merge_windows(gap_merge(speech), min_width=0.0))
        gaps = []
        prev_end = 0.0
        for s, e in spans:
            if s > prev_end:
                l, r = max(0.0, prev_end - pad), min(track_end + pad, s + pad)
                if r > l:
                    gaps.append((l, r))
            prev_end = max(prev_end, e)
        if track_end > prev_end:
            l = max(0.0, prev_end - pad); r = max(l, track_end + pad)
            if r > l:
                gaps.append((l, r))
        if not gaps:
            gaps.append((0.0, track_end + pad))
        return FocusWindowList(min_width(self._merge_windows(gap_merge(gaps), min_width=0.0)), speech_w)

    @staticmethod
    def _merge_windows(windows, min_width: float = 0.0):
        merged = []
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

    # ------------------------------------------------------------------------------------------
    def _detect_energy_valleys(self, vocal_dev, peak_ratio: float, rms_ratio: float, focus_windows=None) -> List[PureVocalPause]:
        """`:1096-1235` — series on the GPU, run-length logic on the host."""
        sr = self.sample_rate
        ctx = self._context()
        frame_length = int(sr * 0.025); hop = int(sr * 0.01)
        rms = ctx.frame_rms(vocal_dev, frame_length, hop).cpu().numpy()
        try:
            flat_t, _ = ctx.stft2048_features(vocal_dev, hop, want_flat=True, want_mel=False)
            flat = flat_t.cpu().numpy()
        except _native.NativeError:
            raise
        peak_threshold = np.max(rms) * peak_ratio
        rms_threshold = np.mean(rms) * rms_ratio
        thr = min(peak_threshold, rms_threshold)
        low = rms < thr
        with np.errstate(divide="ignore", invalid="ignore"):
            self.last_decision_margin = float(np.min(np.abs(rms.astype(np.float64) - float(thr)) / max(float(thr), 1e-30)))
        times = (np.arange(len(rms)) * hop).astype(int) / float(sr)
        if focus_windows:
            valid = np.zeros_like(low, dtype=bool)
            for a, b in focus_windows:
                if b <= a:
                    continue
                valid |= (times >= a) & (times <= b)
            if np.any(valid):
                low &= valid
            else:
                logger.warning("focus windows cover no frame; scanning the whole track")
        w_len = get_config("pure_vocal_detection.valley_scoring.w_len", 0.6)
        w_quiet = get_config("pure_vocal_detection.valley_scoring.w_quiet", 0.4)
        w_flat = get_config("pure_vocal_detection.valley_scoring.w_flat", 0.1)
        pauses: List[PureVocalPause] = []
        for a, b, is_low in _bool_runs(low):
            if not is_low:
                continue
            if b == len(low):       # still "in pause" at the end of the track (`:1218-1232`)
                ps, pe = times[a], times[-1]
                if pe - ps >= 0.2:
                    pauses.append(PureVocalPause(ps, pe, pe - ps, "energy_valley", 0.8,
                                                 {"energy": 0.0, "threshold": thr}, cut_point=(ps + pe) / 2))
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
            sa = max(0, int(ps * sr / hop)); sb = min(len(flat), int(pe * sr / hop))
            if sb > sa:
                hint = float(np.clip(1.0 - float(np.mean(flat[sa:sb])), 0.0, 1.0))
            conf = max(0.1, min(0.99, (w_len * len_score) + (w_quiet * quiet) + (w_flat * hint)))
            pauses.append(PureVocalPause(ps, pe, dur, "energy_valley", conf, {"energy": energy, "threshold": thr},
                                         cut_point=(ps + pe) / 2))
        return pauses

    def _compress_pauses(self, pauses: List[PureVocalPause]) -> List[PureVocalPause]:
        """`:503-547`."""
        if not pauses:
            return pauses
        gap_s = float(get_config("pure_vocal_detection.valley_scoring.merge_close_ms", 80)) / 1000.0
        if gap_s > 0 and len(pauses) > 1:
            pauses = sorted(pauses, key=lambda p: p.start_time)
            merged: List[PureVocalPause] = []
            cur = pauses[0]
            for nxt in pauses[1:]:
                if nxt.start_time - cur.end_time <= gap_s:
                    end = max(cur.end_time, nxt.end_time)
                    cur = PureVocalPause(cur.start_time, end, end - cur.start_time, cur.pause_type,
                                         max(cur.confidence, nxt.confidence), cur.features, cut_point=0.0,
                                         quality_grade=cur.quality_grade)
                else:
                    merged.append(cur)
                    cur = nxt
            merged.append(cur)
            pauses = merged
        cap = int(get_config("pure_vocal_detection.valley_scoring.max_raw_candidates", 1200))
        if len(pauses) > cap:
            pauses = sorted(pauses, key=lambda p: p.confidence, reverse=True)[:cap]
        return pauses

    def _apply_total_valley_cap(self, pauses: List[PureVocalPause], duration_s: float) -> List[PureVocalPause]:
        """`:461-501`."""
        if not pauses:
            return pauses
        seg_min = float(get_config("quality_control.segment_min_duration", 4.0))
        if seg_min <= 0:
            seg_min = 4.0
        limit = max(1, int(math.floor(duration_s / seg_min)))
        if len(pauses) <= limit:
            return pauses

        def quiet_key(p: PureVocalPause):
            q = float(p.features.get("threshold", 0.0)) - float(p.features.get("energy", 0.0))
            if not np.isfinite(q):
                q = 0.0
            return (q, float(p.confidence))

        return sorted(sorted(pauses, key=quiet_key, reverse=True)[:limit], key=lambda p: p.start_time)

    def _apply_mdd_enhancement(self, pauses, original_audio, feature_cache, focus_windows, *, original_dev=None):
        """`:1237-1368`."""
        if not pauses:
            return pauses
        sr = self.sample_rate
        if feature_cache is not None:
            hop_s = float(feature_cache.hop_s)
            rms = np.asarray(feature_cache.rms_series, dtype=np.float32)
            flat = np.asarray(feature_cache.spectral_flatness, dtype=np.float32)
            onset_frames = np.asarray(feature_cache.onset_frames, dtype=np.int64)
            times = np.arange(feature_cache.frame_count(), dtype=np.float32) * hop_s
            rms_max = float(feature_cache.rms_max)
        else:
            ctx = self._context()
            x = self._dev(original_audio, original_dev)
            frame_length = int(sr * 0.1); hop = int(sr * 0.05)
            rms = ctx.frame_rms(x, frame_length, hop).cpu().numpy()
            flat_t, mel = ctx.stft2048_features(x, hop, want_flat=True, want_mel=True)
            flat = flat_t.cpu().numpy()
            strength = ctx.onset_strength(mel, hop, "mean").cpu().numpy()
            onset_frames = np.asarray(onset_detect(strength, sr, hop), dtype=np.int64)
            times = (np.arange(len(rms)) * hop).astype(int) / float(sr)
            rms_max = float(np.max(rms)) if rms.size else 0.0
        fmask = None
        if focus_windows:
            fmask = np.zeros_like(times, dtype=bool)
            for a, b in focus_windows:
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
        if rms_max <= 0:
            rms_max = 1.0
        out: List[PureVocalPause] = []
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
            out.append(PureVocalPause(p.start_time, p.end_time, p.duration, f"{p.pause_type}_mdd", p.confidence * mult,
                                      {**p.features, "mdd_score": score, "confidence_multiplier": mult},
                                      cut_point=p.cut_point, quality_grade=p.quality_grade))
        return out

    def _calculate_precise_cut_points(self, pauses: List[PureVocalPause], vocal_dev) -> List[PureVocalPause]:
        """`:1020-1094` — one `ac_pause_cut_points` launch for all pauses."""
        sr = self.sample_rate
        n = int(vocal_dev.numel())
        win = max(1, int(float(get_config("vocal_pause_splitting.local_rms_window_ms", 25)) / 1000.0 * sr))
        guard = max(0, int(float(get_config("vocal_pause_splitting.lookahead_guard_ms", 120)) / 1000.0 * sr))
        pct = float(get_config("vocal_pause_splitting.silence_floor_percentile", 5))
        allow = float(get_config("vocal_pause_splitting.silence_floor_allowance", 1.5))
        a = np.array([max(0, int(round(p.start_time * sr))) for p in pauses], dtype=np.int64)
        b = np.array([min(n, int(round(p.end_time * sr))) for p in pauses], dtype=np.int64)
        live = (b - a) > 1
        if not np.any(live):
            return pauses
        ctx = self._context()
        if win < 2:
            raise _native.NativeError("local_rms_window_ms below 2 samples is not supported by ac_pause_cut_points")
        cut, aux = ctx.pause_cut_points(vocal_dev, a[live], b[live], win, guard)
        k = 0
        for i, p in enumerate(pauses):
            if not live[i]:
                continue
            m = int(b[i] - a[i])
            c = int(cut[k]); zeros = int(aux[k, 0]); nonzero_at_cut = bool(aux[k, 1])
            k += 1
            # np.percentile(|segment|, pct) > 0  <=>  the order statistics it interpolates are not all zero
            pos = (pct / 100.0) * (m - 1)
            lo_i = int(math.floor(pos)); frac = pos - lo_i
            floor_positive = (zeros <= lo_i) or (frac > 0.0 and zeros <= lo_i + 1 and lo_i + 1 < m)
            fallback = False
            if allow == 0.0:
                exceeds = nonzero_at_cut                    # |x[cut]| > floor * 0.0
            else:
                floor_val = self._segment_percentile(vocal_dev, int(a[i]), int(b[i]), pct)
                exceeds = abs(float(vocal_dev[c].item())) > floor_val * allow
                floor_positive = floor_val > 0.0
            if floor_positive and exceeds:
                c = int(a[i]) + m // 2
                fallback = True
            p.cut_point = c / float(sr)
            p.quality_grade = "B" if fallback else "A"
        return pauses

    @staticmethod
    def _segment_percentile(vocal_dev, a: int, b: int, pct: float) -> float:
        """Only needed when `silence_floor_allowance` is non-zero (not the live configuration)."""
        import torch
        seg = vocal_dev[a:b].abs().to(torch.float64)
        return float(torch.quantile(seg, pct / 100.0).item()) if seg.numel() <= 16_000_000 else float(
            np.percentile(seg.cpu().numpy(), pct))


__all__ = ["PureVocalPauseDetector", "PureVocalPause", "FocusWindowList", "AdaptStats", "DerivedThresholds",
           "resolve_threshold", "resolve_min_pause"]
