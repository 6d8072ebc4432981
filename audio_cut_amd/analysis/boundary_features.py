"""Normalised boundary features for VPBD scoring — mirrors the reference's
`src/audio_cut/analysis/boundary_features.py:16-167`.  The lyrics timeline is always empty on the
acoustic path (`vpbd_acoustic`; the ASR providers are out of scope): the four lyrics-derived
features (`asr_gap`, `sentence_end`, `inside_word_penalty`, `singing_penalty`) are therefore constants 0 here and their
evaluators are not built; the fields stay because the scorer's weight table names them."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List

import numpy as np


def _clamp01(v: float) -> float:
    return 0.0 if v < 0.0 else (1.0 if v > 1.0 else v)


@dataclass
class LyricsTimeline:
    """The slice of `audio_cut.lyrics.models.LyricsTimeline` the acoustic path touches: an empty timeline."""

    duration_s: float = 0.0
    source: str = "none"
    warnings: List[str] = field(default_factory=list)

    def to_dict(self) -> Dict:
        return {"duration_s": self.duration_s, "source": self.source, "words": [], "sentences": [], "vad_regions": [],
                "warnings": list(self.warnings)}


_FEATURE_NAMES = ("acoustic_pause", "asr_gap", "sentence_end", "inside_word_penalty", "singing_penalty", "beat_affinity",
                  "mdd_affinity", "breath", "vocal_cut_risk", "beat_conflict")


@dataclass
class BoundaryFeatures:
    acoustic_pause: float = 0.0
    asr_gap: float = 0.0
    sentence_end: float = 0.0
    inside_word_penalty: float = 0.0
    singing_penalty: float = 0.0
    beat_affinity: float = 0.0
    mdd_affinity: float = 0.0
    breath: float = 0.0
    vocal_cut_risk: float = 0.0
    beat_conflict: float = 0.0

    def __post_init__(self) -> None:
        for name in _FEATURE_NAMES:
            setattr(self, name, _clamp01(float(getattr(self, name))))

    def to_dict(self) -> Dict[str, float]:
        return {name: getattr(self, name) for name in _FEATURE_NAMES}


@dataclass
class BoundaryFeatureExtractor:
    timeline: LyricsTimeline
    beat_times: Iterable[float] = field(default_factory=list)
    mdd_times: Iterable[float] = field(default_factory=list)
    rms_series: Iterable[float] = field(default_factory=list)
    hop_s: float = 0.0
    high_confidence: float = 0.85
    affinity_tolerance_s: float = 0.12
    vocal_risk_window_s: float = 0.08

    def __post_init__(self) -> None:
        self.beat_times = list(self.beat_times)
        self.mdd_times = list(self.mdd_times)
        self.rms_series = list(self.rms_series)
        self._rms = np.asarray(self.rms_series, dtype=np.float32)
        self._rms_p99 = float(np.percentile(self._rms, 99)) if self._rms.size else 0.0

    def extract(self, t: float, *, acoustic_pause: float = 0.0) -> BoundaryFeatures:
        return BoundaryFeatures(
            acoustic_pause=acoustic_pause,            # asr_gap / sentence_end / inside_word / singing: no lyrics timeline on this path -> 0
            beat_affinity=self._affinity(t, self.beat_times), mdd_affinity=self._affinity(t, self.mdd_times),
            vocal_cut_risk=self._vocal_cut_risk(t), beat_conflict=self._beat_conflict(t))

    # -- acoustic terms ---------------------------------------------------------------------------------
    def _vocal_cut_risk(self, t: float) -> float:
        """local mean of the cached RMS around t / its 99th percentile (reference `:129-143`)."""
        rms = self._rms
        if rms.size == 0 or self.hop_s <= 0.0:
            return 0.0
        c = int(round(t / self.hop_s))
        half = max(1, int(round(self.vocal_risk_window_s / self.hop_s)))
        a, b = max(0, c - half), min(rms.size, c + half + 1)
        if a >= b or self._rms_p99 <= 1e-9:
            return 0.0
        return _clamp01(float(np.mean(rms[a:b])) / self._rms_p99)

    def _beat_conflict(self, t: float) -> float:
        if not self.beat_times:
            return 0.0
        nearest = min(abs(t - float(b)) for b in self.beat_times)
        return _clamp01(nearest / max(self.affinity_tolerance_s, 1e-6))

    def _affinity(self, t: float, anchors: Iterable[float]) -> float:
        best = 0.0
        for a in anchors:
            d = abs(t - float(a))
            if d <= self.affinity_tolerance_s:
                best = max(best, 1.0 - d / max(self.affinity_tolerance_s, 1e-6))
        return _clamp01(best)


__all__ = ["BoundaryFeatures", "BoundaryFeatureExtractor", "LyricsTimeline"]
