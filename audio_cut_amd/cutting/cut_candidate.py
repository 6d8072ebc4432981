"""Candidate boundary record shared by the VPBD pool, scorer and planner — mirrors the reference's
`src/audio_cut/cutting/cut_candidate.py:14-51` (same enum values, field names, clamping and `to_dict`)."""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, List


_SOURCE_NAMES = ("acoustic_pause", "mdd_valley", "breath", "lyrics_gap", "sentence_end", "mvad_boundary", "beat", "rescue")
# str-valued enum, members named after their values in upper case (ACOUSTIC_PAUSE = "acoustic_pause", ...)
CandidateSource = Enum("CandidateSource", {name.upper(): name for name in _SOURCE_NAMES}, type=str, module=__name__)

_COPIED_FIELDS = (("reasons", list), ("features", dict), ("meta", dict))


@dataclass
class CutCandidate:
    """A boundary proposal at `t` seconds, `score` kept inside [0, 1], `source` coerced to the enum."""

    t: float
    score: float
    source: CandidateSource
    reasons: List[str] = field(default_factory=list)
    features: Dict[str, float] = field(default_factory=dict)
    meta: Dict[str, Any] = field(default_factory=dict)

    def __post_init__(self) -> None:
        self.t, score = float(self.t), float(self.score)
        score = score if score > 0.0 else 0.0            # NaN lands on 0, as min(1, max(0, nan)) does
        self.score = score if score < 1.0 else 1.0
        self.source = self.source if isinstance(self.source, CandidateSource) else CandidateSource(str(self.source))

    def to_dict(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {"t": self.t, "score": self.score, "source": self.source.value}
        out.update((name, kind(getattr(self, name))) for name, kind in _COPIED_FIELDS)
        return out


def adapt_legacy_acoustic_candidates(raw_candidates, *, source: CandidateSource = CandidateSource.ACOUSTIC_PAUSE,
                                     breath_score_scale: float = 0.6) -> List[CutCandidate]:
    """`(time, score[, meta])` tuples or CutPoint-likes -> candidates; `pause_type` starting with "breath"
    re-labels the source and scales the score (reference `candidate_adapters.py:14-48`)."""
    out: List[CutCandidate] = []
    for raw in raw_candidates:
        if hasattr(raw, "t") and hasattr(raw, "score") and not isinstance(raw, (tuple, list)):
            t, score, meta = raw.t, raw.score, {"legacy_kind": getattr(raw, "kind", "pause")}
        else:
            t, score = float(raw[0]), float(raw[1])
            meta = dict(raw[2]) if len(raw) > 2 and isinstance(raw[2], dict) else {}
        src = source
        if str(meta.get("pause_type", "")).startswith("breath"):
            if breath_score_scale <= 0.0:
                continue
            src = CandidateSource.BREATH
            score *= float(breath_score_scale)
        out.append(CutCandidate(t=t, score=score, source=src, reasons=["legacy_acoustic"], meta=meta))
    return out


__all__ = ["CandidateSource", "CutCandidate", "adapt_legacy_acoustic_candidates"]
