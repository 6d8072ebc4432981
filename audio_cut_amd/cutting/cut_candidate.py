"""Candidate boundary record shared by the VPBD pool, scorer and planner — mirrors the reference's
`src/audio_cut/cutting/cut_candidate.py:14-51` (same enum values, field names, clamping and `to_dict`)."""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, List


class CandidateSource(str, Enum):
    ACOUSTIC_PAUSE = "acoustic_pause"
    MDD_VALLEY = "mdd_valley"
    BREATH = "breath"
    LYRICS_GAP = "lyrics_gap"
    SENTENCE_END = "sentence_end"
    MVAD_BOUNDARY = "mvad_boundary"
    BEAT = "beat"
    RESCUE = "rescue"


@dataclass
class CutCandidate:
    t: float
    score: float
    source: CandidateSource
    reasons: List[str] = field(default_factory=list)
    features: Dict[str, float] = field(default_factory=dict)
    meta: Dict[str, Any] = field(default_factory=dict)

    def __post_init__(self) -> None:
        self.t = float(self.t)
        self.score = min(1.0, max(0.0, float(self.score)))
        if not isinstance(self.source, CandidateSource):
            self.source = CandidateSource(str(self.source))

    def to_dict(self) -> Dict[str, Any]:
        return {"t": self.t, "score": self.score, "source": self.source.value, "reasons": list(self.reasons),
                "features": dict(self.features), "meta": dict(self.meta)}


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
