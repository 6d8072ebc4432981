"""Weighted scorer over normalised boundary features — mirrors the reference's
`src/audio_cut/cutting/phrase_boundary_scorer.py:15-87` (default weights, penalty keys, clamp)."""
from __future__ import annotations

import json
from dataclasses import replace
from pathlib import Path
from typing import Dict, Iterable, Mapping, Optional

from ..analysis.boundary_features import BoundaryFeatures
from .cut_candidate import CutCandidate

DEFAULT_BOUNDARY_WEIGHTS: Dict[str, float] = {
    "acoustic_pause": 0.35, "asr_gap": 0.20, "sentence_end": 0.15, "beat_affinity": 0.08, "mdd_affinity": 0.10,
    "breath": 0.12, "inside_word_penalty": 0.80, "singing_penalty": 0.50,
}
_PENALTIES = frozenset({"inside_word_penalty", "singing_penalty"})


class PhraseBoundaryScorer:
    def __init__(self, weights: Optional[Mapping[str, float]] = None) -> None:
        self.weights = dict(DEFAULT_BOUNDARY_WEIGHTS)
        if weights:
            self.weights.update({k: float(v) for k, v in weights.items()})

    @classmethod
    def from_config(cls, cfg: Optional[Mapping[str, object]] = None) -> "PhraseBoundaryScorer":
        if cfg is None:
            from ..config import get_config
            cfg = {"weights": get_config("phrase_boundary.weights", DEFAULT_BOUNDARY_WEIGHTS)}
        w = cfg.get("weights", DEFAULT_BOUNDARY_WEIGHTS) if isinstance(cfg, Mapping) else DEFAULT_BOUNDARY_WEIGHTS
        return cls(weights=w if isinstance(w, Mapping) else DEFAULT_BOUNDARY_WEIGHTS)

    def score(self, features: BoundaryFeatures) -> float:
        total = 0.0
        for name, value in features.to_dict().items():
            w = float(self.weights.get(name, 0.0))
            total = total - w * value if name in _PENALTIES else total + w * value
        return 0.0 if total < 0.0 else (1.0 if total > 1.0 else total)

    def score_candidate(self, candidate: CutCandidate, features: BoundaryFeatures) -> CutCandidate:
        reasons = list(candidate.reasons)
        if "vpbd_score" not in reasons:
            reasons.append("vpbd_score")
        return replace(candidate, score=self.score(features), features=features.to_dict(), reasons=reasons)


def write_candidate_debug_json(candidates: Iterable[CutCandidate], path) -> None:
    p = Path(path)
    p.parent.mkdir(parents=True, exist_ok=True)
    p.write_text(json.dumps({"candidates": [c.to_dict() for c in candidates]}, indent=2, ensure_ascii=False, default=float),
                 encoding="utf-8")


__all__ = ["PhraseBoundaryScorer", "DEFAULT_BOUNDARY_WEIGHTS", "write_candidate_debug_json"]
