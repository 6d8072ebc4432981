"""Global cut planner: per-second pruning + O(n^2) dynamic programme over candidate times — mirrors the
reference's `src/audio_cut/cutting/global_cut_planner.py:16-232` (config defaults, value function, duration
score, strict-improvement tie rule, rescue grid, `planner_result_to_cut_points`, `apply_guard_shift_metadata`).
Scalar host logic over <= 2 candidates per second (keep on host)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Sequence, Tuple

from .cut_candidate import CutCandidate
from .refine import CutAdjustment, CutPoint


def _clamp01(v: float) -> float:
    return 0.0 if v < 0.0 else (1.0 if v > 1.0 else v)


@dataclass(frozen=True)
class GlobalCutPlannerConfig:
    hard_min_s: float = 2.0
    hard_max_s: float = 18.0
    target_min_s: float = 5.0
    target_max_s: float = 12.0
    duration_penalty_weight: float = 0.15
    vocal_risk_weight: float = 0.25
    beat_conflict_weight: float = 0.15
    max_candidates_per_second: float = 2.0
    rescue_enabled: bool = True


@dataclass(frozen=True)
class GlobalCutPlanResult:
    cut_times: List[float]
    selected_candidates: List[CutCandidate]
    suppressed_candidates: List[CutCandidate] = field(default_factory=list)
    rescue_points: List[float] = field(default_factory=list)
    feasible: bool = True
    metadata: Dict[str, object] = field(default_factory=dict)


class GlobalCutPlanner:
    def __init__(self, config: Optional[GlobalCutPlannerConfig] = None) -> None:
        self.config = config or GlobalCutPlannerConfig()

    # value of passing through a candidate
    def _value(self, c: Optional[CutCandidate]) -> float:
        if c is None:
            return 0.0
        risk = float(c.features.get("vocal_cut_risk", c.meta.get("vocal_cut_risk", 0.0)))
        conflict = float(c.features.get("beat_conflict", c.meta.get("beat_conflict", 0.0)))
        return c.score - self.config.vocal_risk_weight * _clamp01(risk) - self.config.beat_conflict_weight * _clamp01(conflict)

    def _length_score(self, seg_s: float) -> float:
        cfg = self.config
        if cfg.target_min_s <= seg_s <= cfg.target_max_s:
            return 0.1
        gap = (cfg.target_min_s - seg_s) if seg_s < cfg.target_min_s else (seg_s - cfg.target_max_s)
        return -cfg.duration_penalty_weight * gap / max(cfg.target_max_s, 1e-6)

    def _allowed(self, seg_s: float, duration_s: float) -> bool:
        if duration_s <= self.config.hard_min_s:
            return True
        return self.config.hard_min_s <= seg_s <= self.config.hard_max_s

    def plan(self, candidates: Sequence[CutCandidate], *, duration_s: float) -> GlobalCutPlanResult:
        duration_s = float(duration_s)
        if duration_s <= 0.0:
            return GlobalCutPlanResult([0.0], [], feasible=True, metadata={"planner": "empty", "selected_count": 0, "suppressed_count": 0})
        kept, dropped = self._prune(candidates, duration_s)
        path = self._dp(kept, duration_s)
        if path is None:
            if not self.config.rescue_enabled:
                return GlobalCutPlanResult([0.0, duration_s], [], suppressed_candidates=list(candidates), feasible=False,
                                           metadata={"planner": "dynamic_programming", "selected_count": 0,
                                                     "suppressed_count": len(candidates)})
            return self._rescue(duration_s, list(candidates))
        chosen, times = path
        chosen_ids = {id(c) for c in chosen}
        dropped.extend(c for c in kept if id(c) not in chosen_ids)
        return GlobalCutPlanResult(times, chosen, suppressed_candidates=sorted(dropped, key=lambda c: (c.t, c.score)), feasible=True,
                                   metadata={"planner": "dynamic_programming", "selected_count": len(chosen),
                                             "suppressed_count": len(dropped)})

    def _prune(self, candidates: Sequence[CutCandidate], duration_s: float) -> Tuple[List[CutCandidate], List[CutCandidate]]:
        per_second = max(1, int(math.floor(self.config.max_candidates_per_second)))
        buckets: Dict[int, List[CutCandidate]] = {}
        dropped: List[CutCandidate] = []
        for c in candidates:
            if c.t <= 0.0 or c.t >= duration_s:
                dropped.append(c)
            else:
                buckets.setdefault(int(math.floor(c.t)), []).append(c)
        kept: List[CutCandidate] = []
        for group in buckets.values():          # insertion order of first appearance, as the reference iterates
            ranked = sorted(group, key=self._value, reverse=True)
            kept.extend(ranked[:per_second])
            dropped.extend(ranked[per_second:])
        return sorted(kept, key=lambda c: c.t), dropped

    def _dp(self, candidates: Sequence[CutCandidate], duration_s: float) -> Optional[Tuple[List[CutCandidate], List[float]]]:
        nodes: List[Optional[CutCandidate]] = [None, *candidates, None]
        times = [0.0, *(c.t for c in candidates), duration_s]
        n = len(times)
        best = [-math.inf] * n
        parent = [-1] * n
        best[0] = 0.0
        for i in range(1, n):
            gain = self._value(nodes[i]) if nodes[i] is not None else 0.0
            for j in range(i):
                seg = times[i] - times[j]
                if not self._allowed(seg, duration_s):
                    continue
                total = best[j] + gain + self._length_score(seg)
                if total > best[i]:           # strict: the earliest best predecessor wins
                    best[i] = total
                    parent[i] = j
        if parent[-1] < 0:
            return None
        chosen: List[CutCandidate] = []
        path: List[float] = []
        i = n - 1
        while i >= 0:
            path.append(times[i])
            if nodes[i] is not None:
                chosen.append(nodes[i])
            i = parent[i]
            if i < 0 and path[-1] != 0.0:
                return None
        chosen.reverse()
        path.reverse()
        return chosen, path

    def _rescue(self, duration_s: float, suppressed: List[CutCandidate]) -> GlobalCutPlanResult:
        cfg = self.config
        count = max(1, int(math.ceil(duration_s / max(cfg.hard_max_s, 1e-6))))
        step = duration_s / float(count)
        if step < cfg.hard_min_s and count > 1:
            count = max(1, int(math.floor(duration_s / max(cfg.hard_min_s, 1e-6))))
            step = duration_s / float(count)
        times = [round(k * step, 9) for k in range(count + 1)]
        times[-1] = duration_s
        return GlobalCutPlanResult(times, [], suppressed_candidates=suppressed, rescue_points=times[1:-1], feasible=True,
                                   metadata={"planner": "rescue", "selected_count": 0, "suppressed_count": len(suppressed)})


def planner_result_to_cut_points(result: GlobalCutPlanResult) -> List[CutPoint]:
    return [CutPoint(t=c.t, score=c.score, kind=c.source.value) for c in result.selected_candidates]


def apply_guard_shift_metadata(result: GlobalCutPlanResult, adjustments: Sequence[CutAdjustment]) -> GlobalCutPlanResult:
    meta = dict(result.metadata)
    meta["guard_shift_ms_by_raw_time"] = {a.raw_time: a.guard_shift_ms for a in adjustments}
    meta["final_time_by_raw_time"] = {a.raw_time: a.final_time for a in adjustments}
    return replace(result, metadata=meta)


__all__ = ["GlobalCutPlanner", "GlobalCutPlannerConfig", "GlobalCutPlanResult", "planner_result_to_cut_points",
           "apply_guard_shift_metadata"]
