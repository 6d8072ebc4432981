"""Derived QA figures of a SegmentManifest — mirrors the reference's `src/audio_cut/qa_report.py:14-241`
(`build_qa_report`, called from `_build_manifest`, `api.py:251`).  Pure bookkeeping over the manifest dict: nothing here
changes a cut.  The lyrics terms read `lyrics_alignment.timeline` when a caller supplies one (the ASR providers that would
fill it are out of scope; on the acoustic path they evaluate over empty lists and come out 0 / None like the reference's).
"""
from __future__ import annotations

from statistics import median
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

_EDGE = 1e-9


def _num(v: Any) -> Optional[float]:
    try:
        return None if v is None else float(v)
    except (TypeError, ValueError):
        return None


def _r12(v: Optional[float]) -> Optional[float]:
    return None if v is None else round(float(v), 12)


def _ratio(k: int, n: int) -> float:
    return (_r12(k / float(n)) or 0.0) if n > 0 else 0.0


def _mean(vals: Sequence[float]) -> Optional[float]:
    return _r12(sum(vals) / len(vals)) if vals else None


def _quantile(vals: Sequence[float], q: float) -> Optional[float]:
    """Linear interpolation between order statistics at (n-1)*q."""
    if not vals:
        return None
    s = sorted(vals)
    pos = (len(s) - 1) * q
    lo = int(pos)
    hi = min(lo + 1, len(s) - 1)
    return s[lo] * (1.0 - (pos - lo)) + s[hi] * (pos - lo)


def _mappings(seq: Any) -> List[Mapping[str, Any]]:
    return [x for x in (seq or []) if isinstance(x, Mapping)]


def _spans(items: Iterable[Mapping[str, Any]]) -> List[Tuple[float, float]]:
    out = []
    for it in items:
        a, b = _num(it.get("start_s")), _num(it.get("end_s"))
        if a is not None and b is not None and b > a:
            out.append((a, b))
    return out


def _has_source(item: Mapping[str, Any], name: str) -> bool:
    if str(item.get("source", "")) == name:
        return True
    meta = item.get("meta")
    pooled = meta.get("sources") if isinstance(meta, Mapping) else None
    if isinstance(pooled, Iterable) and not isinstance(pooled, (str, bytes)):
        return name in {str(s) for s in pooled}
    return False


def build_qa_report(manifest: Mapping[str, Any]) -> Dict[str, Any]:
    audio = manifest.get("audio")
    total_s = _num(audio.get("duration")) if isinstance(audio, Mapping) else None
    interior = lambda t: t is not None and t > _EDGE and not (total_s is not None and t >= total_s - _EDGE)

    seg_s = [d for d in (_num(s.get("duration")) for s in _mappings(manifest.get("segments"))) if d is not None]
    cuts_block = manifest.get("cuts")
    final = list(cuts_block.get("final", []) or []) if isinstance(cuts_block, Mapping) else []
    annotated = _mappings(final)                                    # cuts that carry planner / guard annotations
    cut_times = [t for t in (_num(c.get("t") if isinstance(c, Mapping) else c) for c in final) if interior(t)]
    inner_items = [c for c in annotated if interior(_num(c.get("t")))]

    lyrics = manifest.get("lyrics_alignment")
    timeline = lyrics.get("timeline") if isinstance(lyrics, Mapping) else None
    timeline = timeline if isinstance(timeline, Mapping) else {}
    words, sung = _mappings(timeline.get("words")), _mappings(timeline.get("vad_regions"))

    def inside(intervals: List[Tuple[float, float]]) -> float:
        return _ratio(sum(1 for t in cut_times if any(a < t < b for a, b in intervals)), len(cut_times)) if cut_times else 0.0

    def coverage() -> Optional[float]:
        if total_s is None or total_s <= 0.0:
            return None
        clipped = sorted((max(0.0, a), min(total_s, b)) for a, b in _spans(words))
        covered, edge = 0.0, None
        for a, b in clipped:                                        # union length of the word intervals
            if edge is None or a > edge:
                covered, edge = covered + (b - a), b
            elif b > edge:
                covered, edge = covered + (b - edge), b
        return _r12(covered / total_s) if clipped else 0.0

    def on_beat(c: Mapping[str, Any]) -> bool:
        feats = c.get("features")
        aff = _num(feats.get("beat_affinity")) if isinstance(feats, Mapping) else None
        return _has_source(c, "beat") or (aff is not None and aff >= 0.8)

    shifts = [abs(v) for v in (_num(c.get("guard_shift_ms")) for c in annotated) if v is not None]
    fallback = lyrics.get("fallback_reason") if isinstance(lyrics, Mapping) else None
    return {
        "segments_count": len(_mappings(manifest.get("segments"))),
        "median_segment_s": _r12(median(seg_s) if seg_s else None),
        "segment_5_15_pass_rate": _ratio(sum(1 for d in seg_s if 5.0 <= d <= 15.0), len(seg_s)),
        "cut_inside_word_rate": inside(_spans(words)),
        "cut_inside_singing_rate": inside(_spans(sung)),
        "avg_boundary_score": _mean([v for v in (_num(c.get("score")) for c in annotated) if v is not None]),
        "lyrics_coverage_ratio": coverage(),
        "asr_avg_confidence": _mean([v for v in (_num(w.get("confidence")) for w in words) if v is not None]),
        "guard_shift_p50_ms": _r12(_quantile(shifts, 0.50)),
        "guard_shift_p95_ms": _r12(_quantile(shifts, 0.95)),
        "breath_cut_ratio": _ratio(sum(1 for c in inner_items if _has_source(c, "breath")), len(cut_times)) if cut_times else 0.0,
        "beat_aligned_ratio": _ratio(sum(1 for c in inner_items if on_beat(c)), len(cut_times)) if cut_times else 0.0,
        "fallback_reason": None if fallback is None else str(fallback),
    }


__all__ = ["build_qa_report"]
