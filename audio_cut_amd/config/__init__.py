"""Hot-path configuration: the effective values of the reference's `config/expert.yaml` merged under
`config/unified.yaml` (what `get_config` returns there, `src/vocal_smart_splitter/utils/config_manager.py:485-495`),
restricted to the keys SURVEY.md Appendix A lists for the separate+detect path.

`get_config(path, default)` has the reference's semantics: dotted lookup, `default` when the key is absent
(several call sites rely on their literal default because the YAML omits the key).  `set_runtime_config`
/ `reset_runtime_config` mirror `config_manager.py:497-509` (dotted-key overrides layered on top).
"""
from __future__ import annotations

import copy
from typing import Any, Dict

DEFAULTS: Dict[str, Any] = {
    "audio": {"sample_rate": 44100, "channels": 1},
    "gpu_pipeline": {
        "enable": True, "prefer_device": "cuda", "strict_gpu": False,
        "chunk_seconds": 10.0, "overlap_seconds": 2.5, "halo_seconds": 0.5, "align_hop": 4096,
        "use_cuda_streams": True, "prefetch_pinned_buffers": 2, "inflight_chunks_limit": 2,
    },
    "enhanced_separation": {
        "backend": "mdx23", "enable_fallback": True,
        "mdx23": {"model_filename": "Kim_Vocal_1.onnx", "output_type": "auto"},
    },
    "analysis": {"features_cache": {"device": "auto"}},
    "musical_dynamic_density": {
        "energy_weight": 0.5, "spectral_weight": 0.3, "onset_weight": 0.2,
        "threshold_multiplier": 0.2, "max_multiplier": 1.4, "min_multiplier": 0.6,
    },
    "advanced_vad": {
        "focus_window_pad_s": 0.2, "focus_window_min_width_s": 0.0, "silero_merge_gap_ms": 120.0,
        "focus_merge_gap_s": 0.12, "silero_length_bucket": 4096,
    },
    "pure_vocal_detection": {
        "enable": True, "min_pause_duration": 0.5, "breath_duration_range": [0.1, 0.3],
        "f0_weight": 0.3, "formant_weight": 0.25, "spectral_weight": 0.25, "duration_weight": 0.2,
        "enable_relative_energy_mode": True,
        "peak_relative_threshold_ratio": 0.26, "rms_relative_threshold_ratio": 0.3,
        "relative_threshold_adaptation": {
            "enable": True, "clamp_min": 0.85, "clamp_max": 1.15,
            "bpm": {"slow_multiplier": 1.08, "medium_multiplier": 1.0, "fast_multiplier": 0.92},
            "mdd": {"base": 1.0, "gain": 0.2},
            "pause_stats_multipliers": {"slow": 1.08, "medium": 1.0, "fast": 0.92},
        },
        "pause_stats_adaptation": {
            "enable": True, "delta_db": 3.0, "morph_close_ms": 150, "morph_open_ms": 50,
            "sing_block_min_s": 2.0, "interlude_min_s": 4.0,
            "classify_thresholds": {
                "slow": {"mpd": 0.6, "p95": 1.2, "rr": 0.35},
                "fast": {"mpd": 0.25, "pr": 18, "rr": 0.15},
            },
        },
        "valley_scoring": {
            "w_len": 0.7, "w_quiet": 0.3, "w_flat": 0.5, "use_weighted_nms": True,
            "merge_close_ms": 450, "max_raw_candidates": 1200, "max_kept_after_nms": 200,
        },
    },
    "vocal_pause_splitting": {
        "local_rms_window_ms": 25, "silence_floor_percentile": 5, "silence_floor_allowance": 0.0,
        "lookahead_guard_ms": 120, "head_offset": 0.0, "tail_offset": 0.0, "voice_threshold": 0.5,
    },
    "quality_control": {
        "min_split_gap": 1.2, "segment_min_duration": 2.0, "segment_max_duration": 18.0,
        "pure_music_min_duration": 6.0, "segment_vocal_threshold_db": -50.0, "segment_min_mix_piece": 2.0,
        "local_boundary_refine": {"enable": True, "search_radius_ms": 500, "window_ms": 5, "min_drop_db": 5.0},
        "enforce_quiet_cut": {
            "enable": True, "win_ms": 80, "guard_db": 1.5, "search_right_ms": 450,
            "floor_percentile": 0.5, "floor_db_override": None,
        },
    },
    "segment_layout": {"enable": True, "micro_merge_s": 2.0, "soft_min_s": 5.0, "soft_max_s": 12.0, "min_gap_s": 1.0, "beat_snap_ms": 50},
    "vpbd": {
        "enabled": True, "candidate_pool": "unified", "candidate_debug_json": True, "breath_score_scale": 0.6,
        "beat_candidates": {"enable": True, "bars_per_cut": 2, "base_score": 0.3},
    },
    "lyrics_alignment": {"enabled": False, "provider": "disabled", "strict": False},
    "phrase_boundary": {
        "word_edge_tolerance_ms": 60.0,
        "weights": {"acoustic_pause": 0.35, "asr_gap": 0.2, "sentence_end": 0.15, "beat_affinity": 0.08, "mdd_affinity": 0.1,
                    "breath": 0.12, "inside_word_penalty": 0.8, "singing_penalty": 0.5},
    },
    "global_planner": {
        "enable": True, "hard_min_s": 2.0, "hard_max_s": 18.0, "target_min_s": 5.0, "target_max_s": 12.0,
        "vocal_risk_weight": 0.25, "beat_conflict_weight": 0.15, "max_candidates_per_second": 2.0, "rescue_enabled": True,
    },
}

_runtime: Dict[str, Any] = {}
_MISSING = object()


def _lookup(tree: Dict[str, Any], path: str) -> Any:
    node: Any = tree
    for part in path.split("."):
        if isinstance(node, dict) and part in node:
            node = node[part]
        else:
            return _MISSING
    return node


def get_config(path: str, default: Any = None) -> Any:
    """The reference's `set_runtime_config` writes the override INTO the config tree (`config_manager.py:497-509`), so an
    override of a section (`{'segment_layout': {...}}`) replaces that subtree for every later child lookup, and an override
    of a child key set afterwards lands inside it.  `_runtime` keeps insertion order: the last write that covers `path`
    (the key itself, an ancestor, or descendants on top of either) wins."""
    parts = path.split(".")
    keys = list(_runtime)
    # the most recently written ancestor section (or the key itself) that is a dict-valued override, if any
    anc = [(keys.index(k), k) for k in (".".join(parts[:i]) for i in range(len(parts), 0, -1)) if k in _runtime]
    if anc:
        pos, key = max(anc)
        rest = path[len(key) + 1:] if len(path) > len(key) else ""
        node = _runtime[key]
        sub = _lookup(node, rest) if rest else node
        if rest and not isinstance(node, dict):
            sub = _MISSING
        value = copy.deepcopy(sub) if sub is not _MISSING else _MISSING
        prefix = path + "."
        later = {k[len(prefix):]: v for k, v in _runtime.items() if k.startswith(prefix) and keys.index(k) > pos}
        exact_later = path in _runtime and keys.index(path) > pos
        if exact_later:
            value = copy.deepcopy(_runtime[path])
        if later:
            value = {} if value is _MISSING or not isinstance(value, dict) else value
            for subkey, v in later.items():
                tgt = value
                sp = subkey.split(".")
                for p_ in sp[:-1]:
                    tgt = tgt.setdefault(p_, {})
                tgt[sp[-1]] = copy.deepcopy(v)
        return default if value is _MISSING else value
    # an override of a child key only
    base = _lookup(DEFAULTS, path)
    prefix = path + "."
    children = {k[len(prefix):]: v for k, v in _runtime.items() if k.startswith(prefix)}
    if base is _MISSING and not children:
        return default
    value = copy.deepcopy(base) if base is not _MISSING else {}
    for sub, v in children.items():
        node = value
        parts = sub.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = copy.deepcopy(v)
    return value


def set_runtime_config(overrides: Dict[str, Any]) -> None:
    """Dotted-key overrides (reference: config_manager.py:497-509).  A key written again moves to the end (last write wins)."""
    for k, v in (overrides or {}).items():
        _runtime.pop(str(k), None)
        _runtime[str(k)] = v


def reset_runtime_config() -> None:
    _runtime.clear()


def snapshot() -> Dict[str, Any]:
    return dict(_runtime)


def restore(saved: Dict[str, Any]) -> None:
    _runtime.clear()
    _runtime.update(saved)
