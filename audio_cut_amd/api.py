"""Public entry point — keeps the reference's `audio_cut.api.separate_and_segment` signature
(`src/audio_cut/api.py:31-45`): load -> (resample) -> separate -> detect -> finalize -> boundary policy -> export ->
SegmentManifest.

Modes `v2.2_mdd` (default when no intent arguments are given, `api.py:74-75`), `v2.1`, `vpbd_acoustic`.
Loader: PCM WAV / .npy, channel mean like `librosa.load(mono=True)`; a file whose rate differs from `audio.sample_rate`
is resampled on the GPU with `ac_resample_poly` (= scipy.signal.resample_poly; the reference's soxr_hq is not
available offline, so this row's parity definition is the scipy filter — SURVEY.md §8(f) row 2).
Export (`seamless_splitter.py:674-731`): `segment_NNN_{human|music}_D.D.wav` mix segments, `segments_vocal/..._vocal_D.D.wav`,
`<name>_<mode>_vocal_full_D.D.wav`, `<name>_<mode>_instrumental_D.D.wav`, all PCM_24 packed on the GPU (`ac_pack_pcm24`).
Manifest: `_build_manifest` (`api.py:178-263`) key for key, QA report included; only the lyrics attachment to segments (ASR
layer) is absent.  `separate_and_segment` returns the manifest like the reference's does.
"""
from __future__ import annotations

import json
import wave
from pathlib import Path
from typing import Any, Dict, Mapping, Optional, Sequence

import numpy as np

from . import config as _config
from .core.seamless_splitter import SeamlessSplitter


def load_audio_mono(path: str) -> tuple:
    """PCM16/24/32 WAV or .npy -> (mono float32 in [-1, 1], sample_rate).  Channel mean like `librosa.load(mono=True)`."""
    p = Path(path)
    if p.suffix.lower() == ".npy":
        arr = np.load(p)
        return (np.mean(arr, axis=0) if arr.ndim == 2 else arr).astype(np.float32), 44100
    with wave.open(str(p), "rb") as w:
        sr, ch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        data = v.astype(np.float32) / 8388608.0
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported WAV sample width {width}")
    data = data.reshape(-1, ch)
    return np.mean(data, axis=1).astype(np.float32) if ch > 1 else data[:, 0].copy(), sr


def _sha256(path: Path) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _rel(path: str, root: Path) -> str:
    try:
        return Path(path).resolve().relative_to(Path(root).resolve()).as_posix()
    except Exception:
        return Path(path).as_posix()


def _normalize_export_plan(export_types: Optional[Sequence[str]]) -> list:
    """`_normalize_export_plan` (`seamless_splitter.py:107-153`): the four artifact kinds, default all."""
    allowed = ("mix_segments", "vocal_segments", "full_vocal", "full_instrumental")
    if not export_types:
        return list(allowed)
    plan = []
    for item in export_types:
        key = str(item).strip().lower()
        if key in ("all", "*"):
            return list(allowed)
        if key not in allowed:
            raise ValueError(f"unknown export type {item!r}; choose from {allowed}")
        if key not in plan:
            plan.append(key)
    return plan


def separate_and_segment(*, input_uri: str, export_dir: str, mode: Optional[str] = None, segments: Optional[Any] = None,
                         alignment: Optional[Any] = None, device: Optional[str] = None,
                         export_types: Optional[Sequence[str]] = None, layout: Optional[Any] = None,
                         strict_gpu: Optional[bool] = None, export_manifest: bool = False,
                         manifest_filename: str = "SegmentManifest.json",
                         runtime_overrides: Optional[Dict[str, Any]] = None) -> Dict:
    """Returns the SegmentManifest dict (`api.py:115-131`); with `export_manifest` it is also written under `export_dir`
    and carries `manifest_path`.  The splitter's own result (`split_audio_seamlessly`'s dict) is `last_result()`."""
    if segments is not None or alignment is not None:
        raise NotImplementedError("intent routing (segments/alignment) belongs to the product layers outside the "
                                  "separate+detect hot path (SURVEY.md §2 #13,#15)")
    in_path = Path(input_uri).expanduser().resolve()
    if not in_path.exists():
        raise FileNotFoundError(f"input audio not found: {in_path}")
    out_dir = Path(export_dir).expanduser().resolve()
    out_dir.mkdir(parents=True, exist_ok=True)
    resolved_mode = mode or "v2.2_mdd"
    saved = _config.snapshot()
    try:
        overrides: Dict[str, Any] = {}
        if device:
            overrides["gpu_pipeline.prefer_device"] = device          # api.py:155-156
        if strict_gpu is not None:
            overrides["gpu_pipeline.strict_gpu"] = bool(strict_gpu)
        if layout:                                                       # api.py:161-166
            lay = dict(layout)
            overrides["segment_layout.enable"] = bool(lay.pop("enable", True))
            for k, v in lay.items():
                overrides[f"segment_layout.{k}"] = v
        overrides.update(dict(runtime_overrides or {}))                 # explicit dotted overrides are applied last (`:168-175`)
        _config.set_runtime_config(overrides)
        layout_cfg = dict(_config.get_config("segment_layout", {}) or {})
        sr = int(_config.get_config("audio.sample_rate", 44100))
        channels = int(_config.get_config("audio.channels", 1))
        result = _split_and_export(in_path, out_dir, resolved_mode, export_types, sr, device)
    finally:
        _config.restore(saved)
    global _LAST_RESULT
    _LAST_RESULT = result
    manifest = _build_manifest(result=result, input_path=in_path, export_dir=out_dir, mode=resolved_mode, sample_rate=sr,
                               channels=channels, layout_cfg=layout_cfg)
    if export_manifest:
        path = out_dir / manifest_filename
        path.write_text(json.dumps(manifest, ensure_ascii=False, indent=2, default=str), encoding="utf-8")
        manifest["manifest_path"] = path.as_posix()
    return manifest


_LAST_RESULT: Optional[Dict[str, Any]] = None


def last_result() -> Optional[Dict[str, Any]]:
    """The result dict behind the most recent manifest of this process (cut samples, file lists, per-phase timings)."""
    return _LAST_RESULT


def _split_and_export(in_path: Path, out_dir: Path, mode: str, export_types: Optional[Sequence[str]], sr: int,
                      device: Optional[str]) -> Dict[str, Any]:
    """The reference's `split_audio_seamlessly` for the modes built here (`seamless_splitter.py:171-253,270-760`): load,
    split, export, and the result dict `_build_manifest` reads."""
    import time
    from .utils.audio_export import ExportResult, PackedTrack, SegmentExporter
    t_start = time.time()
    audio, file_sr = load_audio_mono(str(in_path))
    splitter = SeamlessSplitter(sample_rate=sr, device=device)
    hip = splitter._context()
    audio_dev = None
    if file_sr != sr:
        audio_dev = hip.resample_poly(hip.to_device(audio), sr, file_sr)     # e.g. 48 kHz -> 44.1 kHz = up 147 / down 160
        audio = audio_dev.cpu().numpy()
    res = splitter.split_track(audio, mode=mode, audio_dev=audio_dev)
    single = bool(res.get("single_segment"))            # `_create_single_segment_result`: only the mix, no duration tag
    plan = _normalize_export_plan(export_types) if (export_types or not single) else ["mix_segments"]
    cuts = [int(c) for c in res.get("cuts_samples", res["sample_boundaries"])]
    spans = [tuple(sp) for sp in res.get("segment_spans", list(zip(cuts[:-1], cuts[1:])))]
    flags = list(res.get("segment_vocal_flags", [True] * len(spans)))
    durations = [(hi - lo) / float(sr) for lo, hi in spans]
    dmap = None if single else {i: d for i, d in enumerate(durations)}
    exp = ExportResult()
    exporter = SegmentExporter(sr)
    state = res.get("device_state") or {}
    if "mix_segments" in plan:
        mix_pk = PackedTrack(audio, sr, hip=hip, dev=state.get("mix", audio_dev))
        exp.mix_segment_files = exporter.export_spans(mix_pk, spans, str(out_dir), segment_is_vocal=flags, duration_map=dmap)
        exp.saved_files += exp.mix_segment_files
    vocal = res.get("vocal_track")
    voc_pk = PackedTrack(vocal, sr, hip=hip, dev=state.get("vocal")) if (vocal is not None and ("vocal_segments" in plan or "full_vocal" in plan)) else None
    if "vocal_segments" in plan and voc_pk is not None:
        exp.vocal_segment_files = exporter.export_spans(voc_pk, spans, str(out_dir), segment_is_vocal=flags, subdir="segments_vocal",
                                                        file_suffix="_vocal", duration_map=dmap)
        exp.saved_files += exp.vocal_segment_files
    if "full_vocal" in plan and voc_pk is not None:
        exp.full_vocal_file = exporter.export_full_track(voc_pk, out_dir / f"{in_path.stem}_{mode}_vocal_full_{len(vocal) / float(sr):.1f}")
        exp.saved_files.append(exp.full_vocal_file)
    inst = res.get("instrumental_track")
    if "full_instrumental" in plan and inst is not None:
        inst_pk = PackedTrack(inst, sr, hip=hip, dev=state.get("instrumental"))
        exp.full_instrumental_file = exporter.export_full_track(inst_pk, out_dir / f"{in_path.stem}_{mode}_instrumental_{len(inst) / float(sr):.1f}")
        exp.saved_files.append(exp.full_instrumental_file)
    out: Dict[str, Any] = {
        "success": True, "mode": mode, "method": f"pure_vocal_split_{mode}", "input_file": str(in_path), "output_dir": str(out_dir),
        "sample_rate": sr, "guard_boundaries_samples": [int(b) for b in res["sample_boundaries"]],
        "cut_points_samples": cuts, "cut_points_sec": [c / float(sr) for c in cuts],
        "num_segments": len(spans), "segment_durations": durations, "segment_vocal_flags": flags,
        "segment_labels": ["human" if f else "music" for f in flags],
        "segment_classification_debug": list(res.get("segment_classification_debug", [])),
        "segment_layout_applied": bool(res.get("segment_layout_applied", False)),
        "suppressed_cut_points_sec": list(res.get("suppressed_cut_points_sec", [])),
        "guard_adjustments": [dict(getattr(a, "__dict__", a)) for a in res.get("guard_adjustments", [])],
        "guard_shift_stats": dict(res.get("guard_shift_stats", SeamlessSplitter._guard_shift_stats([]))),
        "precision_guard_ok": bool(res.get("precision_guard_ok", True)),
        "precision_guard_threshold_ms": dict(res.get("precision_guard_threshold_ms", {})),
        "separation_confidence": res.get("separation_confidence"), "backend_used": res.get("backend_used"),
        "export_plan": sorted(plan), "saved_files": list(exp.saved_files), "mix_segment_files": list(exp.mix_segment_files),
        "vocal_segment_files": list(exp.vocal_segment_files), "full_vocal_file": exp.full_vocal_file,
        "full_instrumental_file": exp.full_instrumental_file,
        "timings": res.get("timings", {}), "processing_time": time.time() - t_start,
    }
    if res.get("note"):
        out["note"] = res["note"]
    if res.get("boundary_detection") is not None:
        out["boundary_detection"] = res["boundary_detection"]
        out["lyrics_alignment"] = res.get("lyrics_alignment")
    out.update(res.get("gpu_meta", {}))
    return out


def _to_ms(seconds: Any) -> Optional[int]:
    try:
        return None if seconds is None else int(round(float(seconds) * 1000.0))
    except (TypeError, ValueError):
        return None


def _annotated_cuts(result: Mapping[str, Any]) -> list:
    """`_build_final_cuts` (`api.py:307-372`): with a VPBD planner in the result, every cut that is a selected candidate
    (followed through `final_time_by_raw_time`) becomes `{t, score, source, features, reasons, meta[, guard_shift_ms]}`;
    otherwise `cuts.final` is the plain list of seconds."""
    times = list(result.get("cut_points_sec", []))
    vpbd = result.get("boundary_detection")
    if not isinstance(vpbd, Mapping):
        return times
    key = lambda v: round(float(v), 6)
    planner = vpbd.get("planner") if isinstance(vpbd.get("planner"), Mapping) else {}

    def keyed(table: Any) -> Dict[float, Any]:
        out: Dict[float, Any] = {}
        for k, v in (table.items() if isinstance(table, Mapping) else ()):
            try:
                out[key(k)] = v
            except (TypeError, ValueError):
                continue
        return out

    moved: Dict[float, float] = {}
    for raw, t in keyed(planner.get("final_time_by_raw_time")).items():
        try:
            moved[raw] = float(t)
        except (TypeError, ValueError):
            continue
    landed = lambda raw: key(moved.get(raw, raw))
    chosen: Dict[float, Mapping[str, Any]] = {}
    for cand in vpbd.get("selected", []) or []:
        if isinstance(cand, Mapping):
            try:
                chosen[landed(key(cand.get("t")))] = cand
            except (TypeError, ValueError):
                continue
    shifts = {landed(raw): ms for raw, ms in keyed(planner.get("guard_shift_ms_by_raw_time")).items()}
    if not chosen and not shifts:
        return times
    final: list = []
    for t in times:
        try:
            k, entry = key(t), {"t": float(t)}
        except (TypeError, ValueError):
            final.append(t)
            continue
        cand = chosen.get(k)
        if cand is not None:
            entry.update({"score": cand.get("score"), "source": cand.get("source"), "features": dict(cand.get("features") or {}),
                          "reasons": list(cand.get("reasons") or []), "meta": dict(cand.get("meta") or {})})
        if k in shifts:
            entry["guard_shift_ms"] = shifts[k]
        final.append(entry)
    return final


def _manifest_segments(result: Mapping[str, Any], export_dir: Path) -> list:
    """`_build_segments` (`api.py:266-304`) without the lyrics attachment (ASR layer, out of scope)."""
    times = list(result.get("cut_points_sec", []))
    durations = list(result.get("segment_durations", []))
    mix, voc = list(result.get("mix_segment_files", [])), list(result.get("vocal_segment_files", []))
    debug = list(result.get("segment_classification_debug", []))
    rows = []
    for i, label in enumerate(result.get("segment_labels", [])):
        start = times[i] if i < len(times) else sum(durations[:i])
        end = times[i + 1] if i + 1 < len(times) else start + (durations[i] if i < len(durations) else 0.0)
        row: Dict[str, Any] = {"id": f"{i + 1:04d}", "start": start, "end": end,
                               "duration": durations[i] if i < len(durations) else end - start, "label": label}
        if i < len(mix):
            row["mix_path"] = _rel(mix[i], export_dir)
        if i < len(voc):
            row["vocal_path"] = _rel(voc[i], export_dir)
        if i < len(debug) and debug[i]:
            row["debug"] = debug[i]
        rows.append(row)
    return rows


def _track_seconds(result: Mapping[str, Any], input_path: Path) -> Optional[float]:
    """`_estimate_duration` (`api.py:405-431`): the last cut, else the file header, else the summed segment durations."""
    times = result.get("cut_points_sec")
    if times:
        try:
            return float(times[-1])
        except (TypeError, ValueError):
            pass
    try:
        with wave.open(str(input_path), "rb") as w:
            if w.getnframes() and w.getframerate():
                return w.getnframes() / float(w.getframerate())
    except Exception:
        pass
    durations = result.get("segment_durations")
    return float(sum(durations)) if durations else None


def _build_manifest(*, result: Mapping[str, Any], input_path: Path, export_dir: Path, mode: str, sample_rate: int, channels: int,
                    layout_cfg: Mapping[str, Any]) -> Dict[str, Any]:
    """`_build_manifest` (`api.py:178-263`): same keys, same optional blocks, QA report included."""
    from .qa_report import build_qa_report
    input_path, export_dir = Path(input_path), Path(export_dir)
    artifacts: Dict[str, Any] = {}
    for name, field in (("music_segments", "mix_segment_files"), ("human_segments", "vocal_segment_files")):
        if result.get(field):
            artifacts[name] = [_rel(p, export_dir) for p in result[field]]
    for name, field in (("vocal_full", "full_vocal_file"), ("instrumental_full", "full_instrumental_file")):
        if result.get(field):
            artifacts[name] = _rel(result[field], export_dir)
    if result.get("saved_files"):
        artifacts["all"] = [_rel(p, export_dir) for p in result["saved_files"]]
    artifacts["output_dir"] = export_dir.as_posix()
    manifest: Dict[str, Any] = {
        "version": str(mode), "success": bool(result.get("success", False)), "job": {"source": input_path.as_posix()},
        "export_plan": result.get("export_plan") or [],
        "audio": {"sr": sample_rate, "channels": channels, "duration": _track_seconds(result, input_path),
                  "hash": f"sha256:{_sha256(input_path)}"},
        "layout_cfg": dict(layout_cfg) | {"applied": bool(result.get("segment_layout_applied", False))},
        "cuts": {"final": _annotated_cuts(result), "samples": result.get("cut_points_samples", []),
                 "suppressed": result.get("suppressed_cut_points_sec", [])},
        "segments": _manifest_segments(result, export_dir), "artifacts": artifacts,
        "guard": {"shift_stats": result.get("guard_shift_stats", {}), "adjustments": result.get("guard_adjustments", []),
                  "precision_ok": bool(result.get("precision_guard_ok", True)), "threshold_ms": result.get("precision_guard_threshold_ms", {})},
        "separation": {"backend": result.get("backend_used"), "confidence": result.get("separation_confidence")},
        "timings_ms": {"total": _to_ms(result.get("processing_time"))},
        "stats": {"num_segments": int(result.get("num_segments", 0))},
    }
    if result.get("note"):
        manifest["note"] = result["note"]
    for block in ("lyrics_alignment", "boundary_detection", "auto_profile", "intent"):
        if result.get(block) is not None:
            manifest[block] = result[block]
    gpu = {k: result[k] for k in result if str(k).startswith("gpu_pipeline_")}
    if gpu:
        manifest["gpu"] = gpu
    manifest["qa_report"] = build_qa_report(manifest)
    if result.get("bpm") is not None or result.get("method") == "smart_segment_v2":       # `librosa_onset` results, passed through
        manifest["smart_segmentation"] = {"method": result.get("method"), "bpm": result.get("bpm"),
                                          "bar_duration_s": result.get("bar_duration_s"), "density": result.get("density"),
                                          "silence_boundaries": result.get("silence_boundaries", [])}
    return manifest


__all__ = ["separate_and_segment", "load_audio_mono", "last_result"]
