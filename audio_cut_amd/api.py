"""Public entry point — keeps the reference's `audio_cut.api.separate_and_segment` signature
(`src/audio_cut/api.py:31-45`): load -> (resample) -> separate -> detect -> finalize -> boundary policy -> export ->
SegmentManifest.

Modes `v2.2_mdd` (default when no intent arguments are given, `api.py:74-75`), `v2.1`, `vpbd_acoustic`.
Loader: PCM WAV / .npy, channel mean like `librosa.load(mono=True)`; a file whose rate differs from `audio.sample_rate`
is resampled on the GPU with `ac_resample_poly` (= scipy.signal.resample_poly; the reference's soxr_hq is not
available offline, so this row's parity definition is the scipy filter — SURVEY.md §8(f) row 2).
Export (`seamless_splitter.py:674-731`): `segment_NNN_{human|music}_D.D.wav` mix segments, `segments_vocal/..._vocal_D.D.wav`,
`<name>_<mode>_vocal_full_D.D.wav`, `<name>_<mode>_instrumental_D.D.wav`, all PCM_24 packed on the GPU (`ac_pack_pcm24`).
Manifest: the key set of `_build_manifest` (`api.py:178-263`) without the lyrics / QA-report layers.
"""
from __future__ import annotations

import json
import wave
from pathlib import Path
from typing import Any, Dict, Optional, Sequence

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
    import time
    from .utils.audio_export import ExportResult, PackedTrack, SegmentExporter
    if segments is not None or alignment is not None:
        raise NotImplementedError("intent routing (segments/alignment) belongs to the product layers outside the "
                                  "separate+detect hot path (SURVEY.md §2 #13,#15)")
    resolved_mode = mode or "v2.2_mdd"
    t_start = time.time()
    in_path = Path(input_uri)
    out_dir = Path(export_dir)
    saved = _config.snapshot()
    try:
        overrides = dict(runtime_overrides or {})
        if device is not None:
            overrides["gpu_pipeline.prefer_device"] = device          # api.py:155-156
        if strict_gpu is not None:
            overrides["gpu_pipeline.strict_gpu"] = bool(strict_gpu)
        if layout:                                                       # api.py:161-166
            lay = dict(layout)
            overrides["segment_layout.enable"] = bool(lay.pop("enable", True))
            for k, v in lay.items():
                overrides[f"segment_layout.{k}"] = v
        _config.set_runtime_config(overrides)
        sr = int(_config.get_config("audio.sample_rate", 44100))
        audio, file_sr = load_audio_mono(input_uri)
        splitter = SeamlessSplitter(sample_rate=sr, device=device)
        hip = splitter._context()
        audio_dev = None
        if file_sr != sr:
            audio_dev = hip.resample_poly(hip.to_device(audio), sr, file_sr)     # e.g. 48 kHz -> 44.1 kHz = up 147 / down 160
            audio = audio_dev.cpu().numpy()
        res = splitter.split_track(audio, mode=resolved_mode, audio_dev=audio_dev)
        layout_cfg = dict(_config.get_config("segment_layout", {}) or {})
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
        out_dir.mkdir(parents=True, exist_ok=True)
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
            exp.full_vocal_file = exporter.export_full_track(voc_pk, out_dir / f"{in_path.stem}_{resolved_mode}_vocal_full_{len(vocal) / float(sr):.1f}")
            exp.saved_files.append(exp.full_vocal_file)
        inst = res.get("instrumental_track")
        if "full_instrumental" in plan and inst is not None:
            inst_pk = PackedTrack(inst, sr, hip=hip, dev=state.get("instrumental"))
            exp.full_instrumental_file = exporter.export_full_track(inst_pk, out_dir / f"{in_path.stem}_{resolved_mode}_instrumental_{len(inst) / float(sr):.1f}")
            exp.saved_files.append(exp.full_instrumental_file)
    finally:
        _config.restore(saved)
    bounds = [int(b) for b in res["sample_boundaries"]]
    labels = ["human" if f else "music" for f in flags]
    out: Dict[str, Any] = {
        "success": True, "mode": resolved_mode, "method": f"pure_vocal_split_{resolved_mode}", "input_file": input_uri, "sample_rate": sr,
        "guard_boundaries_samples": bounds,
        "cut_points_samples": cuts, "cut_points_sec": [c / float(sr) for c in cuts],
        "num_segments": len(spans), "segment_durations": durations, "segment_vocal_flags": flags, "segment_labels": labels,
        "segment_layout_applied": bool(res.get("segment_layout_applied", False)),
        "suppressed_cut_points_sec": list(res.get("suppressed_cut_points_sec", [])),
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
    out.update(res.get("gpu_meta", {}))
    if export_manifest:
        segs = []
        csec = out["cut_points_sec"]
        for i, label in enumerate(labels):
            entry = {"id": f"{i + 1:04d}", "start": spans[i][0] / float(sr), "end": spans[i][1] / float(sr), "duration": durations[i], "label": label}
            if i < len(exp.mix_segment_files):
                entry["mix_path"] = _rel(exp.mix_segment_files[i], out_dir)
            if i < len(exp.vocal_segment_files):
                entry["vocal_path"] = _rel(exp.vocal_segment_files[i], out_dir)
            segs.append(entry)
        artifacts: Dict[str, Any] = {}
        if exp.mix_segment_files:
            artifacts["music_segments"] = [_rel(p, out_dir) for p in exp.mix_segment_files]
        if exp.vocal_segment_files:
            artifacts["human_segments"] = [_rel(p, out_dir) for p in exp.vocal_segment_files]
        if exp.full_vocal_file:
            artifacts["vocal_full"] = _rel(exp.full_vocal_file, out_dir)
        if exp.full_instrumental_file:
            artifacts["instrumental_full"] = _rel(exp.full_instrumental_file, out_dir)
        manifest: Dict[str, Any] = {
            "version": str(resolved_mode), "success": True, "job": {"source": in_path.as_posix()}, "export_plan": out["export_plan"],
            "audio": {"sr": sr, "channels": 1, "duration": len(audio) / float(sr), "hash": f"sha256:{_sha256(in_path)}"},
            "layout_cfg": dict(layout_cfg) | {"applied": out["segment_layout_applied"]},
            "cuts": {"final": csec, "samples": cuts, "suppressed": out["suppressed_cut_points_sec"]},
            "segments": segs, "artifacts": artifacts,
            "guard": {"adjustments": [getattr(a, "__dict__", a) for a in res.get("guard_adjustments", [])]},
            "separation": {"backend": out["backend_used"], "confidence": out["separation_confidence"]},
            "timings_ms": {"total": out["processing_time"] * 1000.0},
            "stats": {"num_segments": len(spans)},
        }
        if out.get("note"):
            manifest["note"] = out["note"]
        if out.get("boundary_detection") is not None:
            manifest["boundary_detection"] = out["boundary_detection"]
        gpu = {k: v for k, v in out.items() if k.startswith("gpu_pipeline_")}
        if gpu:
            manifest["gpu"] = gpu
        (out_dir / manifest_filename).write_text(json.dumps(manifest, indent=1, default=str))
        out["manifest_path"] = str(out_dir / manifest_filename)
    return out


__all__ = ["separate_and_segment", "load_audio_mono"]
