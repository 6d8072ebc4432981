"""Public entry point — keeps the reference's `audio_cut.api.separate_and_segment` signature
(`src/audio_cut/api.py:31-45`) for the hot path: load -> separate -> detect -> finalize.

Scope this round: modes `v2.2_mdd` (default when no intent arguments are given,
`api.py:74-75`) and `v2.1`; the result carries the integer cut points and the `gpu` metadata block.
Segment export / SegmentManifest writing (`api.py:115-129,178-263`), layout refinement and the
resampling loader (`audio_processor.py:45-49`) are the "next" rows of §8f: a track whose sample rate
is not 44.1 kHz is rejected here rather than resampled with an unpinned resampler.
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


def separate_and_segment(*, input_uri: str, export_dir: str, mode: Optional[str] = None, segments: Optional[Any] = None,
                         alignment: Optional[Any] = None, device: Optional[str] = None,
                         export_types: Optional[Sequence[str]] = None, layout: Optional[Any] = None,
                         strict_gpu: Optional[bool] = None, export_manifest: bool = False,
                         manifest_filename: str = "SegmentManifest.json",
                         runtime_overrides: Optional[Dict[str, Any]] = None) -> Dict:
    if segments is not None or alignment is not None or layout is not None:
        raise NotImplementedError("intent routing (segments/alignment/layout) belongs to the product layers outside the "
                                  "separate+detect hot path (SURVEY.md §2 #13,#15)")
    resolved_mode = mode or "v2.2_mdd"
    saved = _config.snapshot()
    try:
        overrides = dict(runtime_overrides or {})
        if device is not None:
            overrides["gpu_pipeline.prefer_device"] = device          # api.py:155-156
        if strict_gpu is not None:
            overrides["gpu_pipeline.strict_gpu"] = bool(strict_gpu)
        _config.set_runtime_config(overrides)
        sr = int(_config.get_config("audio.sample_rate", 44100))
        audio, file_sr = load_audio_mono(input_uri)
        if file_sr != sr:
            raise NotImplementedError(f"{input_uri}: {file_sr} Hz input needs the soxr_hq resampler of the loader "
                                      f"(SURVEY.md §8f next-2); supply {sr} Hz audio")
        splitter = SeamlessSplitter(sample_rate=sr, device=device)
        res = splitter.split_track(audio, mode=resolved_mode)
    finally:
        _config.restore(saved)
    bounds = [int(b) for b in res["sample_boundaries"]]
    out: Dict[str, Any] = {
        "success": True, "mode": resolved_mode, "input_file": input_uri, "sample_rate": sr,
        "cut_points_samples": bounds, "cut_points_sec": [b / float(sr) for b in bounds],
        "num_segments": max(0, len(bounds) - 1), "separation_confidence": res.get("separation_confidence"),
        "backend_used": res.get("backend_used"), "timings": res.get("timings", {}),
    }
    out.update(res.get("gpu_meta", {}))
    if export_manifest:
        Path(export_dir).mkdir(parents=True, exist_ok=True)
        manifest = {"version": "hot-path-1", "cuts": {"samples": bounds, "sample_rate": sr},
                    "gpu": {k: v for k, v in out.items() if k.startswith("gpu_pipeline_")}}
        (Path(export_dir) / manifest_filename).write_text(json.dumps(manifest, indent=1, default=str))
        out["manifest_path"] = str(Path(export_dir) / manifest_filename)
    return out


__all__ = ["separate_and_segment", "load_audio_mono"]
