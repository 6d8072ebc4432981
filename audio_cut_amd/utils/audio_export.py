"""Audio export — mirror of the reference's `src/vocal_smart_splitter/utils/audio_export.py` (`ensure_supported_format`,
`build_export_options`, `export_audio`) and `core/utils/segment_exporter.py` (`ExportResult`, `SegmentExporter`) for the
formats this build writes: WAV PCM_24 (the reference default, `audio_export.py:109-111`) and PCM_16.  SURVEY.md §8(f) row 4.

The float -> integer conversion runs on the GPU (`ac_pack_pcm24`) on the whole resident track once and is libsndfile's
clipping conversion - python-soundfile sets SFC_SET_CLIPPING on every file, so `soundfile.write` goes through pcm.c
`f2let_clip_array`: `lrintf(x * 2^31) >> 8`, saturating at 0x7FFFFF / 0x800000 (`>> 16` for PCM_16); segment files are byte slices of that buffer behind a 44-byte RIFF header.  MP3 needs
pydub + FFmpeg in the reference (`:114-135`) and is refused here.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

_DEFAULTS: Dict[str, Dict[str, object]] = {"wav": {"subtype": "PCM_24"}}


def ensure_supported_format(name: Optional[str]) -> str:
    key = (name or "wav").strip().lower()
    if key not in _DEFAULTS:
        raise ValueError(f"unsupported export format {name!r}: this build writes {sorted(_DEFAULTS)}")
    return key


def build_export_options(format_name: str, *overrides: Optional[Dict[str, object]]) -> Dict[str, object]:
    opts: Dict[str, object] = dict(_DEFAULTS[ensure_supported_format(format_name)])
    for o in overrides:
        if o:
            opts.update(o)
    return opts


def _export_path(base_path: Path, ext: str) -> Path:
    """`export_audio` (`:70-90`): a base name that already contains a dot keeps it (durations like `_12.3`)."""
    base_path = Path(base_path)
    return base_path.parent / f"{base_path.name}.{ext}" if base_path.suffix else base_path.with_suffix(f".{ext}")


def wav_header(n_frames: int, sample_rate: int, channels: int, bytes_per_sample: int) -> bytes:
    data = n_frames * channels * bytes_per_sample
    return (b"RIFF" + struct.pack("<I", 36 + data) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, sample_rate,
            sample_rate * channels * bytes_per_sample, channels * bytes_per_sample, 8 * bytes_per_sample) + b"data" + struct.pack("<I", data))


def _sndfile_clip_int32(x: np.ndarray) -> np.ndarray:
    """libsndfile pcm.c f2le{s,t}_clip_array (python-soundfile enables SFC_SET_CLIPPING on every file): the sample times 2^31
    in float32, saturated at 0x7FFFFFFF / -2^31, then lrintf (half to even).  The PCM word is its top 2 or 3 bytes."""
    s = np.asarray(x, dtype=np.float32) * np.float32(2147483648.0)
    v = np.rint(np.nan_to_num(s.astype(np.float64), nan=0.0, posinf=3e9, neginf=-3e9))
    v = np.where(s >= np.float32(2147483647.0), 2147483647.0, np.where(s <= np.float32(-2147483648.0), -2147483648.0, v))
    return v.astype(np.int64)


def pcm_bytes_host(audio: np.ndarray, subtype: str) -> Tuple[np.ndarray, int]:
    """Host conversion for arrays that never were on the device (tests, tiny inputs): same arithmetic as the kernel, which is
    libsndfile's clipping float -> PCM conversion: `lrintf(x * 2^31) >> 16` (PCM_16) / `>> 8` (PCM_24)."""
    x = np.asarray(audio, dtype=np.float32).reshape(-1)
    if subtype == "PCM_16":
        return (_sndfile_clip_int32(x) >> 16).astype("<i2").view(np.uint8), 2
    v = (_sndfile_clip_int32(x) >> 8).astype(np.int32)
    out = np.empty((x.size, 3), dtype=np.uint8)
    out[:, 0] = v & 0xFF; out[:, 1] = (v >> 8) & 0xFF; out[:, 2] = (v >> 16) & 0xFF
    return out.reshape(-1), 3


class PackedTrack:
    """A whole mono track converted once (on the GPU when a context and device tensor are given); slices are cheap."""

    def __init__(self, audio: np.ndarray, sample_rate: int, subtype: str = "PCM_24", *, hip=None, dev=None) -> None:
        self.sample_rate = int(sample_rate)
        self.subtype = subtype
        self.n = int(len(audio))
        if subtype == "PCM_24" and hip is not None and self.n > 0:
            d = dev if dev is not None else hip.to_device(np.ascontiguousarray(audio, dtype=np.float32))
            self.bytes, self.width = hip.pack_pcm24(d), 3
        else:
            self.bytes, self.width = pcm_bytes_host(audio, subtype)

    def write(self, path: Path, start: int = 0, end: Optional[int] = None) -> Path:
        end = self.n if end is None else end
        start = max(0, min(int(start), self.n)); end = max(start, min(int(end), self.n))
        with open(path, "wb") as fh:
            fh.write(wav_header(end - start, self.sample_rate, 1, self.width))
            fh.write(memoryview(self.bytes)[start * self.width: end * self.width])
        return Path(path)


def export_audio(audio: np.ndarray, sample_rate: int, base_path: Path, format_name: str, *, options: Optional[Dict[str, object]] = None) -> Path:
    key = ensure_supported_format(format_name)
    opts = build_export_options(key, options)
    path = _export_path(Path(base_path), key)
    arr = np.asarray(audio)
    if arr.ndim != 1:
        raise ValueError("this build exports mono tracks")
    PackedTrack(arr, sample_rate, str(opts.get("subtype", "PCM_24"))).write(path)
    return path


@dataclass
class ExportResult:
    saved_files: List[str] = field(default_factory=list)
    mix_segment_files: List[str] = field(default_factory=list)
    vocal_segment_files: List[str] = field(default_factory=list)
    full_vocal_file: Optional[str] = None
    full_instrumental_file: Optional[str] = None


class SegmentExporter:
    """`segment_exporter.py:25-110`: `segment_{index:03d}_{human|music}[_lib]{suffix}[_{duration:.1f}].{ext}`.

    `export_segments` / `export_full_track` keep the reference's array-in signatures; `export_spans` is the form the
    device path uses (one PackedTrack converted on the GPU, every piece a byte range of it)."""

    def __init__(self, sample_rate: int = 44100) -> None:
        self.sample_rate = sample_rate

    @staticmethod
    def _stem(i: int, n_samples: int, sample_rate: int, *, segment_is_vocal, lib_flags, lib_suffix: str, file_suffix: str,
              duration_map, index_offset: int, always_append_duration: bool) -> str:
        label = "human" if (bool(segment_is_vocal[i]) if i < len(segment_is_vocal) else True) else "music"
        lib = lib_suffix if (lib_flags is not None and i < len(lib_flags) and bool(lib_flags[i])) else ""
        seconds: Optional[float] = None
        if duration_map is not None and i in duration_map:
            seconds = max(0.0, float(duration_map[i]))
        elif always_append_duration:
            seconds = n_samples / float(sample_rate)
        tail = file_suffix if seconds is None else f"{file_suffix}_{seconds:.1f}"
        return f"segment_{i + index_offset:03d}_{label}{lib}{tail}"

    def export_segments(self, segments: Sequence[np.ndarray], output_dir: str, *, segment_is_vocal: Sequence[bool], export_format: str,
                        export_options: Optional[Dict[str, object]], lib_flags: Optional[Sequence[bool]] = None, lib_suffix: str = "_lib",
                        subdir: Optional[str] = None, file_suffix: str = "", duration_map: Optional[Dict[int, float]] = None,
                        index_offset: int = 1, always_append_duration: bool = False) -> List[str]:
        base = Path(output_dir) / subdir if subdir else Path(output_dir)
        base.mkdir(parents=True, exist_ok=True)
        saved: List[str] = []
        for i, piece in enumerate(segments):
            stem = self._stem(i, len(piece), self.sample_rate, segment_is_vocal=segment_is_vocal, lib_flags=lib_flags,
                              lib_suffix=lib_suffix, file_suffix=file_suffix, duration_map=duration_map, index_offset=index_offset,
                              always_append_duration=always_append_duration)
            saved.append(str(export_audio(piece, self.sample_rate, base / stem, export_format, options=export_options)))
        return saved

    def export_spans(self, track: PackedTrack, spans: Sequence[Tuple[int, int]], output_dir: str, *, segment_is_vocal: Sequence[bool],
                     lib_flags: Optional[Sequence[bool]] = None, lib_suffix: str = "_lib", subdir: Optional[str] = None,
                     file_suffix: str = "", duration_map: Optional[Dict[int, float]] = None, index_offset: int = 1,
                     always_append_duration: bool = False) -> List[str]:
        base = Path(output_dir) / subdir if subdir else Path(output_dir)
        base.mkdir(parents=True, exist_ok=True)
        saved: List[str] = []
        for i, (lo, hi) in enumerate(spans):
            stem = self._stem(i, hi - lo, track.sample_rate, segment_is_vocal=segment_is_vocal, lib_flags=lib_flags,
                              lib_suffix=lib_suffix, file_suffix=file_suffix, duration_map=duration_map, index_offset=index_offset,
                              always_append_duration=always_append_duration)
            saved.append(str(track.write(_export_path(base / stem, "wav"), lo, hi)))
        return saved

    def export_full_track(self, audio, output_base: Path, *, export_format: str = "wav",
                          export_options: Optional[Dict[str, object]] = None) -> str:
        """`audio`: a PackedTrack (device path) or a mono array (the reference's form)."""
        Path(output_base).parent.mkdir(parents=True, exist_ok=True)
        if isinstance(audio, PackedTrack):
            return str(audio.write(_export_path(Path(output_base), ensure_supported_format(export_format))))
        return str(export_audio(audio, self.sample_rate, Path(output_base), export_format, options=export_options))


__all__ = ["ensure_supported_format", "build_export_options", "export_audio", "ExportResult", "SegmentExporter", "PackedTrack",
           "wav_header", "pcm_bytes_host"]
