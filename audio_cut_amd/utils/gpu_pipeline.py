"""Chunk planning and per-device pipeline context: the MI355X mirror of the reference's
`src/audio_cut/utils/gpu_pipeline.py` plug point (same `__all__`, same names / keyword names /
defaults / `to_meta()` key set, `gpu_pipeline.py:645-660,546-572`).

What differs underneath: streams are HIP streams (PyTorch-ROCm exposes them as `torch.cuda.Stream`),
device telemetry comes from `amdsmi` / `rocm-smi` instead of NVML / `nvidia-smi`
(`gpu_pipeline.py:191-269`), and `OrtExecutionConfig` / `ensure_ort_dependencies` remain only as
inert configuration shells so existing config mappings keep loading — this build has no ONNX Runtime.
"""
from __future__ import annotations

import logging
import shutil
import subprocess
import threading
from contextlib import contextmanager, nullcontext
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Sequence

logger = logging.getLogger(__name__)

try:
    import torch
except Exception:  # pragma: no cover
    torch = None  # type: ignore


@dataclass
class Streams:
    """HIP stream triple: separation / VAD / features (reference: gpu_pipeline.py:42-51)."""

    s_sep: Optional["torch.cuda.Stream"] = None
    s_vad: Optional["torch.cuda.Stream"] = None
    s_feat: Optional["torch.cuda.Stream"] = None

    def as_tuple(self) -> Sequence[Optional["torch.cuda.Stream"]]:
        return (self.s_sep, self.s_vad, self.s_feat)


@dataclass
class ChunkPlan:
    """One chunk of the schedule (reference: gpu_pipeline.py:54-84)."""

    index: int
    start_s: float
    end_s: float
    halo_left_s: float
    halo_right_s: float

    @property
    def duration_s(self) -> float:
        return max(0.0, self.end_s - self.start_s)

    @property
    def effective_start_s(self) -> float:
        return self.start_s + self.halo_left_s

    @property
    def effective_end_s(self) -> float:
        return self.end_s - self.halo_right_s

    def as_slice(self, sample_rate: int) -> slice:
        a = max(0, int(round(self.start_s * sample_rate)))
        return slice(a, max(a, int(round(self.end_s * sample_rate))))

    def halo_slices(self, sample_rate: int) -> tuple:
        left = max(0, int(round(self.halo_left_s * sample_rate)))
        right = max(0, int(round(self.halo_right_s * sample_rate)))
        return (slice(None, left if left > 0 else None), slice(-right if right > 0 else None, None))


def select_device(preferred: Optional[str] = None) -> str:
    """Parse "cuda:N" | "cuda" | "gpu:N" | "N" | "cpu" (reference: gpu_pipeline.py:87-130).
    On PyTorch-ROCm "cuda:N" addresses HIP device N."""
    if torch is None or not torch.cuda.is_available():
        return "cpu"
    text = (preferred or "cuda").strip().lower() or "cuda"
    if text in {"cpu", "none"}:
        return "cpu"
    if text in {"cuda", "gpu"}:
        try:
            return f"cuda:{torch.cuda.current_device()}"
        except Exception:  # pragma: no cover
            return "cuda:0"
    if text.startswith(("cuda:", "gpu:")):
        idx_text = text.partition(":")[2]
    elif text.isdigit():
        idx_text = text
    else:
        idx_text = "0"
    try:
        index = int(idx_text)
    except ValueError:
        index = 0
    count = torch.cuda.device_count()
    if count == 0:
        return "cpu"
    if index < 0 or index >= count:
        logger.warning("[GPU Pipeline] requested device cuda:%s does not exist, using cuda:0 (device_count=%s)", index, count)
        index = 0
    return f"cuda:{index}"


def create_streams(device: str, enable: bool = True) -> Streams:
    if torch is None or not enable or not device.startswith("cuda"):
        return Streams()
    dev = torch.device(device)
    return Streams(torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))


def record_event(stream, *, enable_timing: bool = False):
    if torch is None or stream is None:
        return None
    ev = torch.cuda.Event(blocking=False, enable_timing=enable_timing)
    ev.record(stream)
    return ev


def wait_event(stream, event) -> None:
    if torch is None or stream is None or event is None:
        return
    stream.wait_event(event)


def _parse_device_index(device: str) -> Optional[int]:
    if not device:
        return None
    device = device.lower()
    if device == "cuda":
        try:
            return torch.cuda.current_device() if torch is not None and torch.cuda.is_available() else None
        except Exception:  # pragma: no cover
            return 0
    if device.startswith("cuda:"):
        try:
            return int(device.partition(":")[2])
        except ValueError:
            return None
    return None


def _collect_device_metrics(device: str) -> Optional[Dict[str, float]]:
    """amdsmi first, `rocm-smi` second (the reference tries NVML then nvidia-smi, :262-269).
    Keys keep the `gpu_pipeline_` prefix so the manifest's `gpu` block has the same shape."""
    index = _parse_device_index(device)
    if index is None or index < 0:
        return None
    try:
        import amdsmi  # type: ignore
        amdsmi.amdsmi_init()
        try:
            handle = amdsmi.amdsmi_get_processor_handles()[index]
            act = amdsmi.amdsmi_get_gpu_activity(handle)
            vram = amdsmi.amdsmi_get_gpu_vram_usage(handle)
            return {
                "gpu_pipeline_amdsmi_gpu_util_percent": float(act.get("gfx_activity", 0) or 0),
                "gpu_pipeline_amdsmi_mem_util_percent": float(act.get("umc_activity", 0) or 0),
                "gpu_pipeline_amdsmi_mem_used_bytes": float(vram.get("vram_used", 0)) * 1024.0 * 1024.0,
                "gpu_pipeline_amdsmi_mem_total_bytes": float(vram.get("vram_total", 0)) * 1024.0 * 1024.0,
            }
        finally:
            amdsmi.amdsmi_shut_down()
    except Exception:
        pass
    exe = shutil.which("rocm-smi")
    if not exe:
        return None
    try:
        out = subprocess.run([exe, "-d", str(index), "--showuse", "--showmeminfo", "vram", "--csv"],
                             capture_output=True, text=True, check=True, timeout=10).stdout
        rows = [r for r in out.strip().splitlines() if r and not r.startswith("WARNING")]
        if len(rows) < 2:
            return None
        header = rows[0].split(","); vals = rows[1].split(",")
        row = dict(zip(header, vals))
        metrics: Dict[str, float] = {}
        for key, name in (("GPU use (%)", "gpu_pipeline_rocm_smi_gpu_util_percent"),
                          ("VRAM Total Used Memory (B)", "gpu_pipeline_rocm_smi_mem_used_bytes"),
                          ("VRAM Total Memory (B)", "gpu_pipeline_rocm_smi_mem_total_bytes")):
            if key in row:
                metrics[name] = float(row[key])
        return metrics or None
    except Exception:  # pragma: no cover
        return None


@dataclass
class OrtExecutionConfig:
    """Inert on this build (no ONNX Runtime); kept so `gpu_pipeline.ort.*` mappings still parse
    (reference: gpu_pipeline.py:271-309)."""

    graph_optimization_level: str = "basic"
    cudnn_conv_algo_search: str = "HEURISTIC"
    disable_trt: bool = True

    def providers(self, *, prefer: Optional[str] = None) -> List[object]:
        return ["HIPKernels+PyTorchROCm"]


def ensure_ort_dependencies() -> None:
    """No-op: there is no ORT DLL search path to prepare on ROCm (reference: gpu_pipeline.py:312-330)."""


def chunk_schedule(total_s: float, *, chunk_s: float = 10.0, overlap_s: float = 2.5, halo_s: float = 0.5) -> List[ChunkPlan]:
    """10 s chunks / 2.5 s overlap / 0.5 s halo on interior edges (reference: gpu_pipeline.py:333-375; the arithmetic - clamps,
    repeated addition of the stride, the 1 us end tolerance - is the reference's, so the schedule is bit-identical: tests/golden)."""
    total = max(0.0, float(total_s))
    length = max(0.1, float(chunk_s))
    lap = max(0.0, min(float(overlap_s), length * 0.9))
    halo = max(0.0, min(float(halo_s), length * 0.5))
    if total <= length:
        return [ChunkPlan(0, 0.0, total, 0.0, 0.0)]
    step = length - lap
    if step <= 0:
        step = length
    horizon = total - 1e-6               # a chunk ending within a microsecond of the track's end is the last one
    plans: List[ChunkPlan] = []
    t = 0.0
    while t < horizon:
        stop = min(total, t + length)
        last = not (stop < horizon)
        plans.append(ChunkPlan(len(plans), t, stop, halo if plans else 0.0, 0.0 if last else halo))
        if last:
            break
        t += step
    return plans


@dataclass
class PinnedBufferPool:
    """Pinned host tensor cache (reference: gpu_pipeline.py:378-421)."""

    dtype: "torch.dtype"
    capacity: int = 2
    _buffers: List["torch.Tensor"] = field(default_factory=list)

    def __post_init__(self) -> None:
        if torch is None:
            self.capacity = 0

    def acquire(self, num_elements: int):
        want = int(num_elements) if torch is not None else 0
        if want <= 0:
            return None
        while self._buffers:             # newest first; a cached buffer that is too small is dropped on the way
            cand = self._buffers.pop()
            if cand.numel() >= want:
                return cand[:want]
        return torch.empty(want, dtype=self.dtype, pin_memory=torch.cuda.is_available())

    def acquire_view(self, shape: Sequence[int]):
        count = 1
        for extent in shape:
            count *= int(extent)
        flat = self.acquire(count)
        return flat.view(*shape) if flat is not None else None

    def release(self, tensor) -> None:
        if torch is not None and tensor is not None and len(self._buffers) < self.capacity:
            self._buffers.append(tensor.reshape(-1))

    def clear(self) -> None:
        del self._buffers[:]


@dataclass
class InflightLimiter:
    """Thread-safe in-flight counter (reference: gpu_pipeline.py:428-465)."""

    limit: int
    _condition: threading.Condition = field(default_factory=threading.Condition, init=False)
    _inflight: int = field(default=0, init=False)

    def __post_init__(self) -> None:
        self.limit = max(0, int(self.limit))

    @contextmanager
    def acquire(self, timeout: Optional[float] = None) -> Iterator[None]:
        gated = self.limit > 0           # limit 0: no gate at all
        if gated:
            self._enter(timeout)
        try:
            yield
        finally:
            if gated:
                self._leave()

    def _enter(self, timeout: Optional[float]) -> None:
        with self._condition:            # timeout None waits for a free slot for good
            if not self._condition.wait_for(lambda: self._inflight < self.limit, timeout=timeout):
                raise RuntimeError("inflight limit exceeded")
            self._inflight += 1

    def _leave(self) -> None:
        with self._condition:
            self._inflight = max(0, self._inflight - 1)
            self._condition.notify()


@dataclass
class PipelineConfig:
    enable: bool = False
    prefer_device: str = "cuda"
    chunk_s: float = 10.0
    overlap_s: float = 2.5
    halo_s: float = 0.5
    align_hop: int = 4096
    use_cuda_streams: bool = True
    prefetch_pinned_buffers: int = 2
    inflight_chunks_limit: int = 2
    strict_gpu: bool = False
    ort_config: OrtExecutionConfig = field(default_factory=OrtExecutionConfig)

    # field -> (accepted keys, first present wins; type): the reference's accepted spellings (gpu_pipeline.py:482-504)
    _MAPPING_KEYS = (("enable", ("enable",), bool), ("prefer_device", ("prefer_device",), str),
                     ("chunk_s", ("chunk_seconds", "chunk_s"), float), ("overlap_s", ("overlap_seconds", "overlap_s"), float),
                     ("halo_s", ("halo_seconds", "halo_s"), float), ("align_hop", ("align_hop", "align_hop_samples"), int),
                     ("use_cuda_streams", ("use_cuda_streams",), bool), ("prefetch_pinned_buffers", ("prefetch_pinned_buffers",), int),
                     ("inflight_chunks_limit", ("inflight_chunks_limit",), int), ("strict_gpu", ("strict_mode", "strict_gpu"), bool))
    _ORT_KEYS = (("graph_optimization_level", str), ("cudnn_conv_algo_search", str), ("disable_trt", bool))

    @classmethod
    def from_mapping(cls, mapping: Optional[dict]) -> "PipelineConfig":
        cfg = cls()
        if not mapping:
            return cfg
        for name, spellings, kind in cls._MAPPING_KEYS:
            for key in spellings:
                if key in mapping:
                    setattr(cfg, name, kind(mapping[key]))
                    break
        ort = mapping.get("ort", {}) if isinstance(mapping, dict) else {}
        for name, kind in cls._ORT_KEYS:
            if name in ort:
                setattr(cfg.ort_config, name, kind(ort[name]))
        return cfg


@dataclass
class PipelineContext:
    """Run-time context (reference: gpu_pipeline.py:507-577)."""

    device: str
    streams: Streams
    plans: List[ChunkPlan]
    pinned_pool: Optional[PinnedBufferPool]
    limiter: Optional[InflightLimiter]
    config: PipelineConfig = field(repr=False)
    use_streams: bool = False
    strict_gpu: bool = False
    mdx23_input: Optional[Dict[str, List[int]]] = None
    gpu_meta: Dict[str, object] = field(default_factory=dict)
    failures: List[Dict[str, str]] = field(default_factory=list)
    device_index: Optional[int] = None
    device_name: Optional[str] = None

    @property
    def enabled(self) -> bool:
        is_gpu = isinstance(self.device, str) and self.device.startswith("cuda")
        return bool(self.config.enable and is_gpu and self.use_streams and self.streams.s_sep)

    @contextmanager
    def acquire_inflight(self, timeout: Optional[float] = None) -> Iterator[None]:
        if self.limiter is None:
            yield
        else:
            with self.limiter.acquire(timeout=timeout):
                yield

    def register_mdx23_input(self, info: Dict[str, List[int]]) -> None:
        self.mdx23_input = info

    def mark_failure(self, stage: str, reason: str) -> None:
        self.failures.append({"stage": stage, "reason": reason})

    def to_meta(self) -> Dict[str, object]:
        """The manifest's `gpu` block: what the run recorded wins, the context fills in the rest (key set: gpu_pipeline.py:546-572)."""
        cfg = self.config
        derived = [("enabled", bool(cfg.enable)), ("used", bool(self.enabled)), ("device", self.device),
                   ("device_index", None if self.device_index is None else int(self.device_index)), ("device_name", self.device_name or None),
                   ("chunks", len(self.plans)), ("streams", bool(self.use_streams)),
                   ("inflight_limit", int(self.limiter.limit) if self.limiter else 0),
                   ("prefetch", int(self.pinned_pool.capacity) if self.pinned_pool else 0), ("align_hop", int(cfg.align_hop)),
                   ("config", {"chunk_seconds": float(cfg.chunk_s), "overlap_seconds": float(cfg.overlap_s), "halo_seconds": float(cfg.halo_s)}),
                   ("mdx23_input", self.mdx23_input or None), ("failures", list(self.failures) or None)]
        meta = dict(self.gpu_meta)
        for key, value in derived:
            if value is not None:
                meta.setdefault("gpu_pipeline_" + key, value)
        return meta

    def capture_device_metrics(self) -> None:
        snap = _collect_device_metrics(self.device)
        if snap:
            self.gpu_meta.update(snap)


def build_pipeline_context(duration_s: float, cfg: PipelineConfig) -> PipelineContext:
    """Reference: gpu_pipeline.py:580-642."""
    device = select_device(cfg.prefer_device)
    dev_ctx = nullcontext()
    if torch is not None and device.startswith("cuda"):
        try:
            torch.cuda.set_device(torch.device(device))
            dev_ctx = torch.cuda.device(torch.device(device))
        except Exception:  # pragma: no cover
            logger.warning("[GPU Pipeline] cannot switch to %s", device, exc_info=True)
    with dev_ctx:
        streams = create_streams(device, cfg.use_cuda_streams)
    plans = chunk_schedule(duration_s, chunk_s=cfg.chunk_s, overlap_s=cfg.overlap_s, halo_s=cfg.halo_s)
    pool = None
    if torch is not None and device.startswith("cuda") and cfg.prefetch_pinned_buffers > 0:
        pool = PinnedBufferPool(dtype=torch.float32, capacity=max(1, cfg.prefetch_pinned_buffers))
    limiter = InflightLimiter(limit=cfg.inflight_chunks_limit) if device.startswith("cuda") and cfg.inflight_chunks_limit > 0 else None
    index = _parse_device_index(device)
    name = None
    if index is not None and torch is not None and torch.cuda.is_available():
        try:
            name = torch.cuda.get_device_name(index)
        except Exception:  # pragma: no cover
            name = None
    ctx = PipelineContext(device=device, streams=streams, plans=plans, pinned_pool=pool, limiter=limiter, config=cfg,
                          use_streams=bool(device.startswith("cuda") and cfg.use_cuda_streams),
                          strict_gpu=bool(cfg.strict_gpu), device_index=index, device_name=name)
    ctx.gpu_meta = {"gpu_pipeline_enabled": bool(cfg.enable), "gpu_pipeline_device": device, "gpu_pipeline_chunks": len(plans)}
    if index is not None:
        ctx.gpu_meta["gpu_pipeline_device_index"] = index
    if name:
        ctx.gpu_meta["gpu_pipeline_device_name"] = name
    return ctx


__all__ = [
    "Streams", "ChunkPlan", "PipelineConfig", "PipelineContext", "PinnedBufferPool", "InflightLimiter",
    "OrtExecutionConfig", "ensure_ort_dependencies", "build_pipeline_context", "chunk_schedule",
    "create_streams", "record_event", "select_device", "wait_event",
]
