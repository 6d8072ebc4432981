"""EnhancedVocalSeparator — drop-in for the reference's
`src/vocal_smart_splitter/core/enhanced_vocal_separator.py` plug point
(`separate_for_detection(audio, *, gpu_context=None) -> SeparationResult`, `:155-205`; result fields
`:45-58`; `quality_metrics` keys consumed at `seamless_splitter.py:354-359,525-526`).

The reference's hot loop (`_separate_with_pipeline`, `:300-488`) walks the chunk plans one by one on
the host: pin/copy, `infer_chunk`, VAD, effective-region `+=` into three N-sample accumulators,
`ChunkFeatureBuilder.add_chunk`.  Here the mono track is uploaded once and

  * every sub-window of every chunk goes through `MDX23HipBackend.separate_track` in large batches
    (STFT-6144 -> TFC-TDF U-Net -> iSTFT -> stem algebra + uniform overlap-add, all in HBM);
  * the chunked VAD reads the per-chunk vocals from HBM (`SileroChunkVAD.process_chunk` per plan, as in
    `:412-417`, with the halo clipping / merge bookkeeping of `silero_chunk_vad.py` on the host);
  * `ChunkFeatureBuilder` evaluates all chunks in three launches on the resident mix;
  * vocal-presence markers (`vocal_separator.py:460-529`) and the confidence ratio (`:490-501`) come
    from `ac_frame_rms` / `ac_sum_squares` on the resident stems.

Failure contract (`:173-188`): the reference falls back to a CPU backend unless `strict_gpu`.  This
build has no CPU backend by design, so a failure is recorded in `gpu_pipeline_failures` and re-raised.
"""
from __future__ import annotations

import logging
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _native
from ..analysis.features_cache import ChunkFeatureBuilder, TrackFeatureCache
from ..config import get_config
from ..detectors.silero_chunk_vad import EnergyGateVad, SileroChunkVAD, VadFn
from ..separation.backends import IVocalSeparatorBackend, MDX23HipBackend
from ..utils.gpu_pipeline import PipelineConfig, PipelineContext, Streams, build_pipeline_context, chunk_schedule

logger = logging.getLogger(__name__)


@dataclass
class SeparationResult:
    vocal_track: np.ndarray
    instrumental_track: Optional[np.ndarray]
    separation_confidence: float
    backend_used: str
    processing_time: float
    quality_metrics: Dict
    feature_cache: Optional[TrackFeatureCache] = None
    vad_segments: Optional[List[Dict[str, float]]] = None
    gpu_meta: Dict = field(default_factory=dict)
    pipeline_used: bool = False
    # extension (not in the reference dataclass): stems / mix still resident in HBM for the detector and guard
    device_state: Optional[Dict[str, object]] = None


def compute_vocal_presence_markers(hip: "_native.Context", vocal_dev: torch.Tensor, sr: int) -> Dict:
    """`vocal_separator.py:460-529`: RMS(2205/882) -> dB > -50 -> state runs -> marker cut times."""
    empty = {"vocal_presence_cut_points_sec": [], "vocal_presence_cut_points_samples": [],
             "vocal_presence_segments": [], "pure_music_segments": []}
    n = int(vocal_dev.numel())
    if sr <= 0 or n == 0:
        return empty
    duration = float(n) / sr
    thr_db = float(get_config("quality_control.segment_vocal_threshold_db", -50.0))
    music_min = float(get_config("quality_control.pure_music_min_duration", 0.0))
    hop = max(1, int(0.02 * sr)); frame_length = max(hop * 2, int(0.05 * sr))
    rms = hip.frame_rms(vocal_dev, frame_length, hop).cpu().numpy()
    mask = (20.0 * np.log10(rms + 1e-12)) > thr_db
    if mask.size == 0:
        return empty
    times = (np.arange(len(mask)) * hop).astype(int) / float(sr)
    segs: List[Dict] = []
    cur_state = bool(mask[0]); cur_start = 0.0
    change = np.flatnonzero(mask[1:] != mask[:-1]) + 1
    for idx in change:
        t = float(times[idx])
        segs.append({"start": cur_start, "end": t, "is_vocal": cur_state})
        cur_start, cur_state = t, bool(mask[idx])
    segs.append({"start": cur_start, "end": duration, "is_vocal": cur_state})

    def clamp(v: float) -> float:
        return float(min(max(v, 0.0), duration))

    cuts = set()
    first = next((s for s in segs if s["is_vocal"] and s["end"] > s["start"]), None)
    if first is not None:
        cuts.add(clamp(first["start"] - 1.0))
    for prev, nxt in zip(segs, segs[1:]):
        if not prev["is_vocal"] and nxt["is_vocal"] and (prev["end"] - prev["start"]) >= music_min:
            c = clamp(nxt["start"] - 1.0)
            if c >= prev["start"]:
                cuts.add(c)
    last = next((s for s in reversed(segs) if s["is_vocal"] and s["end"] > s["start"]), None)
    if last is not None:
        cuts.add(clamp(last["end"] + 1.0))
    secs = sorted(c for c in cuts if 0.0 <= c <= duration)
    return {"vocal_presence_cut_points_sec": secs, "vocal_presence_cut_points_samples": [int(round(c * sr)) for c in secs],
            "vocal_presence_segments": segs, "pure_music_segments": [s for s in segs if not s["is_vocal"] and s["end"] > s["start"]]}


class EnhancedVocalSeparator:
    def __init__(self, sample_rate: int = 44100, *, backend: Optional[IVocalSeparatorBackend] = None,
                 device: Optional[str] = None, vad_inference_fn: Optional[VadFn] = None) -> None:
        self.sample_rate = sample_rate
        self._pipeline_cfg = PipelineConfig.from_mapping(get_config("gpu_pipeline", {}))
        if device is not None:
            self._pipeline_cfg.prefer_device = device
        self.backend_pref = str(get_config("enhanced_separation.backend", "mdx23")).lower()
        self.min_confidence_threshold = float(get_config("enhanced_separation.min_separation_confidence", 0.7))
        self._vad_inference_fn = vad_inference_fn
        self._primary_backend: Optional[IVocalSeparatorBackend] = backend
        if self._primary_backend is None:
            self._init_backends()

    def _init_backends(self) -> None:
        if self.backend_pref not in {"mdx23", "auto"}:
            raise RuntimeError(f"backend {self.backend_pref!r} is not built (only the MDX23 path is in scope, SURVEY.md §2 #2)")
        from ..utils.gpu_pipeline import select_device
        device = select_device(self._pipeline_cfg.prefer_device)
        if not device.startswith("cuda"):
            raise _native.NativeError("no HIP device available: the separator has no CPU path")
        backend = MDX23HipBackend(device=device, align_hop=self._pipeline_cfg.align_hop)
        backend.load_model()
        self._primary_backend = backend

    def _ensure_pipeline_context(self, audio: np.ndarray, gpu_context: Optional[PipelineContext]) -> PipelineContext:
        cfg = self._pipeline_cfg
        duration_s = float(len(audio)) / float(self.sample_rate) if self.sample_rate > 0 else 0.0
        if gpu_context and gpu_context.enabled:
            if not gpu_context.plans:
                gpu_context.plans = chunk_schedule(duration_s, chunk_s=cfg.chunk_s, overlap_s=cfg.overlap_s, halo_s=cfg.halo_s)
            return gpu_context
        run_cfg = PipelineConfig(**{**cfg.__dict__, "enable": True})
        ctx = build_pipeline_context(duration_s, run_cfg)
        if not ctx.device.startswith("cuda"):
            raise _native.NativeError("no HIP device available: the separator has no CPU path")
        ctx.register_mdx23_input({"name": "input", "shape": [1, 4, 3072, 256]})
        return ctx

    # ------------------------------------------------------------------------------------------
    def separate_for_detection(self, audio: np.ndarray, *, gpu_context: Optional[PipelineContext] = None,
                               audio_dev: Optional[torch.Tensor] = None, separation_gate=None, unet_stream=None) -> SeparationResult:
        """`audio_dev` (extension): the same mono track already resident in HBM; skips the upload.
        `unet_stream` (extension, `batch.TrackPipeline.unet_stream`): the one stream all workers queue their separations on; the
        gate is then held only while this track's launches are being queued, and the next track's U-Net sits in the queue behind
        this one (no idle GPU between two tracks).
        `separation_gate` (extension, a lock shared by the workers of a `batch.TrackPipeline`): taken right before this
        track's first separation kernel is queued (its index tables are already uploaded) and released once that work
        has left the GPU (the VAD results are back), before the host-bound tail."""
        if unet_stream is not None and separation_gate is None:
            # the backend, its network and its scratch are shared by every worker: without the gate two threads interleave one
            # track's STFT / U-Net / iSTFT launches with another's on that one stream
            raise ValueError("unet_stream needs separation_gate (batch.TrackPipeline hands out both)")
        backend = self._primary_backend
        if backend is None:
            raise RuntimeError("separator backend not initialised")
        start = time.time()
        ctx = self._ensure_pipeline_context(audio, gpu_context)
        try:
            vocal, inst, cache, vad_segments, markers, confidence, state = self._separate_with_pipeline(
                audio, backend, ctx, audio_dev, separation_gate, unet_stream)
        except Exception as exc:
            ctx.mark_failure("separation", str(exc))
            raise
        meta = ctx.to_meta()
        return SeparationResult(
            vocal_track=vocal, instrumental_track=inst, separation_confidence=confidence,
            backend_used=type(backend).__name__, processing_time=time.time() - start, quality_metrics=markers,
            feature_cache=cache, vad_segments=vad_segments, gpu_meta=meta, pipeline_used=ctx.enabled, device_state=state)

    def _separate_with_pipeline(self, audio: np.ndarray, backend: IVocalSeparatorBackend, gpu_context: PipelineContext,
                                audio_dev: Optional[torch.Tensor] = None, separation_gate=None, unet_stream=None):
        if not isinstance(backend, MDX23HipBackend):
            raise RuntimeError("only MDX23HipBackend drives the batched device path")
        sr = self.sample_rate
        hip = backend.hip
        plans = gpu_context.plans
        total = len(audio)
        timings: Dict[str, float] = {}
        # per-call metrics only: under batch.TrackPipeline another track shares this backend and this device, so nothing
        # backend- or device-global is reset here (the stage timings of THIS call come back through `timings`)
        t0 = time.perf_counter()
        if audio_dev is not None:
            if audio_dev.numel() != total or audio_dev.dtype != torch.float32:
                raise ValueError("audio_dev must be the float32 device copy of `audio`")
            mix_dev = audio_dev
        else:
            mix_dev = hip.to_device(np.ascontiguousarray(audio, dtype=np.float32))
        torch.cuda.current_stream(hip.device).synchronize()     # this stream only: another track may be in flight on the device
        h2d_ms = (time.perf_counter() - t0) * 1000.0

        # 1. queue the whole separation (no host synchronisation inside)
        gate_held = []

        def take_gate() -> None:
            if separation_gate is not None:
                separation_gate.acquire()
                gate_held.append(True)

        def drop_gate() -> None:
            if gate_held:
                gate_held.pop()
                separation_gate.release()
        try:
            return self._separate_gated(audio, backend, gpu_context, mix_dev, plans, timings, h2d_ms, take_gate, drop_gate, unet_stream)
        finally:
            drop_gate()

    def _separate_gated(self, audio, backend, gpu_context, mix_dev, plans, timings, h2d_ms, take_gate, drop_gate, unet_stream=None):
        sr = self.sample_rate
        hip = backend.hip
        total = len(audio)
        mix_ready = torch.cuda.Event()
        mix_ready.record()
        sep = backend.separate_track(mix_dev, sr, plans, timings, defer_sync=True, before_launch=take_gate, unet_stream=unet_stream,
                                     after_launch=drop_gate if unet_stream is not None else None)
        sep_done = torch.cuda.Event()
        sep_done.record()
        # the track-global kernels whose parameters no host decision touches (analysis/prefetch.py): the mix's on the side stream
        # beside the U-Net, the stems' right behind the separation - their consumers (markers, detector, guards) find the results
        from ..analysis import prefetch as PF
        hip.prefetch_begin()
        side = self._side_stream(hip)
        with torch.cuda.stream(side):
            side.wait_event(mix_ready)
            mix_dev.record_stream(side)
            guard_floor = PF.queue_mix_globals(hip, mix_dev, sr)
        PF.queue_vocal_globals(hip, sep.vocal, sep.instrumental, sr, guard_floor)

        # 2. while the U-Net runs: everything that only needs the MIX (`ChunkFeatureBuilder`, BPM / beat analysis) on a
        #    second, high-priority stream - its small kernels and downloads slot in between the U-Net's launches
        live_plans = [p for p in plans if min(total, int(round(p.end_s * sr))) > max(0, int(round(p.start_s * sr)))]
        with torch.cuda.stream(side):
            feature_builder = ChunkFeatureBuilder(sr=sr, use_gpu=True, device=str(hip.device), ctx=hip)
            feature_builder.attach_track(hip, mix_dev)
            for plan, (cs, ce, es, ee) in zip(live_plans, sep.chunk_ranges):
                if ee > es:
                    feature_builder.add_chunk_range(plan, cs, ce)
            cache = feature_builder.finalize(audio)
            # the stem downloads (pinned host memory) follow on the same side stream as soon as the separation is done,
            # beside the VAD / marker / detector kernels of the main stream
            t1 = time.perf_counter()
            side.wait_event(sep_done)
            sep.vocal.record_stream(side); sep.instrumental.record_stream(side)
            vocal_h = torch.empty(sep.vocal.shape, dtype=torch.float32, pin_memory=True)
            inst_h = torch.empty(sep.instrumental.shape, dtype=torch.float32, pin_memory=True)
            vocal_h.copy_(sep.vocal, non_blocking=True)
            inst_h.copy_(sep.instrumental, non_blocking=True)
            stems_on_host = torch.cuda.Event()
            stems_on_host.record()

        gpu_context.capture_device_metrics()      # an SMI query costs the host ~1 ms: taken here, while it waits for the U-Net anyway (and the GPU is under load)

        # 3. chunked VAD on the per-chunk vocals (enhanced_vocal_separator.py:331-333,412-417)
        if self._vad_inference_fn is None:        # Silero network when weights are configured, else the no-weights energy gate
            from ..detectors.silero_vad import default_vad
            self._vad_inference_fn = default_vad(sr, hip)
        vad_fn = self._vad_inference_fn
        chunk_vad = SileroChunkVAD(sample_rate=sr, merge_gap_ms=float(get_config("advanced_vad.silero_merge_gap_ms", 120.0)),
                                   focus_pad_s=float(get_config("advanced_vad.focus_window_pad_s", 0.2)), inference_fn=vad_fn)
        if hasattr(vad_fn, "precompute"):         # every chunk's windows in one batch of launches and one download
            pre = vad_fn.precompute(sep.chunk_vocal, sep.chunk_offsets, [ce - cs for cs, ce, _, _ in sep.chunk_ranges])
            for plan, chunk in zip(live_plans, pre):
                chunk_vad.process_chunk(plan, chunk, sr)
        else:                                     # injected VadFn contract: host float32 chunks
            host = sep.chunk_vocal.cpu().numpy()
            for plan, off, (cs, ce, es, ee) in zip(live_plans, sep.chunk_offsets, sep.chunk_ranges):
                chunk_vad.process_chunk(plan, host[off: off + (ce - cs)], sr)
        vad_segments = chunk_vad.finalize()
        sep.finish()
        drop_gate()

        inst_energy = hip.mean_square(sep.instrumental) if sep.instrumental.numel() else 0.0
        has_inst = inst_energy > 0.0                                            # `:458` (`np.any(instrumental)`), reduced on the GPU
        confidence = self._estimate_confidence_device(hip, sep.vocal, sep.instrumental if has_inst else None, mix_dev, inst_energy)
        markers = compute_vocal_presence_markers(hip, sep.vocal, sr)
        stems_on_host.synchronize()
        dtoh_ms = (time.perf_counter() - t1) * 1000.0
        vocal = vocal_h.numpy()
        inst = inst_h.numpy() if has_inst else None

        gm = gpu_context.gpu_meta
        gm["gpu_pipeline_processed_chunks"] = len(sep.chunk_ranges)
        gm["gpu_pipeline_used"] = bool(gpu_context.enabled)
        gm["silero_vad_segments"] = len(vad_segments)
        gm["gpu_pipeline_h2d_ms"] = float(h2d_ms)
        gm["gpu_pipeline_dtoh_ms"] = float(dtoh_ms)
        gm["gpu_pipeline_compute_ms"] = float(timings.get("stft_ms", 0.0) + timings.get("unet_ms", 0.0) + timings.get("istft_ms", 0.0))
        gm["gpu_pipeline_peak_mem_bytes"] = float(torch.cuda.max_memory_allocated(hip.device))     # device-wide high-water mark since process start
        gm["gpu_pipeline_chunk_invocations"] = len(sep.chunk_ranges)
        gm["mdx23_output_type"] = backend.get_output_type()
        gm["gpu_pipeline_stage_ms"] = dict(timings)
        state = {"hip": hip, "mix": mix_dev, "vocal": sep.vocal, "instrumental": sep.instrumental, "timings": timings}
        return vocal, inst, cache, vad_segments, markers, confidence, state

    def _side_stream(self, hip) -> "torch.cuda.Stream":
        """A second HIP stream (high priority) for the mix-only feature path that overlaps the U-Net."""
        st = getattr(self, "_side", None)
        if st is None or st.device != hip.device:
            st = torch.cuda.Stream(device=hip.device, priority=-1)
            self._side = st
        return st

    @staticmethod
    def _estimate_confidence_device(hip, vocal_dev, inst_dev, mix_dev, inst_energy: Optional[float] = None) -> float:
        """`:490-501` with the three mean squares reduced on the GPU (float64 partials)."""
        ve = hip.mean_square(vocal_dev) if vocal_dev.numel() else 0.0
        me = hip.mean_square(mix_dev) if mix_dev.numel() else 1e-8
        ratio = float(np.clip(ve / (me + 1e-8), 0.0, 1.0))
        if inst_dev is not None and inst_dev.numel():
            bal = ve / ((hip.mean_square(inst_dev) if inst_energy is None else inst_energy) + 1e-8)
            return float(np.clip(0.5 * ratio + 0.5 * np.clip(bal / (1.0 + bal), 0.0, 1.0), 0.0, 1.0))
        return float(np.clip(ratio, 0.0, 1.0))


__all__ = ["EnhancedVocalSeparator", "SeparationResult", "compute_vocal_presence_markers"]
