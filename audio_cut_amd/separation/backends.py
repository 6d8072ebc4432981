"""Separation backends — drop-in for the reference's `src/audio_cut/separation/backends.py` plug point
(`IVocalSeparatorBackend.{load_model, sample_rate, infer_chunk, flush}` `:69-87`, `SeparationOutputs`
`:61-66`, optional `reset_performance_metrics / get_performance_metrics / get_output_type /
describe_input / fallback_to_cpu` discovered with `getattr`, `:134-135,183-208,294-297`).

`MDX23HipBackend` replaces `MDX23OnnxBackend` (`:90-406`):

  reference, per chunk                                   here, per track (all chunks batched)
  ---------------------------------------------------   -------------------------------------------------
  numpy pad/window copies (`:268-281,306-330`)           index math inside ac_mdx_stft (no copies)
  torch.stft -> .cpu().numpy() (`:355-356`)              ac_mdx_stft, stays in HBM
  ORT session.run (`:358`)                               TfcTdfNet (one MFMA kernel per layer)
  torch.from_numpy().to(device); torch.istft (`:375-376`) ac_mdx_istft, stays in HBM
  numpy crop / mix - stem / mean (`:389-406`)            ac_mdx_assemble_ola (fused with the OLA of
  + host OLA in enhanced_vocal_separator.py:423-458       enhanced_vocal_separator.py)

`infer_chunk` keeps the reference's per-chunk contract (mono float32 in, mono vocal/instrumental of
the same length out) on top of the same kernels; `separate_track` is the batched fast path the
separator uses.  There is no CPU execution provider to fall back to: `fallback_to_cpu()` raises.
"""
from __future__ import annotations

import abc
import logging
import os
import contextlib
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _native
from ..config import get_config
from ..utils.gpu_pipeline import ChunkPlan
from .tfc_tdf import TfcTdfNet, TfcTdfSpec, synth_weights

logger = logging.getLogger(__name__)

N_FFT = 6144
HOP = 1024
ITEM_LEN = 261120
TRIM = N_FFT // 2
GEN = ITEM_LEN - 2 * TRIM


@dataclass
class SeparationOutputs:
    vocal: np.ndarray
    instrumental: np.ndarray


class IVocalSeparatorBackend(abc.ABC):
    @abc.abstractmethod
    def load_model(self) -> None: ...

    @abc.abstractmethod
    def sample_rate(self) -> int: ...

    @abc.abstractmethod
    def infer_chunk(self, mix_chunk: np.ndarray, **kwargs) -> SeparationOutputs: ...

    def flush(self) -> Optional[SeparationOutputs]:
        return None


@dataclass
class TrackSeparation:
    """Device-resident result of `MDX23HipBackend.separate_track`."""

    vocal: torch.Tensor            # [N] f32
    instrumental: torch.Tensor     # [N] f32
    chunk_vocal: torch.Tensor      # concatenation of the per-chunk mono vocals (VAD input)
    chunk_offsets: List[int]       # start of chunk c inside chunk_vocal
    chunk_ranges: List[Tuple[int, int, int, int]]   # (chunk_start, chunk_end, eff_start, eff_end)
    n_items: int
    finish: Optional[object] = None   # callable: waits for the queued work and fills the stage timings (separate_track(defer_sync=True))


def items_per_chunk(chunk_len: int, align_hop: int) -> int:
    """backends.py:277-281,310-312: align to `align_hop`, then pad to a multiple of GEN."""
    aligned = chunk_len + ((-chunk_len) % max(1, align_hop))
    pad = (GEN - aligned % GEN) % GEN
    return (aligned + pad) // GEN


class MDX23HipBackend(IVocalSeparatorBackend):
    def __init__(self, model_dir: Optional[Path] = None, *, device: str = "cuda:0", align_hop: Optional[int] = None,
                 weights: Optional[Dict[str, np.ndarray]] = None, spec: TfcTdfSpec = TfcTdfSpec(), seed: int = 0,
                 max_items_per_forward: int = 64, ctx: Optional["_native.Context"] = None) -> None:
        self._model_dir = Path(model_dir) if model_dir else None
        self._device = device
        self._align_hop = int(align_hop if align_hop is not None else os.getenv("MDX23_ALIGN_HOP", 4096))
        self._weights = weights
        self._spec = spec
        self._seed = seed
        self._sr = 44100
        self._net: Optional[TfcTdfNet] = None
        self._ctx = ctx
        # items (3 s U-Net windows) per forward: 64 = a whole 4-min track in one forward, 42 GB of activations of the 288 GB; measured 1.2 % faster
        # than two forwards of 32 on the same box (the deep levels' grids fill the chip better), bit-identical per item (profiles/r04ab)
        self.max_items_per_forward = int(max_items_per_forward)
        pref = str(get_config("enhanced_separation.mdx23.output_type", "auto")).strip().lower() or "auto"
        self._output_type_pref = pref if pref in {"auto", "vocal", "instrumental"} else "auto"
        self._model_name = str(get_config("enhanced_separation.mdx23.model_filename", "Kim_Vocal_1.onnx"))
        self._perf: Dict[str, float] = {}
        self.reset_performance_metrics()
        if spec.dim_f != 3072 or spec.dim_t != 256:
            raise ValueError("the MDX23 front end is built for dim_f=3072, dim_t=256 (backends.py:260-265)")

    # -- interface ---------------------------------------------------------------------------------
    def sample_rate(self) -> int:
        return self._sr

    def load_model(self) -> None:
        """Weights: an explicit name->ndarray dict, else `<model_dir>/<model_filename>` (the ONNX file the reference
        hands to onnxruntime, `backends.py:222-255`; its initializers are read by separation/onnx_weights.py), else
        `<model_dir>/<model>.npz`, else seeded synthetic weights of the Kim_Vocal_1 architecture (the ONNX file cannot
        be fetched offline; see tfc_tdf.py)."""
        if self._ctx is None:
            self._ctx = _native.Context(self._device)
        w = self._weights
        if w is None and self._model_dir is not None and (self._model_dir / self._model_name).exists():
            from .onnx_weights import load_tfc_tdf_weights
            w = load_tfc_tdf_weights(self._model_dir / self._model_name, self._spec)
            logger.info("[MDX23Hip] loaded %d tensors from %s", len(w), self._model_dir / self._model_name)
        if w is None and self._model_dir is not None:
            cand = self._model_dir / (Path(self._model_name).stem + ".npz")
            if cand.exists():
                with np.load(cand) as z:
                    w = {k: z[k] for k in z.files}
        if w is None:
            logger.warning("[MDX23Hip] no weights supplied: using seeded synthetic TFC-TDF weights (seed=%d)", self._seed)
            w = synth_weights(self._spec, seed=self._seed)
        self._weights = w
        self._net = TfcTdfNet(w, self._spec, hip=self._ctx).to(self._ctx.device).eval()
        self.reset_performance_metrics()

    def describe_input(self) -> Optional[dict]:
        return {"name": "input", "shape": [1, 4, 3072, 256]}

    def get_output_type(self) -> str:
        """backends.py:198-208: 'auto' resolves from the model file name."""
        if self._output_type_pref == "auto":
            name = self._model_name.lower()
            if any(t in name for t in ("vocal", "vocals")) and not any(t in name for t in ("inst", "instrumental", "accomp")):
                return "vocal"
            return "instrumental"
        return self._output_type_pref

    def reset_performance_metrics(self) -> None:
        self._perf = {"h2d_ms": 0.0, "dtoh_ms": 0.0, "compute_ms": 0.0, "chunks": 0.0, "max_alloc_bytes": 0.0}

    def get_performance_metrics(self, *, reset: bool = False) -> Dict[str, float]:
        out = dict(self._perf)
        if reset:
            self.reset_performance_metrics()
        return out

    def fallback_to_cpu(self) -> None:
        raise _native.NativeError("MDX23HipBackend has no CPU execution path")

    @property
    def hip(self) -> "_native.Context":
        if self._ctx is None:
            self._ctx = _native.Context(self._device)
        return self._ctx

    @property
    def net(self) -> TfcTdfNet:
        if self._net is None:
            raise RuntimeError("MDX23HipBackend not initialised: call load_model()")
        return self._net

    def unet_flops_per_item(self) -> float:
        return self._spec.flops_per_item()

    # -- batched fast path ------------------------------------------------------------------------
    def separate_track(self, track_dev: torch.Tensor, sr: int, plans: Sequence[ChunkPlan],
                       timings: Optional[Dict[str, float]] = None, defer_sync: bool = False, before_launch=None,
                       unet_stream=None, after_launch=None) -> TrackSeparation:
        """All chunks of a resident track: STFT -> U-Net -> iSTFT -> stem assembly + OLA, no host bounce.
        Everything is queued on the current stream without a host synchronisation; `defer_sync=True` returns at once
        (the caller overlaps host work and calls `result.finish()` later), otherwise the timings are read before returning.
        `before_launch` is called after the index tables are on the device and before the first kernel is queued
        (`batch.TrackPipeline` waits there for the previous track's U-Net to leave the GPU).
        `unet_stream`: queue the whole separation on THAT stream instead of the current one (the current stream waits for its
        end); `batch.TrackPipeline` gives every worker the same one, so the separations of consecutive tracks are ordered by
        the stream itself and the next one can be queued while this one still runs - `after_launch` is then called as soon as
        everything is queued (the pipeline's gate only has to keep two tracks' launches from interleaving)."""
        hip = self.hip
        net = self.net
        n = int(track_dev.numel())
        ranges: List[Tuple[int, int, int, int]] = []
        for p in plans:        # enhanced_vocal_separator.py:367-368,423-425
            cs = max(0, int(round(p.start_s * sr)))
            ce = min(n, int(round(p.end_s * sr)))
            if ce <= cs:
                continue
            es = cs + int(round(p.halo_left_s * sr))
            ee = ce - int(round(p.halo_right_s * sr))
            ee = max(es, min(n, ee))
            ranges.append((cs, ce, es, ee))
        cs_items: List[int] = []; cl_items: List[int] = []; wi_items: List[int] = []; base: List[int] = []
        for cs, ce, _, _ in ranges:
            base.append(len(cs_items))
            for k in range(items_per_chunk(ce - cs, self._align_hop)):
                cs_items.append(cs); cl_items.append(ce - cs); wi_items.append(k)
        n_items = len(cs_items)
        # every index table goes up before the first launch: a pageable upload behind queued kernels would stall the host
        d_cs = hip.to_device(np.asarray(cs_items, np.int64)); d_cl = hip.to_device(np.asarray(cl_items, np.int64))
        d_wi = hip.to_device(np.asarray(wi_items, np.int32))
        d_chunk_start = hip.to_device(np.asarray([r[0] for r in ranges], np.int64))
        d_chunk_len = hip.to_device(np.asarray([r[1] - r[0] for r in ranges], np.int64))
        d_es = hip.to_device(np.asarray([r[2] for r in ranges], np.int64)); d_ee = hip.to_device(np.asarray([r[3] for r in ranges], np.int64))
        d_base = hip.to_device(np.asarray(base, np.int32))
        offsets = np.concatenate(([0], np.cumsum([r[1] - r[0] for r in ranges]))).astype(np.int64)
        d_offsets = hip.to_device(offsets[:-1])
        step = max(1, self.max_items_per_forward)
        caller_stream = torch.cuda.current_stream(hip.device)
        if unet_stream is not None:
            tables_up = torch.cuda.Event()
            tables_up.record(caller_stream)
        if before_launch is not None:
            before_launch()
        if unet_stream is not None:
            unet_stream.wait_event(tables_up)
            track_dev.record_stream(unet_stream)
            for t in (d_cs, d_cl, d_wi, d_chunk_start, d_chunk_len, d_es, d_ee, d_base, d_offsets):
                t.record_stream(unet_stream)
        with (torch.cuda.stream(unet_stream) if unet_stream is not None else contextlib.nullcontext()):
            sep_out = self._queue_separation(track_dev, n_items, step, timings, ranges, offsets,
                                             (d_cs, d_cl, d_wi, d_chunk_start, d_chunk_len, d_es, d_ee, d_base, d_offsets))
            if unet_stream is not None:
                queued = torch.cuda.Event()
                queued.record()
        wave, vocal, inst, chunk_vocal, events = sep_out
        if unet_stream is not None:
            if after_launch is not None:
                after_launch()
            caller_stream.wait_event(queued)
            for t in (wave, vocal, inst, chunk_vocal):
                t.record_stream(caller_stream)
        n_ranges = len(ranges)

        def finish() -> None:      # one synchronisation for the whole track, after everything has been queued
            t_stft = t_net = t_istft = 0.0
            if events:
                events[-1][3].synchronize()
                for ev in events:
                    t_stft += ev[0].elapsed_time(ev[1]); t_net += ev[1].elapsed_time(ev[2]); t_istft += ev[2].elapsed_time(ev[3])
            self._perf["chunks"] += float(n_ranges)
            self._perf["compute_ms"] += t_stft + t_net + t_istft
            self._perf["max_alloc_bytes"] = max(self._perf["max_alloc_bytes"], float(torch.cuda.max_memory_allocated(hip.device)))
            if timings is not None:
                timings.update({"stft_ms": t_stft, "unet_ms": t_net, "istft_ms": t_istft, "n_items": float(n_items)})

        if not defer_sync:
            finish()
        return TrackSeparation(vocal, inst, chunk_vocal, [int(o) for o in offsets[:-1]], ranges, n_items,
                               finish if defer_sync else None)

    def _queue_separation(self, track_dev, n_items, step, timings, ranges, offsets, tables):
        """Queues STFT -> U-Net -> iSTFT of every sub-batch and the stem assembly on the CURRENT stream; no host synchronisation."""
        hip = self.hip
        net = self.net
        d_cs, d_cl, d_wi, d_chunk_start, d_chunk_len, d_es, d_ee, d_base, d_offsets = tables
        wave = torch.empty((n_items, 2, ITEM_LEN), dtype=torch.float32, device=hip.device)
        events: List[List[torch.cuda.Event]] = []     # per sub-batch: [before stft, before net, before istft, after istft]
        for a in range(0, n_items, step):
            b = min(n_items, a + step)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timings is not None else None
            if ev: ev[0].record()
            amax = torch.zeros((b - a, 256), dtype=torch.float32, device=hip.device)  # max |spec| per item and frame: the first conv's activation scale
            spec = hip.mdx_stft(track_dev, d_cs[a:b].contiguous(), d_cl[a:b].contiguous(), d_wi[a:b].contiguous(), amax=amax)
            if ev: ev[1].record()
            out = net.forward_tf(spec, amax)
            del spec
            if ev: ev[2].record()
            wave[a:b] = hip.mdx_istft(out.contiguous())
            del out
            if ev:
                ev[3].record()
                events.append(ev)
        vocal_like, other = hip.mdx_assemble_ola(track_dev, wave, d_chunk_start, d_chunk_len, d_es, d_ee, d_base)
        chunk_vocal = hip.mdx_chunk_vocal(wave, d_chunk_len, d_offsets, d_base, int(offsets[-1]))
        if self.get_output_type() == "vocal":
            vocal, inst = vocal_like, other
        else:
            # the network output is the instrumental: vocal = mix - wave, and the VAD input follows
            vocal, inst = other, vocal_like
            chunk_mix = torch.cat([track_dev[cs:ce] for cs, ce, _, _ in ranges])
            chunk_vocal = chunk_mix - chunk_vocal
        return wave, vocal, inst, chunk_vocal, events

    # -- reference-shaped per-chunk call ----------------------------------------------------------
    def infer_chunk(self, mix_chunk: np.ndarray, **kwargs) -> SeparationOutputs:
        """backends.py:299-406 for one chunk (mono, or 2-D with identical rows as the reference feeds)."""
        if self._net is None:
            raise RuntimeError("MDX23HipBackend not initialised: call load_model()")
        chunk = np.asarray(mix_chunk, dtype=np.float32)
        if chunk.ndim == 2:
            if chunk.shape[0] == 2 and not np.array_equal(chunk[0], chunk[1]):
                raise NotImplementedError("true-stereo chunks: the hot path feeds mono duplicated to 2 channels "
                                          "(backends.py:269-270); only that case is built")
            chunk = chunk[0]
        elif chunk.ndim != 1:
            raise ValueError("mix_chunk shape invalid")
        hip = self.hip
        t0 = time.perf_counter()
        dev = hip.to_device(np.ascontiguousarray(chunk))
        torch.cuda.synchronize(hip.device)
        self._perf["h2d_ms"] += (time.perf_counter() - t0) * 1000.0
        n = chunk.shape[0]
        plan = ChunkPlan(index=0, start_s=0.0, end_s=n / float(self._sr), halo_left_s=0.0, halo_right_s=0.0)
        sep = self.separate_track(dev, self._sr, [plan])
        t1 = time.perf_counter()
        vocal = sep.vocal.cpu().numpy()
        inst = sep.instrumental.cpu().numpy()
        self._perf["dtoh_ms"] += (time.perf_counter() - t1) * 1000.0
        return SeparationOutputs(vocal=vocal.astype(np.float32), instrumental=inst.astype(np.float32))


__all__ = ["IVocalSeparatorBackend", "SeparationOutputs", "MDX23HipBackend", "TrackSeparation", "items_per_chunk"]
