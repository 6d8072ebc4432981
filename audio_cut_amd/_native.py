"""ctypes binding of libaudiocut_hip.so (include/audiocut_hip.h) over PyTorch-owned device memory.

PyTorch is plumbing here: it allocates HBM, owns the HIP stream and moves host arrays; every
per-sample computation of the hot path happens in the HIP kernels behind these wrappers.  There is
no CPU fallback: a missing library or a failing call raises `NativeError`.
"""
from __future__ import annotations

import ctypes as C
import threading
import os
from pathlib import Path
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

_LIB_NAME = "libaudiocut_hip.so"
_lib: Optional[C.CDLL] = None


class NativeError(RuntimeError):
    pass


def library_path() -> Path:
    # AUDIOCUT_HIP_LIBNAME: another build of the library next to this file (A/B runs of kernel variants: make OUT=../libaudiocut_hip_<tag>.so EXTRA=-D...; the objects go to their own build_<hash of EXTRA>/)
    return Path(__file__).resolve().parent / os.environ.get("AUDIOCUT_HIP_LIBNAME", _LIB_NAME)


def load() -> C.CDLL:
    """Load the HIP library (built in-tree by `__graft_entry__.build()` / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise NativeError(f"{path} is missing: build it with `make -C audio_cut_amd/csrc` "
                          f"(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(str(path))
    _declare(lib)
    if lib.ac_abi_version() != 6:
        raise NativeError("libaudiocut_hip.so ABI version mismatch")
    _lib = lib
    return lib


_P = C.c_void_p
_I64 = C.c_int64
_I = C.c_int

SIGNATURES = {
    "ac_abi_version": (C.c_int, []),
    "ac_last_error": (C.c_char_p, []),
    "ac_ctx_create": (C.c_int, [_I, C.POINTER(_P)]),
    "ac_ctx_destroy": (C.c_int, [_P]),
    "ac_frame_rms": (C.c_int, [_P, _P, _I64, _I, _I, _I, _P, _I64, _P]),
    "ac_frame_rms_multi": (C.c_int, [_P, _P, _I64, _I, _P, _P, _P, _P, _P]),
    "ac_stft2048_features": (C.c_int, [_P, _P, _I64, _I, _P, _P, _P, _P, _P, _I64, _P]),
    "ac_onset_strength": (C.c_int, [_P, _P, _I64, _P, _I, _I, _I, _P, _P, _P]),
    "ac_tempogram_parts": (C.c_int, [_I64]),
    "ac_tempogram_reduce": (C.c_int, [_P, _P, _I64, _I, _P, _P, _P, _P, _P]),
    "ac_yin_f0": (C.c_int, [_P, _P, _I64, _I, _I, _I, _I, C.c_double, _P, _P, _I64, _P]),
    "ac_moving_meansq_db_f64": (C.c_int, [_P, _P, _I64, _I, _P, _P]),
    "ac_next_leq_scratch": (C.c_int64, [_I64]),
    "ac_next_leq_scan": (C.c_int, [_P, _P, _I64, C.c_double, _P, _P, _P]),
    "ac_window_argmin_f64": (C.c_int, [_P, _P, _I64, _P, _P, _I, _P, _P, _P]),
    "ac_zero_cross_nearest": (C.c_int, [_P, _P, _I64, _P, _I, _I, _P, _P]),
    "ac_quiet_guard_slow": (C.c_int, [_P, _P, _I64, _P, _I, _I, _I, _P, _P, _P]),
    "ac_pause_cut_points": (C.c_int, [_P, _P, _I64, _P, _P, _I, _I, _I, _P, _P, _P]),
    "ac_mdx_stft": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I, _P, _P, _P]),
    "ac_mdx_istft": (C.c_int, [_P, _P, _I, _P, _P, _P]),
    "ac_mdx_assemble_ola": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "ac_mdx_chunk_vocal": (C.c_int, [_P, _P, _P, _P, _P, _I, _P, _P]),
    "ac_sum_squares": (C.c_int, [_P, _P, _I64, _P, _I, _P]),
    "ac_window_sum_squares": (C.c_int, [_P, _P, _I64, _P, _P, _I, _P, _P]),
    "ac_conv3x3_f16x3": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P, _P]),
    "ac_conv3x3_f16x3_w96": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P, _P]),
    "ac_conv3x3_f16x3_s8": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P, _P]),
    "ac_conv3x3_f16x3_first": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _I, _P, C.c_float, C.c_float,
                                         _P, _P]),
    "ac_tdf_linear_f16x3": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _I, _I, C.c_float, _P, _P, _P]),
    "ac_tdf_small_fused": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _I, _I, _P, _P]),
    "ac_conv1x1_small": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I64, _I, _P]),
    "ac_down2x_f16x3": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.c_float, _P, _P, _P]),
    "ac_up2x_f16x3": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.c_float, _P, _P, _P]),
    "ac_pyin_observe": (C.c_int, [_P, _P, _I64, _I, _I, C.c_double, C.c_double, _I, _I, _P, _P, _P, _P, _P, C.c_double, C.c_double,
                                  _P, _P, _P, _P]),
    "ac_pyin_viterbi": (C.c_int, [_P, _P, _P, _I64, _I, _I, _P, _P, C.c_double, _P, _P, _P, _P]),
    "ac_lpc_formants": (C.c_int, [_P, _P, _I64, _I, _I, _I, C.c_float, _P, _P, _I64, _P]),
    "ac_zero_crossing_rate": (C.c_int, [_P, _P, _I64, _I, _I, _P, _I64, _P]),
    "ac_stft2048_spectral": (C.c_int, [_P, _P, _I64, _I, C.c_double, _P, _P, _I64, _P]),
    "ac_segment_frame_rms": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I, _I, _I, _I, _P, _I64, _P]),
    "ac_segment_sumsq_peak": (C.c_int, [_P, _P, _I64, _P, _P, _I, _P, _P, _P]),
    "ac_local_valley_tiles": (C.c_int, [_I, _I]),
    "ac_local_valley": (C.c_int, [_P, _P, _I64, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "ac_resample_poly": (C.c_int, [_P, _P, _I64, _I, _I, _P, _I64, _I64, _P, _I64, _P]),
    "ac_pack_pcm24": (C.c_int, [_P, _P, _I64, _P, _P]),
    "ac_resample_poly_segments": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I64, _I64, _P, _I64, _P]),
    "ac_silero_frontend": (C.c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ac_silero_lstm": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, _P]),
    "ac_silero_out": (C.c_int, [_P, _P, _P, C.c_float, _I, _P, _P]),
    "ac_host_beat_dp": (C.c_int, [_P, _I64, C.c_double, C.c_double, _P, _P]),
}


def _declare(lib: C.CDLL) -> None:
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args


def _check(rc: int) -> None:
    if rc != 0:
        msg = load().ac_last_error()
        raise NativeError(f"libaudiocut_hip call failed ({rc}): {msg.decode() if msg else '?'}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_contiguous():
        raise NativeError("non-contiguous tensor passed to the HIP ABI")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class Context:
    """Per-device handle (ac_ctx) + thin typed wrappers; tensors must live on this device."""

    def __init__(self, device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise NativeError("no HIP device visible to PyTorch: the audio-cut HIP path needs a GPU")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise NativeError(f"device {device!r} is not a HIP device")
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self.index)
        self.lib = load()
        handle = _P()
        with torch.cuda.device(self.index):
            _check(self.lib.ac_ctx_create(self.index, C.byref(handle)))
        self._h = handle
        self._rs_filters: dict = {}          # (up, down) -> (polyphase rows on the device, n_pre_remove), designed once per context
        self._pf = threading.local()         # prefetch cache (one track per worker thread)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.ac_ctx_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------------
    SMALL_UPLOAD_BYTES = 1 << 20

    def to_device(self, arr, dtype=None) -> torch.Tensor:
        """Host array -> device tensor, ordered on the CURRENT stream.  Small arrays (index tables, candidate lists: ~50 per track) go
        through pinned staging and an asynchronous copy - a pageable upload makes the host wait for the copy each time (20-30 us apiece
        in the tail of a track); PyTorch's caching host allocator keeps the staging block alive until the copy has run."""
        t = torch.as_tensor(np.ascontiguousarray(arr))
        if dtype is not None:
            t = t.to(dtype)
        if 0 < t.numel() * t.element_size() <= self.SMALL_UPLOAD_BYTES and not t.is_cuda:
            staged = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            staged.copy_(t)
            return staged.to(self.device, non_blocking=True)
        return t.to(self.device, non_blocking=False).contiguous()

    # -- prefetch: kernels whose launch parameters do not depend on host decisions, queued AHEAD of the host logic that consumes them ------
    # prefetch("frame_rms", x, 2205, 882) runs the wrapper now (on the current stream) and keeps the result; the next frame_rms(x, 2205, 882)
    # of the SAME thread on the same device memory returns it (after making the current stream wait for the producer) instead of
    # launching again.  Keys hold the operands' addresses and the entry holds the operands themselves, so the memory cannot be recycled
    # under a live entry; a parameter the consumer derives differently simply misses (the work is repeated, the result never differs).
    def prefetch_begin(self) -> None:
        """Start of a track on this thread: drop whatever an earlier track left unused."""
        self._pf.entries = {}
        self._pf.stats = {"queued": 0, "hits": 0}

    def prefetch(self, op: str, *args):
        ent = getattr(self._pf, "entries", None)
        if ent is None:
            self.prefetch_begin(); ent = self._pf.entries
        self._pf.busy = True
        try:
            out = getattr(self, op)(*args)
        finally:
            self._pf.busy = False
        ev = torch.cuda.Event(); ev.record()
        ent[self._pf_key(op, args)] = (args, out, ev)
        self._pf.stats["queued"] += 1
        return out

    @staticmethod
    def _pf_key(op: str, args) -> tuple:
        return (op,) + tuple((a.data_ptr(), a.numel()) if isinstance(a, torch.Tensor) else a for a in args)

    def _pf_take(self, op: str, *args):
        ent = getattr(self._pf, "entries", None)
        if not ent or getattr(self._pf, "busy", False):
            return None
        hit = ent.pop(self._pf_key(op, args), None)
        if hit is None:
            return None
        torch.cuda.current_stream().wait_event(hit[2])
        self._pf.stats["hits"] += 1
        return hit[1]

    def prefetch_stats(self) -> dict:
        st = dict(getattr(self._pf, "stats", {"queued": 0, "hits": 0}))
        st["unused"] = [k[0] for k in getattr(self._pf, "entries", {})]
        return st

    def _chk_f32(self, x: torch.Tensor) -> None:
        if x.dtype != torch.float32 or x.device != self.device or x.dim() != 1:
            raise NativeError("expected a 1-D float32 tensor on the context's device")

    # -- framewise ---------------------------------------------------------------------------------
    def frame_rms(self, x: torch.Tensor, frame: int, hop: int, center: bool = True) -> torch.Tensor:
        self._chk_f32(x)
        hit = self._pf_take("frame_rms", x, int(frame), int(hop), bool(center))
        if hit is not None:
            return hit
        n = x.numel()
        pad = frame // 2 if center else 0
        if n + 2 * pad < frame:
            raise NativeError("signal shorter than one frame")
        nf = 1 + (n + 2 * pad - frame) // hop
        out = torch.empty(nf, dtype=torch.float32, device=self.device)
        _check(self.lib.ac_frame_rms(self._h, _ptr(x), n, frame, hop, int(center), _ptr(out), nf, _stream()))
        return out

    def frame_rms_multi(self, x: torch.Tensor, configs: Sequence[Tuple[int, int]]) -> list:
        """`frame_rms(x, frame, hop, center=True)` for up to four (frame, hop) pairs in ONE pass over x (bit-identical series)."""
        self._chk_f32(x)
        n = x.numel()
        k = len(configs)
        frames = (C.c_int * k)(*[int(f) for f, _ in configs]); hops = (C.c_int * k)(*[int(h) for _, h in configs])
        nfs = [1 + (n + 2 * (int(f) // 2) - int(f)) // int(h) for f, h in configs]
        if min(nfs) < 1:
            raise NativeError("signal shorter than one frame")
        outs = [torch.empty(nf, dtype=torch.float32, device=self.device) for nf in nfs]
        ptrs = (C.c_void_p * k)(*[o.data_ptr() for o in outs]); nfa = (C.c_int64 * k)(*nfs)
        _check(self.lib.ac_frame_rms_multi(self._h, _ptr(x), n, k, frames, hops, ptrs, nfa, _stream()))
        return outs

    def prefetch_frame_rms_multi(self, x: torch.Tensor, configs: Sequence[Tuple[int, int]]) -> None:
        """One fused pass now; each series is then found by the `frame_rms(x, frame, hop)` call that wants it."""
        ent = getattr(self._pf, "entries", None)
        if ent is None:
            self.prefetch_begin(); ent = self._pf.entries
        outs = self.frame_rms_multi(x, configs)
        ev = torch.cuda.Event(); ev.record()
        for (f, h), o in zip(configs, outs):
            args = (x, int(f), int(h), True)
            ent[self._pf_key("frame_rms", args)] = (args, o, ev)
            self._pf.stats["queued"] += 1

    def stft2048_features(self, x: torch.Tensor, hop: int, *, want_flat: bool = True, want_mel: bool = False,
                          frame_center: Optional[torch.Tensor] = None, frame_lo: Optional[torch.Tensor] = None,
                          frame_hi: Optional[torch.Tensor] = None):
        self._chk_f32(x)
        n = x.numel()
        if frame_center is None and want_flat and not want_mel:
            hit = self._pf_take("stft2048_flatness", x, int(hop))
            if hit is not None:
                return hit, None
        nf = frame_center.numel() if frame_center is not None else 1 + n // hop
        flat = torch.empty(nf, dtype=torch.float32, device=self.device) if want_flat else None
        mel = torch.empty((nf, 128), dtype=torch.float32, device=self.device) if want_mel else None
        _check(self.lib.ac_stft2048_features(self._h, _ptr(x), n, hop, _ptr(frame_center), _ptr(frame_lo), _ptr(frame_hi),
                                             _ptr(flat), _ptr(mel), nf, _stream()))
        return flat, mel

    def onset_strength(self, mel: torch.Tensor, hop: int, aggregate: str = "mean",
                       group_start: Optional[Sequence[int]] = None) -> torch.Tensor:
        nf = mel.shape[0]
        gs = [0, nf] if group_start is None else list(group_start)
        gs_t = self.to_device(np.asarray(gs, dtype=np.int64))
        env = torch.empty(nf, dtype=torch.float32, device=self.device)
        scratch = torch.empty(len(gs) - 1, dtype=torch.float32, device=self.device)
        _check(self.lib.ac_onset_strength(self._h, _ptr(mel), nf, _ptr(gs_t), len(gs) - 1, hop,
                                          0 if aggregate == "mean" else 1, _ptr(env), _ptr(scratch), _stream()))
        return env

    def tempogram_reduce(self, env: torch.Tensor, win: int, logprior: np.ndarray, want_argmax: bool = True):
        self._chk_f32(env)
        n = env.numel()
        parts = self.lib.ac_tempogram_parts(n)
        lp = self.to_device(np.asarray(logprior, dtype=np.float64))
        mean = torch.empty(win, dtype=torch.float64, device=self.device)
        arg = torch.empty(n, dtype=torch.int32, device=self.device) if want_argmax else None
        scratch = torch.empty(parts * win, dtype=torch.float64, device=self.device)
        _check(self.lib.ac_tempogram_reduce(self._h, _ptr(env), n, win, _ptr(lp), _ptr(mean), _ptr(arg), _ptr(scratch), _stream()))
        return mean, arg

    def yin_f0(self, x: torch.Tensor, sr: int, fmin: float, fmax: float, frame_length: int = 2048, hop: int = 512,
               threshold: float = 0.1, want_cmnd: bool = False):
        """librosa.yin on the GPU: (f0 [frames] float64 on the host, optional cmnd device tensor [frames, lags])."""
        self._chk_f32(x)
        n = x.numel()
        min_period = max(int(np.floor(sr / fmax)), 1)
        max_period = min(int(np.ceil(sr / fmin)), frame_length - frame_length // 2 - 1)
        nf = 1 + n // hop
        period = torch.empty(nf, dtype=torch.float64, device=self.device)
        cmnd = torch.empty((nf, max_period - min_period + 1), dtype=torch.float64, device=self.device) if want_cmnd else None
        _check(self.lib.ac_yin_f0(self._h, _ptr(x), n, frame_length, hop, min_period, max_period, float(threshold),
                                  _ptr(period), _ptr(cmnd), nf, _stream()))
        return float(sr) / period.cpu().numpy(), cmnd

    # -- loader / exporter ---------------------------------------------------
    # libsoxr's HQ recipe as published (soxr.c `soxr_quality_spec(SOXR_HQ)`, rate.h): 20-bit precision, linear phase, stop band from
    # the Nyquist frequency of the lower rate, pass band to 1 - 0.05 / TO_3dB(20 bits * 6.0206 dB) = 0.91363 of it, low-pass designed
    # for (20 + 1) * 6.0206 = 126.43 dB.  The reference resamples with it twice (librosa's default res_type "soxr_hq":
    # audio_processor.py:44-48 on load, vocal_pause_detector.py:189 in front of Silero).
    SOXR_HQ_PASSBAND_END = 1.0 - 0.05 / ((1.6e-6 * 120.41199826559248 - 7.5e-4) * 120.41199826559248 + 0.646)
    SOXR_HQ_ATT_DB = 126.43259817887210

    @staticmethod
    def _resample_filter(up: int, down: int):
        """Anti-alias / anti-image low-pass of a rational resampler to libsoxr's HQ specification (pass band flat to 0.9136 of the
        lower Nyquist frequency, stop band from that Nyquist on, 126 dB): one Kaiser-windowed sinc at the common rate, sized by
        `scipy.signal.kaiserord`, in scipy.signal.resample_poly's framing (scaled by `up`, front-padded so that output sample 0 is
        aligned with input sample 0).  Returns (h float32, n_pre_remove).  libsoxr's own coefficients cannot be reproduced offline
        (DESIGN.md 6 row 2): this matches its published response, not its bits."""
        import scipy.signal
        rate = max(up, down)
        f_pass, f_stop = Context.SOXR_HQ_PASSBAND_END / rate, 1.0 / rate            # in units of the common rate's Nyquist
        n_taps, beta = scipy.signal.kaiserord(Context.SOXR_HQ_ATT_DB, f_stop - f_pass)
        n_taps += 1 - (n_taps & 1)                                                  # odd length: zero phase, centred on a tap
        half_len = (n_taps - 1) // 2
        h = scipy.signal.firwin(n_taps, 0.5 * (f_pass + f_stop), window=("kaiser", beta), scale=True)    # float64, unit DC gain
        h = (h * up).astype(np.float32)
        n_pre_pad = down - half_len % down
        n_pre_remove = (half_len + n_pre_pad) // down
        return np.concatenate((np.zeros(n_pre_pad, dtype=np.float32), h)), n_pre_remove

    @staticmethod
    def _polyphase_rows(h_full: np.ndarray, up: int) -> np.ndarray:
        """hp[p][t] = h_full[p + t * up], rows zero-padded: the layout ac_resample_poly takes (include/audiocut_hip.h)."""
        tpp = -(-h_full.size // up)
        hp = np.zeros(tpp * up, dtype=np.float32)
        hp[:h_full.size] = h_full
        return np.ascontiguousarray(hp.reshape(tpp, up).T)

    def _resample_filter_dev(self, up: int, down: int):
        """(polyphase rows on the device, n_pre_remove) of the (up, down) resampler; designed and uploaded once per context."""
        cache = self._rs_filters
        if (up, down) not in cache:
            h, n_pre_remove = self._resample_filter(up, down)
            cache[(up, down)] = (self.to_device(self._polyphase_rows(h, up).reshape(-1)), n_pre_remove)
        return cache[(up, down)]

    def resample_poly(self, x: torch.Tensor, up: int, down: int) -> torch.Tensor:
        """x at fs -> fs * up / down on the device: poly-phase FIR in scipy.signal.resample_poly's framing (zero padding, aligned, ceil(n up / down)
        outputs) with the soxr-HQ-specification low-pass of `_resample_filter`."""
        import math
        self._chk_f32(x)
        g = math.gcd(int(up), int(down))
        up, down = int(up) // g, int(down) // g
        if up == down == 1:
            return x.clone()
        n = x.numel()
        n_out = n * up
        n_out = n_out // down + bool(n_out % down)
        hd, n_pre_remove = self._resample_filter_dev(up, down)
        out = torch.empty(n_out, dtype=torch.float32, device=self.device)
        _check(self.lib.ac_resample_poly(self._h, _ptr(x), n, up, down, _ptr(hd), hd.numel(), n_pre_remove, _ptr(out), n_out, _stream()))
        return out

    def resample_poly_segments(self, x: torch.Tensor, offsets: Sequence[int], lengths: Sequence[int], up: int, down: int, bucket: int = 0):
        """`resample_poly` of every segment x[offsets[s] : offsets[s] + lengths[s]] on its own, one launch.  Returns (out, out_off,
        out_len): segment s's result is out[out_off[s] : out_off[s] + out_len[s]], followed by zeros up to the next multiple of
        `bucket` (0 = no padding)."""
        import math
        self._chk_f32(x)
        g = math.gcd(int(up), int(down))
        up, down = int(up) // g, int(down) // g
        lengths = [int(v) for v in lengths]
        out_len = [(n * up) // down + bool((n * up) % down) for n in lengths]
        padded = [n + ((-n) % bucket if bucket > 0 else 0) for n in out_len]
        out_off = np.concatenate(([0], np.cumsum(padded))).astype(np.int64)
        total = int(out_off[-1])
        out = torch.zeros(max(total, 1), dtype=torch.float32, device=self.device)
        if total == 0:
            return out[:0], out_off[:-1], out_len
        if up == down == 1:
            for o, n, oo in zip(offsets, lengths, out_off[:-1]):
                out[int(oo): int(oo) + n] = x[int(o): int(o) + n]
            return out, out_off[:-1], out_len
        hd, n_pre_remove = self._resample_filter_dev(up, down)
        d_io = self.to_device(np.asarray(offsets, dtype=np.int64)); d_il = self.to_device(np.asarray(lengths, dtype=np.int64))
        d_oo = self.to_device(out_off[:-1].copy()); d_ol = self.to_device(np.asarray(out_len, dtype=np.int64))
        _check(self.lib.ac_resample_poly_segments(self._h, _ptr(x), _ptr(d_io), _ptr(d_il), _ptr(d_oo), _ptr(d_ol), len(lengths), up, down,
                                                  _ptr(hd), hd.numel(), n_pre_remove, _ptr(out), total, _stream()))
        return out, out_off[:-1], out_len

    # -- Silero VAD network (detectors/silero_vad.py packs the weights) ------------------------------------
    def silero_probs(self, x16: torch.Tensor, win_start: np.ndarray, seg_first: np.ndarray, seg_count: np.ndarray, packed: dict) -> torch.Tensor:
        """Speech probability of every 512-sample window (include/audiocut_hip.h: ac_silero_frontend / _lstm / _out).
        `win_start[w]` = index in x16 of the window's first new sample, `-(index) - 1` for the first window of a chunk;
        chunk s owns windows seg_first[s] .. + seg_count[s] (LSTM state reset per chunk)."""
        self._chk_f32(x16)
        n_win = int(len(win_start))
        d_ws = self.to_device(np.asarray(win_start, dtype=np.int64))
        d_sf = self.to_device(np.asarray(seg_first, dtype=np.int32)); d_sc = self.to_device(np.asarray(seg_count, dtype=np.int32))
        gates = torch.empty((n_win, 512), dtype=torch.float32, device=self.device)
        hs = torch.empty((n_win, 128), dtype=torch.float32, device=self.device)
        probs = torch.empty(n_win, dtype=torch.float32, device=self.device)
        p = packed
        _check(self.lib.ac_silero_frontend(self._h, _ptr(x16), _ptr(d_ws), n_win, _ptr(p["basis_t"]), _ptr(p["c1"]), _ptr(p["b1"]),
                                           _ptr(p["c2"]), _ptr(p["b2"]), _ptr(p["c3"]), _ptr(p["b3"]), _ptr(p["c4"]), _ptr(p["b4"]),
                                           _ptr(p["wih_t"]), _ptr(p["bias_sum"]), _ptr(gates), _stream()))
        _check(self.lib.ac_silero_lstm(self._h, _ptr(gates), _ptr(d_sf), _ptr(d_sc), len(seg_first), _ptr(p["whh_t"]), _ptr(hs), _stream()))
        _check(self.lib.ac_silero_out(self._h, _ptr(hs), _ptr(p["w_out"]), float(p["b_out"]), n_win, _ptr(probs), _stream()))
        return probs

    def pack_pcm24(self, x: torch.Tensor) -> np.ndarray:
        """float32 device track -> host uint8 array of 3 * n little-endian PCM_24 bytes."""
        self._chk_f32(x)
        n = x.numel()
        out = torch.empty(3 * n + 3, dtype=torch.uint8, device=self.device)
        _check(self.lib.ac_pack_pcm24(self._h, _ptr(x), n, _ptr(out), _stream()))
        return out[: 3 * n].cpu().numpy()

    # -- post-path boundary policy ----------------------------------------------
    def segment_frame_rms(self, x: torch.Tensor, seg_start: np.ndarray, seg_end: np.ndarray, frame: int, hop: int, center: bool = True):
        """Framed RMS of every segment in one launch -> list of host float32 arrays.  center=True:
        librosa.feature.rms(y=x[a:b], frame, hop)[0]; center=False: windows from the segment start, the last zero padded."""
        self._chk_f32(x)
        a = np.asarray(seg_start, dtype=np.int64); b = np.asarray(seg_end, dtype=np.int64)
        counts = (1 + (b - a) // hop) if center else -((a - b) // hop)
        off = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        nf = int(off[-1])
        out = torch.empty(nf, dtype=torch.float32, device=self.device)
        da, db_, do = self.to_device(a), self.to_device(b), self.to_device(off)
        _check(self.lib.ac_segment_frame_rms(self._h, _ptr(x), x.numel(), _ptr(da), _ptr(db_), _ptr(do), len(a), frame, hop, int(center),
                                             _ptr(out), nf, _stream()))
        host = out.cpu().numpy()
        return [host[off[i]: off[i + 1]] for i in range(len(a))]

    def segment_sumsq_peak(self, x: torch.Tensor, seg_start: np.ndarray, seg_end: np.ndarray):
        """(sum of squares float64, peak |x| float32) per segment, host arrays."""
        self._chk_f32(x)
        a = self.to_device(np.asarray(seg_start, dtype=np.int64)); b = self.to_device(np.asarray(seg_end, dtype=np.int64))
        k = a.numel()
        ss = torch.empty((k, 16), dtype=torch.float64, device=self.device)
        pk = torch.empty((k, 16), dtype=torch.float32, device=self.device)
        _check(self.lib.ac_segment_sumsq_peak(self._h, _ptr(x), x.numel(), _ptr(a), _ptr(b), k, _ptr(ss), _ptr(pk), _stream()))
        parts = ss.cpu().numpy()
        total = np.zeros(k, dtype=np.float64)
        for p in range(16):                       # fixed order: deterministic
            total += parts[:, p]
        return total, pk.cpu().numpy().max(axis=1)

    def local_valley(self, x: torch.Tensor, centers: np.ndarray, radius: int, win: int):
        """(orig_db, min_db, min_idx) per boundary, host arrays (see ac_local_valley)."""
        self._chk_f32(x)
        c = self.to_device(np.asarray(centers, dtype=np.int64))
        k = c.numel()
        od = torch.empty(k, dtype=torch.float64, device=self.device)
        md = torch.empty(k, dtype=torch.float64, device=self.device)
        mi = torch.empty(k, dtype=torch.int64, device=self.device)
        nt = int(self.lib.ac_local_valley_tiles(int(radius), int(win)))
        pv = torch.empty(k * nt, dtype=torch.float64, device=self.device); pi = torch.empty(k * nt, dtype=torch.int64, device=self.device)
        _check(self.lib.ac_local_valley(self._h, _ptr(x), x.numel(), _ptr(c), k, int(radius), int(win), _ptr(od), _ptr(md), _ptr(mi),
                                        _ptr(pv), _ptr(pi), _stream()))
        return od.cpu().numpy(), md.cpu().numpy(), mi.cpu().numpy()

    # -- multi-feature detector branch ---------------------------------------------
    _PYIN_TABLES = None

    @classmethod
    def _pyin_tables(cls):
        """Host tables of librosa.pyin (numpy / scipy values so the kernel's products are the ones librosa forms)."""
        if cls._PYIN_TABLES is None:
            import scipy.stats
            thresholds = np.linspace(0, 1, 101)
            beta_probs = np.diff(scipy.stats.beta.cdf(thresholds, 2, 18))
            beta_cum = np.array([np.sum(beta_probs[:k]) for k in range(101)], dtype=np.float64)
            nn = np.arange(0, 513, dtype=np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                fact = (1.0 - np.exp(-2.0)) / (1.0 - np.exp(-2.0 * nn))
            fact[0] = 0.0
            ek = np.exp(-2.0 * np.arange(0, 512, dtype=np.float64))
            cls._PYIN_TABLES = (thresholds, beta_probs, beta_cum, fact, ek)
        return cls._PYIN_TABLES

    def pyin(self, x: torch.Tensor, sr: int, fmin: float, fmax: float, frame_length: int = 2048, hop: Optional[int] = None,
             resolution: float = 0.1, max_transition_rate: float = 35.92, switch_prob: float = 0.01, no_trough_prob: float = 0.01):
        """librosa.pyin on the GPU -> (f0 [frames] float64 with NaN where unvoiced, voiced_flag, voiced_prob), host arrays.
        CMND curves (ac_yin_f0) -> trough probabilities / pitch-bin observations (ac_pyin_observe) -> Viterbi (ac_pyin_viterbi)."""
        import scipy.signal
        self._chk_f32(x)
        hop = frame_length // 4 if hop is None else int(hop)
        n = x.numel()
        _, cmnd = self.yin_f0(x, sr, fmin, fmax, frame_length, hop, want_cmnd=True)
        nf, n_lags = cmnd.shape
        min_period = max(int(np.floor(sr / fmax)), 1)
        bps = int(np.ceil(1.0 / resolution))
        n_bins = int(np.floor(12 * bps * np.log2(fmax / fmin))) + 1
        thresholds, beta_probs, beta_cum, fact, ek = self._pyin_tables()
        tiny = float(np.finfo(np.float64).tiny)
        d = lambda a: self.to_device(np.ascontiguousarray(a, dtype=np.float64))
        logv = torch.empty((nf, n_bins), dtype=torch.float64, device=self.device)
        logu = torch.empty(nf, dtype=torch.float64, device=self.device)
        vp = torch.empty(nf, dtype=torch.float64, device=self.device)
        tabs = [d(thresholds), d(beta_probs), d(beta_cum), d(fact), d(ek)]     # keep the device tables alive across the launch
        _check(self.lib.ac_pyin_observe(self._h, _ptr(cmnd), nf, n_lags, min_period, float(sr), float(fmin), n_bins, bps,
                                        _ptr(tabs[0]), _ptr(tabs[1]), _ptr(tabs[2]), _ptr(tabs[3]), _ptr(tabs[4]),
                                        float(no_trough_prob), tiny, _ptr(logv), _ptr(logu), _ptr(vp), _stream()))
        # banded log-transition tables, built exactly like librosa.sequence.transition_local(window="triangle", wrap=False)
        max_semitones = round(max_transition_rate * 12 * hop / sr)
        width = max_semitones * bps + 1
        half = width // 2
        win = scipy.signal.get_window("triangle", width, fftbins=False)
        base = np.zeros(n_bins)
        lpad = (n_bins - width) // 2
        base[lpad: lpad + width] = win
        dense = np.zeros((n_bins, n_bins))
        for i in range(n_bins):                       # same steps (pad_center, roll, clip, row-normalise) as librosa: same rounding
            row = np.roll(base, n_bins // 2 + i + 1)
            row[min(n_bins, i + width // 2 + 1):] = 0
            row[: max(0, i - width // 2)] = 0
            dense[i] = row
        dense /= dense.sum(axis=1, keepdims=True)
        idx = np.arange(n_bins)
        src = idx[:, None] - half + np.arange(width)[None, :]                  # per DESTINATION j, tap d <-> source i = j - half + d
        ok = (src >= 0) & (src < n_bins)
        t_in = np.where(ok, dense[np.clip(src, 0, n_bins - 1), idx[:, None]], 0.0)
        lt_same = np.log((1 - switch_prob) * t_in + tiny)
        lt_cross = np.log(switch_prob * t_in + tiny)
        p_init = np.zeros(2 * n_bins); p_init[n_bins:] = 1 / n_bins
        ptr = torch.empty((nf, 2 * n_bins), dtype=torch.int16, device=self.device)
        states = torch.empty(nf, dtype=torch.int32, device=self.device)
        vt = [d(lt_same), d(lt_cross), d(np.log(p_init + tiny))]
        _check(self.lib.ac_pyin_viterbi(self._h, _ptr(logv), _ptr(logu), nf, n_bins, half, _ptr(vt[0]), _ptr(vt[1]),
                                        float(np.log(tiny)), _ptr(vt[2]), _ptr(ptr), _ptr(states), _stream()))
        st = states.cpu().numpy().astype(np.int64)
        freqs = fmin * 2 ** (np.arange(n_bins) / (12 * bps))
        f0 = freqs[st % n_bins]
        voiced = st < n_bins
        f0 = f0.copy(); f0[~voiced] = np.nan
        return f0, voiced, vp.cpu().numpy()

    def lpc_formants(self, x: torch.Tensor, frame_len: int, hop: int, order: int = 12, preemph: float = 0.95):
        """`_extract_formants`: per-frame LPC peak magnitudes -> (count [frames] int32, mags [frames, 3] float64), host arrays."""
        self._chk_f32(x)
        n = x.numel()
        nf = len(range(0, n - frame_len, hop))
        if nf <= 0:
            return np.zeros(0, np.int32), np.zeros((0, 3), np.float64)
        cnt = torch.empty(nf, dtype=torch.int32, device=self.device)
        mag = torch.empty((nf, 3), dtype=torch.float64, device=self.device)
        _check(self.lib.ac_lpc_formants(self._h, _ptr(x), n, frame_len, hop, order, float(preemph), _ptr(cnt), _ptr(mag), nf, _stream()))
        return cnt.cpu().numpy(), mag.cpu().numpy()

    def zero_crossing_rate(self, x: torch.Tensor, frame_len: int, hop: int) -> np.ndarray:
        self._chk_f32(x)
        n = x.numel()
        nf = 1 + n // hop
        out = torch.empty(nf, dtype=torch.float64, device=self.device)
        _check(self.lib.ac_zero_crossing_rate(self._h, _ptr(x), n, frame_len, hop, _ptr(out), nf, _stream()))
        return out.cpu().numpy()

    def stft2048_spectral(self, x: torch.Tensor, sr: int, hop: int):
        """(spectral centroid float64 [frames], low-third magnitude ratio float32 [frames]), host arrays."""
        self._chk_f32(x)
        n = x.numel()
        nf = 1 + n // hop
        cen = torch.empty(nf, dtype=torch.float64, device=self.device)
        rat = torch.empty(nf, dtype=torch.float32, device=self.device)
        _check(self.lib.ac_stft2048_spectral(self._h, _ptr(x), n, hop, float(sr), _ptr(cen), _ptr(rat), nf, _stream()))
        return cen.cpu().numpy(), rat.cpu().numpy()

    # -- guard ---------------------------------------------------------------------------------------
    def stft2048_flatness(self, x: torch.Tensor, hop: int) -> torch.Tensor:
        """Spectral flatness per frame of the whole wave (the prefetchable form of stft2048_features(x, hop, want_flat=True))."""
        return self.stft2048_features(x, hop, want_flat=True, want_mel=False)[0]

    def moving_meansq_db(self, x: torch.Tensor, win: int) -> torch.Tensor:
        self._chk_f32(x)
        hit = self._pf_take("moving_meansq_db", x, int(win))
        if hit is not None:
            return hit
        out = torch.empty(x.numel(), dtype=torch.float64, device=self.device)
        _check(self.lib.ac_moving_meansq_db_f64(self._h, _ptr(x), x.numel(), win, _ptr(out), _stream()))
        return out

    def next_leq_scan(self, db: torch.Tensor, floor_db: float) -> torch.Tensor:
        hit = self._pf_take("next_leq_scan", db, float(floor_db))
        if hit is not None:
            return hit
        n = db.numel()
        out = torch.empty(n, dtype=torch.int64, device=self.device)
        scratch = torch.empty(int(self.lib.ac_next_leq_scratch(n)), dtype=torch.int64, device=self.device)
        _check(self.lib.ac_next_leq_scan(self._h, _ptr(db), n, float(floor_db), _ptr(out), _ptr(scratch), _stream()))
        return out

    def window_argmin(self, db: torch.Tensor, start: np.ndarray, length: np.ndarray):
        k = len(start)
        s = self.to_device(np.asarray(start, dtype=np.int64)); ln = self.to_device(np.asarray(length, dtype=np.int64))
        arg = torch.empty(k, dtype=torch.int64, device=self.device)
        val = torch.empty((k, 2), dtype=torch.float64, device=self.device)
        _check(self.lib.ac_window_argmin_f64(self._h, _ptr(db), db.numel(), _ptr(s), _ptr(ln), k, _ptr(arg), _ptr(val), _stream()))
        return arg.cpu().numpy(), val.cpu().numpy()

    def zero_cross_nearest(self, x: torch.Tensor, idx: np.ndarray, half: int) -> np.ndarray:
        self._chk_f32(x)
        k = len(idx)
        i = self.to_device(np.asarray(idx, dtype=np.int64))
        pos = torch.empty(k, dtype=torch.float64, device=self.device)
        _check(self.lib.ac_zero_cross_nearest(self._h, _ptr(x), x.numel(), _ptr(i), int(half), k, _ptr(pos), _stream()))
        return pos.cpu().numpy()

    def quiet_guard_slow(self, x: torch.Tensor, idx: np.ndarray, span: int, win: int):
        self._chk_f32(x)
        k = len(idx)
        i = self.to_device(np.asarray(idx, dtype=np.int64))
        arg = torch.empty(k, dtype=torch.int64, device=self.device)
        val = torch.empty((k, 2), dtype=torch.float64, device=self.device)
        _check(self.lib.ac_quiet_guard_slow(self._h, _ptr(x), x.numel(), _ptr(i), int(span), int(win), k, _ptr(arg), _ptr(val), _stream()))
        return arg.cpu().numpy(), val.cpu().numpy()

    def pause_cut_points(self, x: torch.Tensor, a: np.ndarray, b: np.ndarray, win: int, guard: int):
        self._chk_f32(x)
        k = len(a)
        ta = self.to_device(np.asarray(a, dtype=np.int64)); tb = self.to_device(np.asarray(b, dtype=np.int64))
        cut = torch.empty(k, dtype=torch.int64, device=self.device)
        aux = torch.empty((k, 2), dtype=torch.int64, device=self.device)
        _check(self.lib.ac_pause_cut_points(self._h, _ptr(x), x.numel(), _ptr(ta), _ptr(tb), k, int(win), int(guard),
                                            _ptr(cut), _ptr(aux), _stream()))
        return cut.cpu().numpy(), aux.cpu().numpy()

    # -- MDX23 ---------------------------------------------------------------------------------------
    def mdx_stft(self, track: torch.Tensor, chunk_start: torch.Tensor, chunk_len: torch.Tensor,
                 win_index: torch.Tensor, out: Optional[torch.Tensor] = None, amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`amax` [n_items, 256] float32, zeroed by the caller: receives max |spectrogram| per item and frame (the first conv's
        time-local activation scale)."""
        self._chk_f32(track)
        n_items = chunk_start.numel()
        if out is None:
            out = torch.empty((n_items, 4, 256, 3072), dtype=torch.float32, device=self.device)
        _, pa = self._amax_args(n_items, None, amax, 256, 256)
        _check(self.lib.ac_mdx_stft(self._h, _ptr(track), track.numel(), _ptr(chunk_start), _ptr(chunk_len), _ptr(win_index),
                                    n_items, _ptr(out), pa, _stream()))
        return out

    def mdx_istft(self, spec: torch.Tensor) -> torch.Tensor:
        if spec.dtype != torch.float32 or tuple(spec.shape[1:]) != (4, 256, 3072):
            raise NativeError("mdx_istft expects float32 [items, 4, 256, 3072]")
        n_items = spec.shape[0]
        wave = torch.empty((n_items, 2, 261120), dtype=torch.float32, device=self.device)
        scratch = torch.empty((n_items * 2 * 256 * 6144,), dtype=torch.float32, device=self.device)
        _check(self.lib.ac_mdx_istft(self._h, _ptr(spec), n_items, _ptr(wave), _ptr(scratch), _stream()))
        return wave

    def mdx_assemble_ola(self, track: torch.Tensor, wave: torch.Tensor, chunk_start, chunk_len, eff_start, eff_end, item_base):
        self._chk_f32(track)
        n = track.numel()
        vocal = torch.empty(n, dtype=torch.float32, device=self.device)
        inst = torch.empty(n, dtype=torch.float32, device=self.device)
        _check(self.lib.ac_mdx_assemble_ola(self._h, _ptr(track), n, _ptr(wave), _ptr(chunk_start), _ptr(chunk_len),
                                            _ptr(eff_start), _ptr(eff_end), _ptr(item_base), chunk_start.numel(),
                                            _ptr(vocal), _ptr(inst), _stream()))
        return vocal, inst


    def mdx_chunk_vocal(self, wave: torch.Tensor, chunk_len: torch.Tensor, out_offset: torch.Tensor, item_base: torch.Tensor,
                        total: int) -> torch.Tensor:
        out = torch.empty(int(total), dtype=torch.float32, device=self.device)
        _check(self.lib.ac_mdx_chunk_vocal(self._h, _ptr(wave), _ptr(chunk_len), _ptr(out_offset), _ptr(item_base),
                                           chunk_len.numel(), _ptr(out), _stream()))
        return out

    # -- U-Net layers (NCHW float32; `in_amax` / `out_amax`: per-item max |x| of the input / output tensor, include/audiocut_hip.h) --
    AMAX_ROWS = 1       # AC_AMAX_ROWS: rows of the time axis per amax entry (one maximum per item and time row)

    def _amax_args(self, b: int, in_amax: Optional[torch.Tensor], out_amax: Optional[torch.Tensor], h_in: int, h_out: int):
        """amax tensors are float32 [batch, H] (H = the tensor's time axis): max |x| per item and time row."""
        for t, h in ((in_amax, h_in), (out_amax, h_out)):
            if t is None:
                continue
            if h % self.AMAX_ROWS or t.dtype != torch.float32 or t.numel() != b * (h // self.AMAX_ROWS) or not t.is_contiguous() \
                    or t.device != self.device:
                raise NativeError(f"amax tensors must be contiguous float32 [batch, H] on the context's device (H = {h})")
        return _ptr(in_amax), _ptr(out_amax)

    def _conv3x3(self, fn, name: str, x, w_packed, bias, c_out, w_unscale, relu, out, in_amax, out_amax) -> torch.Tensor:
        if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
            raise NativeError(f"{name} expects a contiguous float32 NCHW tensor")
        b, c_in, h, w = x.shape
        if out is None:
            out = torch.empty((b, c_out, h, w), dtype=torch.float32, device=self.device)
        pi, po = self._amax_args(b, in_amax, out_amax, h, h)
        _check(fn(self._h, _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(out), b, c_in, c_out, h, w, float(w_unscale), int(relu), pi, po,
                  _stream()))
        return out

    def conv3x3_f16x3(self, x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, c_out: int, w_unscale: float = 1.0,
                      relu: bool = True, out: Optional[torch.Tensor] = None, in_amax: Optional[torch.Tensor] = None,
                      out_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """3x3 / stride 1 / pad 1 conv on the f16 matrix cores (3-term hi/lo split), fused bias (+ReLU).  NCHW float32."""
        return self._conv3x3(self.lib.ac_conv3x3_f16x3, "conv3x3_f16x3", x, w_packed, bias, c_out, w_unscale, relu, out, in_amax, out_amax)

    def conv3x3_f16x3_w96(self, x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, c_out: int, w_unscale: float = 1.0,
                          relu: bool = True, out: Optional[torch.Tensor] = None, in_amax: Optional[torch.Tensor] = None,
                          out_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """conv3x3_f16x3 with 96 output channels per workgroup (C_in % 16 == 0, C_out % 96 == 0; `pack_conv3x3_w96` weights)."""
        return self._conv3x3(self.lib.ac_conv3x3_f16x3_w96, "conv3x3_f16x3_w96", x, w_packed, bias, c_out, w_unscale, relu, out,
                             in_amax, out_amax)

    def conv3x3_f16x3_s8(self, x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, c_out: int, w_unscale: float = 1.0,
                         relu: bool = True, out: Optional[torch.Tensor] = None, in_amax: Optional[torch.Tensor] = None,
                         out_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The 8-channel-stage conv kernel with 48 output channels per workgroup (`pack_conv3x3_w96(w, cob=48)` weights)."""
        return self._conv3x3(self.lib.ac_conv3x3_f16x3_s8, "conv3x3_f16x3_s8", x, w_packed, bias, c_out, w_unscale, relu, out,
                             in_amax, out_amax)

    def conv3x3_f16x3_first(self, spec: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor,
                            c_out: int, w_unscale: float, relu: bool = True, spec_amax: Optional[torch.Tensor] = None,
                            amax_gain: float = 1.0, amax_offs: float = 0.0, out_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """relu(conv3x3(relu(conv1x1(spec, w1) + b1))) in one kernel: the 1x1 convolution's output never touches HBM.
        `spec_amax * amax_gain + amax_offs` bounds the generated tensor (gain = max row L1 norm of w1, offs = max |b1|)."""
        if spec.dtype != torch.float32 or spec.dim() != 4 or not spec.is_contiguous():
            raise NativeError("conv3x3_f16x3_first expects a contiguous float32 NCHW tensor")
        b, c0, h, w = spec.shape
        w1 = w1.reshape(w1.shape[0], -1)
        if w1.shape[1] != c0 or not w1.is_contiguous():
            raise NativeError("conv3x3_f16x3_first: w1 must be [C_in, C0(,1,1)] contiguous")
        out = torch.empty((b, c_out, h, w), dtype=torch.float32, device=self.device)
        pi, po = self._amax_args(b, spec_amax, out_amax, h, h)
        _check(self.lib.ac_conv3x3_f16x3_first(self._h, _ptr(spec), _ptr(w1), _ptr(b1), _ptr(w_packed), _ptr(bias), _ptr(out), b, c0,
                                               w1.shape[0], c_out, h, w, float(w_unscale), int(relu), pi, float(amax_gain),
                                               float(amax_offs), po, _stream()))
        return out

    def tdf_linear_f16x3(self, x: torch.Tensor, w_packed: torch.Tensor, n_out: int, scale: torch.Tensor, shift: torch.Tensor,
                         w_unscale: float, resid: Optional[torch.Tensor] = None, in_amax: Optional[torch.Tensor] = None,
                         out_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """TDF layer on the f16 matrix cores: relu(scale[c] * (x @ W^T) + shift[c]) (+ resid) over the last axis of an
        NCHW float32 tensor [B, C, T, K] -> [B, C, T, n_out].  Shapes the kernel cannot tile raise NativeError."""
        if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
            raise NativeError("tdf_linear_f16x3 expects a contiguous float32 NCHW tensor")
        b, c, t, k = x.shape
        out = torch.empty((b, c, t, n_out), dtype=torch.float32, device=self.device)
        if resid is not None and (resid.shape != out.shape or not resid.is_contiguous() or resid.dtype != torch.float32):
            raise NativeError("tdf_linear_f16x3: residual must match the output")
        pi, po = self._amax_args(b, in_amax, out_amax, t, t)
        _check(self.lib.ac_tdf_linear_f16x3(self._h, _ptr(x), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(resid), _ptr(out),
                                            b * c * t, n_out, k, t, c, float(w_unscale), pi, po, _stream()))
        return out

    def tdf_small_fused(self, x: torch.Tensor, w1_packed: torch.Tensor, w2_packed: torch.Tensor, hidden: int, scale1: torch.Tensor,
                        shift1: torch.Tensor, scale2: torch.Tensor, shift2: torch.Tensor,
                        out_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x + relu(bn(linear(relu(bn(linear(x)))))) over the last axis, both narrow TDF layers in one exact-float32 kernel
        (`conv_pack.pack_tdf_small` weights; F % 32 == 0, hidden <= 48)."""
        if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
            raise NativeError("tdf_small_fused expects a contiguous float32 NCHW tensor")
        b, c, t, f = x.shape
        out = torch.empty_like(x)
        _, po = self._amax_args(b, None, out_amax, t, t)
        _check(self.lib.ac_tdf_small_fused(self._h, _ptr(x), _ptr(w1_packed), _ptr(w2_packed), _ptr(scale1), _ptr(shift1), _ptr(scale2),
                                           _ptr(shift2), _ptr(out), b * c * t, f, int(hidden), t, c, po, _stream()))
        return out

    def conv1x1_small(self, x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, relu: bool) -> torch.Tensor:
        """1x1 conv with min(C_in, C_out) <= 8 (+ bias, optional ReLU) as a streaming float32 kernel.  NCHW float32."""
        if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
            raise NativeError("conv1x1_small expects a contiguous float32 NCHW tensor")
        b, c, h, w = x.shape
        w2 = weight.reshape(weight.shape[0], -1)
        if w2.shape[1] != c or not w2.is_contiguous():
            raise NativeError("conv1x1_small: weight must be [C_out, C_in(,1,1)] contiguous")
        out = torch.empty((b, w2.shape[0], h, w), dtype=torch.float32, device=self.device)
        _check(self.lib.ac_conv1x1_small(self._h, _ptr(x), _ptr(w2), _ptr(bias), _ptr(out), b, c, w2.shape[0], h * w, int(relu), _stream()))
        return out

    def down2x_f16x3(self, x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, c_out: int, w_unscale: float,
                     in_amax: Optional[torch.Tensor] = None, out_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """2x2 / stride-2 conv + bias + ReLU, one fused MFMA kernel (space-to-depth gather in the loader).  NCHW float32."""
        if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
            raise NativeError("down2x_f16x3 expects a contiguous float32 NCHW tensor")
        b, c, h, w = x.shape
        out = torch.empty((b, c_out, h // 2, w // 2), dtype=torch.float32, device=self.device)
        pi, po = self._amax_args(b, in_amax, out_amax, h, h // 2)
        _check(self.lib.ac_down2x_f16x3(self._h, _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(out), b, c, c_out, h, w, float(w_unscale),
                                        pi, po, _stream()))
        return out

    def up2x_f16x3(self, x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, c_out: int, w_unscale: float,
                   skip: Optional[torch.Tensor] = None, in_amax: Optional[torch.Tensor] = None,
                   out_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """2x2 / stride-2 transposed conv + bias + ReLU (* skip), one fused MFMA kernel (depth-to-space in the epilogue)."""
        if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
            raise NativeError("up2x_f16x3 expects a contiguous float32 NCHW tensor")
        b, c, h, w = x.shape
        out = torch.empty((b, c_out, 2 * h, 2 * w), dtype=torch.float32, device=self.device)
        if skip is not None and (skip.shape != out.shape or not skip.is_contiguous() or skip.dtype != torch.float32):
            raise NativeError("up2x_f16x3: skip must match the output")
        pi, po = self._amax_args(b, in_amax, out_amax, h, 2 * h)
        _check(self.lib.ac_up2x_f16x3(self._h, _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(skip), _ptr(out), b, c, c_out, h, w,
                                      float(w_unscale), pi, po, _stream()))
        return out

    def sum_squares_parts(self, x: torch.Tensor) -> torch.Tensor:
        """Block partial sums of x^2 (float64) on the device; the host adds them in index order (mean_square)."""
        self._chk_f32(x)
        hit = self._pf_take("sum_squares_parts", x)
        if hit is not None:
            return hit
        n = x.numel()
        parts = min(1024, max(1, n // 4096))
        buf = torch.empty(parts, dtype=torch.float64, device=self.device)
        _check(self.lib.ac_sum_squares(self._h, _ptr(x), n, _ptr(buf), parts, _stream()))
        return buf

    WINDOW_MEAN_SQUARE_MAX = 8191          # longest window `window_mean_squares` takes: below 8192 samples `mean_square` sums ONE partial

    def window_mean_squares(self, x: torch.Tensor, starts, ends) -> np.ndarray:
        """`[mean_square(x[a:b]) for a, b in zip(starts, ends)]` in one launch and one download, bit-identical to the per-window calls
        (0.0 for an empty window).  Every window must be at most WINDOW_MEAN_SQUARE_MAX samples long."""
        self._chk_f32(x)
        a = np.asarray(starts, dtype=np.int64); b = np.asarray(ends, dtype=np.int64)
        n = int(x.numel())
        if a.shape != b.shape or a.ndim != 1:
            raise ValueError("starts / ends must be equally long 1-D sequences")
        if a.size == 0:
            return np.zeros(0, dtype=np.float64)
        if np.any(a < 0) or np.any(b > n) or np.any(b < a) or np.any(b - a > self.WINDOW_MEAN_SQUARE_MAX):
            raise ValueError("windows must lie inside the wave and be at most WINDOW_MEAN_SQUARE_MAX samples long")
        out = torch.empty(a.size, dtype=torch.float64, device=self.device)
        a_dev, b_dev = self.to_device(a), self.to_device(b)       # named: a temporary would hand its block back to the allocator before the launch
        _check(self.lib.ac_window_sum_squares(self._h, _ptr(x), n, _ptr(a_dev), _ptr(b_dev), int(a.size), _ptr(out), _stream()))
        sums = out.cpu().numpy()
        length = (b - a).astype(np.float64)
        return np.where(length > 0, sums / np.where(length > 0, length, 1.0), 0.0)

    def mean_square(self, x: torch.Tensor) -> float:
        self._chk_f32(x)
        n = x.numel()
        if n == 0:
            return 0.0
        return float(np.sum(self.sum_squares_parts(x).cpu().numpy())) / float(n)


def host_beat_dp(localscore: np.ndarray, period: float, tightness: float):
    """librosa's beat-tracking DP, sequential, in C++ on the host (ac_host_beat_dp)."""
    lib = load()
    ls = np.ascontiguousarray(localscore, dtype=np.float64)
    n = ls.size
    back = np.empty(n, dtype=np.int64)
    cum = np.zeros(n, dtype=np.float64)
    _check(lib.ac_host_beat_dp(ls.ctypes.data, n, float(period), float(tightness), back.ctypes.data, cum.ctypes.data))
    return back, cum
