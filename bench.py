#!/usr/bin/env python3
"""bench.py — audio-seconds processed per wall-second (separate + detect) on MI355X.

A "step" is one pass of the hot path over one synthetic track of the
BASELINE.json configs[1] shape: 4-min 44.1 kHz song (stereo generator, mono down-mix as the reference's
loader produces), chunked MDX23 separation (full-size TFC-TDF U-Net, seeded synthetic weights — the
Kim_Vocal_1.onnx file cannot be fetched offline) + TrackFeatureCache + chunked VAD + pause detection +
quiet-guard cut refinement.  The track is resident in HBM when the timed region starts.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Every step is a DIFFERENT track.  N = 1: seed 2 (the C2 track) first, then the C3 seeds 100, 101, ...  N > 1: one
process per GPU; the N * K tracks of the job are the C3 seeds 100 .. 100 + N * K - 1 (N = 8, K = 4 is
BASELINE configs[2] itself: 32 x 4-min tracks), dealt to the ranks by `audio_cut_amd.batch.assign_tracks` (weak
scaling; tracks are independent, no data-path collective), batch completion = RCCL barrier + all_gather_object of
per-track summaries.  Every track's boundary SHA-1 is compared with the committed single-GPU result of the same seed
(tests/golden/c3_n1_sha1.json, written by `bench.py --config c3 --write-golden PATH`): "per-track boundaries identical to
the single-GPU run" is checked inside the bench (`parity_vs_single_gpu`).  `--gpus N` without a launcher starts the
N ranks itself (child processes, before anything touches the GPU).
  --config c3   32 tracks (seeds 100-131) over the ranks present, K = 32 / N per rank
  --config c5   BASELINE configs[4]: one 30-min track per step (240 chunks, 480 U-Net items)
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np
import torch

METRIC = "audio-seconds processed/sec (separate+detect) per GPU; cut-point index parity"
F32_MATRIX_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak FP32 (matrix), v_mfma_f32_32x32x2_f32


def cpu_baseline(sample_s: float, weights, spec, single_thread_sample_s: float = 10.0) -> dict:
    """The oracle (CPU restatement of the reference path: torch-CPU STFT/U-Net/iSTFT + numpy detection and
    guard) timed on this host's cores on a bounded sample of the same workload: once on the GPU's share of the host cores
    (`value`) and once single-threaded (the reference's published detection figure is "single core"),
    the latter on one 10 s chunk so the default run stays within minutes."""
    from audio_cut_amd.testing import signals
    from oracle import e2e as OE, refine as OR
    OR.LEGACY_PROMOTION = True
    # the GPU box shares its host: one GPU's CPU share is 16 cores, whatever os.cpu_count() says
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = int(os.environ.get("AC_CPU_BASELINE_THREADS", min(16, avail)))
    saved = torch.get_num_threads()

    def leg(seconds: float, n_threads: int) -> dict:
        torch.set_num_threads(n_threads)
        mix = signals.c2_song(seconds, seed=2)
        t0 = time.perf_counter()
        res = OE.run_track(mix, 44100, weights)
        dt = time.perf_counter() - t0
        return {"value": round(seconds / dt, 4), "cores": n_threads, "sample_audio_s": seconds, "seconds": round(dt, 2),
                "phases_s": {k: round(v, 3) for k, v in res.timings.items()}, "n_boundaries": len(res.sample_boundaries)}

    try:
        multi = leg(sample_s, threads)
        single = leg(single_thread_sample_s, 1) if single_thread_sample_s > 0 else None
    finally:
        torch.set_num_threads(saved)
    out = {"value": multi["value"], "unit": "audio-s/s", "cores": threads, "kind": "port",
           "sample": f"first {sample_s:g} s of the 240 s C2 track (generator seed 2) = {sample_s / 240.0:.1%} of BASELINE configs[1]: "
                     f"{int(np.ceil(max(sample_s - 2.5, 0.0) / 7.5 + 1e-9)) if sample_s > 10 else 1} chunks through the full oracle path "
                     "(torch-CPU STFT / U-Net / iSTFT, numpy features, detection, guard), one pass; the path is linear in "
                     "track length (per-chunk U-Net dominates)",
           "seconds": multi["seconds"], "phases_s": multi["phases_s"], "n_boundaries": multi["n_boundaries"]}
    if single is not None:
        out["single_thread"] = {"value": single["value"], "unit": "audio-s/s", "cores": 1,
                                "sample": f"first {single_thread_sample_s:g} s of the same track (one chunk), torch / OpenMP pinned to 1 thread",
                                "seconds": single["seconds"], "phases_s": single["phases_s"]}
    return out


class SocketSampler:
    """Power cap, and power draw / shader clock sampled every 50 ms from the amdgpu hwmon node while the timed region runs
    (a side thread reading three sysfs files).  The U-Net kernels run on the package power cap: boxes of the
    pool that hold different clocks under it give different `value`s, and this block says which kind the run was on."""

    def __init__(self, device_index: int = 0) -> None:
        import glob
        import threading
        nodes = []
        try:                                   # the hwmon node of the PCI function torch's device sits on
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            nodes = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*"))
            self.pci = bdf
        except Exception:
            self.pci = None
        if not nodes:
            nodes = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.hw = nodes[0] if nodes else None
        self.samples: list = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._loop, daemon=True)

    def _read(self, name: str):
        try:
            with open(f"{self.hw}/{name}") as fh:
                return int(fh.read().strip())
        except Exception:
            return None

    def _loop(self) -> None:
        while not self._stop.wait(0.05):
            p = self._read("power1_average") or self._read("power1_input")
            f = self._read("freq1_input")
            if p is not None or f is not None:
                self.samples.append((p, f))

    def start(self) -> "SocketSampler":
        if self.hw:
            self._thread.start()
        return self

    def stop(self) -> dict:
        if not self.hw:
            return {}
        self._stop.set()
        self._thread.join(timeout=1.0)
        cap = self._read("power1_cap")
        pw = sorted(p / 1e6 for p, _ in self.samples if p)
        fr = sorted(f / 1e6 for _, f in self.samples if f)
        out = {"pci": self.pci, "power_cap_w": cap / 1e6 if cap else None, "samples": len(self.samples)}
        if pw:
            out["power_w_median"] = round(pw[len(pw) // 2], 1)
        if fr:
            out["sclk_mhz_median"] = round(fr[len(fr) // 2], 1)
        return out


DOMINANT = "k_conv3x3_f16x3"           # audio_cut_amd/csrc/ac_conv96.hip: k_conv3x3_f16x3_w96<MT, OCC, RELU, FIRST>
F16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md: BF16/F16 MFMA ~2.5 PF dense (v_mfma_f32_16x16x32_f16)
# MFMA FLOPs the split issues per algorithmic FLOP: 3 products (hi*hi + hi*lo + lo*hi); tap 8 of two consecutive 16-channel
# blocks shares one k-step, so only a trailing unpaired block (C/16 odd: C = 48, 144, 240) pads - 1.5 % FLOP-weighted over the U-Net
F16X3_ISSUE_FACTOR = 3.0 * 1.015


def roofline_conv(probe, conv_ms: float, conv_flops: float, elapsed: float) -> dict:
    """The dominant kernel: the f16x3 split-MFMA 3x3 conv, one launch per 3x3 conv of the U-Net (33 per forward).
    `achieved` = ALGORITHMIC FLOPs (2*B*C*C*9*H*W per launch, the f32 convolution the kernel replaces) / HIP-event
    time around the launches of the timed region (events recorded on the stream the kernel is launched on);
    `traffic` = HBM bytes per launch from the committed PMC passes (profiles/*_pmc_summary.json)."""
    n = max(1, len(probe))
    achieved = conv_flops / (conv_ms / 1e3) / 1e12 if conv_ms > 0 else 0.0
    traffic = None
    for cand in sorted((ROOT / "profiles").glob("*_pmc_summary.json"), reverse=True):
        try:
            ks = json.loads(cand.read_text())["kernels"]
            hit = [v for k, v in ks.items() if DOMINANT in k]
            if hit:
                traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit))
                break
        except Exception:
            pass
    return {
        "kernel": DOMINANT + "<relu> (hand-written HIP: float16 hi/lo split, v_mfma_f32_16x16x32_f16, f32 accumulate; "
                             "one launch per 3x3 conv of the U-Net)",
        "bound": "mfma", "achieved": round(achieved, 2), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / F16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
        "f32_matrix_peak": F32_MATRIX_PEAK_TFLOPS, "frac_of_f32_matrix_peak": round(achieved / F32_MATRIX_PEAK_TFLOPS, 3),
        "mfma_issued_tflops": round(achieved * F16X3_ISSUE_FACTOR, 2),
        "frac_mfma_issued": round(achieved * F16X3_ISSUE_FACTOR / F16_MFMA_PEAK_TFLOPS, 4),
        "note": "achieved/frac count algorithmic f32-conv FLOPs against the dense f16 MFMA peak the kernel runs on; the split issues "
                "3.05x as many f16 MFMA FLOPs (mfma_issued_tflops) to deliver f32-class products, i.e. 2x the chip's f32 MFMA "
                "peak (frac_of_f32_matrix_peak); a bare v_mfma_f32_16x16x32_f16 loop sustains 1880 TFLOP/s under this socket's "
                "power cap (profiles/r01f_mfma_power_probe.log, DESIGN.md 7)",
        "launches": len(probe), "avg_launch_ms": round(conv_ms / n, 4), "flops_per_launch": conv_flops / n,
        "share_of_step": round(conv_ms / 1e3 / max(1e-9, elapsed), 3),
    }


HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (about 6.3 TB/s achievable)


def framewise_rooflines(hip, mix_dev, reps: int = 8) -> list:
    """the framewise / scan kernels are HBM-bound streaming passes; report achieved GB/s per kernel on the
    C2 track (algorithmic bytes per launch / HIP-event time on the launch stream), outside the timed region.  Each launch
    reads a DIFFERENT copy of the track: 8 x 42 MB of float32 (and 8 x 85 MB of float64 for the scan) exceed the 256 MB
    Infinity Cache, so these are HBM rates, not cache rates."""
    n = int(mix_dev.numel())
    copies = [mix_dev] + [mix_dev.clone() for _ in range(7)]
    turn = [0]

    def nxt():
        turn[0] = (turn[0] + 1) % len(copies)
        return copies[turn[0]]

    def timed(fn) -> float:
        fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps

    rows = []

    def add(name, fn, nbytes):
        ms = timed(fn)
        gbs = nbytes / (ms * 1e-3) / 1e9
        rows.append({"kernel": name, "bytes": int(nbytes), "ms": round(ms, 4), "GB/s": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)})

    for frame, hop in ((4410, 2205), (2048, 441), (2205, 882), (1102, 441)):
        nf = 1 + n // hop
        add(f"ac_frame_rms({frame},{hop})", lambda f=frame, h=hop: hip.frame_rms(nxt(), f, h), 4 * n + 4 * nf)
    add("ac_frame_rms_multi(2205/882 + 1102/441 + 2048/441: the vocal stem's three series in one pass)",
        lambda: hip.frame_rms_multi(nxt(), [(2205, 882), (1102, 441), (2048, 441)]), 4 * n + 4 * (2 * (1 + n // 441) + 1 + n // 882))
    add("ac_stft2048_features(hop 441, flatness)", lambda: hip.stft2048_features(nxt(), 441, want_flat=True, want_mel=False),
        4 * n + 4 * (1 + n // 441))
    add("ac_stft2048_features(hop 512, mel-128)", lambda: hip.stft2048_features(nxt(), 512, want_flat=False, want_mel=True),
        4 * n + 512 * (1 + n // 512))
    _, mel = hip.stft2048_features(mix_dev, 512, want_flat=False, want_mel=True)
    add("ac_onset_strength(mean)", lambda: hip.onset_strength(mel, 512, "mean"), 2 * 512 * mel.shape[0] + 4 * mel.shape[0])
    add("ac_moving_meansq_db_f64(W 3528)", lambda: hip.moving_meansq_db(nxt(), 3528), 12 * n)
    dbs = [hip.moving_meansq_db(c, 3528) for c in copies]
    k = [0]

    def scan():
        k[0] = (k[0] + 1) % len(dbs)
        return hip.next_leq_scan(dbs[k[0]], -40.0)
    add("ac_next_leq_scan", scan, 16 * n)
    return rows


def framewise_accounting(n: int) -> dict:
    """SURVEY.md 8(d) asks for the fused-pass count: how many times a track's full-length waves cross HBM for the framewise / scan
    kernels, by construction of the product path (analysis/prefetch.py, analysis/features_cache.py, analysis/rhythm.py,
    cutting/refine.py), beside the reference's count for the same series.  n = samples per track."""
    w = 4 * n                                                     # one float32 wave
    rows = [
        # (wave, kernel, series it yields, float32 wave reads, other bytes, what the reference does)
        ("vocal", "ac_frame_rms_multi", "RMS 2205/882 (markers) + 1102/441 (detector) + 2048/441 (no-vocal runs)", 1, 0, "3 passes (+ a 4th, 2048/441 for the VPP multiplier, which is dead code and not computed here)"),
        ("vocal", "ac_stft2048_features hop 441", "spectral flatness (detector)", 1, 0, "1 pass"),
        ("vocal", "ac_moving_meansq_db_f64 + ac_next_leq_scan", "quiet-guard lookup (dB series f64 + next-quiet i64)", 1, 32 * n, "1 pass + an O(N * 3528) convolution + a 2N-step Python loop"),
        ("vocal, instrumental", "ac_sum_squares x2", "separation confidence", 2, 0, "2 passes"),
        ("mix", "ac_sum_squares", "separation confidence", 1, 0, "1 pass"),
        ("mix", "ac_frame_rms 2048/441", "guard floor percentile", 1, 0, "1 pass"),
        ("mix", "ac_moving_meansq_db_f64 + ac_next_leq_scan", "quiet-guard lookup", 1, 32 * n, "as for the vocal stem"),
        ("mix", "ac_segment_frame_rms 4410/2205 (all chunks, one launch)", "cache RMS", 1, 0, "1 pass (per chunk)"),
        ("mix", "ac_stft2048_features hop 2205 (all chunks, one launch)", "cache flatness + mel-128 onset", 1, 0, "2 passes (flatness, onset_strength)"),
        ("mix", "ac_stft2048_features hop 512, mel-128", "BPM onset envelope (median) + tempogram", 1, 0, "2 passes (mean and median onset envelopes)"),
    ]
    reads = sum(r[3] for r in rows)
    return {"framewise_fused_passes": reads, "framewise_passes_reference": 15,
            "framewise_fused_note": "full-length float32 wave reads per track by the framewise / scan kernels (the reference path: 6 RMS + 2 STFT-2048 + "
                                    "2 mel-onset + 2 guard + 3 mean-square passes = 15); the three RMS series of the vocal stem are ONE pass (ac_frame_rms_multi), "
                                    "all of them are queued ahead of the host logic that consumes them (analysis/prefetch.py)",
            "framewise_hbm_bytes_per_track": int(reads * w + sum(r[4] for r in rows)),
            "framewise_passes": [{"wave": r[0], "kernel": r[1], "yields": r[2], "wave_reads": r[3], "other_bytes": int(r[4]), "reference": r[5]} for r in rows]}


C3_SEED0 = 100                         # C3 = 32 x C2 with seeds 100-131
C3_GOLDEN = ROOT / "tests" / "golden" / "c3_n1_sha1.json"


def _launch_ranks(n: int, argv: list) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of a torch.distributed.run child
    (never an exec of this process; nothing here has touched the GPU yet) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + argv
    return subprocess.run(cmd).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("c2", "c3", "c4", "c5"), default="c2",
                    help="c2: 4-min tracks, K per rank (default; N > 1 deals the C3 seeds); c3: exactly the 32 C3 tracks over the ranks; "
                         "c4: the c2 tracks in vpbd_acoustic mode with the Silero network as the chunked VAD; c5: 30-min tracks")
    ap.add_argument("--track-seconds", type=float, default=None, help="default 240 (c2 / c3) or 1800 (c5)")
    ap.add_argument("--items-per-forward", type=int, default=None,
                    help="U-Net windows per forward; default 64 (a 4-min track in one forward: -1.2 %% per track), 32 for --config c4: with two "
                         "tracks in flight the long VPBD / Silero tail of one track runs beside the other's U-Net, and behind 2x longer launches "
                         "its small kernels wait longer (profiles/r04ak)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=240.0,
                    help="length of the C2 track's prefix the CPU oracle is timed on (default: the whole 240 s track, about 170 s of CPU "
                         "on 16 threads); 0 disables the CPU baseline leg")
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="tracks in flight per GPU (audio_cut_amd.batch.TrackPipeline): 1 = strictly one after the other")
    ap.add_argument("--no-separation-gate", action="store_true",
                    help="experiment: let the U-Nets of the tracks in flight run concurrently (default: one separation on the GPU at a time)")
    ap.add_argument("--shared-unet-stream", type=int, default=1,
                    help="1 (default): every worker queues its separation on the pipeline's one U-Net stream, the gate only serialises the "
                         "queueing; 0: each worker on its own stream, the gate held until the U-Net has left the GPU (round 2's scheme)")
    ap.add_argument("--write-golden", default=None, metavar="PATH",
                    help="(c3, N = 1) write the per-track SHA-1 table of this run to PATH (committed as tests/golden/c3_n1_sha1.json)")
    args = ap.parse_args()
    if args.items_per_forward is None:
        args.items_per_forward = 32 if args.config == "c4" else 64

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(_launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or run `python bench.py --gpus {args.gpus}` and let it start the ranks")
    import torch.distributed as dist
    # Rehearsal switch for a ONE-GPU box only (never set by the driver): AC_BENCH_REHEARSAL=1 runs all ranks on cuda:0 with
    # the gloo backend, so the N > 1 control flow (barriers, MAX over ranks, summary gather) can be exercised without N GPUs.
    rehearsal = os.environ.get("AC_BENCH_REHEARSAL", "") == "1"
    dev_index = 0 if rehearsal else local_rank
    # AC_BENCH_FORCE_DIST=1 (tests/test_distributed_gpu.py): take the distributed branch with ONE rank too, so that RCCL's
    # initialisation, a barrier, the MAX all-reduce, all_gather_object over a device and the shutdown run on a one-GPU box
    use_dist = world > 1 or os.environ.get("AC_BENCH_FORCE_DIST", "") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # N processes share the host: keep each rank's CPU-side torch / OpenMP pools to its share of the cores
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // world)))
        torch.cuda.set_device(dev_index)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    device = f"cuda:{dev_index}"
    torch.cuda.set_device(dev_index)

    from audio_cut_amd import _native, batch
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    from audio_cut_amd.testing import signals

    sr = 44100
    spec = TfcTdfSpec()
    weights = synth_weights(spec, seed=0)
    hip = _native.Context(device)
    backend = MDX23HipBackend(weights=weights, ctx=hip, max_items_per_forward=args.items_per_forward)
    backend.load_model()

    # ---- the job: which tracks, which of them are mine ------------------------------------------------------------------
    track_s = float(args.track_seconds if args.track_seconds is not None else (1800.0 if args.config == "c5" else 240.0))
    if args.config == "c3":
        if 32 % world:
            raise SystemExit("--config c3 needs a rank count that divides 32")
        steps = 32 // world
        seeds = [C3_SEED0 + i for i in range(32)]
    elif args.config == "c5":
        steps = args.steps
        seeds = [5 + i for i in range(world * steps)]
    else:
        steps = args.steps
        seeds = ([2] + [C3_SEED0 + i for i in range(steps - 1)]) if world == 1 else [C3_SEED0 + i for i in range(world * steps)]
    mine = batch.assign_tracks([track_s] * len(seeds), world)[rank]            # longest-processing-time-first (equal lengths: round robin)
    assert len(mine) == steps
    make = (lambda sd: signals.c5_long_form(track_s, seed=sd)) if args.config == "c5" else (lambda sd: signals.c2_song(track_s, seed=sd))
    depth = max(1, min(int(args.pipeline_depth), steps))
    mode = "vpbd_acoustic" if args.config == "c4" else "v2.2_mdd"
    c4_fixture = ROOT / "tests" / "golden" / "c4_full_oracle.npz"
    vad_fn = None
    if args.config == "c4":
        # BASELINE configs[3]: SileroChunkVAD focus windows.  The real Silero weights cannot be fetched offline: seeded synthetic
        # weights of the v5 architecture (audio_cut_amd/testing/silero_synth.py), output layer calibrated once with the CPU oracle - the
        # two calibration numbers travel in the committed fixture, so nothing under oracle/ or tests/ is imported here
        from audio_cut_amd.testing.silero_synth import synth_silero_weights
        from audio_cut_amd.detectors.silero_vad import SileroHipVad
        fx = np.load(c4_fixture)
        vad_fn = SileroHipVad(sr, synth_silero_weights(int(fx["silero_seed"]), affine=fx["silero_affine"]), hip)
    splitters = [SeamlessSplitter(sr, separator=EnhancedVocalSeparator(sr, backend=backend, vad_inference_fn=vad_fn)) for _ in range(depth)]
    pipeline = batch.TrackPipeline(splitters, device)
    gate = pipeline.separation_gate if (depth > 1 and not args.no_separation_gate) else None
    unet_stream = pipeline.unet_stream if (depth > 1 and args.shared_unet_stream and gate is not None) else None

    def job_for(mix, mix_dev):
        return lambda sp: sp.split_track(mix, mode=mode, audio_dev=mix_dev, separation_gate=gate, unet_stream=unet_stream)

    # every timed track is generated and made resident in HBM before the clock starts (BASELINE contract); warm-up tracks are
    # tracks of their own (seeds 50, 51, ...) so nothing of a timed track has been seen before
    tracks = [make(seeds[i]) for i in mine]
    tracks_dev = [hip.to_device(m) for m in tracks]
    n_warm = max(args.warmup, depth) if args.warmup > 0 else 0
    if n_warm:
        warm = [make(50 + (k % 2)) for k in range(min(n_warm, 2))]
        warm_dev = [hip.to_device(m) for m in warm]
        pipeline.run([job_for(warm[k % len(warm)], warm_dev[k % len(warm)]) for k in range(n_warm)])   # every worker (stream, allocator pools) warms up
        del warm, warm_dev
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()

    backend.net.conv_probe = []        # HIP events around every 3x3 conv launch of the timed region (same stream)
    unet_ms = stft_ms = istft_ms = 0.0
    items = 0
    phases = {"separate_s": 0.0, "detect_s": 0.0, "finalize_s": 0.0}
    policy_s = 0.0
    summaries = []
    sampler = SocketSampler(dev_index).start() if rank == 0 else None
    t0 = time.perf_counter()
    step_results = pipeline.run([job_for(m, d) for m, d in zip(tracks, tracks_dev)])      # exactly K steps (tracks), `depth` of them in flight
    torch.cuda.synchronize()
    t_done = time.perf_counter()
    socket_state = sampler.stop() if sampler else {}
    for step, res in enumerate(step_results):
        st = res["gpu_meta"].get("gpu_pipeline_stage_ms", {})
        unet_ms += st.get("unet_ms", 0.0); stft_ms += st.get("stft_ms", 0.0); istft_ms += st.get("istft_ms", 0.0)
        items += int(st.get("n_items", 0))
        for k in phases:
            phases[k] += res["timings"].get(k, 0.0)
        policy_s += float(res.get("timings_policy_s", 0.0))
        sm = batch.summarize(mine[step], res["sample_boundaries"], track_s, {"step_s": (t_done - t0) / steps})
        sm["seed"] = seeds[mine[step]]; sm["rank"] = rank
        sm["cuts_sha1"] = batch.summarize(0, res.get("cuts_samples", []), 0.0)["boundaries_sha1"]
        summaries.append(sm)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    all_summaries = batch.gather_summaries(summaries)      # batch completion (RCCL barrier + all_gather_object)
    probe = backend.net.conv_probe
    backend.net.conv_probe = None
    conv_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in probe)
    conv_flops = sum(f for _, _, f in probe)

    if rank == 0:
        res = step_results[-1]
        total_audio = track_s * steps * world
        flops = spec.flops_per_item() * items
        achieved = flops / (unet_ms / 1e3) / 1e12 if unet_ms > 0 else 0.0
        # "per-track boundaries identical to the single-GPU run" (SURVEY.md 8d C3): every track against the committed N = 1 result
        golden = {}
        if C3_GOLDEN.exists() and args.config in ("c2", "c3") and track_s == 240.0:
            golden = json.loads(C3_GOLDEN.read_text()).get("tracks", {})
        checked = [(d, golden[str(d["seed"])]) for d in all_summaries if str(d["seed"]) in golden]
        bad = [d["seed"] for d, g in checked if d["boundaries_sha1"] != g["boundaries_sha1"] or d["cuts_sha1"] != g["cuts_sha1"]]
        parity = {"reference": "tests/golden/c3_n1_sha1.json (bench.py --config c3 --write-golden on one GPU)" if golden else None,
                  "tracks_checked": len(checked), "tracks_identical": len(checked) - len(bad), "mismatched_seeds": bad,
                  "unchecked_seeds": [d["seed"] for d in all_summaries if str(d["seed"]) not in golden]}
        # the seed-2 track is BASELINE configs[1] / [3] itself: its guard boundaries and manifest cuts against the CPU oracle's
        # committed result (tests/golden/c2_full_oracle.npz, c4_full_oracle.npz: data, written by tests/golden/make_*.py)
        oracle_fx = {"c2": ROOT / "tests" / "golden" / "c2_full_oracle.npz", "c4": c4_fixture}.get(args.config)
        vs_oracle = None
        if oracle_fx is not None and oracle_fx.exists() and track_s == 240.0:
            for st_res, d in zip(step_results, summaries):
                if d["seed"] == 2:
                    fx = np.load(oracle_fx)
                    vs_oracle = {"fixture": str(oracle_fx.relative_to(ROOT)), "seed": 2,
                                 "boundaries_exact": [int(v) for v in st_res["sample_boundaries"]] == fx["sample_boundaries"].tolist(),
                                 "manifest_cuts_exact": [int(v) for v in st_res.get("cuts_samples", [])] == fx["cuts"].tolist(),
                                 "n_boundaries": int(len(fx["sample_boundaries"]))}
                    if args.config == "c4":
                        vs_oracle["vpbd_selected_exact"] = list(st_res.get("vpbd_selected_times", [])) == fx["vpbd_selected"][:, 0].tolist()
                        vs_oracle["vad_segments"] = len(st_res.get("vad_segments") or [])
        parity["vs_cpu_oracle_fixture"] = vs_oracle
        parity_ok = (not bad) and (vs_oracle is None or all(v for k, v in vs_oracle.items() if k.endswith("_exact")))
        if args.write_golden and args.config == "c3" and world == 1:
            Path(args.write_golden).write_text(json.dumps({"what": "per-track SHA-1 of the guard boundaries and of the manifest cuts of the 32 C3 tracks "
                                                     "(c2_song 240 s, seeds 100-131, synth weights seed 0), one MI355X, bench.py --config c3",
                                             "tracks": {str(d["seed"]): {"boundaries_sha1": d["boundaries_sha1"], "cuts_sha1": d["cuts_sha1"],
                                                                         "n_boundaries": d["n_boundaries"]} for d in all_summaries}}, indent=1) + "\n")
        workloads = {
            "c2": "BASELINE configs[1]: 4-min 44.1 kHz synthetic song (stereo generator, mono down-mix), chunked MDX23 separation "
                  "(full-size TFC-TDF, seeded synthetic weights) + TrackFeatureCache + chunked VAD + PureVocalPauseDetector + "
                  "quiet-guard + boundary policy; one track per step, every step a different seed, resident in HBM"
                  + ("" if world == 1 else f"; N > 1: the C3 seeds 100..{C3_SEED0 + world * steps - 1} dealt by assign_tracks"),
            "c3": "BASELINE configs[2]: the 32 x 4-min C3 tracks (seeds 100-131) dealt over the ranks by assign_tracks, same path as configs[1]",
            "c4": "BASELINE configs[3]: the configs[1] tracks in vpbd_acoustic mode - chunked MDX23 separation + TrackFeatureCache + SileroChunkVAD "
                  "(the Silero v5 network as HIP kernels, seeded synthetic weights) focus windows + PureVocalPauseDetector + VPBD candidate pool / "
                  "scoring / global planner + quiet-guard + boundary policy; one track per step, resident in HBM",
            "c5": "BASELINE configs[4] after the loader: 30-min 44.1 kHz long-form track (C2 generator looped with per-section seeds), "
                  "240 chunks / 480 U-Net items per track, same path as configs[1]",
        }
        out = {
            "metric": METRIC, "value": round(total_audio / elapsed, 3), "unit": "audio-s/s", "n_gpus": world,
            "value_is": "WHOLE-JOB aggregate over n_gpus (all ranks' audio seconds / max-over-ranks wall time), as the bench contract "
                        "prescribes; divide by n_gpus for the per-GPU figure the metric's name refers to",
            "value_per_gpu": round(total_audio / elapsed / world, 3), "parity_ok": bool(parity_ok),
            "batch_completion": ({"backend": dist.get_backend(), "collectives": "barrier, all_reduce(MAX), all_gather_object",
                                  "summaries_gathered": len(all_summaries)} if use_dist else None),
            "steps": steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workloads[args.config],
                "track_seconds": track_s, "chunks_per_track": res["gpu_meta"].get("gpu_pipeline_chunks"),
                "unet_items_per_track": items // max(1, steps),
                "items_per_forward": args.items_per_forward, "tracks_per_gpu": steps, "sharding": "track-per-rank (assign_tracks, LPT)",
                "track_seeds": [d["seed"] for d in all_summaries], "pipeline_depth": depth, "shared_unet_stream": unet_stream is not None,
                "real_time_factor": round(total_audio / elapsed / world, 2),
            },
            "roofline": roofline_conv(probe, conv_ms, conv_flops, elapsed),
            "unet_forward": {
                "what": "whole TFC-TDF forward (one hand-written MFMA kernel per layer: f16x3 convs / TDF GEMMs / 2x2 resampling, exact-f32 "
                        f"narrow TDF pairs), {args.items_per_forward} items per forward; algorithmic f32 FLOPs",
                "achieved": round(achieved, 2), "unit": "TFLOP/s", "f32_matrix_peak": F32_MATRIX_PEAK_TFLOPS, "flops_per_item": spec.flops_per_item(), "items": items,
                "ms_total": round(unet_ms, 2), "share_of_step": round(unet_ms / 1e3 / max(1e-9, elapsed), 3),
            },
            "phases_note": "per-track phase times; with pipeline_depth > 1 the host-bound tail of one track overlaps the next track's U-Net, "
                           "so they add up to more than ms_per_step",
            "phases_ms_per_step": {"separate": round(phases["separate_s"] / steps * 1e3, 2),
                                   "detect": round(phases["detect_s"] / steps * 1e3, 2),
                                   "finalize": round(phases["finalize_s"] / steps * 1e3, 2),
                                   "boundary_policy": round(policy_s / steps * 1e3, 2),
                                   "mdx_stft": round(stft_ms / steps, 2), "unet": round(unet_ms / steps, 2),
                                   "mdx_istft": round(istft_ms / steps, 2)},
            "n_boundaries": all_summaries[0]["n_boundaries"], "boundaries_sha1": all_summaries[0]["boundaries_sha1"],
            "n_manifest_cuts": len(step_results[0].get("cuts_samples", [])), "segment_layout_applied": bool(step_results[0].get("segment_layout_applied", False)),
            "tracks_completed": len(all_summaries), "parity_vs_single_gpu": parity,
            "track_hashes": {str(d["seed"]): {"boundaries_sha1": d["boundaries_sha1"], "cuts_sha1": d["cuts_sha1"], "n_boundaries": d["n_boundaries"]}
                             for d in all_summaries},
        }
        if world == 1:
            # single-stream latency next to the pipelined throughput: one more track, strictly alone on the GPU (outside the timed region)
            lat_ms = []
            for _ in range(3):           # the median of three: a single probe caught allocator / clock hiccups of 2-3x now and then
                torch.cuda.synchronize()
                tl = time.perf_counter()
                lat = pipeline.run([job_for(tracks[0], tracks_dev[0])])[0]      # the same machinery (worker stream, U-Net stream, allocator pools), one track in flight
                torch.cuda.synchronize()
                lat_ms.append((time.perf_counter() - tl) * 1e3)
            out["single_stream_latency_ms"] = round(sorted(lat_ms)[1], 2)
            out["single_stream_latency_note"] = "one track alone on the GPU (one job through the pipeline: host tail not overlapped), median of " + \
                str([round(v, 1) for v in lat_ms]) + "; same result: " + str(lat["sample_boundaries"] == step_results[0]["sample_boundaries"])
            out["framewise_rooflines"] = framewise_rooflines(hip, tracks_dev[0])
            out.update(framewise_accounting(int(tracks_dev[0].numel())))
        if world == 1 and args.cpu_baseline_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_seconds, weights, spec)
        out["socket_under_load"] = socket_state
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    # a throughput figure on top of boundaries that differ from the committed single-GPU / CPU-oracle results is not a result:
    # fail the run.  That includes the one-GPU rehearsal of the N > 1 control flow (AC_BENCH_REHEARSAL=1, ranks sharing one device):
    # the corruption that mode showed in round 2 was a kernel-level defect, found and removed in round 3.
    if rank == 0 and not parity_ok:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
