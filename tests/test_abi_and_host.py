"""C-ABI library loads and exports every symbol include/audiocut_hip.h declares; host-side logic of the
product (no GPU compute).  CPU only."""
import ctypes
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SR = 44100


@pytest.fixture(scope="module")
def lib():
    from audio_cut_amd import _native
    if not _native.library_path().exists():
        subprocess.run(["make", "-C", str(ROOT / "audio_cut_amd" / "csrc")], check=True)
    return _native.load()


def test_header_symbols_exported(lib):
    header = (ROOT / "include" / "audiocut_hip.h").read_text()
    names = sorted(set(re.findall(r"\b(ac_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 20
    from audio_cut_amd import _native
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_native.SIGNATURES) == set(names)
    assert lib.ac_abi_version() == 6


def test_size_helpers_need_no_gpu(lib):
    assert lib.ac_tempogram_parts(20672) == (20672 + 63) // 64
    assert lib.ac_next_leq_scratch(10_584_000) == 2 * ((10_584_000 + 4095) // 4096) + 2


def test_invalid_arguments_return_error_codes_not_exceptions(lib):
    rc = lib.ac_frame_rms(None, None, 0, 0, 0, 0, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.ac_last_error()


def test_product_fails_loudly_without_gpu():
    import torch
    from audio_cut_amd import _native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.NativeError):
        _native.Context("cuda:0")
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    with pytest.raises(_native.NativeError):
        EnhancedVocalSeparator(SR)
    from audio_cut_amd.detectors.pure_vocal_pause_detector import PureVocalPauseDetector
    with pytest.raises(_native.NativeError):
        PureVocalPauseDetector(SR).detect_pure_vocal_pauses(np.zeros(SR, np.float32))


def test_shared_unet_stream_needs_the_gate():
    """One U-Net stream for every worker is only safe together with the lock that keeps two tracks' launches from interleaving."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    sep = object.__new__(EnhancedVocalSeparator)
    sep._primary_backend = None
    with pytest.raises(ValueError, match="separation_gate"):
        sep.separate_for_detection(np.zeros(SR, np.float32), unet_stream=object())


def test_product_never_imports_the_oracle():
    for path in (ROOT / "audio_cut_amd").rglob("*.py"):
        text = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{path} imports the oracle"


def test_host_beat_dp_matches_oracle_dp(lib):
    from audio_cut_amd import _native
    from oracle import librosa_ops as L
    rng = np.random.default_rng(3)
    env = np.abs(rng.standard_normal(3000)) * (1 + np.sin(np.arange(3000) * 2 * np.pi / 43.0))
    for period in (43, 20, 7):
        score = L._beat_local_score(env, period)
        back_o, cum_o = L._beat_track_dp(score, period, 100.0)
        back, cum = _native.host_beat_dp(score, period, 100.0)
        assert np.array_equal(back, back_o)
        np.testing.assert_allclose(cum, cum_o, rtol=1e-12)


def test_host_beat_dp_breaks_ties_like_numpy():
    """Round-3 soak finding (profiles/r03_parity_soak_k.log: c1_sine_silence 120 s, seed 423): on a burst / digital-silence track the
    beat DP is full of candidates that tie to the last bit (the local score is ~1e-100 for seconds on end), so the transition weights
    must be numpy's to the ulp: `-tightness * log(.) ** 2` squares first - `(-tightness * l) * l` rounds differently and moved a
    backlink, a beat (frame 1404 vs 1403) and, through the layout refiner's beat snap, a manifest cut by 1 744 samples.  Bit-for-bit
    tables, and the beat grid of that track's own onset envelope."""
    from audio_cut_amd import _native
    from audio_cut_amd.analysis import rhythm as R
    from audio_cut_amd.testing import signals
    from oracle import chunking as C, features as OF, librosa_ops as L
    sr = 44100
    mix = signals.c1_sine_silence(120.0, seed=423)
    plans = C.chunk_plan(len(mix) / float(sr), 10.0, 2.5, 0.5)
    feat = OF.ChunkFeatureOracle(sr)
    for plan, (cs, ce, es, ee) in zip(plans, C.plan_sample_ranges(plans, sr, len(mix))):
        if ce > cs and ee > es:
            feat.add_chunk(plan, mix[cs:ce], sr)
    cache = feat.finalize(mix)
    env = np.asarray(cache.onset_envelope, dtype=np.float32)
    hop = cache.hop_length
    bpm = float(L.tempo(env, sr=sr, hop_length=hop)[0])
    period = round(60.0 * (float(sr) / hop) / bpm)
    score = L._beat_local_score(env, period)
    back_o, cum_o = L._beat_track_dp(score, period, 100.0)
    back, cum = _native.host_beat_dp(score, period, 100.0)
    assert np.array_equal(back, back_o) and np.array_equal(cum, cum_o)          # every tie the same way, every sum the same bits
    want = np.round(np.asarray(cache.beat_times) * sr / hop).astype(int)
    assert want.size == 106 and np.array_equal(R.beat_frames(env, bpm, sr, hop), want)


def test_gpu_pipeline_mirror(golden_dir):
    from audio_cut_amd.utils import gpu_pipeline as gp
    assert set(gp.__all__) == {"Streams", "ChunkPlan", "PipelineConfig", "PipelineContext", "PinnedBufferPool", "InflightLimiter",
                               "OrtExecutionConfig", "ensure_ort_dependencies", "build_pipeline_context", "chunk_schedule",
                               "create_streams", "record_event", "select_device", "wait_event"}
    rows = np.load(golden_dir / "chunk_schedule.npz")["rows"]
    for total in np.unique(rows[:, 0]):
        ref = rows[rows[:, 0] == total][:, 1:]
        got = np.array([[p.index, p.start_s, p.end_s, p.halo_left_s, p.halo_right_s] for p in gp.chunk_schedule(float(total))])
        assert np.array_equal(got, ref)
    cfg = gp.PipelineConfig.from_mapping({"enable": True, "chunk_seconds": 8, "overlap_s": 2, "halo_seconds": 1, "strict_mode": True,
                                          "ort": {"disable_trt": False}})
    assert (cfg.enable, cfg.chunk_s, cfg.overlap_s, cfg.halo_s, cfg.strict_gpu, cfg.ort_config.disable_trt) == (True, 8.0, 2.0, 1.0, True, False)
    assert gp.select_device("cpu") == "cpu"
    ctx = gp.PipelineContext(device="cpu", streams=gp.Streams(), plans=gp.chunk_schedule(25.0), pinned_pool=None, limiter=None,
                             config=gp.PipelineConfig())
    meta = ctx.to_meta()
    for key in ("gpu_pipeline_enabled", "gpu_pipeline_used", "gpu_pipeline_device", "gpu_pipeline_chunks", "gpu_pipeline_streams",
                "gpu_pipeline_inflight_limit", "gpu_pipeline_prefetch", "gpu_pipeline_align_hop", "gpu_pipeline_config"):
        assert key in meta
    p = gp.chunk_schedule(25.0)[1]
    assert (p.effective_start_s, p.effective_end_s) == (8.0, 17.0) and p.as_slice(SR) == slice(330750, 771750)
    lim = gp.InflightLimiter(1)
    with lim.acquire():
        with pytest.raises(RuntimeError):
            with lim.acquire(timeout=0.01):
                pass


def test_config_overrides():
    from audio_cut_amd import config as cfg
    assert cfg.get_config("quality_control.enforce_quiet_cut.floor_percentile") == 0.5
    assert cfg.get_config("quality_control.nms_topk_per_10s", None) is None
    saved = cfg.snapshot()
    cfg.set_runtime_config({"quality_control.min_split_gap": 2.0, "gpu_pipeline.prefer_device": "cuda:1"})
    assert cfg.get_config("quality_control.min_split_gap") == 2.0
    assert cfg.get_config("gpu_pipeline")["prefer_device"] == "cuda:1"
    cfg.restore(saved)
    assert cfg.get_config("quality_control.min_split_gap") == 1.2


def test_chunk_vad_bookkeeping_golden(golden_dir):
    from audio_cut_amd.detectors.silero_chunk_vad import SileroChunkVAD, speech_timestamps
    from audio_cut_amd.testing import signals
    from audio_cut_amd.utils.gpu_pipeline import chunk_schedule
    from oracle import detector as OD, vad as OV

    def fake(chunk):
        blk = 2205
        n = len(chunk) // blk
        act = np.abs(chunk[: n * blk]).reshape(n, blk).mean(axis=1) > 0.02
        return [{"start": a * blk, "end": b * blk} for a, b, v in OD._runs(act) if v]

    g = np.load(golden_dir / "chunk_vad.npz")
    voc = signals.vocal_like(40.0, seed=11)
    vad = SileroChunkVAD(sample_rate=SR, merge_gap_ms=120.0, focus_pad_s=0.2, inference_fn=fake)
    for p in chunk_schedule(40.0):
        a = int(round(p.start_s * SR)); b = int(round(p.end_s * SR))
        vad.process_chunk(p, voc[a:b], SR)
    assert np.array_equal(np.array([[s["start"], s["end"]] for s in vad.finalize()]), g["segments"])
    assert np.array_equal(np.array(vad.to_focus_windows()), g["focus"])
    rng = np.random.default_rng(0)
    probs = np.clip(np.convolve(rng.uniform(0, 1, 400), np.ones(9) / 9, "same") * 1.6 - 0.3, 0, 1)
    assert speech_timestamps(probs, 400 * 1411 - 77, 1411, SR, 0.35, 250, 700, 150) == \
        OV.speech_timestamps(probs, 400 * 1411 - 77, 1411, SR, 0.35, 250, 700, 150)


def test_host_rhythm_and_nms_against_oracle():
    from audio_cut_amd.analysis import rhythm
    from audio_cut_amd.cutting.refine import CutPoint, nms_min_gap
    from oracle import librosa_ops as L, refine as OR
    rng = np.random.default_rng(5)
    for hop in (512, 2205):
        env = np.abs(rng.standard_normal(700)).astype(np.float32) * (rng.uniform(0, 1, 700) > 0.7)
        assert np.array_equal(rhythm.onset_detect(env, SR, hop), L.onset_detect(env, SR, hop))
    assert len(rhythm.onset_detect(np.zeros(50, np.float32), SR, 512)) == 0
    env = np.abs(rng.standard_normal(2000)) * (1 + np.sin(np.arange(2000) * 2 * np.pi / 43.07))
    assert np.array_equal(rhythm.beat_frames(env, 120.0, SR, 512), L.beat_tracker(env, 120.0, SR / 512, 100, True))
    bpms, lp = rhythm.tempo_logprior(689, 512, SR)
    assert np.isinf(lp[0]) and bpms[1] == 60.0 * SR / 512
    pts = [(float(t), float(s)) for t, s in zip(rng.uniform(0, 60, 80), np.round(rng.uniform(0, 1, 80), 1))]
    a = nms_min_gap([CutPoint(t, s) for t, s in pts], 1.2, 30, max_per_window=2)
    b = OR.nms_min_gap([OR.Cut(t, s) for t, s in pts], 1.2, 30, max_per_window=2)
    assert [(p.t, p.score) for p in a] == [(p.t, p.score) for p in b]


def test_detector_host_helpers_against_oracle():
    from audio_cut_amd.detectors import pure_vocal_pause_detector as P
    from oracle import detector as OD
    vad = [{"start": 1.0, "end": 6.2}, {"start": 6.25, "end": 13.0}, {"start": 13.6, "end": 20.5}, {"start": 21.4, "end": 26.5}]
    det = P.PureVocalPauseDetector.__new__(P.PureVocalPauseDetector)
    assert list(det._focus_windows_from_vad_segments(vad, pad_s=0.2, min_width_s=0.0)) == OD.focus_windows_from_vad(vad, 0.2, 0.0)
    for bpm in (None, 70.0, 120.0, 150.0):
        for mdd in (None, 0.3, 0.9):
            a = P.resolve_threshold(0.26, {"clamp_min": 0.85, "clamp_max": 1.15}, P.AdaptStats(bpm, mdd))
            b = OD.resolve_threshold(0.26, {"clamp_min": 0.85, "clamp_max": 1.15}, bpm, mdd)
            assert (a.peak_ratio, a.rms_ratio) == (b.peak_ratio, b.rms_ratio)
            assert P.resolve_min_pause(0.5, 1.0, P.AdaptStats(bpm, mdd)) == OD.resolve_min_pause(0.5, 1.0, bpm)


def test_tfc_tdf_folded_net_matches_unfused_graph_on_cpu():
    import torch
    from audio_cut_amd.separation.tfc_tdf import TfcTdfNet, TfcTdfSpec, synth_weights
    from oracle.separator import unet_forward
    full = TfcTdfSpec()
    assert abs(full.param_count() - 16.67e6) < 0.05e6            # = the 66.8 MB Kim_Vocal_1.onnx at 4 bytes/param
    assert abs(full.flops_per_item() - 758.9e9) < 1e9
    spec = TfcTdfSpec(dim_f=256, dim_t=32, g=8)
    w = synth_weights(spec, seed=1, calib_t=32)
    import unet_torch
    net = TfcTdfNet(w, spec)                   # folded weights only (no HIP context): evaluated by the test helper
    # truth = the unfused graph in float64 on seeded inputs (two float32 evaluations of a ReLU network may differ by more than either
    # differs from the truth, and an unseeded input made the comparison depend on the draw)
    w64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in w.items()}
    for seed in (0, 1, 3):
        x = torch.randn(2, 4, 256, 32, generator=torch.Generator().manual_seed(seed)) * 0.5
        a = unet_forward(x.double(), w64); b = unet_torch.forward(net, x)
        assert float((a - b).abs().max() / a.abs().max()) < 1e-5
    assert torch.equal(unet_torch.forward(net, torch.zeros(1, 4, 256, 32)), torch.zeros(1, 4, 256, 32))   # silence in, silence out
    from audio_cut_amd._native import NativeError
    with pytest.raises(NativeError):           # the product has no library / CPU path
        net(x)


def test_track_assignment_lpt():
    from audio_cut_amd import batch
    assert batch.assign_tracks([240.0] * 32, 8) == [[r + 8 * k for k in range(4)] for r in range(8)]
    parts = batch.assign_tracks([1800.0, 240.0, 240.0, 60.0, 600.0], 2)
    assert sorted(i for p in parts for i in p) == [0, 1, 2, 3, 4] and parts[0] == [0]


def test_runtime_override_of_a_section_is_seen_by_child_lookups():
    """Reference semantics (`config_manager.py:497-509`: `set_runtime_config` writes the value into the tree): a dict-valued
    override of a section replaces that subtree, so every dotted child lookup sees it (and falls back to the call-site default
    for keys the override leaves out); a child written afterwards lands inside it; the last write wins."""
    from audio_cut_amd import config as C
    saved = C.snapshot()
    try:
        C.reset_runtime_config()
        assert C.get_config("segment_layout.enable") is True and C.get_config("segment_layout.soft_min_s") == 5.0
        C.set_runtime_config({"segment_layout": {"enable": False, "soft_min_s": 7.0}})
        assert C.get_config("segment_layout.enable") is False and C.get_config("segment_layout.soft_min_s") == 7.0
        assert C.get_config("segment_layout.soft_max_s", 12.5) == 12.5
        assert C.get_config("segment_layout") == {"enable": False, "soft_min_s": 7.0}
        C.set_runtime_config({"segment_layout.soft_max_s": 9.0})
        assert C.get_config("segment_layout") == {"enable": False, "soft_min_s": 7.0, "soft_max_s": 9.0}
        C.set_runtime_config({"segment_layout": {"enable": True}})            # a later section write replaces the earlier child too
        assert C.get_config("segment_layout.soft_max_s", None) is None and C.get_config("segment_layout.enable") is True
        C.reset_runtime_config()
        C.set_runtime_config({"quality_control.enforce_quiet_cut": {"enable": False}})
        assert C.get_config("quality_control.enforce_quiet_cut.enable") is False
        assert C.get_config("quality_control.min_split_gap") == 1.2           # siblings of the overridden subtree keep their defaults
        assert C.get_config("quality_control")["enforce_quiet_cut"] == {"enable": False}
    finally:
        C.restore(saved)


def test_library_has_no_packed_float32_instructions(tmp_path):
    """The device code of libaudiocut_hip.so must not contain packed-float32 VALU instructions (csrc/Makefile NOPK).
    Measured on MI355X (profiles/r03_gpu_sharing_rootcause.log): `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` - what hipcc's SLP
    vectoriser makes of float2 complex arithmetic - returns wrong values for a quarter-wave at a time while a workgroup of the MFMA
    conv kernels shares the compute unit (another stream or another process alike); that was the "two processes on one GPU"
    corruption of round 2.  Guards the build flag: a kernel file compiled without it would bring the instructions back."""
    import shutil
    import subprocess
    from audio_cut_amd import _native
    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("no llvm-objdump in this image")
    lib = tmp_path / "lib.so"
    shutil.copy(_native.library_path(), lib)
    subprocess.run([str(objdump), "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    parts = sorted(tmp_path.glob("lib.so.*gfx950*"))
    assert parts, "no gfx950 code object found in the library"
    n_inst = 0
    for part in parts:
        asm = subprocess.run([str(objdump), "-d", str(part)], check=True, capture_output=True, text=True).stdout
        n_inst += asm.count("v_mfma_f32_16x16x32_f16")
        bad = [ln.strip() for ln in asm.splitlines() if "v_pk_" in ln and "_f32" in ln.split("v_pk_", 1)[1].split()[0]]
        bad += [ln.strip() for ln in asm.splitlines() if "v_pk_mov_b32" in ln]
        assert not bad, f"{len(bad)} packed-float32 instructions in {part.name}, e.g. {bad[:3]}"
    assert n_inst > 1000          # the disassembly really is the kernels (the conv / GEMM MFMA streams are in there)


def test_every_lds_dma_sets_m0_right_in_front_of_it(tmp_path):
    """The 96-channel conv tile issues its weights' LDS-DMA from inline assembly (csrc/ac_conv96.hip `w9_dma16`: hipcc's wait-count pass
    drains lgkmcnt to 0 at every LDS wait while it knows an LDS-DMA is pending), which writes M0 behind the compiler's back.  That is
    only sound if nothing relies on an M0 value across such an instruction: in the disassembly of the built library every
    `global_load_lds` must have its own M0 write close in front of it (measured: 1-8 instructions, the compiler schedules its own a few
    instructions early; never a value carried from one DMA to the next), and nothing else may read M0 (no s_movrel / ds_gws / sendmsg in
    these kernels).  The assembly-issued DMA and the compiler's live in different branches of the kernel (PIPE vs the row-exact path)."""
    import shutil
    import subprocess
    from audio_cut_amd import _native
    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("no llvm-objdump in this image")
    lib = tmp_path / "lib.so"
    shutil.copy(_native.library_path(), lib)
    subprocess.run([str(objdump), "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    n_dma = 0
    for part in sorted(tmp_path.glob("lib.so.*gfx950*")):
        lines = [ln.split("//")[0].strip() for ln in subprocess.run([str(objdump), "-d", str(part)], check=True, capture_output=True, text=True).stdout.splitlines()]
        lines = [ln for ln in lines if ln and not ln.endswith(":")]
        for i, ln in enumerate(lines):
            if "global_load_lds" in ln or ("buffer_load" in ln and " lds" in ln):
                n_dma += 1
                assert any("m0" in p for p in lines[max(0, i - 10): i]), f"LDS-DMA without its own M0 write: {lines[max(0, i - 11): i + 1]}"
            assert not any(op in ln for op in ("s_movrel", "v_movrel", "ds_gws", "s_sendmsg ")), ln
    assert n_dma > 50          # conv, TDF and resampling kernels all stream their weights by LDS-DMA


def test_resampling_filter_meets_the_soxr_hq_specification():
    """the reference resamples with librosa's default soxr_hq (audio_processor.py:44-48, vocal_pause_detector.py:189).
    libsoxr's coefficients cannot be pinned offline; its published HQ specification can: pass band flat to 0.9136 of the lower
    Nyquist frequency, stop band from that Nyquist on, >= 120 dB (20 bits).  The oracle's float64 design (oracle/resample.py, explicit
    Kaiser formulas) meets it, and the product's (audio_cut_amd/_native.py, scipy's kaiserord + firwin, float32) is the same filter."""
    from audio_cut_amd._native import Context
    from oracle import resample as ORS
    assert abs(ORS.SOXR_HQ_PASSBAND_END - 0.91363) < 1e-5 and abs(Context.SOXR_HQ_PASSBAND_END - ORS.SOXR_HQ_PASSBAND_END) < 1e-12
    for up, down in ((147, 160), (160, 441)):            # 48 kHz -> 44.1 kHz (loader), 44.1 kHz -> 16 kHz (Silero input)
        h = ORS.soxr_hq_lowpass(up, down)
        rate = max(up, down)
        assert h.size % 2 == 1 and np.array_equal(h, h[::-1]) and abs(float(np.sum(h)) - 1.0) < 1e-12
        pb = ORS.response_db(h, np.linspace(0.0, ORS.SOXR_HQ_PASSBAND_END, 1501) / rate)
        assert float(np.max(np.abs(pb))) < 1e-4                                     # flat (measured 5e-6 dB)
        sb = ORS.response_db(h, np.linspace(1.0, 6.0, 6001) / rate)
        assert float(np.max(sb)) < -120.0                                           # measured -124.5 dB at the edge, < -133 dB from 1.02 on
        hp, n_pre_remove = Context._resample_filter(up, down)
        n_pre_pad = hp.size - h.size
        assert n_pre_pad == down - ((h.size - 1) // 2) % down and n_pre_remove == ((h.size - 1) // 2 + n_pre_pad) // down
        assert not hp[:n_pre_pad].any()
        assert float(np.max(np.abs(hp[n_pre_pad:].astype(np.float64) - h * up))) < 1e-7 * float(np.max(h * up))
    x = np.random.default_rng(0).standard_normal(3000).astype(np.float32)
    assert ORS.resample(x, 16000, 44100).shape == (-(-3000 * 160 // 441),)          # librosa / resample_poly length rule
    assert np.array_equal(ORS.resample(x, 5, 5), x)
