"""SURVEY.md §8 row a13, the Silero half: `_detect_speech_timestamps` (`vocal_pause_detector.py:175-296`) as HIP kernels behind
`SileroChunkVAD`, against the torch-CPU restatement in oracle/silero.py, with seeded synthetic weights of the v5 architecture
(tests/silero_synth.py; the real weights cannot be fetched offline - parity unpinned)."""
import numpy as np
import pytest

from audio_cut_amd.testing import signals
from silero_synth import synth_silero_weights

SR = 44100


def test_silero_weight_files_and_oracle_contract(tmp_path):
    """CPU: the weight readers (.npz with TorchScript state-dict names, .onnx initializers with the same names, `_model.` prefix)
    validate names and shapes; the oracle's `detect_speech_timestamps` honours the VadFn contract (track-rate sample indices
    relative to the chunk, inside the chunk, non-overlapping and ordered)."""
    from audio_cut_amd.detectors import silero_vad as SV
    from oracle import silero as OS
    from onnx_writer import write_named_initializers_onnx
    w = synth_silero_weights(0)
    assert {k: v.shape for k, v in w.items()} == SV.TENSOR_SHAPES
    np.savez(tmp_path / "silero.npz", **{"_model." + k: v for k, v in w.items()})
    a = SV.load_silero_weights(tmp_path / "silero.npz")
    write_named_initializers_onnx(tmp_path / "silero.onnx", w)
    b = SV.load_silero_weights(tmp_path / "silero.onnx")
    for k in w:
        assert np.array_equal(a[k], w[k]) and np.array_equal(b[k], w[k]), k
    bad = dict(w); bad["decoder.rnn.weight_hh"] = bad["decoder.rnn.weight_hh"][:, :64]
    with pytest.raises(ValueError, match="weight_hh"):
        SV.validate_silero_weights(bad)
    bad = dict(w); del bad["encoder.2.reparam_conv.bias"]
    with pytest.raises(ValueError, match="encoder.2.reparam_conv.bias"):
        SV.validate_silero_weights(bad)
    with pytest.raises(ValueError):
        SV.load_silero_weights(tmp_path / "silero.bin")
    x = signals.c1_sine_silence(12.0, seed=2)
    st = OS.detect_speech_timestamps(x, SR, w)
    assert len(st) >= 2 and all(0 <= s["start"] < s["end"] <= len(x) for s in st)
    assert all(st[i]["end"] <= st[i + 1]["start"] for i in range(len(st) - 1))
    # no weights configured -> the explicit no-weights mode, with weights -> the network
    from audio_cut_amd import config as C
    saved = C.snapshot()
    try:
        C.reset_runtime_config()
        assert SV.configured_weights_path() is None
        C.set_runtime_config({"advanced_vad.silero_weights_path": str(tmp_path / "silero.npz")})
        assert SV.configured_weights_path() == tmp_path / "silero.npz"
        C.set_runtime_config({"advanced_vad.silero_weights_path": str(tmp_path / "missing.npz")})
        with pytest.raises(FileNotFoundError):
            SV.configured_weights_path()
    finally:
        C.restore(saved)


def test_negative_threshold_has_silero_vads_floor():
    """silero_vad.get_speech_timestamps clamps neg_threshold to max(threshold - 0.15, 0.01): with a configured threshold <= 0.15 a
    triggered segment must still be able to close (product hysteresis == oracle hysteresis)."""
    from audio_cut_amd.detectors.silero_chunk_vad import speech_timestamps as product
    from oracle.vad import speech_timestamps as oracle
    probs = np.concatenate([np.full(40, 0.9), np.full(60, 0.005), np.full(30, 0.9), np.full(50, 0.02)]).astype(np.float32)
    for thr in (0.1, 0.15, 0.35):
        got = product(probs, 512 * len(probs), 512, 16000, thr, 250.0, 700.0, 150.0)
        assert got == oracle(probs, 512 * len(probs), 512, 16000, thr, 250.0, 700.0, 150.0)
    low = product(probs, 512 * len(probs), 512, 16000, 0.1, 250.0, 700.0, 0.0)
    assert len(low) == 2 and low[0]["end"] == 40 * 512        # closed at the first window under 0.01, not never


@pytest.mark.gpu
def test_silero_network_and_timestamps_against_oracle(hip_ctx):
    """Window probabilities within 1e-5 of the torch-CPU oracle (float32 both sides; the 16 kHz input differs by the float32
    rounding of the resampling filter), speech timestamps exact, for three kinds of chunk incl. a ragged length and one shorter
    than a window; the batched path (all chunks in one set of launches) equals chunk-by-chunk calls bit for bit."""
    from audio_cut_amd.detectors.silero_vad import SileroHipVad
    from oracle import silero as OS
    w = synth_silero_weights(0)
    vad = SileroHipVad(SR, w, hip_ctx)
    chunks = [signals.c1_sine_silence(10.0, seed=2), signals.voice_with_rests(10.0, seed=3), signals.vocal_like(7.3, seed=5)[:-123],
              np.zeros(4000, np.float32), signals.c1_sine_silence(0.005, seed=1)[:200]]
    packed = hip_ctx.to_device(np.concatenate(chunks))
    offs = np.concatenate(([0], np.cumsum([len(c) for c in chunks])))[:-1]
    pre = vad.precompute(packed, offs.tolist(), [len(c) for c in chunks])
    n_speech = 0
    for c, p in zip(chunks, pre):
        a16 = OS.resample_to_16k(c, SR)
        assert p.n16 == len(a16) and p.n16_padded == len(a16) + (-len(a16)) % 4096
        ref = OS.silero_probs(w, np.pad(a16, (0, p.n16_padded - len(a16))))
        assert p.probs.shape == ref.shape
        assert float(np.max(np.abs(p.probs - ref))) < 1e-5, float(np.max(np.abs(p.probs - ref)))
        got, want = vad(p), OS.detect_speech_timestamps(c, SR, w)
        assert got == want, (got, want)
        n_speech += len(want)
        single = vad(c)                                   # the VadFn contract: one host chunk
        assert single == want
        assert np.array_equal(vad.precompute(hip_ctx.to_device(c), [0], [len(c)])[0].probs, p.probs)
    assert n_speech >= 5                                  # the hysteresis really produced segments
    # the reference's adaptive branch (vocal_pause_detector.py:198-206): threshold, minimum pause and padding per call
    vad.set_adaptive_params(vad_threshold=0.5, min_pause_duration=0.3, speech_pad_ms=60)
    for c, p in zip(chunks[:3], pre[:3]):
        want = OS.detect_speech_timestamps(c, SR, w, adaptive=(0.5, 0.3, 60))
        assert vad(p) == want
        assert want != OS.detect_speech_timestamps(c, SR, w) or not want
    # ... seen by the calling thread only (one SileroHipVad serves every worker of a TrackPipeline), or passed for one call
    import threading
    seen = []
    th = threading.Thread(target=lambda: seen.append(vad(pre[0]))); th.start(); th.join()
    assert seen[0] == OS.detect_speech_timestamps(chunks[0], SR, w)
    vad.set_adaptive_params()
    assert vad(pre[0]) == OS.detect_speech_timestamps(chunks[0], SR, w)
    assert vad(pre[1], adaptive=vad.adaptive_dict(0.5, 0.3, 60)) == OS.detect_speech_timestamps(chunks[1], SR, w, adaptive=(0.5, 0.3, 60))


@pytest.mark.gpu
def test_separator_with_silero_vad_end_to_end(hip_ctx, tmp_path):
    """The whole path with the Silero network as the chunked VAD (weights file configured -> `default_vad` picks
    `SileroHipVad`): VAD segments, focus windows' effect on the pauses, and the integer boundaries equal the oracle's run with
    `oracle.silero.silero_vad_fn` as its `inference_fn`."""
    from audio_cut_amd import config as C
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.detectors.silero_vad import SileroHipVad
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    from oracle import e2e as OE, refine as OR, silero as OS
    sw = synth_silero_weights(0, calib="bursts")
    np.savez(tmp_path / "silero.npz", **sw)
    mix = signals.c1_sine_silence(27.0, seed=3)          # bursts and exact silences: the separated stem keeps the pattern, the VAD segments it
    weights = synth_weights(TfcTdfSpec(), seed=0)
    backend = MDX23HipBackend(weights=weights, ctx=hip_ctx)
    backend.load_model()
    saved = C.snapshot()
    try:
        C.set_runtime_config({"advanced_vad.silero_weights_path": str(tmp_path / "silero.npz")})
        sep = EnhancedVocalSeparator(SR, backend=backend)
        res = SeamlessSplitter(SR, separator=sep).split_track(mix)
        assert isinstance(sep._vad_inference_fn, SileroHipVad)
    finally:
        C.restore(saved)
    OR.LEGACY_PROMOTION = True
    ref = OE.run_track(mix, SR, weights, vad_fn=OS.silero_vad_fn(SR, sw))
    assert res["vad_segments"] == ref.vad_segments and len(ref.vad_segments) >= 4
    assert [p.cut_point for p in res["pauses"]] == [p.cut_point for p in ref.pauses]
    assert res["sample_boundaries"] == ref.sample_boundaries
    assert res["cuts_samples"] == ref.policy.cuts
