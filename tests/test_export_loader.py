"""the resampling loader and the export / manifest layer.
CPU: WAV writer bytes, PCM arithmetic, export naming.  GPU: ac_resample_poly against scipy.signal.resample_poly,
ac_pack_pcm24 against the host arithmetic, and `separate_and_segment` end to end on a 48 kHz stereo WAV."""
import json
import wave

import numpy as np
import pytest

from audio_cut_amd.utils import audio_export as AE

SR = 44100


def test_pcm24_arithmetic_and_wav_round_trip(tmp_path):
    # known answers hand-derived from libsndfile pcm.c f2let_clip_array (what soundfile.write runs: python-soundfile enables
    # SFC_SET_CLIPPING): s = x * 2^31 (float32); s >= 0x7FFFFFFF -> 0x7FFFFF; s <= -2^31 -> 0x800000; else lrintf(s) >> 8
    q = 2.0 ** -23
    x = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 1.5, -1.5, 1e-7, 0.9 * q, -0.1 * q, 1.0 - 2.0 ** -24, 3.0 * q, -3.0 * q, 2.5 * q,
                  -(1.0 - q)], dtype=np.float32)
    b, width = AE.pcm_bytes_host(x, "PCM_24")
    assert width == 3 and b.size == 3 * x.size
    v = b.reshape(-1, 3).astype(np.int32)
    ints = v[:, 0] | (v[:, 1] << 8) | (v[:, 2] << 16)
    ints = np.where(ints & 0x800000, ints - 0x1000000, ints)
    #        0   1.0      -1.0      0.5      -0.5      1.5      -1.5     1e-7 .9q -.1q  1-2^-24  3q  -3q 2.5q -(1-q)
    assert ints.tolist() == [0, 8388607, -8388608, 4194304, -4194304, 8388607, -8388608, 0, 0, -1, 8388607, 3, -3, 2, -8388607]
    b16, w16 = AE.pcm_bytes_host(np.array([0.0, 1.0, -1.0, 0.5, 0.9 * 2.0 ** -15, -0.1 * 2.0 ** -15, 1.5], np.float32), "PCM_16")
    assert w16 == 2 and b16.view("<i2").tolist() == [0, 32767, -32768, 16384, 0, -1, 32767]      # lrintf(x * 2^31) >> 16
    p = AE.export_audio(x, SR, tmp_path / "seg_1.5", "wav")
    assert p.name == "seg_1.5.wav"                                           # a dotted base keeps its dot (`audio_export.py:83-86`)
    with wave.open(str(p), "rb") as w:
        assert (w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()) == (SR, 1, 3, x.size)
        assert w.readframes(x.size) == b.tobytes()
    from audio_cut_amd.api import load_audio_mono
    back, sr = load_audio_mono(str(p))
    assert sr == SR and np.allclose(back[:5], np.clip(x[:5], -1, 1), atol=2.0 / 8388607)
    with pytest.raises(ValueError):
        AE.ensure_supported_format("mp3")
    assert AE.build_export_options("wav", {"subtype": "PCM_16"})["subtype"] == "PCM_16"


def test_segment_exporter_names(tmp_path):
    x = (np.sin(np.arange(SR) * 0.05) * 0.3).astype(np.float32)
    tr = AE.PackedTrack(x, SR)
    ex = AE.SegmentExporter(SR)
    files = ex.export_spans(tr, [(0, 22050), (22050, 44100)], str(tmp_path), segment_is_vocal=[True, False], duration_map={0: 0.5, 1: 0.5})
    assert [f.split("/")[-1] for f in files] == ["segment_001_human_0.5.wav", "segment_002_music_0.5.wav"]
    files = ex.export_spans(tr, [(0, 44100)], str(tmp_path), segment_is_vocal=[True], subdir="segments_vocal", file_suffix="_vocal", duration_map={0: 1.0})
    assert files[0].endswith("segments_vocal/segment_001_human_vocal_1.0.wav")
    with wave.open(files[0], "rb") as w:
        assert w.getnframes() == 44100


@pytest.mark.gpu
def test_resample_poly_kernel_vs_oracle(hip_ctx):
    """ac_resample_poly with the product's soxr-HQ-specification filter against the oracle's independently designed float64 filter
    applied by scipy.signal.resample_poly (oracle/resample.py)."""
    from oracle import resample as ORS
    rng = np.random.default_rng(0)
    t = np.arange(48000 * 2) / 48000.0
    x = (0.4 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 9000 * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    for up, down in ((44100, 48000), (16000, 44100), (2, 1), (3, 7)):
        ref = ORS.resample(x, up, down)
        got = hip_ctx.resample_poly(hip_ctx.to_device(x), up, down).cpu().numpy()
        assert got.shape == ref.shape
        assert float(np.max(np.abs(got - ref))) < 2e-6 * max(1.0, float(np.max(np.abs(ref)))), (up, down)
    assert np.array_equal(hip_ctx.resample_poly(hip_ctx.to_device(x[:100]), 5, 5).cpu().numpy(), x[:100])


@pytest.mark.gpu
def test_pack_pcm24_kernel_vs_host(hip_ctx):
    rng = np.random.default_rng(1)
    for n in (1, 3, 4, 4099):
        x = np.clip(rng.standard_normal(n) * 0.6, -1.3, 1.3).astype(np.float32)
        assert np.array_equal(hip_ctx.pack_pcm24(hip_ctx.to_device(x)), AE.pcm_bytes_host(x, "PCM_24")[0]), n


@pytest.mark.gpu
def test_separate_and_segment_end_to_end_from_48k_stereo_wav(hip_ctx, tmp_path):
    """A 48 kHz stereo PCM_16 WAV through load -> device resample -> separate -> detect -> guard -> boundary policy ->
    export -> SegmentManifest; the exported audio tiles the track and the manifest agrees with the files."""
    import scipy.signal
    from audio_cut_amd import api
    from audio_cut_amd.testing import signals
    st = signals.c2_song(14.0, seed=9, stereo=True)                            # 44.1 kHz generator ...
    st48 = np.stack([scipy.signal.resample_poly(ch, 160, 147) for ch in st]).astype(np.float32)   # ... presented as a 48 kHz file
    src = tmp_path / "song48.wav"
    pcm = np.clip(np.rint(st48.T * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(str(src), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(48000); w.writeframes(pcm.tobytes())
    out_dir = tmp_path / "out"
    ret = api.separate_and_segment(input_uri=str(src), export_dir=str(out_dir), export_manifest=True)
    res = api.last_result()
    n = int(round(st48.shape[1] * 147 / 160 + 0.49999))
    cuts = res["cut_points_samples"]
    assert cuts[0] == 0 and abs(cuts[-1] - n) <= 1 and cuts == sorted(set(cuts))
    assert res["num_segments"] == len(res["mix_segment_files"]) == len(res["vocal_segment_files"]) == len(res["segment_labels"])
    total = 0
    for f, d in zip(res["mix_segment_files"], res["segment_durations"]):
        with wave.open(f, "rb") as w:
            assert (w.getframerate(), w.getsampwidth(), w.getnchannels()) == (SR, 3, 1)
            total += w.getnframes()
            assert abs(w.getnframes() / SR - d) < 1e-9
    assert total == cuts[-1]                                                  # the mix segments tile the whole track
    # the call returns the manifest (reference `api.py:115-131`); the file on disk is the same document
    man = json.loads((out_dir / "SegmentManifest.json").read_text())
    assert ret["manifest_path"] == (out_dir / "SegmentManifest.json").resolve().as_posix()
    assert man["version"] == ret["version"] == "v2.2_mdd" and man["audio"]["sr"] == SR and man["audio"]["hash"].startswith("sha256:")
    assert man["cuts"]["samples"] == cuts and len(man["segments"]) == res["num_segments"] == man["stats"]["num_segments"]
    assert set(man["artifacts"]) == {"music_segments", "human_segments", "vocal_full", "instrumental_full", "all", "output_dir"}
    assert man["segments"][0]["mix_path"].startswith("segment_001_") and "gpu" in man
    assert man["qa_report"]["segments_count"] == res["num_segments"] and man["guard"]["precision_ok"] in (True, False)
    assert man["guard"]["shift_stats"]["count"] == len(man["guard"]["adjustments"]) and man["guard"]["threshold_ms"] == {"avg": 150.0, "p95": 220.0}
    assert man["audio"]["duration"] == cuts[-1] / SR and isinstance(man["timings_ms"]["total"], int)
    # the full vocal file holds the separator's stem, quantised
    voc, sr2 = api.load_audio_mono(res["full_vocal_file"])
    assert sr2 == SR and len(voc) == cuts[-1]
    # VPBD through the same entry: the manifest carries the pool's blocks and annotated cuts, and survives JSON
    out2 = tmp_path / "out_vpbd"
    ret2 = api.separate_and_segment(input_uri=str(src), export_dir=str(out2), mode="vpbd_acoustic", export_types=["mix_segments"],
                                    export_manifest=True, manifest_filename="m.json")
    man2 = json.loads((out2 / "m.json").read_text())
    assert man2["version"] == "vpbd_acoustic" and man2["export_plan"] == ["mix_segments"] and set(man2["artifacts"]) == {"music_segments", "all", "output_dir"}
    assert man2["boundary_detection"]["actual_mode"] == "vpbd_acoustic" and man2["lyrics_alignment"]["fallback_reason"] is None
    planner = ret2["boundary_detection"]["planner"]
    assert "guard_shift_ms_by_raw_time" in planner and "final_time_by_raw_time" in planner
    tagged = [c for c in man2["cuts"]["final"] if isinstance(c, dict) and "source" in c]
    assert len(tagged) == sum(1 for c in ret2["cuts"]["final"] if isinstance(c, dict) and "source" in c)
    assert man2["cuts"]["samples"][0] == 0 and man2["cuts"]["samples"][-1] == cuts[-1]
    assert man2["qa_report"]["fallback_reason"] is None and man2["qa_report"]["segments_count"] == len(man2["segments"])
