"""End-to-end parity of the HIP path against the oracle and the committed goldens (GPU box only).

Bar (BASELINE.json north_star): sample-index cut points bit-exact; stems and RMS / feature series
within 1e-4 relative.
"""
import numpy as np
import pytest

from audio_cut_amd.testing import signals
from oracle import chunking as OC, detector as OD, e2e as OE, features as OF, refine as OR, vad as OV

pytestmark = pytest.mark.gpu
SR = 44100
STEM_RTOL = 1e-4          # relative to the stem's peak, as the north star states ("within 1e-4 relative")
SERIES_RTOL = 1e-4


def _pauses(ps):
    return np.array([[p.start_time, p.end_time, p.confidence, p.cut_point] for p in ps], dtype=np.float64).reshape(-1, 4)


def _assert_pauses_equal(got, ref):
    g, r = _pauses(got), _pauses(ref)
    assert g.shape == r.shape, (g, r)
    assert np.array_equal(g[:, :2], r[:, :2])                  # frame-quantised start/end: exact
    np.testing.assert_allclose(g[:, 2], r[:, 2], rtol=1e-5)    # confidence is a float score
    assert np.array_equal(g[:, 3], r[:, 3])                    # cut_point = integer sample / sr: exact


@pytest.fixture(scope="module")
def splitter(hip_ctx):
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter

    class _NoSeparator:
        _primary_backend = None

    sp = SeamlessSplitter.__new__(SeamlessSplitter)
    sp.sample_rate = SR
    sp.separator = None
    sp._hip = hip_ctx
    from audio_cut_amd.detectors.pure_vocal_pause_detector import PureVocalPauseDetector
    sp.pure_vocal_detector = PureVocalPauseDetector(SR, ctx=hip_ctx)
    sp._last_guard_adjustments_raw = []
    sp._last_suppressed_cut_points = []
    return sp


def test_c1_detector_only_cut_points_bit_exact(hip_ctx, splitter, golden_dir):
    """BASELINE config C1: 60 s mono sine+silence, PureVocalPauseDetector only (no separator, no cache, no VAD)."""
    x = signals.c1_sine_silence(60.0, seed=1)
    got = splitter.pure_vocal_detector.detect_pure_vocal_pauses(x, enable_mdd_enhancement=True, original_audio=x)
    ref = OD.detect_pure_vocal_pauses(x, SR, enable_mdd_enhancement=True, original_audio=x)
    assert len(ref) >= 10
    _assert_pauses_equal(got, ref)
    cands = [(float(p.cut_point), float(p.confidence)) for p in got]
    OR.LEGACY_PROMOTION = True
    ref_b = OE.finalize_and_filter_cuts([(float(p.cut_point), float(p.confidence)) for p in ref], x, x, SR).sample_boundaries
    got_b = splitter._finalize_and_filter_cuts_v2(cands, x, pure_vocal_audio=x).sample_boundaries
    assert got_b == ref_b and len(got_b) >= 8
    # the 30 s prefix case is also pinned by a golden produced with the reference's own detector code
    g = np.load(golden_dir / "features_detector.npz")
    x30 = signals.c1_sine_silence(30.0, seed=1)
    got30 = splitter.pure_vocal_detector.detect_pure_vocal_pauses(x30, enable_mdd_enhancement=True, original_audio=x30)
    gp = _pauses(got30)
    assert np.array_equal(gp[:, [0, 1, 3]], g["c1_pauses"][:, [0, 1, 3]])
    np.testing.assert_allclose(gp[:, 2], g["c1_pauses"][:, 2], rtol=1e-5)


def test_feature_cache_series_and_detector_with_cache(hip_ctx, splitter, golden_dir):
    from audio_cut_amd.analysis.features_cache import ChunkFeatureBuilder
    from audio_cut_amd.utils.gpu_pipeline import chunk_schedule
    mix = signals.c2_song(27.0, seed=21)
    voc = signals.vocal_like(27.0, seed=21)
    b = ChunkFeatureBuilder(SR, ctx=hip_ctx)
    ob = OF.ChunkFeatureOracle(SR)
    for p, op in zip(chunk_schedule(27.0), OC.chunk_plan(27.0)):
        a_, b_ = int(round(p.start_s * SR)), min(len(mix), int(round(p.end_s * SR)))
        b.add_chunk(p, mix[a_:b_], SR)
        ob.add_chunk(op, mix[a_:b_], SR)
    cache = b.finalize(mix)
    ocache = ob.finalize(mix)
    for name in ("rms_series", "spectral_flatness", "onset_envelope", "mdd_series"):
        np.testing.assert_allclose(getattr(cache, name), getattr(ocache, name), rtol=SERIES_RTOL, atol=2e-5, err_msg=name)
    assert np.array_equal(cache.onset_frames, ocache.onset_frames)
    assert np.array_equal(cache.beat_times, ocache.beat_times)
    assert np.array_equal(cache.tempo_curve, ocache.tempo_curve)
    assert float(cache.bpm_features.main_bpm) == float(ocache.bpm_features.main_bpm)
    assert np.array_equal(cache.bpm_features.beat_positions, ocache.bpm_features.beat_positions)
    np.testing.assert_allclose(cache.bpm_features.tempo_variance, ocache.bpm_features.tempo_variance, atol=1e-12)
    np.testing.assert_allclose(cache.global_mdd, ocache.global_mdd, rtol=1e-5)
    # ... and against the golden written by the reference's own ChunkFeatureBuilder control logic
    g = np.load(golden_dir / "features_detector.npz")
    np.testing.assert_allclose(cache.rms_series, g["cache_rms"], rtol=SERIES_RTOL, atol=1e-7)
    np.testing.assert_allclose(cache.spectral_flatness, g["cache_flat"], rtol=SERIES_RTOL, atol=1e-9)
    assert np.array_equal(cache.onset_frames, g["cache_onset_frames"])
    assert np.array_equal(cache.beat_times, g["cache_beat_times"])

    vad = [{"start": 1.0, "end": 6.2, "duration": 5.2}, {"start": 7.1, "end": 13.0, "duration": 5.9},
           {"start": 13.6, "end": 20.5, "duration": 6.9}, {"start": 21.4, "end": 26.5, "duration": 5.1}]
    for vad_segments, key in ((vad, "c2_pauses_vad"), ([], "c2_pauses_novad")):
        got = splitter.pure_vocal_detector.detect_pure_vocal_pauses(voc, enable_mdd_enhancement=True, original_audio=mix,
                                                                    feature_cache=cache, vad_segments=vad_segments)
        ref = OD.detect_pure_vocal_pauses(voc, SR, enable_mdd_enhancement=True, original_audio=mix, feature_cache=ocache,
                                          vad_segments=vad_segments)
        _assert_pauses_equal(got, ref)
        gp = _pauses(got)
        assert np.array_equal(gp[:, [0, 1, 3]], g[key][:, [0, 1, 3]])
    # finalize: markers + pure-music spans + guard, integer boundaries
    from audio_cut_amd.core.enhanced_vocal_separator import compute_vocal_presence_markers
    voc_dev = hip_ctx.to_device(voc)
    markers = compute_vocal_presence_markers(hip_ctx, voc_dev, SR)
    omarkers = OD.vocal_presence_markers(voc, SR)
    assert markers["vocal_presence_cut_points_sec"] == omarkers["vocal_presence_cut_points_sec"]
    assert markers["vocal_presence_segments"] == omarkers["vocal_presence_segments"]
    quiet = voc.copy(); quiet[int(8 * SR): int(17 * SR)] *= 1e-3
    assert splitter._find_no_vocal_runs(quiet, 6.0) == OD.no_vocal_runs(quiet, SR, 6.0)
    assert np.array_equal(np.array(splitter._find_no_vocal_runs(quiet, 6.0)), g["no_vocal_runs"])
    cands = [(float(p.cut_point), float(p.confidence)) for p in got] + [(float(t), 1.0) for t in markers["vocal_presence_cut_points_sec"] if 0 < t < 27.0]
    OR.LEGACY_PROMOTION = True
    ref_b = OE.finalize_and_filter_cuts(cands, mix, voc, SR).sample_boundaries
    got_b = splitter._finalize_and_filter_cuts_v2(cands, mix, pure_vocal_audio=voc).sample_boundaries
    assert got_b == ref_b
    assert got_b == g["final_boundaries_legacy"].tolist()


def test_refine_goldens(hip_ctx, golden_dir):
    """finalize_cut_points against fixtures written by the reference's own refine.py (numpy<2 scalar semantics)."""
    from audio_cut_amd.cutting.refine import CutContext, CutPoint, finalize_cut_points
    g = np.load(golden_dir / "refine.npz")
    for case in range(4):
        seed, n_s, holes, floor_db = g[f"c{case}_params"]
        rng = np.random.default_rng(int(seed))
        n = int(SR * n_s); t = np.arange(n) / SR
        env = np.clip(np.sin(2 * np.pi * 0.13 * (int(seed) + 1) * t), 0, None) ** 2
        mix = (rng.standard_normal(n) * 0.1 * env + 0.3 * env * np.sin(np.arange(n) * 0.05)).astype(np.float32)
        voc = (0.7 * mix + rng.standard_normal(n).astype(np.float32) * 0.01 * env).astype(np.float32)
        if holes:
            a = int(n * 0.3); mix[a:a + SR] = 0; voc[a:a + SR] = 0
        pts = np.stack([rng.uniform(0, n_s, 24), rng.uniform(0, 1, 24)], axis=1)
        res = finalize_cut_points(CutContext(sr=SR, mix_wave=mix, vocal_wave=voc, hip=hip_ctx),
                                  [CutPoint(t=float(a), score=float(b)) for a, b in pts], min_gap_s=1.2, max_keep=200,
                                  guard_db=1.5, search_right_ms=450.0, guard_win_ms=80.0, floor_db=float(floor_db))
        assert res.sample_boundaries == g[f"c{case}_boundaries_legacy"].tolist(), case
        np.testing.assert_allclose([a.final_time for a in res.adjustments], g[f"c{case}_final_times_legacy"], rtol=0, atol=1e-12)
        # decimated lookup arrays of the reference's _prepare_quiet_lookup
        db = hip_ctx.moving_meansq_db(hip_ctx.to_device(voc), 3528)
        np.testing.assert_allclose(db.cpu().numpy()[::997], g[f"c{case}_db_dec"], rtol=0, atol=1e-9)
        nq = hip_ctx.next_leq_scan(db, float(floor_db)).cpu().numpy()[::997]
        assert np.mean(nq == g[f"c{case}_nq_dec"]) > 0.999      # an index can differ only where db ties the floor to 1e-9
    # the reference's own known answer (tests/unit/test_cutting_consistency.py:20-46)
    r = finalize_cut_points(CutContext(sr=10, mix_wave=np.zeros(120, np.float32), hip=hip_ctx),
                            [CutPoint(t=4.0, score=0.9), CutPoint(t=8.0, score=0.9)], min_gap_s=1.0,
                            enable_mix_guard=False, enable_vocal_guard=False, zero_cross_win_ms=0.0)
    assert r.sample_boundaries == [0, 40, 80, 120]


def test_full_path_separate_detect_against_oracle(hip_ctx):
    """Separator (full-size TFC-TDF, seeded synthetic weights) + cache + VAD + detector + guard on a 12.3 s song."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    mix = signals.c2_song(12.3, seed=4)
    w = synth_weights(TfcTdfSpec(), seed=0)
    backend = MDX23HipBackend(weights=w, ctx=hip_ctx)
    backend.load_model()
    sp = SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend))
    res = sp.split_track(mix)
    # analysis/prefetch.py queues the track-global kernels ahead of their consumers: every one of them must have been picked up (a
    # parameter derived differently on the two sides would silently repeat the work), and nothing else may sit in the cache
    st = hip_ctx.prefetch_stats()
    print("prefetch:", st)
    assert st["queued"] >= 10 and st["hits"] == st["queued"] and st["unused"] == [], st
    OR.LEGACY_PROMOTION = True
    ref = OE.run_track(mix, SR, w)
    sep = sp.separator.separate_for_detection(mix)
    peak = float(np.max(np.abs(ref.vocal)))
    err_v = float(np.max(np.abs(sep.vocal_track - ref.vocal))) / peak
    err_i = float(np.max(np.abs(sep.instrumental_track - ref.instrumental))) / float(np.max(np.abs(ref.instrumental)))
    print(f"stem parity: vocal {err_v:.3e}, instrumental {err_i:.3e} (relative to peak)")
    assert err_v < STEM_RTOL and err_i < STEM_RTOL
    assert sep.vad_segments == ref.vad_segments
    np.testing.assert_allclose(sep.feature_cache.rms_series, ref.cache.rms_series, rtol=SERIES_RTOL, atol=1e-7)
    assert sep.quality_metrics["vocal_presence_cut_points_sec"] == ref.markers["vocal_presence_cut_points_sec"]
    _assert_pauses_equal(res["pauses"], ref.pauses)
    assert res["sample_boundaries"] == ref.sample_boundaries
    # the manifest-facing cuts after classification / layout / local valley / weak-tail merge
    assert res["cuts_samples"] == ref.policy.cuts
    assert res["segment_vocal_flags"] == ref.policy.flags
    assert [tuple(p) for p in res["segment_spans"]] == ref.policy.pieces
    assert res["segment_layout_applied"] == ref.policy.layout_applied


def test_vpbd_acoustic_mode_against_oracle(hip_ctx):
    """BASELINE config C4: separator + chunked VAD focus windows + VPBD pool / score / plan, then the guard.
    Expected values: the CPU oracle's stems / cache / pauses pushed through the VPBD host logic (itself pinned to the
    reference by tests/golden/vpbd.npz) and the oracle's finalize."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.core.vocal_phrase_boundary_detector import VocalPhraseBoundaryDetector
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    from audio_cut_amd.testing.vpbd_inputs import FixedPauses
    mix = signals.c2_song(24.0, seed=6)
    w = synth_weights(TfcTdfSpec(), seed=0)
    backend = MDX23HipBackend(weights=w, ctx=hip_ctx)
    backend.load_model()
    sp = SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend))
    res = sp.split_track(mix, mode="vpbd_acoustic")
    OR.LEGACY_PROMOTION = True
    ref = OE.run_track(mix, SR, w)
    vp = VocalPhraseBoundaryDetector(SR).detect(mode="vpbd_acoustic", vocal_track=ref.vocal, original_audio=mix,
                                                pure_vocal_detector=FixedPauses(ref.pauses), feature_cache=ref.cache,
                                                vad_segments=ref.vad_segments)
    exp_times = [c.t for c in vp.selected_candidates]
    assert res["vpbd_selected_times"] == exp_times and len(exp_times) >= 2
    assert res["boundary_detection"]["candidate_counts"] == vp.boundary_detection["candidate_counts"]
    cands = [(c.t, c.score) for c in vp.selected_candidates]
    for a, b in OD.no_vocal_runs(ref.vocal, SR, 6.0):
        cands += [(float(a), 1.0), (float(b), 1.0)]
    protected = set()
    for t in ref.markers["vocal_presence_cut_points_sec"]:
        if 0.0 < t < len(mix) / SR:
            cands.append((float(t), 1.0)); protected.add(int(round(t * SR)))
    exp = set(OE.finalize_and_filter_cuts(cands, mix, ref.vocal, SR).sample_boundaries) | {s for s in protected if 0 < s < len(mix)}
    assert res["sample_boundaries"] == sorted(exp)


def test_c5_long_form_round_trip_and_determinism(hip_ctx):
    """BASELINE config C5 size (30 min, 240 chunks, 480 U-Net items).  With an identity network the whole
    chunked STFT -> iSTFT -> stem assembly -> overlap-add machinery is cheap enough for the CPU oracle at full
    size, so it is compared sample by sample; the round trip must return the mix up to the one frequency bin the
    MDX23 front end drops (bin 3072 of 3073); two runs must agree bit for bit."""
    import torch
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.utils.gpu_pipeline import chunk_schedule

    class _Identity(torch.nn.Module):
        def forward_tf(self, x, spec_amax=None):
            return x

    n = 1800 * SR
    rng = np.random.default_rng(7)
    base = signals.c2_song(60.0, seed=8)
    mix = np.tile(base, 30)[:n].copy()
    mix *= (0.6 + 0.4 * np.sin(np.arange(n) * (2 * np.pi / (97.0 * SR)))).astype(np.float32)     # no two minutes alike
    backend = MDX23HipBackend(weights={}, ctx=hip_ctx, max_items_per_forward=32)
    backend._net = _Identity()
    plans = chunk_schedule(n / SR)
    assert len(plans) == 240
    dev = hip_ctx.to_device(mix)
    sep = backend.separate_track(dev, SR, plans)
    assert sep.n_items == 480
    vocal = sep.vocal.cpu().numpy()
    from oracle import separator as OS
    ref_v, ref_i, _ = OS.separate_track(mix, SR, {}, net_fn=lambda spec: spec)
    peak = float(np.max(np.abs(mix)))
    assert float(np.max(np.abs(vocal - ref_v))) / peak < 2e-6
    assert float(np.max(np.abs(sep.instrumental.cpu().numpy() - ref_i))) / peak < 2e-6
    assert float(np.max(np.abs(vocal - mix))) / peak < 2e-3          # only the dropped top bin is missing
    again = backend.separate_track(dev, SR, plans)
    assert torch.equal(again.vocal, sep.vocal) and torch.equal(again.chunk_vocal, sep.chunk_vocal)
    # guard lookup at full size: idempotent and consistent with its own definition on a decimated probe
    db = hip_ctx.moving_meansq_db(dev, 3528)
    nq = hip_ctx.next_leq_scan(db, -30.0)
    dbh = db.cpu().numpy(); nqh = nq.cpu().numpy()
    probe = rng.integers(0, n, 2000)
    hit = nqh[probe] >= 0
    assert np.all(dbh[nqh[probe][hit]] <= -30.0) and np.all(nqh[probe][hit] >= probe[hit])
    for i in probe[hit][:200]:
        assert not np.any(dbh[i:nqh[i]] <= -30.0)


@pytest.mark.parametrize("tag", ["voice", "bursts"])
def test_dormant_multifeature_branch_against_oracle_and_golden(hip_ctx, golden_dir, tag):
    """the multi-feature (pyin / LPC / centroid / ZCR) branch on the GPU against the CPU oracle and the
    reference-generated fixture: voiced probabilities, pitch states and candidate frames exact; float series within
    tolerance (stated per series); pauses and their integer cut points exact."""
    from audio_cut_amd import config as PCFG
    from audio_cut_amd.detectors.pure_vocal_pause_detector import PureVocalPauseDetector
    from oracle import config as OCFG
    g = np.load(golden_dir / "dormant_branch.npz")
    x = {"voice": signals.voice_with_rests(14.0, seed=3), "bursts": signals.c1_sine_silence(12.0, seed=2)}[tag]
    key = "pure_vocal_detection.enable_relative_energy_mode"
    saved = PCFG.snapshot()
    PCFG.set_runtime_config({key: False}); OCFG.set_runtime_config({key: False})
    try:
        det = PureVocalPauseDetector(SR, ctx=hip_ctx)
        ft = det._extract_vocal_features(hip_ctx.to_device(x))
        # pyin: the Viterbi path decides bins, so f0 is either the same bin or a visible miss
        assert np.array_equal(np.isnan(ft.f0_contour), np.isnan(g[f"{tag}_f0"]))
        v = ~np.isnan(ft.f0_contour)
        assert np.allclose(ft.f0_contour[v], g[f"{tag}_f0"][v], rtol=1e-12, atol=0)
        assert np.allclose(ft.f0_confidence, g[f"{tag}_voiced_prob"], rtol=1e-9, atol=1e-12)
        assert np.allclose(ft.rms_energy, g[f"{tag}_rms"], rtol=1e-5, atol=1e-9)
        assert np.allclose(ft.spectral_centroid, g[f"{tag}_centroid"], rtol=1e-4, atol=1e-2)       # Hz; silence frames are 0/0-guarded
        assert np.allclose(ft.harmonic_ratio, g[f"{tag}_harmonic_ratio"], rtol=1e-4, atol=1e-6)
        assert np.array_equal(ft.zero_crossing_rate, g[f"{tag}_zcr"])
        if tag == "voice":
            for j in range(3):
                ref = g[f"{tag}_formant{j}"]
                assert len(ft.formant_energies[j]) == len(ref)            # same peak counts frame by frame
                # |1/A| near a pole amplifies the float32 Burg recursion's summation-order rounding (numpy pairwise vs tree)
                assert np.allclose(ft.formant_energies[j], ref, rtol=5e-3, atol=1e-6)
        else:
            # pure 440 Hz bursts put the LPC poles ON the unit circle: |1/A| ~ 1e5-1e6 and chaotic in the last float32 bit, in
            # librosa as much as here, so only the structure is compared (first track has one entry per frame)
            assert len(ft.formant_energies[0]) == len(g[f"{tag}_formant0"])
        cand = np.asarray(det._detect_candidate_pauses(ft), dtype=np.int64).reshape(-1, 2)
        assert np.array_equal(cand, g[f"{tag}_candidates"])
        for breath in (0, 1):
            ps = det.detect_pure_vocal_pauses(x, include_breath_candidates=bool(breath))
            ref = g[f"{tag}_pauses_breath{breath}"]
            assert len(ps) == len(ref)
            got = np.array([[p.start_time, p.end_time, p.confidence, p.cut_point] for p in ps], dtype=np.float64).reshape(-1, 4)
            assert np.array_equal(got[:, :2], ref[:, :2])                                       # run boundaries (frames)
            assert np.array_equal((got[:, 3] * SR).astype(np.int64), (ref[:, 3] * SR).astype(np.int64))   # integer cut samples
            assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-3, atol=1e-3)                      # confidence (a float score)
    finally:
        PCFG.restore(saved); OCFG.reset_runtime_config()


def test_degenerate_tracks(hip_ctx, tmp_path):
    """Empty / sub-chunk / silent / constant inputs: the single-segment exits of the reference (`seamless_splitter.py:421-433`,
    `_create_single_segment_result`) and a refusal for an empty track, never a device fault."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip_ctx)
    backend.load_model()
    sp = SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend))
    cases = {"silence": np.zeros(12 * SR, np.float32), "one_second": signals.c2_song(1.0, seed=1),
             "12345_samples": signals.c2_song(0.28, seed=1)[:12345], "dc": np.full(9 * SR, 0.3, np.float32)}
    for name, x in cases.items():
        for mode in ("v2.2_mdd", "vpbd_acoustic"):
            r = sp.split_track(x, mode=mode)
            assert r["sample_boundaries"] == [0, len(x)] and r["cuts_samples"] == [0, len(x)], (name, mode)
            assert r["note"] in ("no_pause_candidates", "no_vpbd_candidates") and r["single_segment"] is True
            assert r["segment_vocal_flags"] == [False] if name in ("silence",) else len(r["segment_vocal_flags"]) == 1
    for x in (signals.c2_song(10.0, seed=2), signals.c2_song(23.7, seed=3)[:-7]):     # exactly one chunk; ragged multi-chunk
        r = sp.split_track(x)
        c = r["cuts_samples"]
        assert c[0] == 0 and c[-1] == len(x) and c == sorted(set(c)) and len(r["segment_vocal_flags"]) == len(r["segment_spans"])
        assert sum(hi - lo for lo, hi in r["segment_spans"]) == len(x)
    with pytest.raises(ValueError):
        sp.split_track(np.zeros(0, np.float32))


@pytest.mark.parametrize("fixture,seconds,seed", [("c2_full_oracle", 240.0, 2), ("c2_150s_seed11_oracle", 150.0, 11)])
def test_c2_full_size_track_against_oracle_fixture(hip_ctx, golden_dir, fixture, seconds, seed):
    """BASELINE configs[1] at full size (4-min C2 song, and a second 150 s song of the same generator): the GPU path against
    the CPU oracle's committed result (tests/golden/make_c2_full.py; the oracle needs minutes per run) - every guard
    boundary, every manifest cut, the segment labels, pause cut points, VAD segments and beat grid exact; stems and RMS
    series within 1e-4."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    g = np.load(golden_dir / f"{fixture}.npz")
    mix = signals.c2_song(seconds, seed=seed)
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip_ctx, max_items_per_forward=32)
    backend.load_model()
    sp = SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend))
    res = sp.split_track(mix)
    assert res["sample_boundaries"] == g["sample_boundaries"].tolist()
    assert res["cuts_samples"] == g["cuts"].tolist()
    assert [int(f) for f in res["segment_vocal_flags"]] == g["flags"].tolist()
    assert [list(p) for p in res["segment_spans"]] == g["pieces"].tolist()
    assert np.array_equal(np.asarray([p.cut_point for p in res["pauses"]]), g["pause_cut_points"])
    assert np.array_equal(np.asarray([[s["start"], s["end"]] for s in res["vad_segments"]]), g["vad_segments"])
    assert np.array_equal(np.asarray(res["feature_cache"].beat_times), g["beat_times"])
    np.testing.assert_allclose(res["feature_cache"].rms_series, g["cache_rms"], rtol=SERIES_RTOL, atol=1e-7)
    voc = res["vocal_track"]
    peak = float(g["vocal_peak"])
    assert float(np.max(np.abs(voc[: 4 * SR: 7] - g["vocal_head"]))) / peak < STEM_RTOL
    nsec = len(mix) // SR
    rms = np.sqrt(np.mean(voc[: nsec * SR].reshape(nsec, SR).astype(np.float64) ** 2, axis=1))
    np.testing.assert_allclose(rms, g["vocal_rms_per_second"], rtol=1e-4, atol=1e-4 * peak)


def test_soak_track_on_a_threshold_crossing_is_exact(hip_ctx, golden_dir):
    """The one round-1 live-soak track (of 21) that was NOT index-identical to the oracle: a sine-burst / silence track whose guard
    boundary #8 is the quiet guard's "first sample under the floor" on a slow decay into a silent stretch.  Round 1's split-f16
    kernels carried an absolute error floor of 2^-25 per activation (the float16 low part went subnormal below 6e-2), so the
    quiet tail of the decay lost its relative precision and the crossing moved by two samples (1947295 vs 1947297).  With the
    per-item activation scale (include/audiocut_hip.h "amax") the stems are float32-class in the quiet region too and every
    index is exact."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    g = np.load(golden_dir / "c1_170s_seed64_w17_oracle.npz")
    mix = signals.c1_sine_silence(170.0, seed=64)
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=17), ctx=hip_ctx, max_items_per_forward=32)
    backend.load_model()
    res = SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend)).split_track(mix)
    voc = res["vocal_track"]
    stem_err = float(np.max(np.abs(voc[: 4 * SR: 7] - g["vocal_head"]))) / float(g["vocal_peak"])
    got, want = res["sample_boundaries"], g["sample_boundaries"].tolist()
    print(f"soak track: stem error {stem_err:.2e} of peak, moved boundaries {[(a, b) for a, b in zip(got, want) if a != b]}")
    assert got == want
    assert res["cuts_samples"] == g["cuts"].tolist()
    assert [int(f) for f in res["segment_vocal_flags"]] == g["flags"].tolist()
    assert np.array_equal(np.asarray([p.cut_point for p in res["pauses"]]), g["pause_cut_points"])
    assert stem_err < 3e-6


@pytest.mark.parametrize("seconds,seed,weight_seed,name", [(200.0, 79, 29, "c1_200s_seed79_w29_oracle"),
                                                          (120.0, 71, 21, "c1_120s_seed71_w21_oracle")])
def test_soak_tracks_decided_in_digital_silence_are_exact(hip_ctx, golden_dir, seconds, seed, weight_seed, name):
    """The two round-2 live-soak tracks that a per-8-row activation scale left open: guard boundaries inside the
    digital silence between sine bursts, where the oracle's own dB series is one flat plateau and np.argmin returns the first
    sample at which the U-Net's leakage - 1e-14 of the neighbouring burst, falling by 0.2 % per sample - stops being resolved
    against the 1e-12 epsilon.  One scale per item moved them by up to 84 samples, one per time row by up to 30; with the conv
    kernels' row-exact path (csrc/ac_common.h) every index is the oracle's."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    g = np.load(golden_dir / f"{name}.npz")
    mix = signals.c1_sine_silence(seconds, seed=seed)
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=weight_seed), ctx=hip_ctx, max_items_per_forward=32)
    backend.load_model()
    res = SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend)).split_track(mix)
    got, want = res["sample_boundaries"], g["sample_boundaries"].tolist()
    stem_err = float(np.max(np.abs(res["vocal_track"][: 4 * SR: 7] - g["vocal_head"]))) / float(g["vocal_peak"])
    print(f"{name}: stem error {stem_err:.2e} of peak, moved boundaries {[(a, b) for a, b in zip(got, want) if a != b]}")
    assert got == want
    assert res["cuts_samples"] == g["cuts"].tolist()
    assert [int(f) for f in res["segment_vocal_flags"]] == g["flags"].tolist()
    assert np.array_equal(np.asarray([p.cut_point for p in res["pauses"]]), g["pause_cut_points"])
    assert stem_err < 5e-6


def test_track_pipeline_matches_sequential_processing(hip_ctx):
    """BASELINE config C3 in miniature: six different tracks through `batch.TrackPipeline` (two in flight, own streams,
    shared U-Net weights) give, track by track, exactly what processing them one after the other gives."""
    from audio_cut_amd import batch
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip_ctx, max_items_per_forward=16)
    backend.load_model()
    mk = lambda: SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend))
    tracks = [signals.c2_song(d, seed=100 + i) for i, d in enumerate((24.0, 31.5, 18.2, 27.7, 12.3, 36.0))]
    solo = mk()
    ref = [solo.split_track(t) for t in tracks]
    pipe = batch.TrackPipeline([mk(), mk()], hip_ctx.device)
    got = pipe.run([(lambda sp, t=t: sp.split_track(t, separation_gate=pipe.separation_gate)) for t in tracks])
    # ... and with every worker's separation queued on the pipeline's ONE U-Net stream (the gate only serialises the queueing)
    got_shared = pipe.run([(lambda sp, t=t: sp.split_track(t, separation_gate=pipe.separation_gate, unet_stream=pipe.unet_stream)) for t in tracks])
    for i, (a, b, c) in enumerate(zip(ref, got, got_shared)):
        assert a["sample_boundaries"] == b["sample_boundaries"] == c["sample_boundaries"], i
        assert a["cuts_samples"] == b["cuts_samples"] == c["cuts_samples"], i
        assert a["segment_vocal_flags"] == b["segment_vocal_flags"] == c["segment_vocal_flags"], i
        assert np.array_equal(a["vocal_track"], b["vocal_track"]), i          # same kernels, same order inside a track: bit-identical stems
        assert np.array_equal(a["vocal_track"], c["vocal_track"]), i
    summaries = [batch.summarize(i, r["sample_boundaries"], len(t) / SR) for i, (r, t) in enumerate(zip(got, tracks))]
    assert [s["track"] for s in batch.gather_summaries(summaries)] == list(range(6))
    with pytest.raises(ValueError):                                           # a failing job surfaces, the pipeline does not hang
        pipe.run([lambda sp: sp.split_track(np.zeros(0, np.float32))])


def _oracle_fixture_asserts(res, g, mix):
    """What every full-size oracle fixture (tests/golden/make_c2_full.py) pins: guard boundaries, manifest cuts, labels, pause cut
    points, VAD segments and beat grid exact; stems and RMS series within the north-star tolerance (1e-4)."""
    assert res["sample_boundaries"] == g["sample_boundaries"].tolist()
    assert res["cuts_samples"] == g["cuts"].tolist()
    assert [int(f) for f in res["segment_vocal_flags"]] == g["flags"].tolist()
    assert [list(p) for p in res["segment_spans"]] == g["pieces"].tolist()
    assert np.array_equal(np.asarray([p.cut_point for p in res["pauses"]]), g["pause_cut_points"])
    assert np.array_equal(np.asarray([[s["start"], s["end"]] for s in res["vad_segments"]]), g["vad_segments"])
    assert np.array_equal(np.asarray(res["feature_cache"].beat_times), g["beat_times"])
    np.testing.assert_allclose(res["feature_cache"].rms_series, g["cache_rms"], rtol=SERIES_RTOL, atol=1e-7)
    voc = res["vocal_track"]
    peak = float(g["vocal_peak"])
    head_err = float(np.max(np.abs(voc[: 4 * SR: 7] - g["vocal_head"]))) / peak
    assert head_err < STEM_RTOL
    nsec = len(mix) // SR
    rms = np.sqrt(np.mean(voc[: nsec * SR].reshape(nsec, SR).astype(np.float64) ** 2, axis=1))
    np.testing.assert_allclose(rms, g["vocal_rms_per_second"], rtol=1e-4, atol=1e-4 * peak)
    return head_err


def test_c5_long_form_end_to_end_against_oracle_fixture(hip_ctx, golden_dir):
    """BASELINE configs[4] end to end AFTER the loader (the parity input is defined after resampling): a 30-min
    track (C2 generator looped with per-section seeds, 240 chunks, 480 U-Net items with the real - synthetic-weight - U-Net)
    through separation, feature cache, VAD, pause detection, quiet guard and the boundary policy, against the CPU oracle's
    committed result (`make_c2_full.py 1800 5 c5_1800s_seed5_oracle c5_long_form 0`: 37 min of oracle time): all 201 guard
    boundaries, 241 manifest cuts, labels, 240 pause cut points, VAD segments and 3599 beats exact; stems within 1e-4."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    g = np.load(golden_dir / "c5_1800s_seed5_oracle.npz")
    mix = signals.c5_long_form(1800.0, seed=5)
    assert len(mix) == 1800 * SR
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip_ctx, max_items_per_forward=32)
    backend.load_model()
    sp = SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend))
    res = sp.split_track(mix)
    assert res["gpu_meta"]["gpu_pipeline_chunks"] == 240 and len(g["sample_boundaries"]) == 201
    err = _oracle_fixture_asserts(res, g, mix)
    print(f"C5: 201 boundaries / 241 cuts exact, vocal stem error {err:.2e} of peak")


def test_c5_loader_leg_48k_stereo_at_full_length(hip_ctx):
    """The load leg of BASELINE configs[4] at full size: 30 min of 48 kHz stereo -> channel mean -> ac_resample_poly 147/160 on
    the device (the reference: librosa.load(sr=44100, mono=True), `audio_processor.py:45-49`; its soxr_hq coefficients cannot be pinned
    offline, so this row's parity definition is the oracle's filter to soxr's published HQ specification, oracle/resample.py - see DESIGN.md).  Properties at 86.4 M input samples:
    exact output length, agreement with the oracle on windows spread over the track (the FIR is local), exact homogeneity under a
    power-of-two gain, and section energies preserved (the source has < 0.1 % of its energy above 20 kHz)."""
    from oracle import resample as ORS
    st = signals.c5_long_form(1800.0, seed=5, sr=48000, stereo=True)
    assert st.shape == (2, 1800 * 48000)
    mono = np.mean(st, axis=0).astype(np.float32)
    del st
    dev = hip_ctx.to_device(mono)
    y = hip_ctx.resample_poly(dev, 147, 160)
    n_out = -(-len(mono) * 147 // 160)
    assert y.numel() == n_out == 1800 * SR
    yh = y.cpu().numpy()
    assert np.all(np.isfinite(yh))
    rng = np.random.default_rng(3)
    for c in [0, len(mono) - 480000] + [int(v) for v in rng.integers(10_000_000, len(mono) - 10_000_000, 6)]:
        c -= c % 160                                     # window starts on a polyphase period: output index = c * 147 / 160 exactly
        lo = max(0, c - 160 * 40); hi = min(len(mono), c + 480000 + 160 * 40)
        ref = ORS.resample(mono[lo:hi], 147, 160)
        o0 = c * 147 // 160; r0 = (c - lo) * 147 // 160
        n_cmp = 400000 * 147 // 160
        assert float(np.max(np.abs(yh[o0:o0 + n_cmp] - ref[r0:r0 + n_cmp]))) < 2e-6 * max(1.0, float(np.max(np.abs(ref)))), c
        if hi == len(mono):                              # the track's tail: the same zero padding past the last sample
            assert float(np.max(np.abs(yh[-2000:] - ref[-2000:]))) < 2e-6
    y2 = hip_ctx.resample_poly(dev * 0.5, 147, 160).cpu().numpy()
    assert np.array_equal(y2, yh * np.float32(0.5))      # homogeneity: float64 accumulation of exactly halved terms
    sec_in = mono[: 30 * 60 * 48000].reshape(30, -1).astype(np.float64); sec_out = yh.reshape(30, -1).astype(np.float64)
    e_in = np.mean(sec_in ** 2, axis=1); e_out = np.mean(sec_out ** 2, axis=1)
    assert np.all(np.abs(e_out / e_in - 1.0) < 2e-3)


def test_c3_batch_of_32_tracks_on_one_gpu(hip_ctx, golden_dir):
    """BASELINE configs[2] on one GPU: the 32 C3 tracks (c2_song, seeds 100-131) through `batch.TrackPipeline` exactly as
    `bench.py --config c3` runs them.  Every track's guard boundaries and manifest cuts hash to the committed single-GPU
    result (tests/golden/c3_n1_sha1.json - the table `bench.py --gpus N` checks every rank's tracks against), the LPT deal
    over 8 ranks is the C3 sharding (4 tracks each), and the first and last track are also checked against the CPU oracle
    (tests/golden/c3_seed100_oracle.npz, c3_seed131_oracle.npz)."""
    import json
    from concurrent.futures import ThreadPoolExecutor
    from audio_cut_amd import batch
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    table = json.loads((golden_dir / "c3_n1_sha1.json").read_text())["tracks"]
    seeds = list(range(100, 132))
    assert sorted(int(k) for k in table) == [2] + seeds      # + the C2 bench track (seed 2), which bench.py's N = 1 run starts with
    assert batch.assign_tracks([240.0] * 32, 8) == [[r + 8 * k for k in range(4)] for r in range(8)]
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=0), ctx=hip_ctx, max_items_per_forward=32)
    backend.load_model()
    sps = [SeamlessSplitter(SR, separator=EnhancedVocalSeparator(SR, backend=backend)) for _ in range(2)]
    pipe = batch.TrackPipeline(sps, hip_ctx.device)
    oracle = {100: np.load(golden_dir / "c3_seed100_oracle.npz"), 131: np.load(golden_dir / "c3_seed131_oracle.npz")}
    for lo in range(0, 32, 8):                       # eight tracks resident at a time
        with ThreadPoolExecutor(8) as pool:          # the generator is numpy-bound (releases the GIL): 8 tracks in the time of 1.5
            mixes = list(pool.map(lambda s: signals.c2_song(240.0, seed=s), seeds[lo:lo + 8]))
        devs = [hip_ctx.to_device(m) for m in mixes]
        out = pipe.run([(lambda sp, m=m, d=d: sp.split_track(m, audio_dev=d, separation_gate=pipe.separation_gate, unet_stream=pipe.unet_stream))
                        for m, d in zip(mixes, devs)])       # bench.py's default scheme: one U-Net stream, the next track queued behind the running one
        for s, m, r in zip(seeds[lo:lo + 8], mixes, out):
            sm = batch.summarize(s, r["sample_boundaries"], 240.0)
            assert sm["boundaries_sha1"] == table[str(s)]["boundaries_sha1"], s
            assert batch.summarize(0, r["cuts_samples"], 0.0)["boundaries_sha1"] == table[str(s)]["cuts_sha1"], s
            if s in oracle:
                _oracle_fixture_asserts(r, oracle[s], m)


def test_build_feature_cache_whole_track_variant(hip_ctx):
    """`build_feature_cache` (`features_cache.py:355-398,483-509`, the fallback the orchestrator takes when the
    separator hands back no cache, `seamless_splitter.py:332-343`) on the full 4-min C2 track against the oracle's whole-track
    builder: float series within the series tolerance, every integer / decision-carrying field exact; and against the chunked
    builder's cache of the same track, from which it may differ only where chunk edges truncate an STFT frame."""
    from audio_cut_amd.analysis.features_cache import build_feature_cache
    mix = signals.c2_song(240.0, seed=2)
    dev = hip_ctx.to_device(mix)
    cache = build_feature_cache(mix, None, SR, ctx=hip_ctx, mix_dev=dev)
    ocache = OF.whole_track_cache(mix, SR)
    assert cache.sr == SR and cache.hop_length == 2205 and abs(cache.duration_s - 240.0) < 1e-9
    for name in ("rms_series", "spectral_flatness", "onset_envelope", "mdd_series"):
        a, b = getattr(cache, name), getattr(ocache, name)
        assert len(a) == len(b) == 4801, name
        np.testing.assert_allclose(a, b, rtol=SERIES_RTOL, atol=2e-5, err_msg=name)
    assert np.array_equal(cache.onset_frames, ocache.onset_frames)
    assert np.array_equal(cache.beat_times, ocache.beat_times)
    assert np.array_equal(cache.tempo_curve, ocache.tempo_curve)
    assert float(cache.bpm_features.main_bpm) == float(ocache.bpm_features.main_bpm)
    assert np.array_equal(cache.bpm_features.beat_positions, ocache.bpm_features.beat_positions)
    np.testing.assert_allclose(cache.global_mdd, ocache.global_mdd, rtol=1e-5)
    np.testing.assert_allclose(cache.rms_max, ocache.rms_max, rtol=SERIES_RTOL)
    # accessors the consumers use (`features_cache.py:60-91`)
    assert cache.frame_index(12.34) == ocache.frame_index(12.34) == int(np.clip(round(12.34 / 0.05), 0, 4800))
    # a call without a context or device copy builds its own (no CPU path: it needs the GPU)
    again = build_feature_cache(mix, None, SR)
    assert np.array_equal(again.rms_series, cache.rms_series) and np.array_equal(again.beat_times, cache.beat_times)
    with pytest.raises(ValueError):
        build_feature_cache(np.zeros(0, np.float32), None, SR, ctx=hip_ctx)


def _run_with_silero(hip_ctx, tmp_path, mix, weight_seed, silero_seed, calib, mode="v2.2_mdd", affine=None):
    """The product path with the Silero network as the chunked VAD (synthetic weights file configured).  `affine`: the output
    layer's calibration as a fixture stores it (no oracle call)."""
    from audio_cut_amd import config as C
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.detectors.silero_vad import SileroHipVad
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    from silero_synth import synth_silero_weights
    np.savez(tmp_path / "silero.npz", **synth_silero_weights(silero_seed, calib=calib, affine=affine))
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=weight_seed), ctx=hip_ctx, max_items_per_forward=32)
    backend.load_model()
    saved = C.snapshot()
    try:
        C.set_runtime_config({"advanced_vad.silero_weights_path": str(tmp_path / "silero.npz")})
        sep = EnhancedVocalSeparator(SR, backend=backend)
        res = SeamlessSplitter(SR, separator=sep).split_track(mix, mode=mode)
        assert isinstance(sep._vad_inference_fn, SileroHipVad)
    finally:
        C.restore(saved)
    return res


def test_soak_track_on_the_epsilon_plateau_is_exact_or_plateau_equivalent(hip_ctx, golden_dir, tmp_path):
    """The one round-2 live-soak track (of 33) whose guard boundary was not the oracle's: c1_sine_silence 60 s, song seed 301,
    U-Net weights 61, Silero weights 1 (`profiles/r02_parity_soak_g_silero.log`: 2407370 vs 2407367).  Every interior boundary of
    this track sits in digital silence, on the epsilon plateau of the guard's dB series (tests/guard_plateau.py), where the
    reference's `np.argmin` lands on the sample at which float32 inverse-FFT rounding noise leaves the 80 ms window.
    Asserted: everything upstream exact (VAD segments, pause cut points); every boundary either EQUAL or - only on the plateau -
    at an index where the oracle's own vocal and mix dB series are bit-equal to their values at the oracle's index, with the
    stem around it within 1e-5 of the peak; manifest cuts equal after that mapping.  This is an equivalence class, not a sample
    tolerance: off the plateau nothing may move."""
    from guard_plateau import classify_boundaries, map_cuts
    g = np.load(golden_dir / "c1_60s_seed301_w61_silero1_oracle.npz")
    assert int(g["on_plateau"].sum()) == 14                      # all interior boundaries of this track are plateau decisions
    mix = signals.c1_sine_silence(60.0, seed=301)
    res = _run_with_silero(hip_ctx, tmp_path, mix, weight_seed=61, silero_seed=1, calib="bursts")
    assert np.array_equal(np.asarray([[s["start"], s["end"]] for s in res["vad_segments"]], dtype=np.float64).reshape(-1, 2), g["vad_segments"])
    assert np.array_equal(np.asarray([p.cut_point for p in res["pauses"]]), g["pause_cut_points"])
    peak = float(g["vocal_peak"])
    stem_err = float(np.max(np.abs(res["vocal_track"][: 4 * SR: 7] - g["vocal_head"]))) / peak
    ctx = {k: g[k] for k in ("on_plateau", "db_vocal", "db_mix", "stem_window", "guard_half_window")}
    exact, equiv, fails = classify_boundaries(res["sample_boundaries"], g["sample_boundaries"].tolist(), ctx, res["vocal_track"], stem_atol=1e-5 * peak)
    print(f"seed 301 / w61 / silero 1: stem error {stem_err:.2e} of peak; {len(exact)} boundaries exact, plateau-equivalent (gpu, oracle): {equiv}")
    assert not fails, fails
    assert len(equiv) <= 2, equiv            # round 2 measured one (3 samples); a second one would be news worth reading, not a pass
    assert map_cuts(res["cuts_samples"], equiv) == g["cuts"].tolist()
    assert [int(f) for f in res["segment_vocal_flags"]] == g["flags"].tolist()
    assert stem_err < 5e-6


def test_soak_track_that_exposed_the_beat_dp_tie_is_exact(hip_ctx, golden_dir, tmp_path):
    """The one live-soak track (round 3, 40 tracks) that exposed a HOST-side defect: c1_sine_silence 120 s, song seed 423, U-Net weights 93,
    Silero weights 10.  Every guard boundary was exact and ONE manifest cut was 1 744 samples off (3 455 696 vs 3 457 440): the layout
    refiner's beat snap, fed by a beat grid that left the oracle's at beat 62 - `ac_host_beat_dp` associated its transition weight as
    (-tightness * l) * l where numpy squares first, one ulp, the other way round at a tie of ~1e-100 local scores
    (`profiles/r03_parity_soak_m.log`; the DP alone: tests/test_abi_and_host.py::test_host_beat_dp_breaks_ties_like_numpy).  The track
    itself is the fixture here: beats, boundaries, manifest cuts, labels, VAD segments and pause cut points equal."""
    g = np.load(golden_dir / "c1_120s_seed423_w93_silero10_oracle.npz")
    assert float(g["seconds"]) == 120.0 and int(g["seed"]) == 423 and int(g["weight_seed"]) == 93 and int(g["silero_seed"]) == 10
    mix = signals.c1_sine_silence(120.0, seed=423)
    res = _run_with_silero(hip_ctx, tmp_path, mix, weight_seed=93, silero_seed=10, calib=str(g["silero_calib"]), affine=g["silero_affine"])
    bt = np.asarray(res["feature_cache"].beat_times, dtype=np.float64)
    assert bt.shape == g["beat_times"].shape and np.array_equal(bt, g["beat_times"]), int(np.argmax(bt[: len(g["beat_times"])] != g["beat_times"][: len(bt)]))
    assert len(bt) > 62                                       # the grid really runs past the beat at which the two once parted
    assert np.array_equal(np.asarray([[s["start"], s["end"]] for s in res["vad_segments"]], dtype=np.float64).reshape(-1, 2), g["vad_segments"])
    assert np.array_equal(np.asarray([p.cut_point for p in res["pauses"]]), g["pause_cut_points"])
    assert res["sample_boundaries"] == g["sample_boundaries"].tolist()
    assert res["cuts_samples"] == g["cuts"].tolist()
    assert [int(f) for f in res["segment_vocal_flags"]] == g["flags"].tolist()
    stem_err = float(np.max(np.abs(res["vocal_track"][: 4 * SR: 7] - g["vocal_head"]))) / float(g["vocal_peak"])
    print(f"seed 423 / w93 / silero 10: {len(bt)} beats, {len(res['sample_boundaries'])} boundaries, {len(res['cuts_samples'])} manifest cuts exact; stem error {stem_err:.2e}")
    assert stem_err < 5e-6


def _vpbd_silero_fixture_asserts(hip_ctx, g, min_vad_segments: int) -> str:
    """One track in `vpbd_acoustic` mode with the Silero network (HIP kernels, the fixture's synthetic weights) as the chunked VAD against
    a fixture written by tests/golden/make_track_fixture.py --mode vpbd_acoustic (the CPU oracle's stems / cache / pauses / VAD segments
    pushed through the REFERENCE's own VocalPhraseBoundaryDetector, then the oracle's guard and boundary policy): VAD segments, the
    planner's selected candidate times and the pool counts, guard boundaries, manifest cuts, segment labels and beats exact."""
    from audio_cut_amd.core.enhanced_vocal_separator import EnhancedVocalSeparator
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.detectors.silero_vad import SileroHipVad
    from audio_cut_amd.separation.backends import MDX23HipBackend
    from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, synth_weights
    from silero_synth import synth_silero_weights
    assert str(g["mode"]) == "vpbd_acoustic"
    mix = getattr(signals, str(g["generator"]))(float(g["seconds"]), seed=int(g["seed"]))
    mix = np.mean(mix, axis=0).astype(np.float32) if mix.ndim == 2 else mix
    sw = synth_silero_weights(int(g["silero_seed"]), str(g["silero_calib"]), affine=g["silero_affine"])       # no oracle call: the affine is data
    backend = MDX23HipBackend(weights=synth_weights(TfcTdfSpec(), seed=int(g["weight_seed"])), ctx=hip_ctx, max_items_per_forward=32)
    backend.load_model()
    sep = EnhancedVocalSeparator(SR, backend=backend, vad_inference_fn=SileroHipVad(SR, sw, hip_ctx))
    res = SeamlessSplitter(SR, separator=sep).split_track(mix, mode="vpbd_acoustic")
    vad = np.asarray([[s["start"], s["end"]] for s in res["vad_segments"]], dtype=np.float64).reshape(-1, 2)
    assert np.array_equal(vad, g["vad_segments"]) and len(vad) >= min_vad_segments
    assert res["vpbd_selected_times"] == g["vpbd_selected"][:, 0].tolist()
    counts = res["boundary_detection"]["candidate_counts"]
    assert [counts[k] for k in ("acoustic", "beat", "merged", "total", "selected", "suppressed")] == g["vpbd_counts"].tolist()
    assert res["sample_boundaries"] == g["sample_boundaries"].tolist()
    assert res["cuts_samples"] == g["cuts"].tolist()
    assert [int(f) for f in res["segment_vocal_flags"]] == g["flags"].tolist()
    peak = float(g["vocal_peak"])
    stem_err = float(np.max(np.abs(res["vocal_track"][: 4 * SR: 7] - g["vocal_head"]))) / peak
    np.testing.assert_allclose(res["feature_cache"].rms_series, g["cache_rms"], rtol=SERIES_RTOL, atol=1e-7)
    assert np.array_equal(np.asarray(res["feature_cache"].beat_times, dtype=np.float64), g["beat_times"])
    assert stem_err < 5e-6
    return (f"{len(vad)} VAD segments, {len(res['vpbd_selected_times'])} selected candidates, {len(res['sample_boundaries'])} boundaries, "
            f"{len(res['cuts_samples'])} manifest cuts exact; stem error {stem_err:.2e} of peak")


def test_c4_full_size_vpbd_acoustic_with_silero_vad_against_oracle_fixture(hip_ctx, golden_dir, tmp_path):
    """BASELINE configs[3] / SURVEY 8d C4 at its stated size: the 4-min C2 track (seed 2) in `vpbd_acoustic` mode with the
    Silero network (HIP kernels, seeded synthetic weights calibrated on the stem the VAD sees) as the chunked VAD, against
    `tests/golden/c4_full_oracle.npz` (see `_vpbd_silero_fixture_asserts`)."""
    g = np.load(golden_dir / "c4_full_oracle.npz")
    assert float(g["seconds"]) == 240.0 and int(g["seed"]) == 2 and str(g["generator"]) == "c2_song"
    assert len(g["sample_boundaries"]) >= 10
    print("C4 full size:", _vpbd_silero_fixture_asserts(hip_ctx, g, min_vad_segments=8))      # real focus windows, not one segment spanning the track


@pytest.mark.parametrize("name", ["c4_90s_seed21_w31_silero5_oracle", "c4_80s_seed23_w33_silero7_oracle", "c4_75s_seed22_w32_silero6_oracle"])
def test_vpbd_acoustic_with_silero_vad_on_more_tracks(hip_ctx, golden_dir, name):
    """C4's path on three more tracks / weight seeds / generators (a song, bursts in digital silence, a voice with rests): fixtures written
    in round 3 with the reference's VPBD on the oracle's stems, same assertions as the full-size test."""
    print(name, _vpbd_silero_fixture_asserts(hip_ctx, np.load(golden_dir / f"{name}.npz"), min_vad_segments=0))
