"""The oracle against the committed golden fixtures (written by tests/golden/make_golden.py from the
reference's own Python in the build container).  CPU only."""
import json

import numpy as np
import pytest

from audio_cut_amd.testing import signals
from oracle import chunking as OC, config as OCFG, detector as OD, e2e as OE, features as OF, refine as OR, vad as OV

SR = 44100


def test_effective_config_matches_reference_dump(golden_dir):
    dump = json.loads((golden_dir / "config_effective.json").read_text())["values"]
    for key, val in dump.items():
        assert OCFG.get_config(key, "<absent>") == val, key
    from audio_cut_amd import config as PCFG          # the product keeps its own copy of the same values
    for key, val in dump.items():
        assert PCFG.get_config(key, "<absent>") == val, key


def test_chunk_schedule_golden(golden_dir):
    g = np.load(golden_dir / "chunk_schedule.npz")
    rows = g["rows"]
    for total in np.unique(rows[:, 0]):
        ref = rows[rows[:, 0] == total][:, 1:]
        got = np.array([[p.index, p.start_s, p.end_s, p.halo_left_s, p.halo_right_s] for p in OC.chunk_plan(float(total))])
        assert np.array_equal(got, ref), total
    alt = np.array([[p.start_s, p.end_s, p.halo_left_s, p.halo_right_s] for p in OC.chunk_plan(33.0, 8.0, 3.0, 1.0)])
    assert np.array_equal(alt, g["alt"])
    assert [len(OC.chunk_plan(t)) for t in (60.0, 240.0, 1800.0)] == [8, 32, 240]


def test_derive_golden(golden_dir):
    rows = np.load(golden_dir / "derive.npz")["rows"]
    adapt = OCFG.get_config("pure_vocal_detection.relative_threshold_adaptation", {})
    for bpm, mdd, peak, rms, min_pause in rows:
        th = OD.resolve_threshold(0.26, adapt, None if bpm < 0 else float(bpm), None if mdd < 0 else float(mdd))
        assert (th.peak_ratio, th.rms_ratio) == (peak, rms)
        assert OD.resolve_min_pause(0.5, 1.0, None if bpm < 0 else float(bpm)) == min_pause


def _refine_case(seed, n_s, holes):
    rng = np.random.default_rng(int(seed))
    n = int(SR * n_s); t = np.arange(n) / SR
    env = np.clip(np.sin(2 * np.pi * 0.13 * (int(seed) + 1) * t), 0, None) ** 2
    mix = (rng.standard_normal(n) * 0.1 * env + 0.3 * env * np.sin(np.arange(n) * 0.05)).astype(np.float32)
    voc = (0.7 * mix + rng.standard_normal(n).astype(np.float32) * 0.01 * env).astype(np.float32)
    if holes:
        a = int(n * 0.3); mix[a:a + SR] = 0; voc[a:a + SR] = 0
    pts = np.stack([rng.uniform(0, n_s, 24), rng.uniform(0, 1, 24)], axis=1)
    return mix, voc, pts


@pytest.mark.parametrize("case", [0, 3])
def test_refine_golden(golden_dir, case):
    g = np.load(golden_dir / "refine.npz")
    seed, n_s, holes, floor_db = g[f"c{case}_params"]
    mix, voc, pts = _refine_case(seed, n_s, bool(holes))
    kw = dict(min_gap_s=1.2, max_keep=200, guard_db=1.5, search_right_ms=450.0, guard_win_ms=80.0, floor_db=float(floor_db))
    for legacy, tag in ((False, "live"), (True, "legacy")):
        OR.LEGACY_PROMOTION = legacy
        out = OR.finalize_cut_points(SR, mix, voc, [OR.Cut(float(a), float(b)) for a, b in pts], **kw)
        assert out.sample_boundaries == g[f"c{case}_boundaries_{tag}"].tolist()
        assert np.array_equal(np.array([a.final_time for a in out.adjustments]), g[f"c{case}_final_times_{tag}"])
    OR.LEGACY_PROMOTION = True
    lk = OR.prepare_quiet_lookup(voc, SR, 80.0, float(floor_db))
    assert np.array_equal(lk.rms_db[::997], g[f"c{case}_db_dec"])
    assert np.array_equal(lk.next_quiet[::997], g[f"c{case}_nq_dec"])


def test_reference_known_answer_boundaries():
    """tests/unit/test_cutting_consistency.py:20-46 of the reference: sample_boundaries == [0, 40, 80, 120]."""
    out = OR.finalize_cut_points(10, np.zeros(120, np.float32), None, [OR.Cut(4.0, 0.9), OR.Cut(8.0, 0.9)], min_gap_s=1.0,
                                 enable_mix_guard=False, enable_vocal_guard=False, zero_cross_win_ms=0.0)
    assert out.sample_boundaries == [0, 40, 80, 120]


def _fake_vad(chunk):
    blk = 2205
    n = len(chunk) // blk
    act = np.abs(chunk[: n * blk]).reshape(n, blk).mean(axis=1) > 0.02
    return [{"start": a * blk, "end": b * blk} for a, b, v in OD._runs(act) if v]


def test_chunk_vad_golden(golden_dir):
    g = np.load(golden_dir / "chunk_vad.npz")
    voc = signals.vocal_like(40.0, seed=11)
    ora = OV.ChunkVadOracle(SR, 120.0, 0.2, _fake_vad)
    for p in OC.chunk_plan(40.0):
        a = int(round(p.start_s * SR)); b = int(round(p.end_s * SR))
        ora.process_chunk(p, voc[a:b], SR)
    segs = ora.finalize()
    assert np.array_equal(np.array([[s["start"], s["end"]] for s in segs]), g["segments"])
    assert np.array_equal(np.array(ora.to_focus_windows()), g["focus"])


def test_features_and_detector_golden(golden_dir):
    g = np.load(golden_dir / "features_detector.npz")
    mix = signals.c2_song(27.0, seed=21)
    ob = OF.ChunkFeatureOracle(SR)
    for p in OC.chunk_plan(27.0):
        a = int(round(p.start_s * SR)); b = min(len(mix), int(round(p.end_s * SR)))
        ob.add_chunk(p, mix[a:b], SR)
    oc = ob.finalize(mix)
    assert np.array_equal(oc.rms_series, g["cache_rms"]) and np.array_equal(oc.spectral_flatness, g["cache_flat"])
    assert np.array_equal(oc.onset_envelope, g["cache_onset"]) and np.array_equal(oc.onset_frames, g["cache_onset_frames"])
    assert np.array_equal(oc.mdd_series, g["cache_mdd"]) and np.array_equal(oc.beat_times, g["cache_beat_times"])
    assert np.array_equal(np.asarray(oc.tempo_curve), g["cache_tempo_curve"])
    sc = g["cache_scalars"]
    assert (float(oc.bpm_features.main_bpm), oc.bpm_features.beat_strength, oc.bpm_features.tempo_variance,
            oc.global_mdd, oc.rms_max, oc.onset_max) == tuple(sc)

    def arr(ps):
        return np.array([[p.start_time, p.end_time, p.confidence, p.cut_point] for p in ps], dtype=np.float64).reshape(-1, 4)

    x = signals.c1_sine_silence(30.0, seed=1)
    assert np.array_equal(arr(OD.detect_pure_vocal_pauses(x, SR, enable_mdd_enhancement=True, original_audio=x)), g["c1_pauses"])
    voc = signals.vocal_like(27.0, seed=21)
    vad = [{"start": 1.0, "end": 6.2, "duration": 5.2}, {"start": 7.1, "end": 13.0, "duration": 5.9},
           {"start": 13.6, "end": 20.5, "duration": 6.9}, {"start": 21.4, "end": 26.5, "duration": 5.1}]
    p_vad = OD.detect_pure_vocal_pauses(voc, SR, enable_mdd_enhancement=True, original_audio=mix, feature_cache=oc, vad_segments=vad)
    p_no = OD.detect_pure_vocal_pauses(voc, SR, enable_mdd_enhancement=True, original_audio=mix, feature_cache=oc, vad_segments=[])
    assert np.array_equal(arr(p_vad), g["c2_pauses_vad"]) and np.array_equal(arr(p_no), g["c2_pauses_novad"])
    markers = OD.vocal_presence_markers(voc, SR)
    assert np.array_equal(np.array(markers["vocal_presence_cut_points_sec"]), g["marker_times"])
    quiet = voc.copy(); quiet[int(8 * SR): int(17 * SR)] *= 1e-3
    assert np.array_equal(np.array(OD.no_vocal_runs(quiet, SR, 6.0)), g["no_vocal_runs"])
    cands = [(p.cut_point, p.confidence) for p in p_no] + [(float(t), 1.0) for t in markers["vocal_presence_cut_points_sec"] if 0 < t < 27.0]
    for legacy, tag in ((False, "live"), (True, "legacy")):
        OR.LEGACY_PROMOTION = legacy
        assert OE.finalize_and_filter_cuts(cands, mix, voc, SR).sample_boundaries == g[f"final_boundaries_{tag}"].tolist()
    OR.LEGACY_PROMOTION = True


def test_vpp_multiplier_is_structurally_one():
    """pure_vocal_pause_detector.py:1464-1509: singing blocks are maximal True-runs, so no rest is ever counted."""
    rng = np.random.default_rng(0)
    for seed in range(3):
        voc = signals.vocal_like(12.0, seed=seed)
        mul, tag = OD.vpp_multiplier(voc, SR, 441, None)
        assert mul == 1.0 and tag.startswith("VPP{")


def _dormant_inputs():
    return {"voice": signals.voice_with_rests(14.0, seed=3), "bursts": signals.c1_sine_silence(12.0, seed=2)}


@pytest.mark.parametrize("tag", ["voice", "bursts"])
def test_dormant_multifeature_branch_golden(golden_dir, tag):
    """oracle.detector.detect_multifeature_pauses against the reference's own control logic
    (fixture written by make_golden.golden_dormant_branch with `enable_relative_energy_mode: false`)."""
    g = np.load(golden_dir / "dormant_branch.npz")
    x = _dormant_inputs()[tag]
    key = "pure_vocal_detection.enable_relative_energy_mode"
    OCFG.set_runtime_config({key: False})
    try:
        ft = OD.extract_vocal_features(x, SR)
        assert np.array_equal(ft.f0_contour, g[f"{tag}_f0"], equal_nan=True)
        assert np.array_equal(ft.f0_confidence, g[f"{tag}_voiced_prob"])
        assert np.array_equal(ft.spectral_centroid, g[f"{tag}_centroid"])
        assert np.array_equal(ft.harmonic_ratio, g[f"{tag}_harmonic_ratio"])
        assert np.array_equal(ft.zero_crossing_rate, g[f"{tag}_zcr"])
        assert np.array_equal(ft.rms_energy, g[f"{tag}_rms"])
        for j in range(3):
            assert np.array_equal(ft.formant_energies[j], g[f"{tag}_formant{j}"])
        assert np.array_equal(np.asarray(OD.detect_candidate_pauses(ft, SR), dtype=np.int64).reshape(-1, 2), g[f"{tag}_candidates"])
        for breath in (0, 1):
            ps = OD.detect_multifeature_pauses(x, SR, include_breath_candidates=bool(breath), features=ft)
            got = np.array([[p.start_time, p.end_time, p.confidence, p.cut_point] for p in ps], dtype=np.float64).reshape(-1, 4)
            assert np.array_equal(got, g[f"{tag}_pauses_breath{breath}"])
    finally:
        OCFG.reset_runtime_config()


def test_pyin_building_blocks_known_answers():
    """Closed-form anchors for the restated librosa.pyin pieces (librosa itself is not installable here)."""
    import scipy.stats
    from oracle import librosa_ops as L
    k = np.arange(0, 30)[:, None] * np.ones((1, 40)); n = np.arange(1, 41)[None, :] * np.ones((30, 1))
    assert np.array_equal(L._boltzmann_pmf(k, 2.0, n), scipy.stats.boltzmann.pmf(k, 2.0, n))
    t = L.transition_local_triangle(25, 7)
    assert np.allclose(t.sum(axis=1), 1.0) and np.all(t[np.abs(np.subtract.outer(np.arange(25), np.arange(25))) > 3] == 0)
    assert abs(L.note_to_hz("C2") - 65.40639132514966) < 1e-12 and abs(L.note_to_hz("A4") - 440.0) < 1e-12
    # a steady 220 Hz harmonic tone is voiced at the bin nearest 220 Hz; digital silence is unvoiced
    sr = 22050
    tt = np.arange(sr) / sr
    y = (0.5 * np.sin(2 * np.pi * 220.0 * tt) + 0.25 * np.sin(2 * np.pi * 440.0 * tt)).astype(np.float32)
    y[sr // 2:] = 0
    f0, voiced, vp = L.pyin(y, 65.0, 2000.0, sr=sr)
    mid = len(f0) // 4
    assert voiced[mid] and abs(f0[mid] - 220.0) / 220.0 < 0.006 and vp[mid] > 0.9
    assert not voiced[-3] and np.isnan(f0[-3]) and vp[-3] == 0.0
    # Burg LPC recovers an AR(2) process
    rng = np.random.default_rng(0)
    e = rng.standard_normal(4000)
    z = np.zeros(4000)
    for i in range(2, 4000):
        z[i] = 1.5 * z[i - 1] - 0.8 * z[i - 2] + e[i]
    a = L.lpc(z, 2)
    assert np.allclose(a, [1.0, -1.5, 0.8], atol=0.05)


def test_guard_plateau_predicate_and_soak_fixture(golden_dir):
    """tests/guard_plateau.py (the equivalence class of a guard boundary decided inside digital silence): the predicate accepts a
    moved boundary only on the epsilon plateau and only where the ORACLE's own dB series are bit-equal; the committed soak
    fixture (seed 301 / weights 61 / Silero 1) is consistent with it and every interior boundary of that track is such a decision."""
    import sys
    sys.path.insert(0, str(golden_dir.parent))
    from guard_plateau import HALF, boundary_context, classify_boundaries, map_cuts, plateau_db
    g = np.load(golden_dir / "c1_60s_seed301_w61_silero1_oracle.npz")
    plat = plateau_db()
    assert float(g["plateau_db"]) == float(plat) and abs(float(plat) + 119.999991) < 1e-5
    b = g["sample_boundaries"]
    assert int(g["guard_half_window"]) == HALF and g["db_vocal"].shape == (len(b), 2 * HALF + 1)
    on = g["on_plateau"]
    assert not on[0] and not on[-1] and int(on.sum()) == len(b) - 2
    assert np.all(g["db_vocal"][on, HALF] == plat) and np.all(g["db_mix"][on, HALF] == plat)
    ctx = {k: g[k] for k in ("on_plateau", "db_vocal", "db_mix", "stem_window", "guard_half_window")}
    n = int(round(float(g["seconds"]) * 44100))
    stem = np.zeros(n, np.float32)
    for k, bb in enumerate(b):                       # a stem that agrees with the oracle's around every boundary
        lo, hi = max(0, bb - HALF), min(n, bb + HALF + 1)
        stem[lo:hi] = g["stem_window"][k][lo - (bb - HALF): hi - (bb - HALF)]
    want = b.tolist()
    exact, equiv, fails = classify_boundaries(want, want, ctx, stem, 1e-9)
    assert len(exact) == len(want) and not equiv and not fails
    # the round-2 GPU result: boundary 2407367 -> 2407370, inside the plateau
    k = want.index(2407367)
    assert g["db_vocal"][k][HALF + 3] == plat and g["db_mix"][k][HALF + 3] == plat
    got = list(want); got[k] = 2407370
    exact, equiv, fails = classify_boundaries(got, want, ctx, stem, 1e-9)
    assert equiv == [(2407370, 2407367)] and not fails and len(exact) == len(want) - 1
    assert map_cuts([0, 2407370, n], equiv) == [0, 2407367, n]
    # moved to where the oracle's series is NOT the plateau value: rejected; so is any move of a boundary off the plateau
    off = int(np.flatnonzero(g["db_vocal"][k] != plat)[0]) - HALF
    got[k] = 2407367 + off
    assert classify_boundaries(got, want, ctx, stem, 1e-9)[2]
    ctx2 = dict(ctx); ctx2["on_plateau"] = np.zeros_like(on)
    got[k] = 2407370
    assert classify_boundaries(got, want, ctx2, stem, 1e-9)[2]
    # and a stem that differs around the boundary is rejected even on the plateau
    bad_stem = stem.copy(); bad_stem[2407367 - 100] += 1e-3
    assert classify_boundaries(got, want, ctx, bad_stem, 1e-5)[2]
    # boundary_context on synthetic data: exact zeros -> plateau, a tone -> not
    x = np.zeros(44100, np.float32); x[:20000] = 0.3 * np.sin(np.arange(20000) * 0.05).astype(np.float32)
    c = boundary_context(x, x, [10000, 35000], 44100, half=64)
    assert c["on_plateau"].tolist() == [False, True]


def test_c4_fixture_is_self_consistent_and_its_vad_calibration_is_the_oracles(golden_dir):
    """tests/golden/c4_full_oracle.npz (BASELINE configs[3] at full size): the stored Silero output-layer calibration is what
    `silero_synth.calibration_affine` derives with the CPU oracle (the GPU test and bench.py rebuild the weights from the two stored
    numbers without importing the oracle), the VAD really segments the track, and the VPBD counts add up."""
    import sys
    sys.path.insert(0, str(golden_dir.parent))
    from silero_synth import calibration_affine, synth_silero_weights
    g = np.load(golden_dir / "c4_full_oracle.npz")
    assert str(g["mode"]) == "vpbd_acoustic" and float(g["seconds"]) == 240.0 and int(g["seed"]) == 2 and str(g["generator"]) == "c2_song"
    a, b = calibration_affine(int(g["silero_seed"]), str(g["silero_calib"]))
    assert (a, b) == tuple(g["silero_affine"].tolist())
    w1 = synth_silero_weights(int(g["silero_seed"]), str(g["silero_calib"]), affine=g["silero_affine"])
    assert w1["decoder.decoder.2.bias"].shape == (1,) and np.isfinite(w1["decoder.decoder.2.weight"]).all()
    vad = g["vad_segments"]
    assert len(vad) >= 8 and np.all(vad[:, 1] > vad[:, 0]) and np.all(vad[1:, 0] >= vad[:-1, 1])
    acoustic, beat, merged, total, selected, suppressed = g["vpbd_counts"].tolist()
    assert selected == len(g["vpbd_selected"]) >= 10 and suppressed == len(g["vpbd_suppressed"]) and selected + suppressed == total
    sb = g["sample_boundaries"]
    assert sb[0] == 0 and sb[-1] == 240 * 44100 and np.all(np.diff(sb) > 0) and len(sb) >= 10
    assert g["cuts"][0] == 0 and g["cuts"][-1] == 240 * 44100 and not g["on_plateau"].any()
