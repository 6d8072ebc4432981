"""SURVEY.md §8(f) row 1 — the post-path boundary policy (classification, layout refiner, local valley, weak-tail merge,
sample-level split).  CPU: oracle.layout against the reference-generated fixture, and the product's host-side layout
refiner against the oracle on random segmentations.  GPU: the product policy end to end against the fixture."""
import numpy as np
import pytest

from audio_cut_amd.testing.policy_inputs import policy_case, random_layout_case
from oracle import layout as OL

SR = 44100


def _segs(arr):
    return [[float(a), float(b), "human" if k > 0.5 else "music"] for a, b, k in arr]


def test_layout_refiner_oracle_matches_reference_fixture(golden_dir):
    g = np.load(golden_dir / "boundary_policy.npz")
    for case in range(12):
        cfg = g[f"layout{case}_cfg"]
        lc = OL.LayoutConfig(enable=True, micro_merge_s=cfg[0], soft_min_s=cfg[1], soft_max_s=cfg[2], min_gap_s=cfg[3], beat_snap_ms=cfg[4])
        total = float(g[f"layout{case}_in"][-1, 1])
        out, _, _ = OL.refine_layout(_segs(g[f"layout{case}_in"]), lc, [tuple(x) for x in g[f"layout{case}_supp"]], g[f"layout{case}_rms"], 0.05,
                                     np.arange(0.3, total, 0.5), midpoint_fallback=bool(cfg[5]))
        ref = g[f"layout{case}_out"]
        assert np.array_equal(np.array([[s[0], s[1], 1.0 if s[2] == "human" else 0.0] for s in out]), ref), case


@pytest.mark.parametrize("seed", [11, 12])
def test_boundary_policy_oracle_matches_reference_fixture(golden_dir, seed):
    g = np.load(golden_dir / "boundary_policy.npz")
    voc, cuts, rms, hop_s, beats, supp = policy_case(seed)
    assert cuts == g[f"policy{seed}_cuts_in"].tolist()
    res = OL.apply_boundary_policy(cuts, voc, len(voc), SR, suppressed=supp, rms_series=rms, hop_s=hop_s, beat_times=beats)
    assert res.cuts == g[f"policy{seed}_cuts_out"].tolist()
    assert [int(f) for f in res.flags] == g[f"policy{seed}_flags"].tolist()
    assert [list(p) for p in res.pieces] == g[f"policy{seed}_pieces"].tolist()
    assert int(res.layout_applied) == int(g[f"policy{seed}_applied"][0])


def test_sample_level_split_edge_cases():
    # sub-10 ms slivers are glued forward, a trailing sliver backward, empty slices vanish
    pieces, flags = OL.split_at_sample_level(10000, [0, 100, 5000, 5000, 9990, 10000], [True, False, True, False, True], SR)
    assert pieces == [(0, 5000), (5000, 10000)] and flags == [True, True]
    pieces, flags = OL.split_at_sample_level(300, [0, 300], [False], SR)
    assert pieces == [(0, 300)] and flags == [False]
    assert OL.split_at_sample_level(0, [0, 0], [True], SR) == ([], [])


def test_product_layout_refiner_matches_oracle_on_random_segmentations():
    from audio_cut_amd.analysis.features_cache import TrackFeatureCache
    from audio_cut_amd.cutting.refine import CutPoint
    from audio_cut_amd.cutting import segment_layout_refiner as PL
    rng = np.random.default_rng(2024)
    for case in range(300):
        edges, kinds, rms, hop_s, beats, supp, cfg, midpoint = random_layout_case(rng, case)
        frames = len(rms)
        cache = TrackFeatureCache(sr=SR, hop_length=int(SR * hop_s), hop_s=hop_s, duration_s=float(edges[-1]), rms_series=rms,
                                  spectral_flatness=np.zeros(frames, np.float32), onset_envelope=np.zeros(frames, np.float32),
                                  onset_strength=np.zeros(frames, np.float32), onset_frames=np.zeros(0, np.int64), rms_max=float(rms.max()),
                                  onset_max=0.0, bpm_features=None, tempo_curve=None, beat_times=beats, global_mdd=0.5,
                                  mdd_series=np.zeros(frames, np.float32))
        got = PL.refine_layout([PL.Segment(float(edges[i]), float(edges[i + 1]), kinds[i]) for i in range(len(kinds))], [],
                               config=PL.LayoutConfig(**cfg), sample_rate=SR, suppressed_cut_points=[CutPoint(t=t, score=sc) for t, sc in supp],
                               features=cache, allow_midpoint_fallback=midpoint)
        ref, rsupp, _ = OL.refine_layout([[float(edges[i]), float(edges[i + 1]), kinds[i]] for i in range(len(kinds))], OL.LayoutConfig(**cfg),
                                         supp, rms, hop_s, beats, midpoint_fallback=midpoint)
        assert [(s.start, s.end, s.kind) for s in got.segments] == [(s[0], s[1], s[2]) for s in ref], case
        assert [(float(p.t), float(p.score)) for p in got.suppressed_points] == [(float(t), float(sc)) for t, sc in rsupp], case
    cfg0 = PL.derive_layout_config({"enable": 1, "micro_merge_s": "2", "soft_max_s": None}, None, sample_rate=SR)
    assert (cfg0.enable, cfg0.micro_merge_s, cfg0.soft_max_s, cfg0.min_gap_s) == (True, 2.0, 0.0, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_product_boundary_policy_on_gpu_matches_reference_fixture(hip_ctx, golden_dir, seed):
    from audio_cut_amd.analysis.features_cache import TrackFeatureCache
    from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
    from audio_cut_amd.cutting.refine import CutPoint
    g = np.load(golden_dir / "boundary_policy.npz")
    voc, cuts, rms, hop_s, beats, supp = policy_case(seed)
    n = len(voc)
    cache = TrackFeatureCache(sr=SR, hop_length=int(SR * hop_s), hop_s=hop_s, duration_s=n / SR, rms_series=rms,
                              spectral_flatness=np.zeros_like(rms), onset_envelope=np.zeros_like(rms), onset_strength=np.zeros_like(rms),
                              onset_frames=np.zeros(0, np.int64), rms_max=float(rms.max()), onset_max=0.0, bpm_features=None,
                              tempo_curve=None, beat_times=beats, global_mdd=0.5, mdd_series=np.zeros_like(rms))
    sp = SeamlessSplitter.__new__(SeamlessSplitter)
    sp.sample_rate = SR
    sp._hip = hip_ctx
    sp._last_guard_adjustments_raw = []
    sp._last_suppressed_cut_points = [CutPoint(t=t, score=sc) for t, sc in supp]
    dev = hip_ctx.to_device(voc)
    # the three device steps against the oracle, then the whole policy against the reference fixture
    flags = sp._classify_segments_vocal_presence(voc, cuts, vocal_dev=dev)
    assert flags == OL.classify_segments(voc, cuts, SR)[0]
    from oracle.config import get_config as oget
    lcl = oget("quality_control.local_boundary_refine", {})
    mid = sorted(set([0, n] + [int(t * SR) for t in (5.0, 13.1, 21.7, 30.2, 44.0, 52.3)]))
    assert sp._refine_boundaries_local_valley(mid, voc, lcl, min_gap_s=1.2, vocal_dev=dev) == OL.refine_local_valley(mid, voc, SR, lcl, 1.2)
    out = sp._apply_boundary_policy(cuts, voc, n, cache, vocal_dev=dev)
    assert out["cuts_samples"] == g[f"policy{seed}_cuts_out"].tolist()
    assert [int(f) for f in out["segment_vocal_flags"]] == g[f"policy{seed}_flags"].tolist()
    assert [list(p) for p in out["segment_spans"]] == g[f"policy{seed}_pieces"].tolist()
    assert int(out["segment_layout_applied"]) == int(g[f"policy{seed}_applied"][0])


def test_layout_refiner_invariants_hold_on_random_inputs():
    """Size-independent properties of the layout policy: the refined segments tile the original span without gaps, no
    interior piece is shorter than min_gap (when a neighbour exists to absorb it), long segments are cut only where the
    min-gap margins allow, and a second pass changes nothing once no segment is short or long (idempotence)."""
    from audio_cut_amd.cutting import segment_layout_refiner as PL
    rng = np.random.default_rng(7)
    for case in range(200):
        edges, kinds, rms, hop_s, beats, supp, cfg, midpoint = random_layout_case(rng, case)
        conf = PL.LayoutConfig(**{**cfg, "beat_snap_ms": 0.0})
        segs = [PL.Segment(float(edges[i]), float(edges[i + 1]), kinds[i]) for i in range(len(kinds))]
        out = PL.refine_layout(segs, [], config=conf, sample_rate=SR, allow_midpoint_fallback=True).segments
        assert out[0].start == segs[0].start and out[-1].end == segs[-1].end
        assert all(out[i].end == out[i + 1].start for i in range(len(out) - 1))
        assert all(s.end > s.start for s in out)
        if len(out) > 1:
            assert all(s.duration >= conf.min_gap_s - 1e-12 for s in out)
        again = PL.refine_layout(out, [], config=conf, sample_rate=SR, allow_midpoint_fallback=True).segments
        if all(conf.soft_min_s <= s.duration <= conf.soft_max_s for s in out):
            assert [(s.start, s.end, s.kind) for s in again] == [(s.start, s.end, s.kind) for s in out]
    off = PL.refine_layout(segs, [], config=PL.LayoutConfig(enable=False), sample_rate=SR).segments
    assert [(s.start, s.end) for s in off] == [(s.start, s.end) for s in segs]


def test_sample_level_split_tiles_the_track():
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(1, 400000))
        k = int(rng.integers(0, 12))
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=k)]))
        flags = [bool(b) for b in rng.integers(0, 2, size=len(cuts) - 1)]
        pieces, merged = OL.split_at_sample_level(n, cuts, flags, SR)
        assert pieces[0][0] == 0 and pieces[-1][1] == n and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        assert len(merged) == len(pieces) and (any(flags) == any(merged))
        assert all(hi - lo >= min(n, int(0.01 * SR)) for lo, hi in pieces) or len(pieces) == 1
