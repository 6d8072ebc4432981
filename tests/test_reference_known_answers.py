"""Known answers the reference's own unit tests hold for the boundary-policy row, replayed against
the product's host logic and, where the case needs no ASR prior, against the oracle.  The cases are the reference's
inputs and expected outputs (`tests/unit/test_segment_layout_refiner.py`,
`tests/unit/test_cpu_baseline_perfect_reconstruction.py`); they are data, rebuilt here from their description.
"""
from typing import List

import numpy as np

from audio_cut_amd.analysis.features_cache import TrackFeatureCache
from audio_cut_amd.core.seamless_splitter import SeamlessSplitter
from audio_cut_amd.cutting.segment_layout_refiner import LayoutConfig, Segment, refine_layout
from oracle import layout as OL


def _cache(rms: np.ndarray, duration_s: float, hop_s: float) -> TrackFeatureCache:
    z = np.zeros(len(rms), dtype=np.float32)
    return TrackFeatureCache(sr=44100, hop_length=int(44100 * hop_s), hop_s=hop_s, duration_s=duration_s, rms_series=rms,
                             spectral_flatness=z.copy(), onset_envelope=z.copy(), onset_strength=z.copy(),
                             onset_frames=np.array([], dtype=np.int64), rms_max=float(rms.max()), onset_max=0.0,
                             bpm_features=None, tempo_curve=None, beat_times=np.array([], dtype=np.float32), global_mdd=0.0,
                             mdd_series=z.copy())


def _one_notch(duration_s: float, valley_s: float, hop_s: float = 0.5) -> TrackFeatureCache:
    n = int(duration_s / hop_s) + 1
    rms = np.ones(n, dtype=np.float32)
    k = int(round(valley_s / hop_s))
    rms[max(0, k - 1): min(n, k + 2)] = np.array([0.08, 0.01, 0.08], dtype=np.float32)
    return _cache(rms, duration_s, hop_s)


def _vee_valleys(duration_s: float, valley_times: List[float], hop_s: float = 0.05) -> TrackFeatureCache:
    n = int(duration_s / hop_s) + 1
    rms = np.ones(n, dtype=np.float32)
    half = max(2, int(round(1.0 / hop_s)))
    for t in valley_times:
        k = int(round(t / hop_s))
        for i in range(max(0, k - half), min(n, k + half + 1)):
            rms[i] = min(rms[i], 0.01 + 0.4 * abs(i - k) / float(half))
    return _cache(rms, duration_s, hop_s)


def _ends(result) -> List[float]:
    return [round(s.end, 1) for s in result.segments[:-1]]


def test_long_segment_splits_at_valley_next_to_asr_boundary():
    res = refine_layout([Segment(0.0, 30.0, "human"), Segment(30.0, 35.0, "music")], [],
                        config=LayoutConfig(enable=True, soft_max_s=15.0, min_gap_s=1.0), sample_rate=44100,
                        features=_one_notch(35.0, 20.0), asr_boundary_times=[20.1])
    assert 20.1 in _ends(res) and 15.0 not in _ends(res)


def test_long_segment_without_valley_is_left_whole():
    feats = _one_notch(35.0, 20.0)
    feats.rms_series[:] = 1.0
    res = refine_layout([Segment(0.0, 30.0, "human"), Segment(30.0, 35.0, "music")], [],
                        config=LayoutConfig(enable=True, soft_max_s=15.0, min_gap_s=1.0), sample_rate=44100,
                        features=feats, asr_boundary_times=[15.0])
    assert _ends(res) == [30.0]
    segs, _, fresh = OL.refine_layout([[0.0, 30.0, "human"], [30.0, 35.0, "music"]],
                                      OL.LayoutConfig(enable=True, soft_max_s=15.0, min_gap_s=1.0), [], feats.rms_series, 0.5, [])
    assert [round(s[1], 1) for s in segs[:-1]] == [30.0] and fresh == []


def test_phrase_boundary_beats_a_deeper_valley_inside_a_word():
    feats = _one_notch(35.0, 20.0)
    feats.rms_series[:] = 1.0
    a, b = int(round(20.0 / feats.hop_s)), int(round(22.0 / feats.hop_s))
    feats.rms_series[a - 1: a + 2] = np.array([0.08, 0.01, 0.08], dtype=np.float32)
    feats.rms_series[b - 1: b + 2] = np.array([0.12, 0.05, 0.12], dtype=np.float32)
    feats.rms_max = float(feats.rms_series.max())
    res = refine_layout([Segment(0.0, 30.0, "human"), Segment(30.0, 35.0, "music")], [],
                        config=LayoutConfig(enable=True, soft_max_s=15.0, min_gap_s=1.0), sample_rate=44100, features=feats,
                        asr_boundary_times=[22.0], asr_word_intervals=[(19.8, 20.2)])
    assert 22.0 in _ends(res) and 20.0 not in _ends(res)


def test_micro_piece_made_by_a_split_rejoins_its_same_kind_neighbour():
    feats = _vee_valleys(30.0, [9.68, 20.70])
    cfg = dict(enable=True, micro_merge_s=2.0, soft_min_s=5.0, soft_max_s=12.0, min_gap_s=1.0)
    start = [(0.0, 21.95, "human"), (21.95, 23.31, "human"), (23.31, 26.47, "music")]
    want = [(0.0, 9.7, "human"), (9.7, 21.95, "human"), (21.95, 26.47, "music")]
    res = refine_layout([Segment(*s) for s in start], [], config=LayoutConfig(**cfg), sample_rate=44100, features=feats)
    assert [(round(s.start, 2), round(s.end, 2), s.kind) for s in res.segments] == want
    assert all(s.duration >= 2.0 for s in res.segments)
    segs, _, _ = OL.refine_layout([list(s) for s in start], OL.LayoutConfig(**cfg), [], feats.rms_series, feats.hop_s, [])
    assert [(round(s[0], 2), round(s[1], 2), s[2]) for s in segs] == want
    # the two restatements agree to the last bit, not just to the printed precision
    assert [(s.start, s.end) for s in res.segments] == [(s[0], s[1]) for s in segs]


def test_sample_level_split_reconstructs_the_track_exactly():
    splitter = SeamlessSplitter.__new__(SeamlessSplitter)
    splitter.sample_rate = 1000
    audio = np.linspace(-1.0, 1.0, 1001, dtype=np.float32)
    cuts, flags = [0, 123, 456, 789, 1001], [True, False, True, False]
    pieces, out_flags, debug = splitter._split_at_sample_level(audio, cuts, segment_flags=list(flags))
    assert out_flags == flags and debug is None
    assert np.array_equal(np.concatenate(pieces), audio)
    spans, oflags = OL.split_at_sample_level(len(audio), cuts, flags, 1000)
    assert oflags == flags and spans == list(zip(cuts[:-1], cuts[1:]))
