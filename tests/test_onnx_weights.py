"""SURVEY.md §8(f) row 3 — reading an MDX-Net ONNX file's initializers into TfcTdfNet parameter names (CPU only)."""
import numpy as np
import pytest

from audio_cut_amd.separation.onnx_weights import load_tfc_tdf_weights, read_onnx_graph
from audio_cut_amd.separation.tfc_tdf import TfcTdfSpec, _fold, synth_weights
from tests.onnx_writer import write_tfc_tdf_onnx

SMALL = TfcTdfSpec(dim_f=64, dim_t=16, n_levels=2, l=2, g=8, bn=4)


@pytest.fixture(scope="module")
def small_weights():
    rng = np.random.default_rng(0)
    w = synth_weights(SMALL, seed=1, calib_t=16)
    for k in list(w):                                     # make every tensor informative (synthetic BN means are zeros)
        if k.endswith("running_mean") or (k.endswith(".bias") and "bn" not in k):
            w[k] = rng.standard_normal(w[k].shape).astype(np.float32) * 0.1
    return w


@pytest.mark.parametrize("raw,gemm", [(True, False), (False, False), (True, True)])
def test_round_trip_with_batchnorm_nodes(tmp_path, small_weights, raw, gemm):
    p = tmp_path / "net.onnx"
    write_tfc_tdf_onnx(p, small_weights, SMALL, fold_conv_bn=False, raw=raw, gemm_for_tdf=gemm)
    nodes, inits = read_onnx_graph(p)
    assert any(n.op_type == "ConvTranspose" for n in nodes) and "onnx::shape_const" not in inits   # int64 constants are skipped
    got = load_tfc_tdf_weights(p, SMALL)
    assert set(got) == set(small_weights)
    for k, v in small_weights.items():
        assert got[k].dtype == np.float32 and np.array_equal(got[k], v), k


def test_folded_conv_batchnorm_is_accepted(tmp_path, small_weights):
    p = tmp_path / "folded.onnx"
    write_tfc_tdf_onnx(p, small_weights, SMALL, fold_conv_bn=True)
    got = load_tfc_tdf_weights(p, SMALL)
    # what the network consumes is the folded pair: identical (to rounding) whether the exporter or the loader folded
    for name, bn, axis in (("first_conv", "first_bn", 0), ("enc.1.tfc.0.conv", "enc.1.tfc.0.bn", 0), ("ds.0.conv", "ds.0.bn", 0),
                           ("us.1.conv", "us.1.bn", 1)):
        wa, ba = _fold(small_weights[name + ".weight"], small_weights[name + ".bias"], small_weights, bn, SMALL.bn_eps, axis)
        wb, bb = _fold(got[name + ".weight"], got[name + ".bias"], got, bn, SMALL.bn_eps, axis)
        assert np.allclose(wa, wb, rtol=1e-6, atol=1e-8) and np.allclose(ba, bb, rtol=1e-6, atol=1e-7), name
    assert np.array_equal(got["dec.0.tdf.1.weight"], small_weights["dec.0.tdf.1.weight"])
    assert np.array_equal(got["dec.0.tdf.1.bn.running_var"], small_weights["dec.0.tdf.1.bn.running_var"])


def test_architecture_mismatches_are_named(tmp_path, small_weights):
    p = tmp_path / "net.onnx"
    write_tfc_tdf_onnx(p, small_weights, SMALL)
    with pytest.raises(ValueError, match="first_conv weight has shape"):
        load_tfc_tdf_weights(p, TfcTdfSpec(dim_f=64, dim_t=16, n_levels=2, l=2, g=16, bn=4))
    with pytest.raises(ValueError, match="expected a"):
        load_tfc_tdf_weights(p, TfcTdfSpec(dim_f=64, dim_t=16, n_levels=2, l=3, g=8, bn=4))
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\x08\x08\x12\x04test")
    with pytest.raises(ValueError, match="no GraphProto"):
        read_onnx_graph(bad)


def test_full_size_spec_parameter_count(tmp_path):
    """The real architecture (g = 48, L = 11): 16.67 M parameters = Kim_Vocal_1.onnx's 66.8 MB."""
    spec = TfcTdfSpec()
    rng = np.random.default_rng(3)
    w = {}
    from audio_cut_amd.separation import tfc_tdf
    # shapes only: cheap random tensors instead of the calibrated synthetic set
    ref = synth_shapes = {}
    def conv(name, co, ci, k, transpose=False):
        synth_shapes[name + ".weight"] = (ci, co, k, k) if transpose else (co, ci, k, k); synth_shapes[name + ".bias"] = (co,)
    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            synth_shapes[f"{name}.{s}"] = (c,)
    def block(prefix, c, f):
        for j in range(spec.l):
            conv(f"{prefix}.tfc.{j}.conv", c, c, spec.k); bn(f"{prefix}.tfc.{j}.bn", c)
        synth_shapes[f"{prefix}.tdf.0.weight"] = (f // spec.bn, f); bn(f"{prefix}.tdf.0.bn", c)
        synth_shapes[f"{prefix}.tdf.1.weight"] = (f, f // spec.bn); bn(f"{prefix}.tdf.1.bn", c)
    conv("first_conv", spec.g, spec.dim_c, 1); bn("first_bn", spec.g)
    f = spec.dim_f
    for i in range(spec.n_levels):
        c = spec.channels(i); block(f"enc.{i}", c, f); conv(f"ds.{i}.conv", c + spec.g, c, 2); bn(f"ds.{i}.bn", c + spec.g); f //= 2
    block("bottleneck", spec.channels(spec.n_levels), f)
    for i in range(spec.n_levels):
        c = spec.channels(spec.n_levels - i); conv(f"us.{i}.conv", c - spec.g, c, 2, transpose=True); bn(f"us.{i}.bn", c - spec.g); f *= 2
        block(f"dec.{i}", c - spec.g, f)
    conv("final_conv", spec.dim_c, spec.g, 1)
    for k, shp in synth_shapes.items():
        w[k] = (rng.standard_normal(shp).astype(np.float32) if not k.endswith("running_var") else rng.uniform(0.5, 1.5, shp).astype(np.float32))
    p = tmp_path / "full.onnx"
    write_tfc_tdf_onnx(p, w, spec)
    got = load_tfc_tdf_weights(p, spec)
    assert all(np.array_equal(got[k], w[k]) for k in w)
    n_par = sum(v.size for k, v in got.items() if k.endswith(".weight") and ".bn." not in k and "first_bn" not in k or (k.endswith(".bias") and "bn" not in k))
    assert abs(p.stat().st_size / 1e6 - 66.8) < 1.5          # the published file size of Kim_Vocal_1.onnx
    assert n_par == spec.param_count()
