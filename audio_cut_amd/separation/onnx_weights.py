"""Real-weights ingestion: read the initializers of an MDX-Net ONNX file (`Kim_Vocal_1.onnx`,
`Kim_Inst.onnx`; the reference hands the file to onnxruntime at `src/audio_cut/separation/backends.py:222-255`) and map
them onto `TfcTdfNet`'s parameter names, without the `onnx` / `onnxruntime` / `protobuf-onnx` packages.

Two layers:

* `read_onnx_graph(path)` — a minimal protobuf wire-format walk of ModelProto.graph: node list (op type, inputs,
  outputs, int attributes) and float32 initializers (raw_data or float_data).  Field numbers are ONNX's
  (onnx.proto3: ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5; NodeProto.input = 1, .output = 2,
  .op_type = 4, .attribute = 5; TensorProto.dims = 1, .data_type = 2, .float_data = 4, .name = 8, .raw_data = 9,
  .data_location = 14; AttributeProto.name = 1, .i = 3, .ints = 8).
* `load_tfc_tdf_weights(path, spec)` — walks the compute nodes in graph order and assigns Conv / ConvTranspose /
  MatMul (Gemm) / BatchNormalization parameters to the layers of the KUIELab TFC-TDF v2 graph in the order its
  `forward` runs them: first conv, `n_levels` x (block, down conv), bottleneck block, `n_levels` x (up conv, block),
  final conv; a block is `l` x (3x3 conv) + two frequency Linear layers.  Exporters that folded a conv's BatchNorm into
  its weights are accepted (identity statistics are filled in); the TDF BatchNorms act on the channel axis of a
  [B, C, T, F] tensor and cannot be folded into a MatMul over F, so they must be present.  Shapes are checked against
  `TfcTdfSpec`; anything unexpected raises `ValueError` naming the layer.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from .tfc_tdf import TfcTdfSpec

Weights = Dict[str, np.ndarray]


# ---------------------------------------------------------------------------------------------------
# protobuf wire format
# ---------------------------------------------------------------------------------------------------

def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    out = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf: memoryview) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value): varint -> int, 64/32-bit -> bytes, length-delimited -> memoryview."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        if pos > end:
            raise ValueError("truncated protobuf message")
        yield num, wt, val


def _packed_varints(val, wt) -> List[int]:
    if wt == 0:
        return [int(val)]
    out, pos = [], 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(v)
    return out


@dataclass
class OnnxNode:
    op_type: str
    inputs: List[str]
    outputs: List[str]
    ints: Dict[str, List[int]] = field(default_factory=dict)


def _parse_tensor(buf: memoryview) -> Tuple[str, Optional[np.ndarray]]:
    dims: List[int] = []
    dtype = 0
    name = ""
    raw: Optional[memoryview] = None
    floats: List[float] = []
    external = False
    for num, wt, val in _fields(buf):
        if num == 1:
            dims += _packed_varints(val, wt)
        elif num == 2:
            dtype = int(val)
        elif num == 4:
            floats += list(struct.unpack(f"<{len(val) // 4}f", bytes(val))) if wt == 2 else [struct.unpack("<f", val)[0]]
        elif num == 8:
            name = bytes(val).decode("utf-8")
        elif num == 9:
            raw = val
        elif num == 14 and int(val) == 1:
            external = True
    if dtype != 1:                       # only FLOAT initializers carry weights; shapes / axes constants are skipped
        return name, None
    if external:
        raise ValueError(f"initializer {name!r} stores its data externally; export the model with embedded weights")
    if raw is not None:
        arr = np.frombuffer(bytes(raw), dtype="<f4")
    else:
        arr = np.asarray(floats, dtype=np.float32)
    count = int(np.prod(dims)) if dims else 1
    if arr.size != count:
        raise ValueError(f"initializer {name!r}: {arr.size} values for dims {dims}")
    return name, arr.reshape(dims).astype(np.float32, copy=True)


def _parse_node(buf: memoryview) -> OnnxNode:
    node = OnnxNode("", [], [])
    for num, wt, val in _fields(buf):
        if num == 1:
            node.inputs.append(bytes(val).decode("utf-8"))
        elif num == 2:
            node.outputs.append(bytes(val).decode("utf-8"))
        elif num == 4:
            node.op_type = bytes(val).decode("utf-8")
        elif num == 5:
            aname, ints = "", []
            for anum, awt, aval in _fields(val):
                if anum == 1:
                    aname = bytes(aval).decode("utf-8")
                elif anum == 3:
                    v = int(aval)
                    ints = [v - (1 << 64) if v >= (1 << 63) else v]
                elif anum == 8:
                    ints += [v - (1 << 64) if v >= (1 << 63) else v for v in _packed_varints(aval, awt)]
            if aname and ints:
                node.ints[aname] = ints
    return node


def read_onnx_graph(path) -> Tuple[List[OnnxNode], Dict[str, np.ndarray]]:
    """(compute nodes in graph order, float32 initializers by name) of an ONNX model file."""
    data = memoryview(Path(path).read_bytes())
    graph = None
    for num, wt, val in _fields(data):
        if num == 7 and wt == 2:
            graph = val
    if graph is None:
        raise ValueError(f"{path}: no GraphProto (ModelProto field 7) found - not an ONNX model?")
    nodes: List[OnnxNode] = []
    inits: Dict[str, np.ndarray] = {}
    for num, wt, val in _fields(graph):
        if num == 1 and wt == 2:
            nodes.append(_parse_node(val))
        elif num == 5 and wt == 2:
            name, arr = _parse_tensor(val)
            if arr is not None:
                inits[name] = arr
    return nodes, inits


# ---------------------------------------------------------------------------------------------------
# graph order -> TfcTdfNet parameter names
# ---------------------------------------------------------------------------------------------------

@dataclass
class _Layer:
    kind: str                      # "conv", "convT", "linear"
    weight: np.ndarray
    bias: Optional[np.ndarray]
    bn: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]] = None   # scale, shift, mean, var


def _collect_layers(nodes: List[OnnxNode], inits: Dict[str, np.ndarray]) -> List[_Layer]:
    layers: List[_Layer] = []
    for nd in nodes:
        par = [inits[i] for i in nd.inputs if i in inits]
        if nd.op_type in ("Conv", "ConvTranspose"):
            if not par:
                raise ValueError(f"{nd.op_type} node without an initializer weight (inputs {nd.inputs})")
            layers.append(_Layer("conv" if nd.op_type == "Conv" else "convT", par[0], par[1] if len(par) > 1 else None))
        elif nd.op_type in ("MatMul", "Gemm"):
            mats = [p for p in par if p.ndim == 2]
            if not mats:
                continue                                   # activation x activation products do not occur in this graph
            w = mats[0]
            if nd.op_type == "Gemm" and nd.ints.get("transB", [0])[0] == 1:
                w = w.T                                    # Gemm(transB=1) already holds [N, K]; normalise to MatMul's [K, N]
            bias = next((p for p in par if p.ndim == 1), None)
            layers.append(_Layer("linear", np.ascontiguousarray(w), bias))
        elif nd.op_type == "BatchNormalization":
            if len(par) != 4:
                raise ValueError("BatchNormalization node without its four initializer inputs")
            if not layers or layers[-1].bn is not None:
                raise ValueError("BatchNormalization that does not follow a Conv / ConvTranspose / MatMul layer")
            layers[-1].bn = (par[0], par[1], par[2], par[3])
    return layers


def _identity_bn(c: int, eps: float):
    return (np.ones(c, np.float32), np.zeros(c, np.float32), np.zeros(c, np.float32), np.full(c, 1.0 - eps, np.float32))


def load_tfc_tdf_weights(path, spec: TfcTdfSpec = TfcTdfSpec()) -> Weights:
    """ONNX file -> `{name: float32 array}` with the names `separation/tfc_tdf.py` uses."""
    nodes, inits = read_onnx_graph(path)
    layers = _collect_layers(nodes, inits)
    it = iter(enumerate(layers))
    out: Weights = {}

    def take(kind: str, what: str) -> _Layer:
        try:
            idx, layer = next(it)
        except StopIteration:
            raise ValueError(f"{path}: graph ends before {what}") from None
        if layer.kind != kind:
            raise ValueError(f"{path}: expected a {kind} layer for {what}, found {layer.kind} (compute layer #{idx})")
        return layer

    def put_bn(prefix: str, layer: _Layer, c: int, *, required: bool, what: str) -> None:
        bn = layer.bn
        if bn is None:
            if required:
                raise ValueError(f"{path}: {what} has no BatchNormalization (it acts on the channel axis and cannot be folded)")
            bn = _identity_bn(c, spec.bn_eps)
        for key, arr in zip(("weight", "bias", "running_mean", "running_var"), bn):
            if arr.shape != (c,):
                raise ValueError(f"{path}: {what} BatchNormalization has shape {arr.shape}, expected ({c},)")
            out[f"{prefix}.{key}"] = np.ascontiguousarray(arr, dtype=np.float32)

    def put_conv(name: str, bn_name: Optional[str], kind: str, shape: Tuple[int, ...], c_out: int) -> None:
        layer = take(kind, name)
        if layer.weight.shape != shape:
            raise ValueError(f"{path}: {name} weight has shape {layer.weight.shape}, expected {shape}")
        out[name + ".weight"] = layer.weight
        out[name + ".bias"] = layer.bias if layer.bias is not None else np.zeros(c_out, np.float32)
        if out[name + ".bias"].shape != (c_out,):
            raise ValueError(f"{path}: {name} bias has shape {out[name + '.bias'].shape}, expected ({c_out},)")
        if bn_name is not None:
            put_bn(bn_name, layer, c_out, required=False, what=name)
        elif layer.bn is not None:
            raise ValueError(f"{path}: unexpected BatchNormalization after {name}")

    def put_block(prefix: str, c: int, f: int) -> None:
        for j in range(spec.l):
            put_conv(f"{prefix}.tfc.{j}.conv", f"{prefix}.tfc.{j}.bn", "conv", (c, c, spec.k, spec.k), c)
        h = f // spec.bn
        for k, (n_out, n_in) in enumerate(((h, f), (f, h))):
            name = f"{prefix}.tdf.{k}"
            layer = take("linear", name)
            w = layer.weight                               # MatMul layout [K, N]; torch Linear keeps [N, K]
            if w.shape == (n_in, n_out):
                w = np.ascontiguousarray(w.T)
            elif w.shape != (n_out, n_in):
                raise ValueError(f"{path}: {name} weight has shape {layer.weight.shape}, expected ({n_in}, {n_out})")
            if layer.bias is not None and np.any(layer.bias != 0):
                raise ValueError(f"{path}: {name} carries a bias; the TFC-TDF v2 TDF layers are bias-free")
            out[name + ".weight"] = w
            put_bn(name + ".bn", layer, c, required=True, what=name)

    g = spec.g
    put_conv("first_conv", "first_bn", "conv", (g, spec.dim_c, 1, 1), g)
    f = spec.dim_f
    for i in range(spec.n_levels):
        c = spec.channels(i)
        put_block(f"enc.{i}", c, f)
        put_conv(f"ds.{i}.conv", f"ds.{i}.bn", "conv", (c + g, c, 2, 2), c + g)
        f //= 2
    put_block("bottleneck", spec.channels(spec.n_levels), f)
    for i in range(spec.n_levels):
        c = spec.channels(spec.n_levels - i)
        put_conv(f"us.{i}.conv", f"us.{i}.bn", "convT", (c, c - g, 2, 2), c - g)
        f *= 2
        put_block(f"dec.{i}", c - g, f)
    put_conv("final_conv", None, "conv", (spec.dim_c, g, 1, 1), spec.dim_c)
    leftover = [layer.kind for _, layer in it]
    if leftover:
        raise ValueError(f"{path}: {len(leftover)} compute layers left over after final_conv ({leftover[:4]} ...)")
    return out
