"""A minimal ONNX *writer* for the weight-ingestion tests: serialises a TFC-TDF parameter dict as a ModelProto the way
an exporter lays the graph out (nodes in forward order, anonymous initializer names, Relu / Transpose / Mul nodes in
between, an int64 shape constant, raw_data or float_data payloads), with or without the convolutions' BatchNorms folded.
Test infrastructure only."""
import struct

import numpy as np


def _vi(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num: int, payload: bytes) -> bytes:
    return _vi((num << 3) | 2) + _vi(len(payload)) + payload


def _tensor(name: str, arr: np.ndarray, raw: bool = True) -> bytes:
    msg = b"".join(_vi((1 << 3) | 0) + _vi(int(d)) for d in arr.shape)
    if arr.dtype == np.int64:
        msg += _vi((2 << 3) | 0) + _vi(7) + _ld(9, arr.astype("<i8").tobytes())
    else:
        msg += _vi((2 << 3) | 0) + _vi(1)
        data = np.ascontiguousarray(arr, dtype="<f4")
        msg += _ld(9, data.tobytes()) if raw else _ld(4, struct.pack(f"<{data.size}f", *data.ravel().tolist()))
    return msg + _ld(8, name.encode())


def _node(op: str, inputs, outputs, ints=None) -> bytes:
    msg = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs) + _ld(4, op.encode())
    for k, v in (ints or {}).items():
        attr = _ld(1, k.encode())
        attr += (_vi((3 << 3) | 0) + _vi(v)) if isinstance(v, int) else _ld(8, b"".join(_vi(x) for x in v))
        msg += _ld(5, attr + _vi((20 << 3) | 0) + _vi(2 if isinstance(v, int) else 7))
    return msg


def write_tfc_tdf_onnx(path, w, spec, *, fold_conv_bn: bool = False, raw: bool = True, gemm_for_tdf: bool = False,
                       shuffle_seed=None) -> None:
    """`shuffle_seed`: write the initializers in a seeded random order (a reader must find them through the nodes' input names, not
    through their position in the file)."""
    nodes, inits = [], []
    counter = [0]
    cur = ["input"]

    def fresh(prefix="onnx::t"):
        counter[0] += 1
        return f"{prefix}_{counter[0]}"

    def init(arr):
        name = fresh("onnx::w")
        inits.append(_tensor(name, np.asarray(arr), raw=raw or counter[0] % 3 != 0))
        return name

    def emit(op, extra_inputs, ints=None):
        out = fresh()
        nodes.append(_node(op, [cur[0]] + extra_inputs, [out], ints))
        cur[0] = out

    def bn(prefix):
        emit("BatchNormalization", [init(w[prefix + ".weight"]), init(w[prefix + ".bias"]), init(w[prefix + ".running_mean"]),
                                    init(w[prefix + ".running_var"])])

    def conv(name, bn_name, op="Conv", stride=1):
        weight, bias = w[name + ".weight"].astype(np.float64), w[name + ".bias"].astype(np.float64)
        if fold_conv_bn and bn_name is not None:
            s = w[bn_name + ".weight"].astype(np.float64) / np.sqrt(w[bn_name + ".running_var"].astype(np.float64) + spec.bn_eps)
            shape = [1] * 4
            shape[1 if op == "ConvTranspose" else 0] = -1
            weight = weight * s.reshape(shape)
            bias = (bias - w[bn_name + ".running_mean"]) * s + w[bn_name + ".bias"]
        emit(op, [init(weight.astype(np.float32)), init(bias.astype(np.float32))], {"strides": [stride, stride], "group": 1})
        if bn_name is not None and not fold_conv_bn:
            bn(bn_name)
        if bn_name is not None:
            emit("Relu", [])

    def block(prefix):
        for j in range(spec.l):
            conv(f"{prefix}.tfc.{j}.conv", f"{prefix}.tfc.{j}.bn")
        for k in range(2):
            wt = w[f"{prefix}.tdf.{k}.weight"]
            if gemm_for_tdf:
                emit("Gemm", [init(wt)], {"transB": 1})
            else:
                emit("MatMul", [init(np.ascontiguousarray(wt.T))])
            bn(f"{prefix}.tdf.{k}.bn")
            emit("Relu", [])
        emit("Add", [cur[0]])

    inits.append(_tensor("onnx::shape_const", np.asarray([1, 4, 3072, 256], dtype=np.int64)))
    conv("first_conv", "first_bn")
    emit("Transpose", [], {"perm": [0, 1, 3, 2]})
    for i in range(spec.n_levels):
        block(f"enc.{i}")
        conv(f"ds.{i}.conv", f"ds.{i}.bn", stride=2)
    block("bottleneck")
    for i in range(spec.n_levels):
        conv(f"us.{i}.conv", f"us.{i}.bn", op="ConvTranspose", stride=2)
        emit("Mul", [cur[0]])
        block(f"dec.{i}")
    emit("Transpose", [], {"perm": [0, 1, 3, 2]})
    conv("final_conv", None)
    if shuffle_seed is not None:
        order = np.random.default_rng(shuffle_seed).permutation(len(inits))
        inits = [inits[i] for i in order]
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"tfc_tdf") + b"".join(_ld(5, t) for t in inits)
    model = _vi((1 << 3) | 0) + _vi(8) + _ld(2, b"audio-cut-test-writer") + _ld(7, graph)
    with open(path, "wb") as fh:
        fh.write(model)


def write_named_initializers_onnx(path, tensors) -> None:
    """A ModelProto whose graph holds only float32 initializers under the given names (the Silero VAD loader test)."""
    graph = b"".join(_ld(5, _tensor(name, np.asarray(arr, dtype=np.float32))) for name, arr in tensors.items())
    with open(path, "wb") as fh:
        fh.write(_vi((1 << 3) | 0) + _vi(8) + _ld(7, graph))           # ir_version = 8, graph
