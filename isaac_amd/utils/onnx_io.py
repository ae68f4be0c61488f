"""Minimal ONNX reader / writer for the actor MLP (no `onnx` package needed; protobuf wire format by hand).

The reference exports its actor with `torch.onnx.export(actor, obs, "locomotion_net.onnx", opset_version=11,
input_names=['obs'], output_names=['action'])` (humanoid/scripts/play.py:89-98) and ships seven such files
(`humanoid/locomotion_net*.onnx`): a chain Gemm(transB=1) -> Elu -> ... -> Gemm with initialisers named after the
`nn.Sequential` indices (`0.weight [512,615]`, `0.bias`, `2.weight`, ... `6.bias [10]`).

  save_actor(path, layers)        writes that graph (weights as raw little-endian fp32)
  load_model(path)                generic parse: initialisers, nodes with attributes, graph inputs/outputs
  load_actor(path)                checks the Gemm/Elu chain and returns [(W[out,in], b[out]), ...]
  actor_state_dict(path)          {'actor.0.weight': ..., ...} for ActorCritic.load_state_dict(strict=False)
"""
import struct

import numpy as np

# ---------------------------------------------------------------------------------------------- wire format
_VARINT, _I64, _LEN, _I32 = 0, 1, 2, 5


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _key(field, wt):
    return _varint((field << 3) | wt)


def _f_varint(field, v):
    return _key(field, _VARINT) + _varint(int(v))


def _f_bytes(field, b):
    if isinstance(b, str):
        b = b.encode()
    return _key(field, _LEN) + _varint(len(b)) + b


def _f_float(field, x):
    return _key(field, _I32) + struct.pack("<f", x)


def _read_varint(buf, pos):
    v = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, pos


def _fields(buf):
    """Yield (field number, wire type, value) of one message; LEN values are memoryviews."""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        k, pos = _read_varint(buf, pos)
        f, wt = k >> 3, k & 7
        if wt == _VARINT:
            v, pos = _read_varint(buf, pos)
        elif wt == _I64:
            v, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == _LEN:
            ln, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == _I32:
            v, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield f, wt, v


# ---------------------------------------------------------------------------------------------- writer
def _tensor(name, arr):
    arr = np.ascontiguousarray(arr, "<f4")
    msg = b"".join(_f_varint(1, d) for d in arr.shape)
    msg += _f_varint(2, 1)                               # data_type FLOAT
    msg += _f_bytes(8, name) + _f_bytes(9, arr.tobytes())
    return msg


def _attr_f(name, x):
    return _f_bytes(1, name) + _f_float(2, x) + _f_varint(20, 1)


def _attr_i(name, v):
    return _f_bytes(1, name) + _f_varint(3, v) + _f_varint(20, 2)


def _node(op, name, inputs, outputs, attrs):
    msg = b"".join(_f_bytes(1, i) for i in inputs) + b"".join(_f_bytes(2, o) for o in outputs)
    msg += _f_bytes(3, name) + _f_bytes(4, op) + b"".join(_f_bytes(5, a) for a in attrs)
    return msg


def _value_info(name, dims):
    shape = b"".join(_f_bytes(1, _f_bytes(2, d) if isinstance(d, str) else _f_varint(1, d)) for d in dims)
    tensor = _f_varint(1, 1) + _f_bytes(2, shape)
    return _f_bytes(1, name) + _f_bytes(2, _f_bytes(1, tensor))


def save_actor(path, layers, input_name="obs", output_name="action", batch=None):
    """layers: [(W[out,in], b[out]), ...]; ELU(alpha=1) between consecutive Gemms, none after the last.
    batch=None writes a symbolic batch dimension."""
    nodes, inits = [], []
    cur = input_name
    last = len(layers) - 1
    for li, (W, b) in enumerate(layers):
        idx = 2 * li                                      # nn.Sequential index of the Linear
        wn, bn = f"{idx}.weight", f"{idx}.bias"
        inits += [_tensor(wn, W), _tensor(bn, b)]
        out = output_name if li == last else f"/{idx}/Gemm_output_0"
        nodes.append(_node("Gemm", f"/{idx}/Gemm", [cur, wn, bn], [out],
                           [_attr_f("alpha", 1.0), _attr_f("beta", 1.0), _attr_i("transB", 1)]))
        cur = out
        if li != last:
            out = f"/{idx + 1}/Elu_output_0"
            nodes.append(_node("Elu", f"/{idx + 1}/Elu", [cur], [out], [_attr_f("alpha", 1.0)]))
            cur = out
    n_in, n_out = int(np.shape(layers[0][0])[1]), int(np.shape(layers[-1][0])[0])
    bdim = "batch" if batch is None else int(batch)
    graph = b"".join(_f_bytes(1, n) for n in nodes) + _f_bytes(2, "main_graph")
    graph += b"".join(_f_bytes(5, t) for t in inits)
    graph += _f_bytes(11, _value_info(input_name, [bdim, n_in])) + _f_bytes(12, _value_info(output_name, [bdim, n_out]))
    model = _f_varint(1, 6) + _f_bytes(2, "isaac_amd") + _f_bytes(3, "1")
    model += _f_bytes(7, graph) + _f_bytes(8, _f_bytes(1, "") + _f_varint(2, 11))
    with open(path, "wb") as f:
        f.write(model)
    return path


# ---------------------------------------------------------------------------------------------- reader
def _parse_tensor(buf):
    dims, name, raw, floats, dtype = [], "", None, [], 1
    for f, wt, v in _fields(buf):
        if f == 1:
            if wt == _VARINT:
                dims.append(v)
            else:                                         # packed
                p, vb = 0, bytes(v)
                while p < len(vb):
                    d, p = _read_varint(vb, p)
                    dims.append(d)
        elif f == 2:
            dtype = v
        elif f == 4:
            floats.append(np.frombuffer(bytes(v), "<f4") if wt == _LEN else np.frombuffer(v, "<f4"))
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
    if dtype != 1:
        return name, None                                 # only fp32 initialisers matter here (skip int64 shapes etc.)
    data = np.frombuffer(raw, "<f4") if raw is not None else (np.concatenate(floats) if floats else np.zeros(0, "<f4"))
    return name, data.reshape(dims).astype(np.float32)


def _parse_attr(buf):
    name, val = "", None
    for f, wt, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = v if v < (1 << 63) else v - (1 << 64)
        elif f == 4:
            val = bytes(v)
    return name, val


def _parse_node(buf):
    node = dict(inputs=[], outputs=[], name="", op="", attrs={})
    for f, wt, v in _fields(buf):
        if f == 1:
            node["inputs"].append(bytes(v).decode())
        elif f == 2:
            node["outputs"].append(bytes(v).decode())
        elif f == 3:
            node["name"] = bytes(v).decode()
        elif f == 4:
            node["op"] = bytes(v).decode()
        elif f == 5:
            k, val = _parse_attr(v)
            node["attrs"][k] = val
    return node


def _parse_value_name(buf):
    for f, wt, v in _fields(buf):
        if f == 1:
            return bytes(v).decode()
    return ""


def load_model(path):
    with open(path, "rb") as fh:
        data = fh.read()
    model = dict(ir_version=None, opset=None, producer="", initializers={}, nodes=[], inputs=[], outputs=[])
    for f, wt, v in _fields(data):
        if f == 1:
            model["ir_version"] = v
        elif f == 2:
            model["producer"] = bytes(v).decode()
        elif f == 8:
            for g, _, w in _fields(v):
                if g == 2:
                    model["opset"] = w
        elif f == 7:
            for g, _, w in _fields(v):
                if g == 1:
                    model["nodes"].append(_parse_node(w))
                elif g == 5:
                    name, arr = _parse_tensor(w)
                    if arr is not None:
                        model["initializers"][name] = arr
                elif g == 11:
                    model["inputs"].append(_parse_value_name(w))
                elif g == 12:
                    model["outputs"].append(_parse_value_name(w))
    return model


def load_actor(path):
    """[(W[out,in], b[out]), ...] of a Gemm/Elu chain; raises ValueError for anything else."""
    m = load_model(path)
    init = m["initializers"]
    graph_inputs = [i for i in m["inputs"] if i not in init]
    if len(graph_inputs) != 1 or len(m["outputs"]) != 1:
        raise ValueError(f"{path}: expected one input and one output, got {graph_inputs} -> {m['outputs']}")
    cur, layers, expect_elu = graph_inputs[0], [], False
    for node in m["nodes"]:
        if node["inputs"][0] != cur:
            raise ValueError(f"{path}: node {node['name']} does not continue the chain at {cur}")
        if node["op"] == "Gemm" and not expect_elu:
            a = node["attrs"]
            if a.get("alpha", 1.0) != 1.0 or a.get("beta", 1.0) != 1.0 or a.get("transA", 0) != 0:
                raise ValueError(f"{path}: unsupported Gemm attributes {a}")
            W, b = init[node["inputs"][1]], init[node["inputs"][2]]
            if not a.get("transB", 0):
                W = W.T
            layers.append((np.ascontiguousarray(W), np.ascontiguousarray(b)))
            expect_elu = True
        elif node["op"] == "Elu" and expect_elu:
            if abs(node["attrs"].get("alpha", 1.0) - 1.0) > 0:
                raise ValueError(f"{path}: ELU alpha != 1")
            expect_elu = False
        else:
            raise ValueError(f"{path}: unexpected op {node['op']} in an MLP chain")
        cur = node["outputs"][0]
    if cur != m["outputs"][0] or not expect_elu:
        raise ValueError(f"{path}: chain must end in a Gemm that produces the graph output")
    for (W0, _), (W1, _) in zip(layers, layers[1:]):
        if W1.shape[1] != W0.shape[0]:
            raise ValueError(f"{path}: layer shapes do not chain")
    return layers


def actor_state_dict(path):
    return {f"actor.{2 * i}.{k}": v for i, (W, b) in enumerate(load_actor(path)) for k, v in (("weight", W), ("bias", b))}


def mlp_forward(layers, x):
    """numpy evaluation of the chain (ELU alpha=1): host-side check / tiny batches."""
    h = np.asarray(x, np.float32)
    for i, (W, b) in enumerate(layers):
        h = h @ W.T + b
        if i + 1 < len(layers):
            h = np.where(h > 0, h, np.expm1(np.minimum(h, 0))).astype(np.float32)
    return h
