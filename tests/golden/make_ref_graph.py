#!/usr/bin/env python
"""Build-container script: decodes the ONE artefact the reference holds for this path -- the saved TensorBoard
GraphDef ``libs/uresnet_graph/events.out.tfevents.1509651705.lee`` -- into ``tests/golden/ref_graph.json``.

    python tests/golden/make_ref_graph.py [/root/reference]

No TensorFlow: a TFRecord stream (u64 length, u32 crc, payload, u32 crc) of ``Event`` protos is read with a
protobuf WIRE-FORMAT reader; Event.graph_def (field 4) is a serialized GraphDef whose NodeDefs (field 1) carry
name (1), op (2), input (3) and attr (5: map<string, AttrValue>).  Only data is extracted (op counts, per-node
attributes, shapes, initializer constants): the fixture holds no reference source text.

The graph was written by an OLDER revision of the reference than the checked-in source (SURVEY.md Appendix C:
7x7 conv0/conv1, pre-activation residual units with biases, RMSProp); what carries over to the current code --
and what tests/test_oracle.py::test_oracle_topology_against_reference_graph asserts -- is: the 53-conv + 5-deconv
topology with its strides / SAME / NHWC attributes, the [deconv_i, skip] concat order and which skip feeds which
concat, the deconv filter layout [k,k,Cout,Cin] with its Conv2DBackpropInput / Conv2D-gradient duality, the
Xavier-uniform bounds, and the BatchNorm variable set (beta + moving averages, no gamma; two-pass moments).
Numerical parity stays UNPINNED: a GraphDef holds no activations.
"""
import json
import os
import struct
import sys


# ---- protobuf wire format --------------------------------------------------------------------------------------
def varint(buf, i):
    r, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, i
        s += 7


def fields(buf):
    """Yields (field number, wire type, value) of one message; length-delimited values as memoryview slices."""
    i, n = 0, len(buf)
    while i < n:
        key, i = varint(buf, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(buf, i)
        elif wt == 1:
            v = bytes(buf[i:i + 8]); i += 8
        elif wt == 2:
            ln, i = varint(buf, i)
            v = buf[i:i + ln]; i += ln
        elif wt == 5:
            v = bytes(buf[i:i + 4]); i += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield f, wt, v


def tfrecords(path):
    with open(path, "rb") as f:
        raw = f.read()
    i = 0
    while i < len(raw):
        (ln,) = struct.unpack_from("<Q", raw, i)
        yield memoryview(raw)[i + 12:i + 12 + ln]
        i += 12 + ln + 4


def shape_proto(buf):   # TensorShapeProto: dim (2) { size (1) }
    dims = []
    for f, _, v in fields(buf):
        if f == 2:
            size = 0
            for g, _, w in fields(v):
                if g == 1:
                    size = w if w < (1 << 63) else w - (1 << 64)
            dims.append(size)
    return dims


def tensor_proto(buf):  # dtype (1), tensor_shape (2), tensor_content (4), float_val (5), int_val (7)
    out = {"dtype": None, "shape": [], "floats": [], "ints": []}
    for f, wt, v in fields(buf):
        if f == 1:
            out["dtype"] = v
        elif f == 2:
            out["shape"] = shape_proto(v)
        elif f == 4:
            raw = bytes(v)
            if out["dtype"] == 1:
                out["floats"] = list(struct.unpack("<%df" % (len(raw) // 4), raw))
            elif out["dtype"] == 3:
                out["ints"] = list(struct.unpack("<%di" % (len(raw) // 4), raw))
        elif f == 5:
            if wt == 5:
                out["floats"].append(struct.unpack("<f", v)[0])
            else:
                raw = bytes(v)
                out["floats"] += list(struct.unpack("<%df" % (len(raw) // 4), raw))
        elif f == 7:
            if wt == 0:
                out["ints"].append(v)
            else:
                j, b = 0, bytes(v)
                while j < len(b):
                    x, j = varint(b, j)
                    out["ints"].append(x)
    return out


def attr_value(buf):    # list (1), s (2), i (3), f (4), b (5), type (6), shape (7), tensor (8)
    out = {}
    for f, wt, v in fields(buf):
        if f == 1:
            lst = {"i": [], "s": []}
            for g, gwt, w in fields(v):
                if g == 3:
                    if gwt == 0:
                        lst["i"].append(w)
                    else:
                        j, b = 0, bytes(w)
                        while j < len(b):
                            x, j = varint(b, j)
                            lst["i"].append(x)
                elif g == 2:
                    lst["s"].append(bytes(w).decode())
            out["list"] = lst
        elif f == 2:
            out["s"] = bytes(v).decode("utf-8", "replace")
        elif f == 3:
            out["i"] = v
        elif f == 4:
            out["f"] = struct.unpack("<f", v)[0]
        elif f == 5:
            out["b"] = bool(v)
        elif f == 7:
            out["shape"] = shape_proto(v)
        elif f == 8:
            out["tensor"] = tensor_proto(v)
    return out


def node_def(buf):
    n = {"name": "", "op": "", "input": [], "attr": {}}
    for f, _, v in fields(buf):
        if f == 1:
            n["name"] = bytes(v).decode()
        elif f == 2:
            n["op"] = bytes(v).decode()
        elif f == 3:
            n["input"].append(bytes(v).decode())
        elif f == 5:
            key, val = None, None
            for g, _, w in fields(v):
                if g == 1:
                    key = bytes(w).decode()
                elif g == 2:
                    val = attr_value(w)
            n["attr"][key] = val
    return n


def read_graph(path):
    nodes, producer, file_version = [], None, None
    for rec in tfrecords(path):
        for f, _, v in fields(rec):
            if f == 3:
                file_version = bytes(v).decode()
            elif f == 4:   # Event.graph_def: serialized GraphDef
                for g, _, w in fields(v):
                    if g == 1:
                        nodes.append(node_def(w))
                    elif g == 4:   # VersionDef
                        for h, _, x in fields(w):
                            if h == 1:
                                producer = x
    return nodes, producer, file_version


# ---- extraction -------------------------------------------------------------------------------------------------
def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "libs", "uresnet_graph", "events.out.tfevents.1509651705.lee")
    nodes, producer, file_version = read_graph(src)
    by_name = {n["name"]: n for n in nodes}
    hist = {}
    for n in nodes:
        hist[n["op"]] = hist.get(n["op"], 0) + 1

    def base(inp):   # strip "^ctrl" / ":k" / "/read"
        x = inp.lstrip("^").split(":")[0]
        return x[:-5] if x.endswith("/read") else x

    var_shape = {n["name"]: n["attr"]["shape"]["shape"] for n in nodes if n["op"] in ("VariableV2", "Variable")
                 and "shape" in n["attr"]}

    convs = []   # forward conv-like ops in graph order
    for n in nodes:
        if n["name"].startswith("UResNet/") and n["op"] in ("Conv2D", "Conv2DBackpropInput"):
            scope = n["name"].rsplit("/", 1)[0]
            filt = [base(i) for i in n["input"] if base(i).endswith("/weights")]
            a = n["attr"]
            convs.append({"scope": scope, "op": n["op"], "strides": a["strides"]["list"]["i"],
                          "padding": a["padding"]["s"], "data_format": a["data_format"]["s"],
                          "filter_shape": var_shape.get(filt[0]) if filt else None})
    # gradient ops: which op differentiates a transposed conv w.r.t. its input
    deconv_grad_ops = sorted(n["op"] for n in nodes if "conv2d_transpose_grad" in n["name"]
                             and n["op"] in ("Conv2D", "Conv2DBackpropFilter", "Conv2DBackpropInput")
                             and "deconv0" in n["name"])
    concats = []
    for n in nodes:
        if n["op"] == "ConcatV2" and n["name"].startswith("UResNet/concat"):
            ins = [base(i) for i in n["input"]][:2]

            def producer_scope(x):
                parts = x.split("/")
                if parts[1].startswith("resnet_module"):
                    return "/".join(parts[:3]) if parts[2].startswith("module") else "/".join(parts[:2])
                return "/".join(parts[:2])
            concats.append({"name": n["name"], "inputs": [producer_scope(x) for x in ins], "raw_inputs": ins})
    xavier = {}
    for n in nodes:
        if n["op"] == "Const" and n["name"].endswith("/weights/Initializer/random_uniform/max"):
            scope = n["name"][:-len("/weights/Initializer/random_uniform/max")]
            xavier[scope] = {"limit": n["attr"]["value"]["tensor"]["floats"][0], "shape": var_shape.get(scope + "/weights")}
    bn_vars = {}
    for name in var_shape:
        if "/BatchNorm/" in name or "/resnet_bn" in name:
            if name.count("/RMSProp") or name.count("/Adam"):
                continue
            scope, leaf = name.rsplit("/", 1)
            bn_vars.setdefault(scope, []).append(leaf)
    bn_leaf_sets = sorted({tuple(sorted(v)) for v in bn_vars.values()})
    moments = {"mean_reduction_axes": None, "has_squared_difference": hist.get("SquaredDifference", 0) > 0,
               "has_stop_gradient_on_mean": any(n["op"] == "StopGradient" and "moments" in n["name"] for n in nodes)}
    for n in nodes:
        if n["op"] == "Const" and n["name"].endswith("moments/mean/reduction_indices") and "tensor" in n["attr"].get("value", {}):
            moments["mean_reduction_axes"] = n["attr"]["value"]["tensor"]["ints"]
            break
    decay = sorted({round(n["attr"]["value"]["tensor"]["floats"][0], 6) for n in nodes if n["op"] == "Const"
                    and n["name"].endswith("AssignMovingAvg/decay") and n["attr"]["value"]["tensor"]["floats"]})
    placeholders = [{"name": n["name"], "shape": n["attr"].get("shape", {}).get("shape")} for n in nodes if n["op"] == "Placeholder"]
    out = {
        "source": "libs/uresnet_graph/events.out.tfevents.1509651705.lee (DeepLearnPhysics/u-resnet)",
        "generator": "tests/golden/make_ref_graph.py (protobuf wire-format reader, no TensorFlow)",
        "file_version": file_version, "graph_producer_version": producer, "num_nodes": len(nodes),
        "op_histogram": {k: hist[k] for k in sorted(hist)},
        "forward_convs": convs, "concats": concats, "deconv0_input_gradient_ops": deconv_grad_ops,
        "xavier_limits": xavier, "batchnorm_variable_sets": [list(s) for s in bn_leaf_sets],
        "num_batchnorm_scopes": len(bn_vars), "moments": moments, "moving_average_decay_constants": decay,
        "placeholders": placeholders,
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_graph.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", dst, ":", len(nodes), "nodes,", len(convs), "forward conv-like ops,", len(concats), "concats,",
          len(xavier), "initializer constants")


if __name__ == "__main__":
    main()
