"""The convolution stack between CPM2Input and CPM2Output (BASELINE config 5): the body network of the
OpenPose COCO-18 model, run layer by layer through ``st_conv2d_nhwc_f32`` / ``st_maxpool2_nhwc_f32``
(MFMA kernels, csrc/st_conv.hip).

In the reference this is a Caffe forward pass: the ``CPM2`` op (scannertools_caffe_cpp/cpm2_kernel.cpp:8-52,
a ``CaffeKernel``) or, in the built library, OpenPose's own wrapper (openpose_kernel.cpp:129-168); prototxt
and caffemodel are downloaded at run time (openpose_kernel.cpp:35-78) and are not in the reference tree.
The layer list below is the published ``pose_deploy_linevec.prototxt`` of that model, written down from
knowledge of it ([EXT]; to be re-checked against the real file), and the weights are RANDOM (there is no
network here): what this module provides is the architecture at full size, float32 like the reference,
checked layer by layer and end to end against ``torch.nn.functional.conv2d`` on the same weights.

Layout: activations NHWC float32, channel counts padded to multiples of 16 with zero channels.  The stage
inputs concat(PAF 38, heat maps 19, features 128) live in one 192-channel buffer, stored as [features |
PAF | heat maps | 7 zero channels] (the features first so that every read starts 16-byte aligned; the
first layer's weights of stages 2..6 are permuted to match), that the branches write their slices of, so
no concatenation pass exists.  Output: (n, H/8, W/8, 57) = 19 heat maps followed by 38
part-affinity planes, the channel order ``cpm2_output_kernel_cpu.cpp:84-88`` indexes (the reference then
up-samples x8 and runs NMS inside the Caffe fork; not built -- profiles/NOTES.md, Part II section 9).
"""
import ctypes

import numpy as np

from . import _native

# (name, cin, cout, kernel, relu); "pool" = 2x2 max pooling
TRUNK = [("conv1_1", 3, 64, 3, 1), ("conv1_2", 64, 64, 3, 1), "pool",
         ("conv2_1", 64, 128, 3, 1), ("conv2_2", 128, 128, 3, 1), "pool",
         ("conv3_1", 128, 256, 3, 1), ("conv3_2", 256, 256, 3, 1), ("conv3_3", 256, 256, 3, 1), ("conv3_4", 256, 256, 3, 1), "pool",
         ("conv4_1", 256, 512, 3, 1), ("conv4_2", 512, 512, 3, 1), ("conv4_3_CPM", 512, 256, 3, 1), ("conv4_4_CPM", 256, 128, 3, 1)]
N_PAF, N_HEAT, N_FEAT = 38, 19, 128
CAT = N_PAF + N_HEAT + N_FEAT          # 185 channels into stages 2..6
CAT_PAD = 192
OFF_FEAT, OFF_PAF, OFF_HEAT = 0, N_FEAT, N_FEAT + N_PAF   # channel offsets inside the stage-input buffer
# prototxt order of the concatenation is (PAF, heat maps, features): buffer channel -> prototxt channel
CAT_PERM = list(range(N_PAF + N_HEAT, CAT)) + list(range(0, N_PAF + N_HEAT))


def branch_layers(stage, out):
    """Layers of one branch (L1: out = 38 part-affinity planes, L2: out = 19 heat maps) of a stage."""
    if stage == 1:
        return [(128, 128, 3, 1)] * 3 + [(128, 512, 1, 1), (512, out, 1, 0)]
    return [(CAT, 128, 7, 1)] + [(128, 128, 7, 1)] * 4 + [(128, 128, 1, 1), (128, out, 1, 0)]


def all_layers():
    """Every convolution as (name, cin, cout, k, relu), in execution order."""
    out = [l for l in TRUNK if l != "pool"]
    for st in range(1, 7):
        for br, n in (("L1", N_PAF), ("L2", N_HEAT)):
            for i, (ci, co, k, r) in enumerate(branch_layers(st, n)):
                out.append(("stage%d_%s_%d" % (st, br, i + 1), ci, co, k, r))
    return out


def flops(h, w):
    """2 * MACs of one forward pass on an (h, w) network input."""
    total, hh, ww = 0, h, w
    for l in TRUNK:
        if l == "pool":
            hh, ww = hh // 2, ww // 2
        else:
            total += 2 * hh * ww * l[1] * l[2] * l[3] * l[3]
    for st in range(1, 7):
        for n in (N_PAF, N_HEAT):
            for ci, co, k, _ in branch_layers(st, n):
                total += 2 * hh * ww * ci * co * k * k
    return total


def caffe_layer_names():
    """Names of the convolution layers in the published prototxt ([EXT]), in all_layers() order: the trunk keeps
    its VGG names; stage 1 is conv5_{1..5}_CPM_L{1,2}; stages 2..6 are Mconv{1..7}_stage{s}_L{1,2}."""
    out = [l[0] for l in TRUNK if l != "pool"]
    for st in range(1, 7):
        for br in ("L1", "L2"):
            n = len(branch_layers(st, 1))
            out += [("conv5_%d_CPM_%s" % (i + 1, br)) if st == 1 else ("Mconv%d_stage%d_%s" % (i + 1, st, br)) for i in range(n)]
    return out


# ---- prototxt (the network DESCRIPTION the reference hands Caffe next to the weights: cpm2_kernel.cpp:8-52 through
# CaffeArgs.net_descriptor.model_path; OpenPose reads <model_directory>/pose/coco/pose_deploy_linevec.prototxt) -------------
def parse_prototxt(text):
    """Protobuf text format -> nested {field: [values]} (every field a list: repeated fields are the rule in a NetParameter)."""
    import re
    tok = re.findall(r'#[^\n]*|"(?:[^"\\]|\\.)*"|\'(?:[^\'\\]|\\.)*\'|[{}:]|[^\s{}:#"\']+', text)
    tok = [t for t in tok if not t.startswith("#")]
    pos = 0

    def message(closing):
        nonlocal pos
        out = {}
        while pos < len(tok):
            t = tok[pos]
            if t == "}":
                if not closing:
                    raise ValueError("prototxt: unbalanced '}'")
                pos += 1
                return out
            name = t
            pos += 1
            if pos < len(tok) and tok[pos] == ":":
                pos += 1
            if pos >= len(tok):
                raise ValueError("prototxt: field %r has no value" % name)
            if tok[pos] == "{":
                pos += 1
                val = message(True)
            else:
                val = tok[pos]
                pos += 1
                if val[0] in "\"'":
                    val = val[1:-1]
            out.setdefault(name, []).append(val)
        if closing:
            raise ValueError("prototxt: missing '}'")
        return out

    return message(False)


def layers_from_prototxt(path):
    """The convolutions a deploy prototxt describes, in file order, with the input channel count of each inferred by
    walking the blobs (input_dim / input_shape, Convolution, Pooling, ReLU, Concat; the fork's trailing `resize` / `nms`
    layers and anything after them are ignored): [(name, cin, cout, kernel, relu, bottom blob, top blob)], plus the names
    of the pooling layers' bottoms (where the trunk pools)."""
    net = parse_prototxt(open(path).read())
    chans = {}
    if "input" in net:
        dims = [int(d) for d in net.get("input_dim", [])]
        if not dims and "input_shape" in net:
            dims = [int(d) for d in net["input_shape"][0].get("dim", [])]
        if len(dims) >= 2:
            chans[net["input"][0]] = dims[1]
    convs, pools = [], []
    for layer in net.get("layer", []) + net.get("layers", []):
        typ = str(layer.get("type", [""])[0])
        name = str(layer.get("name", [""])[0])
        bottoms, tops = layer.get("bottom", []), layer.get("top", [])
        if typ in ("Convolution", "CONVOLUTION", "ReLU", "RELU", "Pooling", "POOLING", "Concat", "CONCAT") and not (bottoms and tops):
            raise ValueError("prototxt layer %s (%s) has no %s blob" % (name, typ, "bottom" if not bottoms else "top"))
        if typ in ("Input",):
            shape = layer.get("input_param", [{}])[0].get("shape", [{}])[0].get("dim", [])
            if tops and len(shape) >= 2:
                chans[tops[0]] = int(shape[1])
        elif typ in ("Convolution", "CONVOLUTION"):
            cp = layer.get("convolution_param", [{}])[0]
            cout = int(cp["num_output"][0])
            k = int(cp.get("kernel_size", cp.get("kernel_h", [1]))[0])
            pad = int(cp.get("pad", [0])[0])
            if 2 * pad + 1 != k or int(cp.get("stride", [1])[0]) != 1:
                raise ValueError("prototxt layer %s: only stride-1 'same' convolutions are implemented (kernel %d, pad %d)" % (name, k, pad))
            if bottoms[0] not in chans:
                raise ValueError("prototxt layer %s reads blob %r, which nothing produced" % (name, bottoms[0]))
            convs.append([name, chans[bottoms[0]], cout, k, 0, bottoms[0], tops[0]])
            chans[tops[0]] = cout
        elif typ in ("ReLU", "RELU"):
            for c in convs[::-1]:
                if c[6] == bottoms[0]:
                    c[4] = 1
                    break
            chans[tops[0]] = chans.get(bottoms[0], 0)
        elif typ in ("Pooling", "POOLING"):
            pools.append(bottoms[0])
            chans[tops[0]] = chans.get(bottoms[0], 0)
        elif typ in ("Concat", "CONCAT"):
            chans[tops[0]] = sum(chans.get(b, 0) for b in bottoms)
    return [tuple(c) for c in convs], pools


def names_from_prototxt(path):
    """Checks that the prototxt describes the network the kernels implement and returns its layer names in all_layers()
    order (a model with other layer NAMES and the same structure is usable).  The layers are identified by walking the
    blobs from the input, not by their position in the file: the published ``pose_deploy_linevec.prototxt`` ([EXT])
    interleaves the two branches of a stage layer by layer (``conv5_1_CPM_L1``, ``conv5_1_CPM_L2``, ``conv5_2_CPM_L1`` ...)
    where all_layers() lists branch L1, then L2.  Trunk = the single chain of convolutions / poolings from the input to
    the blob two convolutions read; a stage = the two chains behind that blob, told apart by their output counts (38 = L1,
    19 = L2); next stage input = the Concat of (L1, L2, features), in that order (the order the weights are packed for).
    Raises ValueError naming the first difference."""
    convs, pools = layers_from_prototxt(path)
    arch = all_layers()
    if len(convs) != len(arch):
        raise ValueError("prototxt %s describes %d convolutions, the kernels implement %d" % (path, len(convs), len(arch)))
    if len(pools) != 3:
        raise ValueError("prototxt %s has %d pooling layers, the kernels implement 3" % (path, len(pools)))
    net = parse_prototxt(open(path).read())
    readers, pool_of, concats, alias = {}, {}, [], {}
    layers = net.get("layer", []) + net.get("layers", [])
    def io(layer, key):
        """First `top` / `bottom` blob of a layer; a layer without it is a malformed description (ValueError, like pose_net.h)."""
        v = layer.get(key) or []
        if not v:
            raise ValueError("prototxt %s: layer %s (%s) has no %s blob" % (path, str(layer.get("name", ["?"])[0]),
                                                                          str(layer.get("type", ["?"])[0]), key))
        return v[0]

    for layer in layers:      # a ReLU that is not in place renames its blob
        if str(layer.get("type", [""])[0]) in ("ReLU", "RELU") and io(layer, "top") != io(layer, "bottom"):
            alias[io(layer, "top")] = io(layer, "bottom")

    def blob(b):
        seen = set()
        while b in alias:   # two renaming ReLUs a -> b, b -> a would walk forever (pose_net.h caps its walk as well)
            if b in seen:
                raise ValueError("prototxt %s: the ReLU layers rename blob %r in a cycle" % (path, b))
            seen.add(b)
            b = alias[b]
        return b

    for c in convs:
        readers.setdefault(blob(c[5]), []).append(c)
    for layer in layers:
        typ = str(layer.get("type", [""])[0])
        if typ in ("Pooling", "POOLING"):
            pool_of[blob(io(layer, "bottom"))] = io(layer, "top")
        elif typ in ("Concat", "CONCAT"):
            concats.append(([blob(b) for b in layer.get("bottom", [])], io(layer, "top")))

    def expect(conv, spec):
        aname, aci, aco, ak, arelu = spec
        pname, ci, co, k, relu = conv[:5]
        if (ci, co, k, relu) != (aci, aco, ak, arelu):
            raise ValueError("prototxt layer %s is a %dx%d convolution %d -> %d (relu %d); the kernels implement %s as %dx%d %d -> %d (relu %d)"
                             % (pname, k, k, ci, co, relu, aname, ak, ak, aci, aco, arelu))

    def only_reader(blob, after):
        rs = readers.get(blob, [])
        if len(rs) != 1:
            raise ValueError("prototxt %s: %d convolutions read blob %r behind %s; the kernels implement a single chain there" % (path, len(rs), blob, after))
        return rs[0]

    it = iter(arch)
    names = []
    if "input" in net:
        cur = net["input"][0]
    else:
        cur = next((l["top"][0] for l in layers if str(l.get("type", [""])[0]) == "Input"), None)
    last = "the input"
    for l in TRUNK:
        if l == "pool":
            if cur not in pool_of:
                raise ValueError("prototxt %s: no pooling layer behind %s; the kernels pool there" % (path, last))
            cur = pool_of[cur]
        else:
            if cur in pool_of and readers.get(cur) is None:
                raise ValueError("prototxt %s pools behind %s; the kernels do not" % (path, last))
            c = only_reader(cur, last)
            expect(c, next(it))
            names.append(c[0])
            cur, last = c[6], c[0]
    feat = cur
    src = feat
    for st in range(1, 7):
        heads = readers.get(src, [])
        if len(heads) != 2:
            raise ValueError("prototxt %s: %d convolutions read the input of stage %d (blob %r); the kernels implement two branches" % (path, len(heads), st, src))
        depth = len(branch_layers(st, 1))
        chains = []
        for head in heads:
            chain = [head]
            while len(chain) < depth:
                chain.append(only_reader(chain[-1][6], chain[-1][0]))
            chains.append(chain)
        by_out = {ch[-1][2]: ch for ch in chains}
        if set(by_out) != {N_PAF, N_HEAT}:
            raise ValueError("prototxt %s: the branches of stage %d end in %s outputs; the kernels implement %d (L1) and %d (L2)"
                             % (path, st, sorted(ch[-1][2] for ch in chains), N_PAF, N_HEAT))
        for n in (N_PAF, N_HEAT):
            for c in by_out[n]:
                expect(c, next(it))
                names.append(c[0])
        if st < 6:
            want = [by_out[N_PAF][-1][6], by_out[N_HEAT][-1][6], feat]
            cat = [top for bottoms, top in concats if sorted(bottoms) == sorted(want)]
            if not cat:
                raise ValueError("prototxt %s: no Concat of the two branches of stage %d and the features" % (path, st))
            order = next(bottoms for bottoms, top in concats if top == cat[0])
            if order != want:
                raise ValueError("prototxt %s: stage %d's input concatenates %s; the kernels implement the order (L1, L2, features) = %s" % (path, st + 1, order, want))
            src = cat[0]
    return names


def write_prototxt(path, names=None, interleaved=False):
    """The deploy description of the built-in architecture in the published model's naming (pose_deploy_linevec.prototxt, [EXT])
    -- input, trunk with ReLU and pooling layers, six two-branch stages with their concatenations; used by tests and the
    benchmark next to write_caffemodel.  ``names`` are given in all_layers() order.  ``interleaved=True`` writes the two
    branches of a stage layer by layer (L1's first, L2's first, L1's second ...), the order of the published file; the
    default writes branch L1, then branch L2."""
    names = names or caffe_layer_names()
    it = iter(names)
    out = ['name: "pose"', 'input: "image"', "input_dim: 1", "input_dim: 3", "input_dim: 368", "input_dim: 656"]

    def conv(name, bottom, top, co, k, relu):
        out.append('layer { name: "%s" type: "Convolution" bottom: "%s" top: "%s" convolution_param { num_output: %d pad: %d kernel_size: %d } }'
                   % (name, bottom, top, co, k // 2, k))
        if relu:
            out.append('layer { name: "relu_%s" type: "ReLU" bottom: "%s" top: "%s" }' % (name, top, top))

    cur, npool = "image", 0
    for l in TRUNK:
        if l == "pool":
            npool += 1
            out.append('layer { name: "pool%d_stage1" type: "Pooling" bottom: "%s" top: "pool%d" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }'
                       % (npool, cur, npool))
            cur = "pool%d" % npool
        else:
            name = next(it)
            conv(name, cur, name, l[2], l[3], l[4])
            cur = name
    feat = cur
    for st in range(1, 7):
        src = feat if st == 1 else "concat_stage%d" % st
        ends, entries = [], []
        for br, n in (("L1", N_PAF), ("L2", N_HEAT)):
            cur, mine = src, []
            for (ci, co, k, r) in branch_layers(st, n):
                name = next(it)
                mine.append((name, cur, name, co, k, r))
                cur = name
            ends.append(cur)
            entries.append(mine)
        order = [e for pair in zip(*entries) for e in pair] if interleaved else entries[0] + entries[1]
        for e in order:
            conv(*e)
        if st < 6:
            out.append('layer { name: "concat_stage%d" type: "Concat" bottom: "%s" bottom: "%s" bottom: "%s" top: "concat_stage%d" concat_param { axis: 1 } }'
                       % (st + 1, ends[0], ends[1], feat, st + 1))
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")


def read_caffemodel(path):
    """Weights of a Caffe model file: {layer name: [blob, ...]} with every blob a float32 array of its stored
    shape.  Reads the NetParameter wire format directly (caffe.proto, [EXT]: NetParameter.layer = 100 and the V1
    `layers` = 2; LayerParameter.name = 1, .blobs = 7 (V1: name = 4, blobs = 6); BlobProto.data = 5 packed float,
    .shape = 7 {dim = 1}, legacy num/channels/height/width = 1..4)."""
    from . import _proto
    with open(path, "rb") as fh:
        buf = fh.read()
    out = {}

    def blob(mv):
        data, dims, legacy = None, [], {}
        for num, wt, v in _proto.fields(mv):
            if num == 5 and wt == 2:
                data = np.frombuffer(v, dtype="<f4")
            elif num == 5 and wt == 5:   # unpacked repeated float
                data = np.append(data if data is not None else np.zeros(0, "<f4"), np.frombuffer(v.to_bytes(4, "little"), "<f4"))
            elif num == 7 and wt == 2:
                for n2, w2, v2 in _proto.fields(v):
                    if n2 == 1 and w2 == 2:   # packed int64 dims
                        dims += _varints(v2)
                    elif n2 == 1 and w2 == 0:
                        dims.append(v2)
            elif num in (1, 2, 3, 4) and wt == 0:
                legacy[num] = v
        if data is None:
            data = np.zeros(0, "<f4")
        if not dims and legacy:
            dims = [legacy.get(k, 1) for k in (1, 2, 3, 4)]
        return np.array(data, dtype=np.float32).reshape(dims) if dims and int(np.prod(dims)) == data.size else np.array(data, dtype=np.float32)

    def _varints(mv):
        vals, v, shift = [], 0, 0
        for b in bytes(mv):
            v |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                vals.append(v)
                v, shift = 0, 0
        return vals

    for num, wt, v in _proto.fields(buf):
        if wt != 2 or num not in (100, 2):
            continue
        name_field, blob_field = (1, 7) if num == 100 else (4, 6)
        name, blobs = None, []
        for n2, w2, v2 in _proto.fields(v):
            if n2 == name_field and w2 == 2:
                name = bytes(v2).decode()
            elif n2 == blob_field and w2 == 2:
                blobs.append(blob(v2))
        if name is not None and blobs:
            out[name] = blobs
    return out


def write_caffemodel(path, weights, names=None, order=None, extra_layers=()):
    """Writes {architecture layer name: (weight (cout, cin, k, k), bias)} (PoseNet.weights) as a caffemodel file with the
    published layer names: NetParameter{name, layer{name, type, blobs{shape, data}}}.  Used by tests and the benchmark
    to hand the kernel classes a model file when only random weights exist.  names: other layer names (all_layers() order);
    order: a permutation of range(92), the order the layers are written in (readers find layers by NAME, a trained file
    also holds its layers in whatever order the training graph had); extra_layers: names of blob-less layers (ReLU,
    Pooling, Concat ...) interleaved as a real file has them."""
    from . import _proto

    def blob(arr):
        arr = np.ascontiguousarray(arr, dtype="<f4")
        return _proto.message(7, _proto.message(1, b"".join(_proto._varint(d) for d in arr.shape))) + _proto.message(5, arr.tobytes())

    with open(path, "wb") as fh:
        fh.write(_proto.message(1, b"pose"))
        entries = list(zip(all_layers(), names or caffe_layer_names()))
        extra = list(extra_layers)
        for i in (order if order is not None else range(len(entries))):
            (name, *_), cname = entries[i]
            wt, b = weights[name]
            wt = wt.numpy() if hasattr(wt, "numpy") else wt
            b = b.numpy() if hasattr(b, "numpy") else b
            fh.write(_proto.message(100, _proto.message(1, cname.encode()) + _proto.message(2, b"Convolution") +
                                    _proto.message(7, blob(wt)) + _proto.message(7, blob(b))))
            if extra:
                fh.write(_proto.message(100, _proto.message(1, extra.pop(0).encode()) + _proto.message(2, b"ReLU")))


def check_caffemodel(path):
    """Number of layers of the architecture whose weights the file holds with the right sizes (92 = usable), through the
    op library's own reader (scannertools_caffe_check_model; no GPU needed).  Raises ValueError with the reader's message."""
    from . import engine
    L = engine._caffe()
    L.scannertools_caffe_check_model.restype = ctypes.c_int
    L.scannertools_caffe_check_model.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    err = ctypes.create_string_buffer(512)
    n = L.scannertools_caffe_check_model(str(path).encode(), err, 512)
    if n < 0:
        raise ValueError(err.value.decode())
    return n


def check_prototxt(prototxt, caffemodel=None):
    """92 when the op library's own reader (scannertools_caffe_check_prototxt; no GPU needed) accepts the deploy description --
    and, with `caffemodel`, finds the weights of every layer the description names.  Raises ValueError with its message."""
    from . import engine
    L = engine._caffe()
    L.scannertools_caffe_check_prototxt.restype = ctypes.c_int
    L.scannertools_caffe_check_prototxt.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    err = ctypes.create_string_buffer(512)
    n = L.scannertools_caffe_check_prototxt(str(prototxt).encode(), str(caffemodel).encode() if caffemodel else None, err, 512)
    if n < 0:
        raise ValueError(err.value.decode())
    return n


def _pad16(c):
    return (c + 15) // 16 * 16


def _pad64(c):
    return (c + 63) // 64 * 64


class PoseNet:
    """Random-weight instance of the network on one GPU."""

    def __init__(self, ctx, seed=0, caffemodel=None, math="f32", prototxt=None):
        """caffemodel: path of the model's weights (pose_iter_440000.caffemodel of the COCO body model); None =
        random weights from `seed` (He initialisation).
        prototxt: the model's deploy description: checked against the architecture the kernels implement
        (names_from_prototxt) and the source of the layer names the weights are looked up by; None = the published names.
        math: "f32" (default; the float32 matrix instruction, what the reference's Caffe pass computes in) or "bf16x3"
        (opt-in: every operand split into three bf16 terms on the bf16 matrix pipe -- float32-grade accuracy at a
        multiple of the float32 matrix rate, not bit-identical to "f32")."""
        import torch
        if math not in ("f32", "bf16x3"):
            raise ValueError("math must be 'f32' or 'bf16x3'")
        self.math = math
        self.ctx, self.torch = ctx, torch
        self.device = ctx.device
        g = torch.Generator().manual_seed(seed)
        blobs = read_caffemodel(caffemodel) if caffemodel else None
        self.weights = {}   # name -> (torch weight (cout, cin, k, k), bias (cout,)) float32 on the CPU
        self.packed = {}    # name -> (w [cout_pad][k][k][cin_pad], bias [cout_pad]) on the device
        for (name, ci, co, k, _), cname in zip(all_layers(), names_from_prototxt(prototxt) if prototxt else caffe_layer_names()):
            if blobs is not None:
                if cname not in blobs or len(blobs[cname]) < 2:
                    raise ValueError("caffemodel %s has no weights for layer %s" % (caffemodel, cname))
                wt = torch.from_numpy(np.ascontiguousarray(blobs[cname][0], dtype=np.float32).reshape(-1))
                b = torch.from_numpy(np.ascontiguousarray(blobs[cname][1], dtype=np.float32).reshape(-1))
                if wt.numel() != co * ci * k * k or b.numel() != co:
                    raise ValueError("layer %s: expected %dx%dx%dx%d weights and %d biases, the file has %d and %d" % (cname, co, ci, k, k, co, wt.numel(), b.numel()))
                wt = wt.reshape(co, ci, k, k)
            else:
                wt = torch.randn((co, ci, k, k), generator=g) * float(np.sqrt(2.0 / (ci * k * k)))   # He initialisation
                b = (torch.rand((co,), generator=g) - 0.5) * 0.1
            self.weights[name] = (wt, b)
            cip = CAT_PAD if ci == CAT else _pad16(ci)
            wp = torch.zeros((_pad64(co), k, k, cip))
            wp[:co, :, :, :ci] = (wt[:, CAT_PERM] if ci == CAT else wt).permute(0, 2, 3, 1)
            bp = torch.zeros((_pad64(co),))
            bp[:co] = b
            self.packed[name] = (wp.contiguous().to(self.device), bp.to(self.device))
        self.packed3 = {}   # name -> weights as bf16 triples (st_conv_pack_weights_bf16x3), math == "bf16x3" only
        if math == "bf16x3":
            ctx._bind()
            for name, (wp, _) in self.packed.items():
                w3 = torch.empty((ctx._L.st_conv_bf16x3_packed_bytes(wp.shape[0], wp.shape[1], wp.shape[2], wp.shape[3]),), dtype=torch.uint8, device=self.device)
                ctx._check(ctx._L.st_conv_pack_weights_bf16x3_n(ctx._h, ctypes.c_void_p(wp.data_ptr()), wp.shape[0], wp.shape[1], wp.shape[2],
                                                                wp.shape[3], ctypes.c_void_p(w3.data_ptr()), w3.numel()))
                self.packed3[name] = w3
            torch.cuda.synchronize(self.device)
        self.packed_tile = {}   # name -> float32 weights in the spatial-tile kernel's operand order (3x3 / 7x7 layers with 128-channel blocks)
        if math == "f32":
            ctx._bind()
            for name, (wp, _) in self.packed.items():
                nb = ctx._L.st_conv_f32_tile_bytes(wp.shape[0], wp.shape[1], wp.shape[2], wp.shape[3])
                if nb > 0:
                    wt_ = torch.empty((nb,), dtype=torch.uint8, device=self.device)
                    ctx._check(ctx._L.st_conv_pack_weights_f32_tile(ctx._h, ctypes.c_void_p(wp.data_ptr()), wp.shape[0], wp.shape[1], wp.shape[2],
                                                                    wp.shape[3], ctypes.c_void_p(wt_.data_ptr())))
                    self.packed_tile[name] = wt_
            torch.cuda.synchronize(self.device)

    # -- plumbing -------------------------------------------------------------------------------------
    def _conv(self, name, x, cin, xoff, y, cout, yoff, k, relu):
        n, h, w, xs = x.shape
        wp, bp = self.packed[name]
        L = self.ctx._L
        self.ctx._bind()
        if self.math == "bf16x3":
            self.ctx._check(L.st_conv2d_nhwc_bf16x3(self.ctx._h, ctypes.c_void_p(x.data_ptr()), n, h, w, cin, xs, xoff,
                                                    ctypes.c_void_p(self.packed3[name].data_ptr()), ctypes.c_void_p(bp.data_ptr()), k, k, cout,
                                                    wp.shape[0], int(relu), ctypes.c_void_p(y.data_ptr()), y.shape[3], yoff))
            return
        wt_ = self.packed_tile.get(name)
        self.ctx._check(L.st_conv2d_nhwc_f32_tiled(self.ctx._h, ctypes.c_void_p(x.data_ptr()), n, h, w, cin, xs, xoff,
                                                   ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(wt_.data_ptr()) if wt_ is not None else None,
                                                   ctypes.c_void_p(bp.data_ptr()), k, k, cout,
                                                   wp.shape[0], int(relu), ctypes.c_void_p(y.data_ptr()), y.shape[3], yoff))

    def _operands(self, name, x, xoff, y, cout, yoff):
        from ._native import ConvOperands
        wp, bp = self.packed[name]
        o = ConvOperands()
        o.x, o.x_stride, o.x_offset = x.data_ptr(), x.shape[3], xoff
        if self.math == "bf16x3":
            o.w, o.w_tile = self.packed3[name].data_ptr(), None
        else:
            wt_ = self.packed_tile.get(name)
            o.w, o.w_tile = wp.data_ptr(), (wt_.data_ptr() if wt_ is not None else None)
        o.bias, o.cout = bp.data_ptr(), cout
        o.y, o.y_stride, o.y_offset = y.data_ptr(), y.shape[3], yoff
        return o

    def _conv_pair(self, a, b, cin, k, relu):
        """Two convolutions of the same geometry (the two branches of a stage) through st_conv2d_nhwc_*_pair; a, b =
        (name, x, xoff, y, cout, yoff).  The same bits as two _conv calls."""
        (na, xa, xoa, ya, coa, yoa), (nb, xb, xob, yb, cob, yob) = a, b
        n, h, w, _ = xa.shape
        cop = self.packed[na][0].shape[0]
        assert cop == self.packed[nb][0].shape[0] and xa.shape[:3] == xb.shape[:3]
        oa, ob = self._operands(na, xa, xoa, ya, coa, yoa), self._operands(nb, xb, xob, yb, cob, yob)
        L = self.ctx._L
        self.ctx._bind()
        fn = L.st_conv2d_nhwc_bf16x3_pair if self.math == "bf16x3" else L.st_conv2d_nhwc_f32_pair
        self.ctx._check(fn(self.ctx._h, n, h, w, cin, k, k, cop, int(relu), ctypes.byref(oa), ctypes.byref(ob)))

    def _pool(self, x, c):
        n, h, w, xs = x.shape
        y = self.torch.zeros((n, h // 2, w // 2, xs), dtype=self.torch.float32, device=self.device)
        self.ctx._bind()
        self.ctx._check(self.ctx._L.st_maxpool2_nhwc_f32(self.ctx._h, ctypes.c_void_p(x.data_ptr()), n, h, w, c, xs,
                                                         ctypes.c_void_p(y.data_ptr()), xs))
        return y

    def forward(self, net_input):
        """net_input: (n, 3, H, W) float32 on the device (CPM2Input's frames; H, W multiples of 8).
        Returns (n, H/8, W/8, 57) float32: heat maps 0..18, part-affinity planes 19..56."""
        final = self.forward_raw(net_input)
        return self.torch.cat([final[..., OFF_HEAT:OFF_HEAT + N_HEAT], final[..., OFF_PAF:OFF_PAF + N_PAF]], dim=3).contiguous()

    def detect(self, net_input, max_peaks=64, nms_threshold=0.05):
        """The whole CPM2 op (cpm2_kernel.cpp:46-49): network, `resize` layer (x8 bicubic up-sampling to the input
        size) and `nms` layer.  Returns (cpm2_resized_map (n, 57, H, W), cpm2_joints (n, 18, max_peaks + 1, 3)),
        the two columns CPM2Output consumes."""
        n, _, H, W = net_input.shape
        final = self.forward_raw(net_input)
        chan = [OFF_HEAT + i for i in range(N_HEAT)] + [OFF_PAF + i for i in range(N_PAF)]
        maps = self.ctx.cpm2_resize_maps(final, H, W, chan_map=chan)
        joints = self.ctx.cpm2_nms(maps, parts=N_HEAT - 1, max_peaks=max_peaks, threshold=nms_threshold)
        return maps, joints

    def forward_raw(self, net_input):
        """The network up to its last stage: the (n, H/8, W/8, 192) stage buffer [features | PAF 38 | heat maps 19 | pad]
        whose PAF / heat-map slices are the outputs of stage 6."""
        torch = self.torch
        n, c, H, W = net_input.shape
        assert c == 3 and H % 8 == 0 and W % 8 == 0 and net_input.is_cuda and net_input.dtype == torch.float32
        x = torch.empty((n, H, W, 16), dtype=torch.float32, device=self.device)
        self.ctx._bind()
        self.ctx._check(self.ctx._L.st_planar_to_nhwc_f32(self.ctx._h, ctypes.c_void_p(net_input.contiguous().data_ptr()), n, 3, H, W,
                                                          ctypes.c_void_p(x.data_ptr()), 16))
        cat = [torch.zeros((n, H // 8, W // 8, CAT_PAD), dtype=torch.float32, device=self.device) for _ in range(2)]
        for l in TRUNK:
            if l == "pool":
                x = self._pool(x, x.shape[3])
                continue
            name, ci, co, k, relu = l
            if name == "conv4_4_CPM":
                # the features go straight into BOTH stage-input buffers (stage s reads buffer (s - 1) & 1)
                for t in range(2):
                    self._conv(name, x, x.shape[3], 0, cat[t], co, OFF_FEAT, k, relu)
                break
            y = torch.empty((x.shape[0], x.shape[1], x.shape[2], _pad16(co)), dtype=torch.float32, device=self.device)
            self._conv(name, x, x.shape[3], 0, y, co, 0, k, relu)
            x = y
        h8, w8 = H // 8, W // 8
        # The two branches of a stage (PAF, heat maps) have the same layer shapes: layer i of both goes out as ONE paired call
        # (one launch where the spatial-tile kernel runs -- at a few frames per call a branch alone leaves CUs idle), each
        # branch with its own temporaries.
        tmp = [[torch.empty((n, h8, w8, 128), dtype=torch.float32, device=self.device) for _ in range(2)] for _ in range(2)]
        wide = [torch.empty((n, h8, w8, 512), dtype=torch.float32, device=self.device) for _ in range(2)]
        branches = (("L1", N_PAF, OFF_PAF), ("L2", N_HEAT, OFF_HEAT))
        for st in range(1, 7):
            src, dst = cat[(st - 1) & 1], cat[st & 1]
            layers = [branch_layers(st, nout) for _, nout, _ in branches]
            cur = [((src, N_FEAT, OFF_FEAT) if st == 1 else (src, CAT_PAD, 0)) for _ in branches]
            for i in range(len(layers[0])):
                ops = []
                for b, (br, nout, boff) in enumerate(branches):
                    ci, co, k, relu = layers[b][i]
                    if i == len(layers[b]) - 1:
                        y, yoff = dst, boff
                    elif co == 512:
                        y, yoff = wide[b], 0
                    else:
                        y, yoff = tmp[b][i & 1], 0
                    x, xc, xoff = cur[b]
                    ops.append(("stage%d_%s_%d" % (st, br, i + 1), x, xoff, y, co, yoff, xc, k, relu))
                    cur[b] = (y, y.shape[3], 0)
                (n0, x0, xo0, y0, c0, yo0, xc0, k0, r0), (n1, x1, xo1, y1, c1, yo1, xc1, k1, r1) = ops
                assert (xc0, k0, r0) == (xc1, k1, r1)
                self._conv_pair((n0, x0, xo0, y0, c0, yo0), (n1, x1, xo1, y1, c1, yo1), xc0, k0, r0)
        return cat[6 & 1]

    # -- float32 reference on the same weights (tests, bench parity) ---------------------------------------
    def reference_forward(self, net_input, device=None):
        """The same network through torch.nn.functional (float32), NCHW: (n, 57, H/8, W/8)."""
        torch, F = self.torch, self.torch.nn.functional
        dev = device or net_input.device
        x = net_input.to(dev)

        def conv(name, x, relu):
            wt, b = self.weights[name]
            y = F.conv2d(x, wt.to(dev), b.to(dev), padding=wt.shape[2] // 2)
            return F.relu(y) if relu else y

        for l in TRUNK:
            x = F.max_pool2d(x, 2) if l == "pool" else conv(l[0], x, l[4])
        feat = x
        l1 = l2 = None
        for st in range(1, 7):
            inp = feat if st == 1 else torch.cat([l1, l2, feat], dim=1)
            outs = []
            for br, nout in (("L1", N_PAF), ("L2", N_HEAT)):
                y = inp
                for i, (_, _, _, relu) in enumerate(branch_layers(st, nout)):
                    y = conv("stage%d_%s_%d" % (st, br, i + 1), y, relu)
                outs.append(y)
            l1, l2 = outs
        return torch.cat([l2, l1], dim=1)
