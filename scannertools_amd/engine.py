"""In-process stand-in for the slice of the scannerpy graph API that the hot path's tests use.

The reference drives its ops through a Scanner ``Client`` (``sc.io.Input`` -> ``sc.ops.X`` ->
``sc.io.Output`` ; ``sc.run`` ; ``stream.load()``; /root/reference/scannertools/tests/
test_all.py:150-177,222-233).  Scanner itself (scanner-research/scanner) is neither vendored nor
installable here, so this module reproduces exactly that surface for in-memory frame streams:

    sc = Client()
    sc.ingest_frames('test1', frames_uint8_nhw3)                 # instead of sc.ingest_videos
    frame = sc.io.Input([NamedVideoStream(sc, 'test1')])
    hist = sc.ops.Histogram(frame=frame, device=DeviceType.GPU, batch=64)
    out = NamedStream(sc, 'hist'); sc.run(sc.io.Output(hist, [out]), PerfParams.estimate())
    next(out.load())

C++ ops (``Histogram``, ``OpticalFlow``, ``FlowHistogram``, ``Blur``, ``Resize``, ``ConvertColor``) are looked up in the kernel registry of
``libscannertools_imgproc.so`` and executed by its mini engine (scanner_shim/shim.cpp): the same
``execute()`` bodies a real Scanner worker would call.  Python ops (``ShotBoundaries``, ``DrawFlow``) are the
functions of this package.  What is deliberately absent: the database, video decode, the
master/worker runtime, scheduling.
"""
import ctypes
import os

import numpy as np

from . import _native
from . import shot_detection as _shot
from . import types as _types

_HERE = os.path.dirname(os.path.abspath(__file__))
IMGPROC_LIB = os.path.join(_HERE, "lib", "libscannertools_imgproc.so")


class DeviceType:
    CPU = 0
    GPU = 1


class CacheMode:
    Error = 0
    Ignore = 1
    Overwrite = 2


class PerfParams:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    @classmethod
    def estimate(cls, **kw):
        return cls(**kw)

    @classmethod
    def manual(cls, work_packet_size=None, io_packet_size=None, **kw):
        return cls(work_packet_size=work_packet_size, io_packet_size=io_packet_size, **kw)


_FRAME_DTYPES = {0: np.uint8, 1: np.float32, 2: np.float64}

CAFFE_LIB = os.path.join(_HERE, "lib", "libscannertools_caffe.so")

_LIBS = {}


def _load_op_library(path):
    """dlopen an op library; its static initialisers register ops and kernels (the mechanism of
    scannertools_infra/__init__.py:90-100 -> scannerpy.op.register_module).  Every library carries
    its own registry and mini engine (scanner_shim/shim.cpp)."""
    if path not in _LIBS:
        if not os.path.exists(path):
            raise RuntimeError("%s is not built (%s); run __graft_entry__.build()" % (os.path.basename(path), path))
        _native.lib()  # loads torch's HIP runtime + libscannertools_hip.so first
        L = ctypes.CDLL(path)
        vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.stshim_num_kernels.restype = ci
        L.stshim_kernel_info.argtypes = [ci, ctypes.c_char_p, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.stshim_op_info.argtypes = [ctypes.c_char_p, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci),
                                     ctypes.POINTER(ci), ci, ctypes.POINTER(ci)]
        L.stshim_kernel_create.restype = vp
        L.stshim_kernel_create.argtypes = [ctypes.c_char_p, ci, ci, ctypes.c_char_p, sz, ctypes.c_char_p, sz]
        L.stshim_kernel_destroy.argtypes = [vp]
        L.stshim_run_frames.restype = vp
        L.stshim_run_frames.argtypes = [vp, ctypes.POINTER(vp), ci, ci, ci, ci, ci, ci, ctypes.POINTER(ci), ci,
                                        ctypes.c_char_p, sz]
        L.stshim_run_columns.restype = vp
        L.stshim_run_columns.argtypes = [vp, ci, ci, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(ci),
                                         ctypes.POINTER(ci), ctypes.POINTER(ci), ci, ctypes.c_char_p, sz]
        L.stshim_outputs_count.argtypes = [vp]
        L.stshim_output_get.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(ci),
                                        ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.stshim_output_copy.argtypes = [vp, ci, vp, sz]
        L.stshim_outputs_columns.argtypes = [vp]
        L.stshim_output_get_col.argtypes = [vp, ci] + list(L.stshim_output_get.argtypes[1:])
        L.stshim_output_copy_col.argtypes = [vp, ci, ci, vp, sz]
        L.stshim_outputs_free.argtypes = [vp]
        L.stshim_live_buffers.restype = sz
        L.stshim_last_execute_seconds.restype = ctypes.c_double
        L.stshim_last_steady_seconds.restype = ctypes.c_double
        L.stshim_last_steady_seconds.argtypes = [ctypes.POINTER(ci)]
        L.stshim_profiler_intervals.restype = ci
        L.stshim_profiler_intervals.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
        L.stshim_live_buffers.argtypes = [ci]
        L.stshim_dev_pool_bytes.restype = sz
        L.stshim_dev_pool_bytes.argtypes = []
        L.stshim_dev_pool_drain.restype = sz
        L.stshim_dev_pool_drain.argtypes = [ci]
        # the library's device-buffer pool is handed back while the HIP runtime is still up (Python's atexit runs before the
        # runtime's own teardown; the library itself no longer frees anything from a static destructor)
        import atexit
        atexit.register(L.stshim_dev_pool_drain, -1)
        _LIBS[path] = L
    return _LIBS[path]


def _imgproc():
    return _load_op_library(IMGPROC_LIB)


def _caffe():
    """libscannertools_caffe.so: the pose ops (scannertools_caffe's op library)."""
    return _load_op_library(CAFFE_LIB)


def _lib_for_op(name):
    """The op library that registers `name` (imgproc first, then caffe)."""
    for get in (_imgproc, _caffe):
        L = get()
        if L.stshim_op_info(name.encode(), None, None, None, None, 0, None) == 0:
            return L
    raise RuntimeError("no loaded op library registers op %r" % name)


def registered_kernels(library="imgproc"):
    """[(op name, device type, kind, can_batch)] of an op library ("imgproc" or "caffe")."""
    L = _caffe() if library == "caffe" else _imgproc()
    out = []
    for i in range(L.stshim_num_kernels()):
        name = ctypes.create_string_buffer(64)
        dev, kind, cb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.stshim_kernel_info(i, name, 64, ctypes.byref(dev), ctypes.byref(kind), ctypes.byref(cb))
        out.append((name.value.decode(), dev.value, kind.value, bool(cb.value)))
    return out


def op_info(name):
    try:
        L = _lib_for_op(name)
    except RuntimeError:
        return None
    n_in, n_out, isf, ns = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    st = (ctypes.c_int * 16)()
    if L.stshim_op_info(name.encode(), ctypes.byref(n_in), ctypes.byref(n_out), ctypes.byref(isf), st, 16, ctypes.byref(ns)):
        return None
    return {"inputs": n_in.value, "outputs": n_out.value, "frame_output": bool(isf.value),
            "stencil": list(st[:ns.value]) or [0]}


# ---------------------------------------------------------------------------------------------
# graph nodes
# ---------------------------------------------------------------------------------------------
class _Node:
    def length(self):
        raise NotImplementedError

    def rows(self, idx):
        """elements for the given sorted list of row indices -> list"""
        raise NotImplementedError


class _InputNode(_Node):
    def __init__(self, stream):
        self.stream = stream

    def length(self):
        return len(self.stream._frames())

    def rows(self, idx):
        fr = self.stream._frames()
        return [fr[i] for i in idx]


class _SampleNode(_Node):
    """Range / Gather / Stride: a row-index mapping onto the parent (sc.streams.*)."""

    def __init__(self, parent, index_fn):
        self.parent = parent
        self._index_fn = index_fn

    def _map(self):
        return self._index_fn(self.parent.length())

    @property
    def reader(self):
        return getattr(self.parent, "reader", None)

    def length(self):
        return len(self._map())

    def rows(self, idx):
        m = self._map()
        return self.parent.rows([m[i] for i in idx]) if idx else []


class _CppOpNode(_Node):
    def __init__(self, client, name, parent, device, batch, stencil, args):
        self.client, self.name, self.parent = client, name, parent
        self.device = DeviceType.CPU if device is None else device
        self.batch = int(batch) if batch else 1
        info = op_info(name)
        self.stencil = list(stencil) if stencil is not None else info["stencil"]
        self.args = args or b""

    def length(self):
        return self.parent.length()

    def rows(self, idx):
        import torch
        if not idx:
            return []
        L = _lib_for_op(self.name)
        n_total = self.parent.length()
        err = ctypes.create_string_buffer(512)
        dev_id = self.client.device_id
        k = L.stshim_kernel_create(self.name.encode(), self.device, dev_id, self.args, len(self.args), err, 512)
        if not k:
            raise RuntimeError("cannot create kernel for op %s on device type %d: %s" % (self.name, self.device, err.value.decode()))
        out = {}
        try:
            smin, smax = min(self.stencil + [0]), max(self.stencil + [0])
            # contiguous runs of requested rows -> one engine stream each
            runs, start = [], 0
            for i in range(1, len(idx) + 1):
                if i == len(idx) or idx[i] != idx[i - 1] + 1:
                    runs.append((idx[start], idx[i - 1]))
                    start = i
            for a, b in runs:
                # input rows this run touches; windows that reach outside the stream clamp to its
                # edges, which is also what the engine does at the edges of the sub-stream
                lo = max(0, a + smin)
                hi = min(n_total - 1, b + smax)
                frames = self.parent.rows(list(range(lo, hi + 1)))
                if self.device == DeviceType.GPU:
                    frames = [f if (isinstance(f, torch.Tensor) and f.is_cuda) else
                              torch.as_tensor(np.ascontiguousarray(f)).cuda(dev_id) for f in frames]
                    ptrs = [f.data_ptr() for f in frames]
                else:
                    frames = [np.ascontiguousarray(f.cpu().numpy() if isinstance(f, torch.Tensor) else f) for f in frames]
                    ptrs = [f.ctypes.data for f in frames]
                h, w, c = frames[0].shape
                dt = frames[0].dtype
                ftype = 0 if dt in (np.uint8, torch.uint8) else 1
                tab = (ctypes.c_void_p * len(ptrs))(*ptrs)
                st = (ctypes.c_int * len(self.stencil))(*self.stencil)
                if self.device == DeviceType.GPU:
                    torch.cuda.synchronize(dev_id)  # kernel contexts run on their own streams
                res = L.stshim_run_frames(k, tab, len(ptrs), h, w, c, ftype, self.batch, st, len(self.stencil), err, 512)
                self.client.execute_seconds += L.stshim_last_execute_seconds()
                srows = ctypes.c_int()
                self.client.steady_seconds += L.stshim_last_steady_seconds(ctypes.byref(srows))
                self.client.steady_rows += srows.value
                if not res or err.value:
                    if res:
                        L.stshim_outputs_free(res)
                    raise RuntimeError("op %s failed: %s" % (self.name, err.value.decode()))
                try:
                    ncols = L.stshim_outputs_columns(res)
                    for r in range(a, b + 1):
                        # ops with several output columns (CPM2) yield a tuple per row; _CppOpColumn picks one
                        out[r] = self._fetch(L, res, r - lo) if ncols == 1 else tuple(self._fetch(L, res, r - lo, c) for c in range(ncols))
                finally:
                    L.stshim_outputs_free(res)
            # what the kernel instance recorded through its Scanner Profiler (caffe_kernel.cpp:387, cpm2_input_kernel_gpu.cpp:154)
            for key in ("cpm2_input", "caffe:net"):
                secs = ctypes.c_double()
                cnt = L.stshim_profiler_intervals(k, key.encode(), ctypes.byref(secs))
                if cnt:
                    rec = self.client.profile.setdefault(key, [0, 0.0])
                    rec[0] += cnt
                    rec[1] += secs.value
        finally:
            L.stshim_kernel_destroy(k)
        return [out[r] for r in idx]

    @staticmethod
    def _fetch(L, res, i, col=0):
        data, size = ctypes.c_void_p(), ctypes.c_size_t()
        isf, typ = ctypes.c_int(), ctypes.c_int()
        shape = (ctypes.c_int * 3)()
        if L.stshim_output_get_col(res, col, i, ctypes.byref(data), ctypes.byref(size), ctypes.byref(isf), shape, ctypes.byref(typ)):
            raise RuntimeError("missing output row %d of column %d" % (i, col))
        if isf.value:
            arr = np.empty((shape[0], shape[1], shape[2]), dtype=_FRAME_DTYPES[typ.value])
            assert arr.nbytes == size.value
            if L.stshim_output_copy_col(res, col, i, arr.ctypes.data_as(ctypes.c_void_p), size.value):
                raise RuntimeError("copying output row %d failed" % i)
            return arr
        buf = ctypes.create_string_buffer(size.value)
        if L.stshim_output_copy_col(res, col, i, buf, size.value):
            raise RuntimeError("copying output row %d failed" % i)
        return buf.raw


class _CppOpColumn(_Node):
    """One output column of a C++ op that declares several (the rows of the op are tuples)."""

    def __init__(self, node, col):
        self.node, self.col = node, col
        self._cache = (None, None)

    def length(self):
        return self.node.length()

    def rows(self, idx):
        if not idx:
            return []
        shared = self.node.__dict__.setdefault("_rows_cache", {})
        key = tuple(idx)
        if key not in shared:
            shared.clear()
            shared[key] = self.node.rows(list(idx))
        return [r[self.col] for r in shared[key]]


class _FrameInfoNode(_Node):
    """InfoFromFrame (scannertools_cpp/misc/info_from_frame_kernel.cpp:17-27): per row the raw
    FrameInfo struct {int32 shape[3]; int32 type} of the parent's frame, as bytes."""

    def __init__(self, parent):
        self.parent = parent

    def length(self):
        return self.parent.length()

    def rows(self, idx):
        import struct
        out = []
        for f in self.parent.rows(idx):
            h, w, c = f.shape
            ftype = 0 if str(f.dtype).endswith("uint8") else 1
            out.append(struct.pack("<4i", h, w, c, ftype))
        return out


class _PoseNetOp:
    """The CPM2 op (scannertools_caffe_cpp/cpm2_kernel.cpp:46-52): cpm2_input -> (cpm2_resized_map, cpm2_joints).
    In the reference a Caffe forward pass; here scannertools_amd.pose_net.PoseNet (MFMA convolution stack, `resize`
    and `nms` layers) with weights from a caffemodel file or, without one, random weights from `seed`.  Both
    outputs stay on the GPU (CUDA tensors), which is what CPM2Output's GPU kernel consumes."""

    def __init__(self, client, parent, weights, seed, batch, max_peaks, nms_threshold):
        self.client, self.parent = client, parent
        self.weights, self.seed, self.batch = weights, seed, max(1, int(batch or 1))
        self.max_peaks, self.nms_threshold = max_peaks, nms_threshold
        self._net, self._cache = None, {}

    def net(self):
        if self._net is None:
            from . import pose_net
            from .hip import HipContext
            ctx = HipContext(self.client.device_id)
            self._net = pose_net.PoseNet(ctx, seed=self.seed, caffemodel=self.weights)
        return self._net

    def compute(self, idx):
        import torch
        key = tuple(idx)
        if key not in self._cache:
            net = self.net()
            frames = self.parent.rows(list(idx))
            maps, joints = [], []
            for i in range(0, len(frames), self.batch):
                x = torch.stack([f if isinstance(f, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(f)) for f in frames[i:i + self.batch]])
                m, j = net.detect(x.to(net.device), self.max_peaks, self.nms_threshold)
                maps += list(m)
                joints += list(j)
            self._cache = {key: (maps, joints)}
        return self._cache[key]


class _PoseNetColumn(_Node):
    def __init__(self, op, which):
        self.op, self.which = op, which

    def length(self):
        return self.op.parent.length()

    def rows(self, idx):
        return list(self.op.compute(idx)[self.which]) if idx else []


class _CppMultiOpNode(_Node):
    """A C++ op with several input columns (frames and/or bytes), no stencil: CPM2Output."""

    def __init__(self, client, name, parents, device, batch, args):
        self.client, self.name, self.parents = client, name, parents
        self.device = DeviceType.CPU if device is None else device
        self.batch = int(batch) if batch else 1
        self.args = args or b""

    def length(self):
        n = {p.length() for p in self.parents}
        if len(n) != 1:
            raise RuntimeError("op %s: input columns have different lengths %s" % (self.name, sorted(n)))
        return n.pop()

    def rows(self, idx):
        import torch
        if not idx:
            return []
        L = _lib_for_op(self.name)
        err = ctypes.create_string_buffer(512)
        dev_id = self.client.device_id
        k = L.stshim_kernel_create(self.name.encode(), self.device, dev_id, self.args, len(self.args), err, 512)
        if not k:
            raise RuntimeError("cannot create kernel for op %s on device type %d: %s" % (self.name, self.device, err.value.decode()))
        try:
            n, nc = len(idx), len(self.parents)
            cols = [p.rows(idx) for p in self.parents]
            keep, ptrs, sizes = [], [], []
            is_frame, shapes, types = [], [], []
            for col in cols:
                first = col[0]
                if isinstance(first, (bytes, bytearray)):
                    is_frame.append(0); shapes += [0, 0, 0]; types.append(0)
                    for e in col:
                        b = ctypes.create_string_buffer(bytes(e), len(e))
                        keep.append(b)
                        ptrs.append(ctypes.addressof(b)); sizes.append(len(e))
                else:
                    if self.device == DeviceType.GPU:
                        fr = [f if (isinstance(f, torch.Tensor) and f.is_cuda) else torch.as_tensor(np.ascontiguousarray(f)).cuda(dev_id) for f in col]
                        p = [f.data_ptr() for f in fr]
                    else:
                        fr = [np.ascontiguousarray(f.cpu().numpy() if isinstance(f, torch.Tensor) else f) for f in col]
                        p = [f.ctypes.data for f in fr]
                    keep.append(fr)
                    sh = tuple(fr[0].shape)
                    if any(tuple(f.shape) != sh for f in fr) or len(sh) != 3:
                        raise ValueError("op %s: frames of a column must share one 3-d shape" % self.name)
                    is_frame.append(1); shapes += list(sh)
                    types.append(0 if fr[0].dtype in (np.uint8, torch.uint8) else 1)
                    ptrs += p; sizes += [0] * n
            if self.device == DeviceType.GPU:
                torch.cuda.synchronize(dev_id)
            res = L.stshim_run_columns(k, nc, n, (ctypes.c_void_p * len(ptrs))(*ptrs), (ctypes.c_size_t * len(sizes))(*sizes),
                                       (ctypes.c_int * nc)(*is_frame), (ctypes.c_int * (3 * nc))(*shapes),
                                       (ctypes.c_int * nc)(*types), self.batch, err, 512)
            self.client.execute_seconds += L.stshim_last_execute_seconds()
            if not res or err.value:
                if res:
                    L.stshim_outputs_free(res)
                raise RuntimeError("op %s failed: %s" % (self.name, err.value.decode()))
            try:
                return [_CppOpNode._fetch(L, res, i) for i in range(n)]
            finally:
                L.stshim_outputs_free(res)
        finally:
            L.stshim_kernel_destroy(k)


class _PyOpNode(_Node):
    """A batched python op over one input column (ShotBoundaries: batch = whole stream)."""

    def __init__(self, fn, parent, deserialize):
        self.fn, self.parent, self.deserialize = fn, parent, deserialize

    def length(self):
        return self.parent.length()

    def rows(self, idx):
        n = self.parent.length()
        if n == 0:
            return []   # an op over an empty stream is never called
        elements = [self.deserialize(e) for e in self.parent.rows(list(range(n)))]
        res = self.fn(None, elements)
        if len(res) != n:
            raise RuntimeError("python op returned %d rows for %d inputs" % (len(res), n))
        return [res[i] for i in idx]


class _PyMapNode(_Node):
    """A per-row python op over several input columns of equal length (DrawFlow)."""

    def __init__(self, fn, parents):
        self.fn, self.parents = fn, parents

    def length(self):
        n = {p.length() for p in self.parents}
        if len(n) != 1:
            raise RuntimeError("python op inputs have different lengths: %s" % sorted(n))
        return n.pop()

    def rows(self, idx):
        if not idx:
            return []
        return self.fn(*[p.rows(idx) for p in self.parents])


class _OutputNode:
    def __init__(self, node, streams):
        self.node, self.streams = node, streams


# ---------------------------------------------------------------------------------------------
# streams
# ---------------------------------------------------------------------------------------------
class NamedVideoStream:
    def __init__(self, sc, name):
        self.sc, self.name = sc, name

    def _frames(self):
        if self.name not in self.sc._videos:
            raise KeyError("no video stream named %r (use Client.ingest_frames)" % self.name)
        return self.sc._videos[self.name]

    def len(self):
        return len(self._frames())

    def load(self, rows=None, ty=None):
        """Frames of the stream: an ingested video, or frames an op wrote to it as an output
        stream (scannertools/tests/test_all.py:185-190 writes Blur's output to one)."""
        if self.name in self.sc._tables and self.name not in self.sc._videos:
            data, _ = self.sc._tables[self.name]
        else:
            data = self._frames()
        sel = range(len(data)) if rows is None else rows
        for i in sel:
            yield data[i]


class NamedStream:
    def __init__(self, sc, name):
        self.sc, self.name = sc, name

    def _rows(self):
        if self.name not in self.sc._tables:
            raise KeyError("stream %r has not been written (run the graph first)" % self.name)
        return self.sc._tables[self.name]

    def len(self):
        return len(self._rows()[0])

    def load(self, rows=None, ty=None):
        data, reader = self._rows()
        sel = range(len(data)) if rows is None else rows
        for i in sel:
            e = data[i]
            yield reader(e) if reader and isinstance(e, (bytes, type(None))) else e


# ---------------------------------------------------------------------------------------------
# client
# ---------------------------------------------------------------------------------------------
class _IO:
    def __init__(self, sc):
        self.sc = sc

    def Input(self, streams):
        if len(streams) != 1:
            raise ValueError("one stream per Input in this stand-in")
        return _InputNode(streams[0])

    def Output(self, node, streams):
        return _OutputNode(node, streams)


class _Streams:
    def Range(self, node, ranges=None, **kw):
        ranges = ranges if ranges is not None else kw.get("ranges")
        r = ranges[0]

        def fn(n):
            return list(range(r["start"], min(r["end"], n)))
        return _SampleNode(node, fn)

    def Gather(self, node, indices):
        rows = list(indices[0])
        return _SampleNode(node, lambda n: rows)

    def Stride(self, node, strides):
        s = strides[0] if isinstance(strides, (list, tuple)) else strides
        s = s["stride"] if isinstance(s, dict) else s
        return _SampleNode(node, lambda n: list(range(0, n, s)))


class _Ops:
    def __init__(self, sc):
        self.sc = sc

    def Histogram(self, frame, device=None, batch=None, bins=None):
        """sc.ops.Histogram(frame=..., device=..., batch=...) (tests/test_all.py:154,
        old/histograms.py:13-14).  ``bins`` is this build's optional extension (default 16)."""
        from . import _proto
        args = _proto.encode([(1, "int32", int(bins))]) if bins is not None else b""   # HistogramArgs
        node = _CppOpNode(self.sc, "Histogram", frame, device, batch, None, args)
        node.reader = _types.histograms
        return node

    def OpticalFlow(self, frame, stencil=None, device=None, batch=None):
        """sc.ops.OpticalFlow(frame=..., stencil=[-1,0], device=...) (tests/test_all.py:166)."""
        return _CppOpNode(self.sc, "OpticalFlow", frame, device, batch, stencil, b"")

    def Blur(self, frame, kernel_size, sigma=0.0, device=None, batch=None):
        """sc.ops.Blur(frame=..., kernel_size=3, sigma=0.1) (tests/test_all.py:184); arguments travel
        as a serialised BlurArgs message, as Scanner passes them."""
        from . import _proto
        args = _proto.encode([(1, "int32", int(kernel_size)), (2, "float", float(sigma))])
        return _CppOpNode(self.sc, "Blur", frame, device, batch, None, args)

    def Resize(self, frame, width=0, height=0, min=False, preserve_aspect=False, interpolation="", device=None,
               batch=None):
        """db.ops.Resize(frame=..., width=426, height=240, device=...) (old/histograms.py:64-68); the
        keyword arguments are the fields of the per-stream ResizeArgs message."""
        from . import _proto
        args = _proto.encode([(1, "int32", int(width)), (2, "int32", int(height)), (3, "bool", bool(min)),
                              (4, "bool", bool(preserve_aspect)), (5, "string", interpolation)])
        return _CppOpNode(self.sc, "Resize", frame, device, batch, None, args)

    def ConvertColor(self, frame, conversion, device=None, batch=None):
        """sc.ops.ConvertColor(frame=..., conversion='COLOR_RGB2GRAY', device=...): the keyword is the
        field of the per-stream ConvertColorArgs message (convert_color_kernel.cpp:224-241)."""
        from . import _proto
        return _CppOpNode(self.sc, "ConvertColor", frame, device, batch, None, _proto.encode([(1, "string", conversion)]))

    def FlowHistogram(self, flow, device=None, batch=None):
        """db.ops.FlowHistogram(flow=flow, device=DeviceType.CPU) (old/histograms.py:74-77)."""
        node = _CppOpNode(self.sc, "FlowHistogram", flow, device, batch, None, b"")
        node.reader = _types.flow_histograms
        return node

    def DrawFlow(self, frame, flow):
        """sc.ops.DrawFlow(frame=frame, flow=flow): the python op of scannertools/vis.py:8-12,
        computed on the GPU (scannertools_amd.vis.draw_flow_rows)."""
        from . import vis as _vis
        dev = self.sc.device_id
        return _PyMapNode(lambda frames, flows: _vis.draw_flow_rows(frames, flows, device=dev), [frame, flow])

    def InfoFromFrame(self, frame):
        """sc.ops.InfoFromFrame(frame=frame): the frame's FrameInfo as a bytes column."""
        return _FrameInfoNode(frame)

    def CPM2Input(self, frame, scale, device=None, batch=None):
        """sc.ops.CPM2Input(frame=..., args CPM2Args{scale}) (cpm2_input_kernel_gpu.cpp:184)."""
        from . import _proto
        return _CppOpNode(self.sc, "CPM2Input", frame, device, batch, None, _proto.encode([(2, "float", float(scale))]))

    def CPM2(self, cpm2_input, weights=None, seed=0, batch=8, max_peaks=64, nms_threshold=0.05, device=None, prototxt=None):
        """sc.ops.CPM2(cpm2_input=...) (cpm2_kernel.cpp:46-52): returns the columns (cpm2_resized_map, cpm2_joints).
        `weights`: path of the model's caffemodel -> the registered C++ kernel class (CPM2KernelHIP, the drop-in; args
        CPM2Args{caffe_args{net_descriptor{model_weights_path}}}); None -> the same layer sequence driven from Python
        (PoseNet) with random weights from `seed`, for exercising the architecture without a model file."""
        if weights is not None:
            from . import _proto
            # CaffeArgs.net_descriptor: model_path = 1 (the deploy prototxt, optional here), model_weights_path = 2
            nd = ([(1, "string", str(prototxt))] if prototxt else []) + [(2, "string", str(weights))]
            args = _proto.message(1, _proto.message(1, _proto.encode(nd)))
            node = _CppOpNode(self.sc, "CPM2", cpm2_input, device, batch, None, args)
            return _CppOpColumn(node, 0), _CppOpColumn(node, 1)
        op = _PoseNetOp(self.sc, cpm2_input, weights, seed, batch, max_peaks, nms_threshold)
        return _PoseNetColumn(op, 0), _PoseNetColumn(op, 1)

    def OpenPose(self, frame, model_directory="", pose_num_scales=1, pose_scale_gap=0.3, compute_hands=False, hand_num_scales=1,
                 hand_scale_gap=0.4, compute_face=False, device=None, batch=None):
        """sc.ops.OpenPose(frame=..., pose_num_scales=..., pose_scale_gap=..., compute_hands=..., compute_face=..., device=...,
        batch=...) (scannertools_caffe/tests/test_all.py:12-22; OpenPoseArgs, scannertools_caffe.proto:50-61); rows read with
        scannertools_amd.pose_detection.pose_list."""
        from . import _proto, pose_detection
        args = _proto.encode([(1, "string", model_directory), (2, "int32", pose_num_scales), (3, "float", pose_scale_gap),
                              (4, "bool", compute_hands), (5, "int32", hand_num_scales), (6, "float", hand_scale_gap),
                              (7, "bool", compute_face)])
        node = _CppOpNode(self.sc, "OpenPose", frame, device, batch, None, args)
        node.reader = pose_detection.pose_list
        return node

    def CPM2Output(self, cpm2_resized_map, cpm2_joints, original_frame_info, scale, device=None, batch=None):
        """sc.ops.CPM2Output(cpm2_resized_map=..., cpm2_joints=..., original_frame_info=...)
        (cpm2_output_kernel_cpu.cpp:805-810); rows read with scannertools_amd.types.poses."""
        from . import _proto
        node = _CppMultiOpNode(self.sc, "CPM2Output", [cpm2_resized_map, cpm2_joints, original_frame_info], device, batch,
                               _proto.encode([(2, "float", float(scale))]))
        node.reader = _types.poses
        return node

    def ShotBoundaries(self, histograms, device=DeviceType.CPU):
        """sc.ops.ShotBoundaries(histograms=hist) (tests/test_all.py:227): the reference's host python op.
        device=DeviceType.GPU: the same decisions computed on the GPU (st_shot_boundaries) from the histogram rows
        uploaded in one piece -- an explicit choice, never a fallback in either direction."""
        if device == DeviceType.GPU:
            sc = self.sc

            def on_device(config, elements):
                import torch
                from .hip import HipContext
                if not elements:
                    return _shot.shot_boundaries(config, elements)   # empty stream: no device work, the host op's answer
                h = torch.from_numpy(np.ascontiguousarray(np.stack([np.stack(e) for e in elements]).astype(np.int32))).to("cuda:%d" % sc.device_id)
                with HipContext(sc.device_id) as ctx:
                    return _shot.shot_boundaries_device(ctx, h)
            return _PyOpNode(on_device, histograms, _types.histograms)
        return _PyOpNode(_shot.shot_boundaries, histograms, _types.histograms)


class Client:
    """Minimal ``scannerpy.Client`` look-alike for this path."""

    def __init__(self, device_id=0, **_ignored):
        self.device_id = device_id
        self._videos, self._tables = {}, {}
        self.execute_seconds = 0.0  # wall time spent inside kernel execute() calls (excludes Python copies)
        # the same without each run's first execute() call (scratch allocation of a fresh kernel instance) + rows covered
        self.steady_seconds, self.steady_rows = 0.0, 0
        self.profile = {}   # Scanner Profiler intervals the kernel instances recorded: key -> [count, seconds]
        self.io, self.ops, self.streams = _IO(self), _Ops(self), _Streams()

    def ingest_frames(self, name, frames):
        """Register decoded RGB frames (n,h,w,3) uint8 -- numpy or a CUDA torch tensor -- as a
        named video stream (stands in for ingest_videos + the H.264 decoder)."""
        self._videos[name] = frames

    def run(self, outputs, perf_params=None, cache_mode=CacheMode.Error, show_progress=False, **_kw):
        outputs = outputs if isinstance(outputs, (list, tuple)) else [outputs]
        for o in outputs:
            for s in o.streams:
                if s.name in self._tables and cache_mode == CacheMode.Error:
                    raise RuntimeError("stream %r exists (CacheMode.Error)" % s.name)
        for o in outputs:
            n = o.node.length()
            rows = o.node.rows(list(range(n)))
            reader = getattr(o.node, "reader", None)
            for s in o.streams:
                if s.name in self._tables and cache_mode == CacheMode.Ignore:
                    continue
                self._tables[s.name] = (rows, reader)

    def live_device_buffers(self):
        return _imgproc().stshim_live_buffers(DeviceType.GPU)

    def device_pool_bytes(self):
        """Bytes idle in the engine's device-buffer pool (scanner_shim/shim.cpp)."""
        return _imgproc().stshim_dev_pool_bytes()

    def drain_device_pool(self, device=-1):
        """Returns the pool's idle blocks to the driver (all devices by default); bytes released."""
        return _imgproc().stshim_dev_pool_drain(device)
