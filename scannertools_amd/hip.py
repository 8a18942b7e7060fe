"""Device-side ops: thin Python over the C ABI (include/scannertools_hip.h).

PyTorch is used only for device memory and streams: inputs/outputs are ``torch`` CUDA tensors
whose ``data_ptr()`` is handed to the library, and kernels are enqueued on torch's current
stream of the context's device.  All arithmetic happens in the HIP kernels under ``csrc/``.
"""
import ctypes

import numpy as np
import torch

from . import _native
from ._native import FbParams, StError, default_params  # noqa: F401


def _require_cuda(t, dtype, name, device=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA tensor (no CPU fallback exists)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if device is not None and t.device != device:
        raise ValueError("%s lives on %s but the context runs on %s" % (name, t.device, device))


def _check_out(out, shape, dtype, device):
    """A caller-supplied output buffer is written through its raw pointer: it must be exactly the
    dense device array the kernel will produce."""
    _require_cuda(out, dtype, "out", device)
    if tuple(out.shape) != tuple(shape):
        raise ValueError("out must have shape %s, got %s" % (tuple(shape), tuple(out.shape)))
    return out


class HipContext:
    """One ``st_ctx``: per-kernel-instance state (stream binding, scratch).  Mirrors the role of
    a Scanner kernel instance's private GPU state (optical_flow_kernel_gpu.cpp:101-107)."""

    def __init__(self, device=0, workspace_limit=None):
        self._L = _native.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("HipContext needs a GPU: torch.cuda.is_available() is False")
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        h = ctypes.c_void_p()
        st = self._L.st_ctx_create(self.device.index, ctypes.byref(h))
        if st != 0:
            raise StError(st, "st_ctx_create(%d) failed" % self.device.index)
        self._h = h
        self._bound_stream = -1
        if workspace_limit:
            self._check(self._L.st_ctx_set_workspace_limit(self._h, int(workspace_limit)))

    # -- plumbing ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.st_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != 0:
            raise StError(st, (self._L.st_ctx_last_error(self._h) or b"").decode() or
                          self._L.st_status_string(st).decode())

    def _bind(self):
        """Enqueue on torch's current stream so that torch ops and events order with ours."""
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._bound_stream:
            self._check(self._L.st_ctx_set_stream(self._h, ctypes.c_void_p(s)))
            self._bound_stream = s

    def sync(self):
        self._check(self._L.st_ctx_sync(self._h))

    def flow_concurrent(self):
        """True if the last optical_flow call chose its kernels for a GPU shared with other kernel instances (diagnostic)."""
        return bool(self._L.st_ctx_flow_concurrent(self._h))

    def release_workspace(self):
        self._check(self._L.st_ctx_release_workspace(self._h))

    def timing_enable(self, kernel_ids):
        mask = 0
        for k in kernel_ids:
            mask |= 1 << k
        self._check(self._L.st_ctx_timing_enable(self._h, mask))

    def timing_reset(self):
        self._check(self._L.st_ctx_timing_reset(self._h))

    def timing_read(self, kernel_id):
        n, ms = ctypes.c_int(), ctypes.c_double()
        self._check(self._L.st_ctx_timing_read(self._h, kernel_id, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    # -- Histogram --------------------------------------------------------------------------
    def histogram(self, frames, bins=16, out=None):
        """Per-channel histograms of U8 RGB frames.

        frames: CUDA uint8 tensor (n,h,w,3) (one contiguous stream) or a list of (h,w,3)
        tensors (one buffer per Scanner element).  Returns int32 (n,3,bins): row i is the
        192-byte element the reference's Histogram op emits for bins=16
        (histogram_kernel_cpu.cpp:20,40-44)."""
        self._bind()
        if isinstance(frames, (list, tuple)):
            n = len(frames)
            if n == 0:
                return torch.zeros((0, 3, bins), dtype=torch.int32, device=self.device)
            for f in frames:
                _require_cuda(f, torch.uint8, "frame", self.device)
            h, w, c = frames[0].shape
            if any(tuple(f.shape) != (h, w, 3) for f in frames):
                raise ValueError("all frames must be (h,w,3) with equal shape")
            out = (torch.empty((n, 3, bins), dtype=torch.int32, device=self.device) if out is None
                   else _check_out(out, (n, 3, bins), torch.int32, self.device))
            table = (ctypes.c_void_p * n)(*[f.data_ptr() for f in frames])
            self._check(self._L.st_hist_u8c3_batch(self._h, table, n, h, w, bins, ctypes.c_void_p(out.data_ptr())))
            return out
        _require_cuda(frames, torch.uint8, "frames", self.device)
        if frames.dim() != 4 or frames.shape[3] != 3:
            raise ValueError("frames must be (n,h,w,3)")
        n, h, w, _ = frames.shape
        out = (torch.empty((n, 3, bins), dtype=torch.int32, device=self.device) if out is None
               else _check_out(out, (n, 3, bins), torch.int32, self.device))
        if n == 0:
            return out
        self._check(self._L.st_hist_u8c3_strided(self._h, ctypes.c_void_p(frames.data_ptr()), 3 * h * w, n, h, w,
                                                 bins, ctypes.c_void_p(out.data_ptr())))
        return out

    # -- flow consumers ---------------------------------------------------------------------
    def shot_boundaries(self, hist, window=500, k_std=2.5, return_diffs=False):
        """ShotBoundaries on device-resident histograms (shot_detection.py:12-28): hist = CUDA int32 (n, 3, bins) as
        `histogram` returns them.  Returns the list of boundary frame indices (the reference op's row 0), bit for bit
        the host op's; with return_diffs also the float64 distances diffs[i]."""
        self._bind()
        _require_cuda(hist, torch.int32, "hist", self.device)
        if hist.dim() != 3 or hist.shape[1] != 3:
            raise ValueError("hist must be (n, 3, bins)")
        hist = hist.contiguous()
        n, _, bins = hist.shape
        flags = torch.empty((n,), dtype=torch.uint8, device=self.device)
        diffs = torch.empty((n,), dtype=torch.float64, device=self.device) if return_diffs else None
        self._check(self._L.st_shot_boundaries(self._h, ctypes.c_void_p(hist.data_ptr()), n, bins, int(window), float(k_std),
                                               ctypes.c_void_p(flags.data_ptr()), ctypes.c_void_p(diffs.data_ptr()) if return_diffs else None))
        self.sync()
        idx = [int(i) for i in torch.nonzero(flags).flatten().cpu()]
        return (idx, diffs) if return_diffs else idx

    def flow_histogram(self, flows, out=None):
        """FlowHistogram (old/cpp_ops/flow_histogram_kernel_cpu.cpp:26-57): magnitude and angle
        histograms (64 bins on [0,64) px and [0,360) degrees) of flow frames.

        flows: CUDA float32 tensor (n,h,w,2) or a list of (h,w,2) tensors.  Returns int32
        (n,2,64): row i is the 512-byte element the reference op emits."""
        self._bind()
        if isinstance(flows, (list, tuple)):
            n = len(flows)
            if n == 0:
                return torch.zeros((0, 2, 64), dtype=torch.int32, device=self.device)
            for f in flows:
                _require_cuda(f, torch.float32, "flow", self.device)
            h, w, _ = flows[0].shape
            if any(tuple(f.shape) != (h, w, 2) for f in flows):
                raise ValueError("all flows must be (h,w,2) with equal shape")
            out = (torch.empty((n, 2, 64), dtype=torch.int32, device=self.device) if out is None
                   else _check_out(out, (n, 2, 64), torch.int32, self.device))
            table = (ctypes.c_void_p * n)(*[f.data_ptr() for f in flows])
            self._check(self._L.st_flow_hist_batch(self._h, table, n, h, w, ctypes.c_void_p(out.data_ptr())))
            return out
        _require_cuda(flows, torch.float32, "flows", self.device)
        if flows.dim() != 4 or flows.shape[3] != 2:
            raise ValueError("flows must be (n,h,w,2)")
        n, h, w, _ = flows.shape
        out = (torch.empty((n, 2, 64), dtype=torch.int32, device=self.device) if out is None
               else _check_out(out, (n, 2, 64), torch.int32, self.device))
        if n == 0:
            return out
        self._check(self._L.st_flow_hist_strided(self._h, ctypes.c_void_p(flows.data_ptr()), 8 * h * w, n, h, w,
                                                 ctypes.c_void_p(out.data_ptr())))
        return out

    def draw_flow(self, frames, flows, out=None):
        """DrawFlow (scannertools/vis.py:8-12): frames (n,h,w,3) uint8 and flows (n,h,w,2) float32
        (tensors or lists of per-row tensors) -> (n,h,2w,3) uint8, the frame beside its flow map."""
        self._bind()
        fr = list(frames) if isinstance(frames, (list, tuple)) else list(frames.unbind(0))
        fl = list(flows) if isinstance(flows, (list, tuple)) else list(flows.unbind(0))
        n = len(fr)
        if len(fl) != n:
            raise ValueError("frames and flows must have the same number of rows")
        if n == 0:
            return torch.zeros((0, 0, 0, 3), dtype=torch.uint8, device=self.device)
        h, w, _ = fr[0].shape
        for f, g in zip(fr, fl):
            _require_cuda(f, torch.uint8, "frame", self.device)
            _require_cuda(g, torch.float32, "flow", self.device)
            if tuple(f.shape) != (h, w, 3) or tuple(g.shape) != (h, w, 2):
                raise ValueError("frames must be (h,w,3) and flows (h,w,2) with equal h,w")
        out = (torch.empty((n, h, 2 * w, 3), dtype=torch.uint8, device=self.device) if out is None
               else _check_out(out, (n, h, 2 * w, 3), torch.uint8, self.device))
        tf = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fr])
        tg = (ctypes.c_void_p * n)(*[g.data_ptr() for g in fl])
        to = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        self._check(self._L.st_draw_flow_batch(self._h, tf, tg, n, h, w, to))
        return out

    # -- sibling imgproc ops ----------------------------------------------------------------
    def box_blur(self, frames, kernel_size, out=None):
        """Blur op (blur_kernel_cpu.cpp:50-81): k x k integer box filter of (n,h,w,3) uint8 frames
        (a tensor or a list of (h,w,3) tensors); border pixels are 0."""
        self._bind()
        fr = list(frames) if isinstance(frames, (list, tuple)) else list(frames.unbind(0))
        n = len(fr)
        if n == 0:
            return torch.zeros((0, 0, 0, 3), dtype=torch.uint8, device=self.device)
        h, w, _ = fr[0].shape
        for f in fr:
            _require_cuda(f, torch.uint8, "frame", self.device)
            if tuple(f.shape) != (h, w, 3):
                raise ValueError("all frames must be (h,w,3) with equal shape")
        out = (torch.empty((n, h, w, 3), dtype=torch.uint8, device=self.device) if out is None
               else _check_out(out, (n, h, w, 3), torch.uint8, self.device))
        tf = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fr])
        to = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        self._check(self._L.st_box_blur_u8c3_batch(self._h, tf, n, h, w, int(kernel_size), to))
        return out

    def resize(self, frames, width, height, interpolation=_native.INTER_LINEAR, out=None):
        """Resize op (resize_kernel.cpp:68-73): (n,h,w,c) uint8 frames (a tensor or a list of (h,w,c)
        tensors) -> (n,height,width,c) uint8, cv::resize semantics for 8-bit frames."""
        self._bind()
        fr = list(frames) if isinstance(frames, (list, tuple)) else list(frames.unbind(0))
        n = len(fr)
        if n == 0:
            return torch.zeros((0, height, width, 3), dtype=torch.uint8, device=self.device)
        h, w, c = fr[0].shape
        for f in fr:
            _require_cuda(f, torch.uint8, "frame", self.device)
            if tuple(f.shape) != (h, w, c):
                raise ValueError("all frames must have the same (h,w,c) shape")
        out = (torch.empty((n, height, width, c), dtype=torch.uint8, device=self.device) if out is None
               else _check_out(out, (n, height, width, c), torch.uint8, self.device))
        tf = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fr])
        to = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        self._check(self._L.st_resize_u8_batch(self._h, tf, n, h, w, c, int(height), int(width), int(interpolation), to))
        return out

    def cvt_color(self, frames, code, gray_bits=15, out=None):
        """ConvertColor op (convert_color_kernel.cpp:268-271): cv::cvtColor on (n,h,w,c) uint8 frames;
        ``code`` is a cv::ColorConversionCodes value or one of the names in _native.COLOR_CODES."""
        self._bind()
        if isinstance(code, str):
            if code not in _native.COLOR_CODES:
                raise ValueError("conversion %r is not implemented" % code)
            code = _native.COLOR_CODES[code]
        fr = list(frames) if isinstance(frames, (list, tuple)) else list(frames.unbind(0))
        n = len(fr)
        if n == 0:
            return torch.zeros((0, 0, 0, 3), dtype=torch.uint8, device=self.device)
        h, w, c = fr[0].shape
        for f in fr:
            _require_cuda(f, torch.uint8, "frame", self.device)
            if tuple(f.shape) != (h, w, c):
                raise ValueError("all frames must have the same (h,w,c) shape")
        oh, ow, oc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if self._L.st_cvt_color_out_shape(int(code), h, w, c, ctypes.byref(oh), ctypes.byref(ow), ctypes.byref(oc)) != 0:
            raise ValueError("conversion code %d on %dx%d frames of %d channel(s) is not implemented" % (code, w, h, c))
        shape = (n, oh.value, ow.value, oc.value)
        out = torch.empty(shape, dtype=torch.uint8, device=self.device) if out is None else _check_out(out, shape, torch.uint8, self.device)
        tf = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fr])
        to = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        self._check(self._L.st_cvt_color_u8_batch(self._h, tf, n, h, w, c, int(code), int(gray_bits), to))
        return out

    # -- pose path (scannertools_caffe) -------------------------------------------------------
    def cpm2_input(self, frames, scale, out=None):
        """CPM2Input (scannertools_caffe_cpp/cpm2_input_kernel_gpu.cpp:104-140): (n,h,w,3) uint8 RGB frames
        -> (n,3,net_h,net_w) float32 network input (planes B,G,R; bicubic resize by ``scale``, padded to
        a multiple of 8 with 128, x/256 - 0.5)."""
        self._bind()
        fr = list(frames) if isinstance(frames, (list, tuple)) else list(frames.unbind(0))
        n = len(fr)
        if n == 0:
            return torch.zeros((0, 3, 0, 0), dtype=torch.float32, device=self.device)
        h, w, _ = fr[0].shape
        for f in fr:
            _require_cuda(f, torch.uint8, "frame", self.device)
            if tuple(f.shape) != (h, w, 3):
                raise ValueError("all frames must be (h,w,3) with equal shape")
        _, _, nh, nw = cpm2_geometry(h, w, scale)
        out = (torch.empty((n, 3, nh, nw), dtype=torch.float32, device=self.device) if out is None
               else _check_out(out, (n, 3, nh, nw), torch.float32, self.device))
        tf = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fr])
        to = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        self._check(self._L.st_cpm2_input_batch(self._h, tf, n, h, w, float(scale), to))
        return out

    def cpm2_limb_scores(self, heatmaps, peaks, inter_threshold=0.05, min_above=9):
        """Candidate-pair scores of CPM2Output (cpm2_output_kernel_cpu.cpp:424-487): heatmaps
        (n,57,H,W) float32, peaks (n,18,max_peaks+1,3) float32 -> (n,19,max_peaks,max_peaks) float32."""
        self._bind()
        _require_cuda(heatmaps, torch.float32, "heatmaps", self.device)
        _require_cuda(peaks, torch.float32, "peaks", self.device)
        n, c, H, W = heatmaps.shape
        if c != 57 or peaks.dim() != 4 or peaks.shape[0] != n or peaks.shape[1] != 18 or peaks.shape[3] != 3:
            raise ValueError("heatmaps must be (n,57,H,W) and peaks (n,18,max_peaks+1,3)")
        mp = peaks.shape[2] - 1
        out = torch.empty((n, 19, mp, mp), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        th = (ctypes.c_void_p * n)(*[heatmaps[i].data_ptr() for i in range(n)])
        tp = (ctypes.c_void_p * n)(*[peaks[i].data_ptr() for i in range(n)])
        self._check(self._L.st_cpm2_limb_scores(self._h, th, tp, n, H, W, mp, float(inter_threshold), int(min_above),
                                                ctypes.c_void_p(out.data_ptr())))
        return out

    def cpm2_resize_maps(self, maps, dst_h, dst_w, chan_map=None, nmaps=None, out=None):
        """The CPM2 network's `resize` layer (cpm2_kernel.cpp:16-23 configures it; [EXT] Caffe fork): maps
        (n,h,w,C) channel-last float32 -> (n,nmaps,dst_h,dst_w) planar float32, bicubic; output plane c reads
        channel chan_map[c] (default: the first nmaps channels)."""
        self._bind()
        _require_cuda(maps, torch.float32, "maps", self.device)
        if maps.dim() != 4:
            raise ValueError("maps must be (n,h,w,C)")
        n, h, w, C = maps.shape
        if chan_map is not None:
            nmaps = len(chan_map)
        elif nmaps is None:
            nmaps = C
        shape = (n, nmaps, int(dst_h), int(dst_w))
        out = torch.empty(shape, dtype=torch.float32, device=self.device) if out is None else _check_out(out, shape, torch.float32, self.device)
        if n == 0:
            return out
        cm = (ctypes.c_int * nmaps)(*[int(v) for v in chan_map]) if chan_map is not None else None
        to = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        self._check(self._L.st_cpm2_resize_maps(self._h, ctypes.c_void_p(maps.data_ptr()), n, h, w, C, cm, nmaps, int(dst_h), int(dst_w), to))
        return out

    def cpm2_resize_merge_maps(self, maps_list, eff_sizes, dst_h, dst_w, chan_map=None, nmaps=None):
        """Maps of several network scales merged (st_cpm2_resize_merge_maps): maps_list[s] (n,h_s,w_s,C) float32 channel-last,
        eff_sizes[s] = (eff_h, eff_w) source pixels of scale s that span the whole output -> (n,nmaps,dst_h,dst_w)."""
        self._bind()
        S = len(maps_list)
        for m in maps_list:
            _require_cuda(m, torch.float32, "maps", self.device)
        n, _, _, C = maps_list[0].shape
        if any(m.dim() != 4 or m.shape[0] != n or m.shape[3] != C for m in maps_list) or len(eff_sizes) != S:
            raise ValueError("every scale needs (n,h,w,C) maps with the same n and C, and one (eff_h, eff_w) pair")
        if chan_map is not None:
            nmaps = len(chan_map)
        elif nmaps is None:
            nmaps = C
        out = torch.empty((n, nmaps, int(dst_h), int(dst_w)), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        cm = (ctypes.c_int * nmaps)(*[int(v) for v in chan_map]) if chan_map is not None else None
        src = (ctypes.c_void_p * S)(*[m.data_ptr() for m in maps_list])
        sh = (ctypes.c_int * S)(*[m.shape[1] for m in maps_list])
        sw = (ctypes.c_int * S)(*[m.shape[2] for m in maps_list])
        eh = (ctypes.c_float * S)(*[float(e[0]) for e in eff_sizes])
        ew = (ctypes.c_float * S)(*[float(e[1]) for e in eff_sizes])
        to = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        self._check(self._L.st_cpm2_resize_merge_maps(self._h, src, sh, sw, eh, ew, S, n, C, cm, nmaps, int(dst_h), int(dst_w), to))
        return out

    def cpm2_nms(self, maps, parts=18, max_peaks=64, threshold=0.05, out=None):
        """The CPM2 network's `nms` layer ([EXT] Caffe fork): maps (n,>=parts,H,W) float32 -> joints
        (n,parts,max_peaks+1,3) float32, row 0 = [count,0,0], rows 1.. = (x, y, score) in raster order."""
        self._bind()
        _require_cuda(maps, torch.float32, "maps", self.device)
        if maps.dim() != 4 or maps.shape[1] < parts:
            raise ValueError("maps must be (n,>=parts,H,W)")
        n, _, H, W = maps.shape
        shape = (n, parts, max_peaks + 1, 3)
        out = torch.empty(shape, dtype=torch.float32, device=self.device) if out is None else _check_out(out, shape, torch.float32, self.device)
        if n == 0:
            return out
        tm = (ctypes.c_void_p * n)(*[maps[i].data_ptr() for i in range(n)])
        to = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        self._check(self._L.st_cpm2_nms(self._h, tm, n, H, W, int(parts), int(max_peaks), float(threshold), to))
        return out

    # -- OpticalFlow ------------------------------------------------------------------------
    def optical_flow(self, frames, pairs=None, params=None, out=None):
        """Farneback flow for a batch of frame pairs.

        frames: CUDA uint8 (n,h,w,3) tensor or list of (h,w,3) tensors.  pairs: (p,2) int
        array-like of indices; default = consecutive pairs (i, i+1), i.e. stencil {0,1} over a
        contiguous run (optical_flow_kernel_cpu.cpp:51-54).  Flow p goes from frame pairs[p][0]
        to frame pairs[p][1].  Returns float32 (p,h,w,2)."""
        self._bind()
        if isinstance(frames, (list, tuple)):
            fl = list(frames)
        else:
            _require_cuda(frames, torch.uint8, "frames", self.device)
            if frames.dim() != 4 or frames.shape[3] != 3:
                raise ValueError("frames must be (n,h,w,3)")
            fl = [frames[i] for i in range(frames.shape[0])]
        n = len(fl)
        for f in fl:
            _require_cuda(f, torch.uint8, "frame", self.device)
        if n:
            h, w, _ = fl[0].shape
            if any(tuple(f.shape) != (h, w, 3) for f in fl):
                raise ValueError("all frames must be (h,w,3) with equal shape")
        if pairs is None:
            pairs = [(i, i + 1) for i in range(max(n - 1, 0))]
        pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
        p = len(pairs)
        if n == 0:
            if p:
                raise ValueError("pairs given but no frames")
            return torch.zeros((0, 0, 0, 2), dtype=torch.float32, device=self.device)
        out = (torch.empty((p, h, w, 2), dtype=torch.float32, device=self.device) if out is None
               else _check_out(out, (p, h, w, 2), torch.float32, self.device))
        if p == 0:
            return out
        prm = params if params is not None else default_params()
        ftab = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fl])
        otab = (ctypes.c_void_p * p)(*[out[i].data_ptr() for i in range(p)])
        self._check(self._L.st_farneback_pairs(
            self._h, ftab, n, pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), p, h, w, ctypes.byref(prm), otab))
        return out

    # -- stage-level entry points (used by the parity tests) ----------------------------------
    def gray(self, rgb, bits=15):
        self._bind()
        _require_cuda(rgb, torch.uint8, "rgb", self.device)
        h, w, _ = rgb.shape
        out = torch.empty((h, w), dtype=torch.uint8, device=self.device)
        self._check(self._L.st_gray_u8(self._h, ctypes.c_void_p(rgb.data_ptr()), h, w, bits,
                                       ctypes.c_void_p(out.data_ptr())))
        return out

    def pyr_image(self, gray, level, params=None):
        self._bind()
        _require_cuda(gray, torch.uint8, "gray", self.device)
        prm = params if params is not None else default_params()
        h, w = gray.shape
        lh, lw = ctypes.c_int(), ctypes.c_int()
        self._check(self._L.st_fb_level_geom(h, w, ctypes.byref(prm), level, ctypes.byref(lh), ctypes.byref(lw),
                                             None, None))
        out = torch.empty((lh.value, lw.value), dtype=torch.float32, device=self.device)
        self._check(self._L.st_fb_pyr_image(self._h, ctypes.c_void_p(gray.data_ptr()), h, w, ctypes.byref(prm), level,
                                            ctypes.c_void_p(out.data_ptr())))
        return out

    def polyexp(self, img, poly_n=5, poly_sigma=1.2):
        """(h,w) f32 -> R (h,w,5) f32, OpenCV's interleaved layout (unpacked from the device layout)."""
        self._bind()
        _require_cuda(img, torch.float32, "img", self.device)
        h, w = img.shape
        out = torch.empty(5 * h * w, dtype=torch.float32, device=self.device)
        self._check(self._L.st_fb_polyexp(self._h, ctypes.c_void_p(img.data_ptr()), h, w, poly_n, poly_sigma,
                                          ctypes.c_void_p(out.data_ptr())))
        return unpack_r(out, h, w)

    def update_matrices(self, r0, r1, flow=None, coarse_flow=None, pyr_scale=0.5):
        """R0,R1 (h,w,5) (+ flow (h,w,2) or coarse flow (ch,cw,2)) -> planar M (5,h,w)."""
        self._bind()
        _require_cuda(r0, torch.float32, "r0", self.device)
        _require_cuda(r1, torch.float32, "r1", self.device)
        h, w, _ = r0.shape
        r0, r1 = pack_r(r0), pack_r(r1)
        out = torch.empty((5, h, w), dtype=torch.float32, device=self.device)
        fp = cp = None
        ch = cw = 0
        if coarse_flow is not None:
            _require_cuda(coarse_flow, torch.float32, "coarse_flow", self.device)
            ch, cw, _ = coarse_flow.shape
            cp = ctypes.c_void_p(coarse_flow.data_ptr())
        elif flow is not None:
            _require_cuda(flow, torch.float32, "flow", self.device)
            fp = ctypes.c_void_p(flow.data_ptr())
        self._check(self._L.st_fb_update_matrices(self._h, ctypes.c_void_p(r0.data_ptr()), ctypes.c_void_p(r1.data_ptr()),
                                                  fp, cp, ch, cw, pyr_scale, h, w, ctypes.c_void_p(out.data_ptr())))
        return out

    def update_flow_blur(self, r0, r1, m, block_size=15, update=True):
        """One FarnebackUpdateFlow_Blur pass over planar M (5,h,w) (R0, R1: (h,w,5), needed when
        update=True).  Returns (flow (h,w,2), M' (5,h,w) or None)."""
        self._bind()
        _require_cuda(m, torch.float32, "m", self.device)
        _, h, w = m.shape
        r0 = pack_r(r0) if r0 is not None else None
        r1 = pack_r(r1) if r1 is not None else None
        flow = torch.empty((h, w, 2), dtype=torch.float32, device=self.device)
        mout = torch.empty_like(m) if update else None
        self._check(self._L.st_fb_update_flow_blur(
            self._h, ctypes.c_void_p(r0.data_ptr()) if r0 is not None else None,
            ctypes.c_void_p(r1.data_ptr()) if r1 is not None else None, ctypes.c_void_p(m.data_ptr()), h, w,
            block_size, 1 if update else 0, ctypes.c_void_p(flow.data_ptr()),
            ctypes.c_void_p(mout.data_ptr()) if update else None))
        return flow, mout


    def flow_iteration(self, r0, r1, flow_in=None, coarse_flow=None, pyr_scale=0.5, block_size=15):
        """One fused iteration (UpdateMatrices + box blur + solve) as the production path runs it."""
        self._bind()
        _require_cuda(r0, torch.float32, "r0", self.device)
        _require_cuda(r1, torch.float32, "r1", self.device)
        h, w, _ = r0.shape
        r0, r1 = pack_r(r0), pack_r(r1)
        out = torch.empty((h, w, 2), dtype=torch.float32, device=self.device)
        fp = cp = None
        ch = cw = 0
        if coarse_flow is not None:
            _require_cuda(coarse_flow, torch.float32, "coarse_flow", self.device)
            ch, cw, _ = coarse_flow.shape
            cp = ctypes.c_void_p(coarse_flow.data_ptr())
        elif flow_in is not None:
            _require_cuda(flow_in, torch.float32, "flow_in", self.device)
            fp = ctypes.c_void_p(flow_in.data_ptr())
        self._check(self._L.st_fb_flow_iteration(self._h, ctypes.c_void_p(r0.data_ptr()), ctypes.c_void_p(r1.data_ptr()),
                                                 fp, cp, ch, cw, pyr_scale, h, w, block_size,
                                                 ctypes.c_void_p(out.data_ptr())))
        return out


def pack_r(r):
    """(h,w,5) polynomial expansion -> the device layout of the st_fb_* stage entry points:
    h*w float4 (channels 0..3) followed by h*w floats (channel 4)."""
    return torch.cat([r[..., :4].reshape(-1), r[..., 4].reshape(-1)]).contiguous()


def unpack_r(flat, h, w):
    """Inverse of :func:`pack_r`."""
    n = h * w
    return torch.cat([flat[:4 * n].view(h, w, 4), flat[4 * n:].view(h, w, 1)], dim=2).contiguous()


def cpm2_geometry(h, w, scale):
    """(resize_h, resize_w, net_h, net_w) of the CPM2 network input for an (h, w) frame at ``scale``."""
    v = [ctypes.c_int() for _ in range(4)]
    st = _native.lib().st_cpm2_geometry(h, w, float(scale), *[ctypes.byref(x) for x in v])
    if st != 0:
        raise StError(st, "st_cpm2_geometry(%d, %d, %r)" % (h, w, scale))
    return tuple(x.value for x in v)


def cpm2_scale_for_height(h, target_h):
    """The float scale at which `cpm2_geometry` resizes a frame of height h to exactly target_h rows (what the OpenPose
    op uses for its fixed network input height: 368.f / 1080 truncates to 367)."""
    v = ctypes.c_float()
    st = _native.lib().st_cpm2_scale_for_height(int(h), int(target_h), ctypes.byref(v))
    if st != 0:
        raise StError(st, "st_cpm2_scale_for_height(%d, %d)" % (h, target_h))
    return float(v.value)


def fb_levels(h, w, params=None):
    prm = params if params is not None else default_params()
    return _native.lib().st_fb_levels(h, w, ctypes.byref(prm))


def fb_level_geom(h, w, level, params=None):
    prm = params if params is not None else default_params()
    lh, lw, ks = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    sg = ctypes.c_double()
    st = _native.lib().st_fb_level_geom(h, w, ctypes.byref(prm), level, ctypes.byref(lh), ctypes.byref(lw),
                                        ctypes.byref(sg), ctypes.byref(ks))
    if st != 0:
        raise StError(st, "st_fb_level_geom")
    return lh.value, lw.value, sg.value, ks.value
