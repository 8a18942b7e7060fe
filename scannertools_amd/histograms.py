"""Front-ends of the histogram pipelines: mirrors of ``compute_histograms``,
``compute_hsv_histograms`` and ``compute_flow_histograms``
(/root/reference/scannertools/scannertools/old/histograms.py:6-81), i.e. of their ``build_pipeline``
graphs; the job machinery around them (Database, sinks, megabatches) is out of scope (SURVEY
section 8).  Each function returns one NamedStream per video.
"""
from . import types as _types
from .engine import CacheMode, DeviceType, NamedStream, NamedVideoStream, PerfParams


def _run(sc, name, suffix, build):
    frame = sc.io.Input([NamedVideoStream(sc, name)])
    out = NamedStream(sc, '%s_%s' % (name, suffix))
    sc.run(sc.io.Output(build(frame), [out]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    return out


def compute_histograms(sc, videos, device=DeviceType.GPU, batch=1):
    """old/histograms.py:10-15: Histogram(frame, device, batch), ``batch=1`` by default as there; rows read with
    scannertools_amd.types.histograms (3 x int32[16]).  ``batch=64`` and up is what fills the GPU from one instance."""
    return [_run(sc, v, 'hist', lambda f: sc.ops.Histogram(frame=f, device=device, batch=batch)) for v in videos]


def compute_hsv_histograms(sc, videos, device=DeviceType.GPU, batch=1):
    """old/histograms.py:32-37: RGB -> HSV conversion, then Histogram.  The reference's
    ConvertToHSVCPP op is cv::cvtColor(COLOR_RGB2HSV) (old/cpp_ops/imgproc.cpp:41): one pass of the
    ConvertColor op of this library with the same code."""
    def build(f):
        hsv = sc.ops.ConvertColor(frame=f, conversion='COLOR_RGB2HSV', device=device, batch=batch)
        return sc.ops.Histogram(frame=hsv, device=device, batch=batch)
    return [_run(sc, v, 'hsv_hist', build) for v in videos]


def compute_flow_histograms(sc, videos, device=DeviceType.GPU, batch=None, width=426, height=240, hist_device=None):
    """old/histograms.py:63-78: Resize(426 x 240) -> OpticalFlow -> FlowHistogram; rows read with
    scannertools_amd.types.flow_histograms (2 x int32[64]).  No op gets a ``batch=`` by default, as there; the reference
    runs FlowHistogram on ``DeviceType.CPU`` (:76-78) -- pass ``hist_device=DeviceType.CPU`` for that placement (the
    flow field then crosses PCIe); the default keeps it on the device the flow was computed on."""
    kw = {} if batch is None else {'batch': batch}

    def build(f):
        small = sc.ops.Resize(frame=f, device=device, width=width, height=height, **kw)
        flow = sc.ops.OpticalFlow(frame=small, device=device, **kw)
        return sc.ops.FlowHistogram(flow=flow, device=hist_device if hist_device is not None else device, **kw)
    return [_run(sc, v, 'flow_hist', build) for v in videos]


flow_hist_reader = _types.flow_histograms  # old/histograms.py:43-46
