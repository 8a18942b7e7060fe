"""Wire-format readers for the hot path's outputs; mirrors
``/root/reference/scannertools/scannertools/types.py:23-41``."""
import numpy as np


def histograms(buf, protobufs=None):
    """One Histogram element: 3 x int32[bins], channel-major (types.py:23-27)."""
    # bufs[0] is None when element is null
    if buf is None:
        return None
    return np.split(np.frombuffer(buf, dtype=np.dtype(np.int32)), 3)


def flow(buf, height, width):
    """One OpticalFlow element: float32 (h, w, 2).  The reference's reader (types.py:36-41)
    takes the shape from a FrameInfo protobuf through an undefined ``db`` (dead code); here the
    shape is passed explicitly."""
    if buf is None:
        return None
    return np.frombuffer(buf, dtype=np.dtype(np.float32)).reshape((height, width, 2))


def flow_histograms(buf, protobufs=None):
    """One FlowHistogram element: 2 x int32[64], magnitude then angle
    (``flow_hist_reader``, scannertools/old/histograms.py:43-46)."""
    if buf is None:
        return None
    return np.split(np.frombuffer(buf, dtype=np.dtype(np.int32)), 2)
