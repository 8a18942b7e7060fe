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


def poses(buf, protobufs=None):
    """Reader of the CPM2Output op's element (cpm2_output_kernel_cpu.cpp:177-180:
    serialize_proto_vector_of_vectors<scanner::Point>): u64 people; per person u64 joints; per joint
    an i32 byte size and the proto3 bytes of Point{float x = 1; y = 2; score = 3}.
    Returns float32 (people, 18, 3)."""
    import struct
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    out = []
    for _ in range(n):
        (m,) = struct.unpack_from("<Q", buf, off)
        off += 8
        person = np.zeros((m, 3), np.float32)
        for j in range(m):
            (sz,) = struct.unpack_from("<i", buf, off)
            off += 4
            end = off + sz
            while off < end:
                tag = buf[off]
                field, wire = tag >> 3, tag & 7
                if wire != 5 or not 1 <= field <= 3:
                    raise ValueError("unexpected field in a Point message")
                person[j, field - 1] = struct.unpack_from("<f", buf, off + 1)[0]
                off += 5
        out.append(person)
    return np.stack(out) if out else np.zeros((0, 18, 3), np.float32)
