"""DrawFlow for MI355X: mirror of ``/root/reference/scannertools/scannertools/vis.py:8-12``.

The reference registers ``DrawFlow`` as a Scanner python op whose body is numpy (average of the
two flow channels, normalised by the frame's maximum, clipped, cast to uint8, stacked beside the
frame).  Here the same op name and signature run the batch on the GPU through
``st_draw_flow_batch`` (include/scannertools_hip.h); outputs are bit-identical to the reference's
on the golden vectors of tests/golden/draw_flow_golden.npz.  There is no CPU fallback: without the
HIP library or a GPU the call raises.
"""
import numpy as np

_CTX = {}


def _ctx(device):
    from .hip import HipContext
    if device not in _CTX:
        _CTX[device] = HipContext(device)
    return _CTX[device]


def draw_flow_rows(frames, flows, device=0):
    """frames: n x (h,w,3) uint8, flows: n x (h,w,2) float32 (numpy arrays or CUDA tensors) ->
    list of n numpy (h,2w,3) uint8 pictures."""
    import torch
    if len(frames) != len(flows):
        raise ValueError("DrawFlow: %d frames for %d flows" % (len(frames), len(flows)))
    if not len(frames):
        return []
    dev = torch.device("cuda", device)

    def up(x, dtype):
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        return t.to(device=dev, dtype=dtype).contiguous()

    fr = [up(f, torch.uint8) for f in frames]
    fl = [up(f, torch.float32) for f in flows]
    out = _ctx(device).draw_flow(fr, fl)
    torch.cuda.synchronize(dev)
    return list(out.cpu().numpy())


def draw_flow(config, frame, flow):
    """Signature of the reference's python op (vis.py:9): one row in, one picture out."""
    return draw_flow_rows([frame], [flow])[0]


try:  # register with Scanner when it is installed, exactly as the reference module does
    import scannerpy as _sp

    draw_flow = _sp.register_python_op(name='DrawFlow')(draw_flow)
except ImportError:  # scannerpy absent: the in-process engine (scannertools_amd.engine) is used
    pass
