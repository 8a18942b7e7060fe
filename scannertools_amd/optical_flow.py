"""Front-end of the OpticalFlow op: mirror of ``compute_flow``
(/root/reference/scannertools/scannertools/old/optical_flow.py:8-26).

The reference's runner wraps one graph -- ``db.ops.OpticalFlow(frame=frame_sampled, device=...)``
(:20-23) -- in its job machinery (Database, sinks, megabatches: out of scope here).
``compute_flow`` builds exactly that graph on a Client (the in-process engine of this package, or
anything with the same ``io/ops/streams/run`` surface) and returns one flow stream per video.
"""
from .engine import CacheMode, DeviceType, NamedStream, NamedVideoStream, PerfParams


def build_pipeline(sc, frame_sampled, device=DeviceType.GPU, batch=None):
    """old/optical_flow.py:19-24.  As there, no ``batch=`` reaches the op by default: Scanner then hands the kernel one
    row (one pair) per ``execute()``.  ``batch=N`` is the speed switch of this build (N pairs per call fill the GPU from
    one kernel instance; without it a Scanner graph fills it with ``pipeline_instances_per_node`` -- DESIGN.md 7)."""
    if batch is None:
        return {'flow': sc.ops.OpticalFlow(frame=frame_sampled, device=device)}
    return {'flow': sc.ops.OpticalFlow(frame=frame_sampled, device=device, batch=batch)}


def compute_flow(sc, videos, frames=None, device=DeviceType.GPU, batch=None, suffix='flow'):
    """videos: names of ingested video streams; frames: optional list (one per video) of frame
    indices to sample (the reference's ``frames=`` argument, prelude.py:267-287).  Returns a list of
    NamedStream, one per video, whose rows are (h, w, 2) float32 flow fields: row i is the flow from
    sampled frame i to sampled frame i+1 (the op's stencil {0, 1}; the last row pairs the last frame
    with itself).  As in the reference the fields are not materialised by this call beyond the
    engine's own output table."""
    outputs = []
    for vi, name in enumerate(videos):
        frame = sc.io.Input([NamedVideoStream(sc, name)])
        sampled = sc.streams.Gather(frame, [list(frames[vi])]) if frames is not None else frame
        out = NamedStream(sc, '%s_%s' % (name, suffix))
        sc.run(sc.io.Output(build_pipeline(sc, sampled, device, batch)['flow'], [out]), PerfParams.estimate(),
               cache_mode=CacheMode.Overwrite)
        outputs.append(out)
    return outputs
