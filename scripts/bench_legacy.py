#!/usr/bin/env python3
"""The pipeline the reference's authors ran (scannertools/old/histograms.py:63-78, OpticalFlowHistogramPipeline):
    1080p frames -> Resize(426x240) -> OpticalFlow (no batch=) -> FlowHistogram
at its own geometry, and Farneback alone at the reference test clip's 640x480 (scannertools_infra/tests.py:17-86) -- both
take other kernels than the 1080p headline (3 pyramid levels with odd halves at 426x240; small launches at 640x480).

  python scripts/bench_legacy.py [--batches 64 256] [--steps 8]      device-resident, through the C ABI
  (host-fed: bench.py's `extra.legacy_flow_hist.host_fed` drives the kernel classes through the engine)

`measure()` is what bench.py's extra record calls."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(torch, ctx, device, frames_1080p, batches=(64, 256), steps=8, warmup=2):
    """frames_1080p: CUDA uint8 (n,1080,1920,3), n >= max(batches) + 1.  Returns the record."""
    from scannertools_amd import _native
    from scannertools_amd.hip import fb_level_geom, fb_levels
    out = {"workload": "1080p device frames -> Resize(426x240, INTER_LINEAR) -> OpticalFlow(3,0.5,false,15,3,5,1.2,0) on consecutive "
                       "pairs -> FlowHistogram (2 x 64 bins), B pairs per call; old/histograms.py:63-78",
           "levels_426x240": [list(fb_level_geom(240, 426, k)[:2]) for k in range(fb_levels(240, 426) + 1)]}
    sync = lambda: torch.cuda.synchronize(device)  # noqa: E731
    for B in batches:
        if B + 1 > frames_1080p.shape[0]:
            continue
        fr = frames_1080p[:B + 1]
        small = torch.empty((B + 1, 240, 426, 3), dtype=torch.uint8, device=device)
        flow = torch.empty((B, 240, 426, 2), dtype=torch.float32, device=device)

        def step():
            ctx.resize(fr, 426, 240, _native.INTER_LINEAR, out=small)
            ctx.optical_flow(small, out=flow)
            return ctx.flow_histogram(flow)

        for _ in range(warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            hist = step()
        sync()
        dt = (time.perf_counter() - t0) / steps
        # the three stages alone (own loops)
        parts = {}
        for name, fn in (("resize", lambda: ctx.resize(fr, 426, 240, _native.INTER_LINEAR, out=small)),
                         ("optical_flow", lambda: ctx.optical_flow(small, out=flow)),
                         ("flow_histogram", lambda: ctx.flow_histogram(flow))):
            fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            sync()
            parts[name + "_ms"] = (time.perf_counter() - t0) / steps * 1e3
        assert 0 < int(hist.sum()) <= B * 2 * 240 * 426   # magnitudes beyond the 64-px range are not counted (cv::calcHist)
        out["pairs_per_call_%d" % B] = {"frames_per_s": B / dt, "ms_per_call": dt * 1e3, **parts}
    # Farneback at the reference test clip's size
    clip = {}
    for B in batches:
        n = B + 1
        g = torch.Generator(device=device).manual_seed(640 + B)
        low = torch.rand((1, 3, 480 // 8 + 12, 640 // 8 + 12), device=device, generator=g)
        tex = torch.nn.functional.interpolate(low, size=(480 + 64, 640 + 64), mode="bicubic", align_corners=False)[0]
        tex = ((tex - tex.amin()) / (tex.amax() - tex.amin()) * 235 + 10).permute(1, 2, 0)
        fr = torch.stack([tex[32 + (i % 5):32 + (i % 5) + 480, 32 - (i % 7):32 - (i % 7) + 640].to(torch.uint8) for i in range(n)]).contiguous()
        flow = torch.empty((B, 480, 640, 2), dtype=torch.float32, device=device)
        for _ in range(warmup):
            ctx.optical_flow(fr, out=flow)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.optical_flow(fr, out=flow)
        sync()
        dt = (time.perf_counter() - t0) / steps
        clip["pairs_per_call_%d" % B] = {"frames_per_s": B / dt, "ms_per_call": dt * 1e3}
    out["optical_flow_640x480"] = clip
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[64, 256])
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    import torch
    from scannertools_amd.hip import HipContext
    sys.path.insert(0, ROOT)
    import bench
    device = torch.device("cuda", 0)
    frames = bench.make_stream(torch, device, max(args.batches) + 1, 1080, 1920, seed=426)
    with HipContext(0) as ctx:
        print(json.dumps(measure(torch, ctx, device, frames, tuple(args.batches), args.steps), indent=1))


if __name__ == "__main__":
    main()
