#!/usr/bin/env python3
"""How coherent is the R1 gather of the flow iteration on the benchmark stream?  (Round 5, CPU only.)

A texel-reuse scheme for k_flow_iter3 takes the second-column texels (t1, b1) of lane i from lane i + 1 and the top-row
texels of row y + 1 from the bottom row of row y.  Both need the gather geometry of neighbouring pixels to line up:
floor(x + fx) of the right-hand neighbour one column further, floor(y + fy) of the lower neighbour one row further.
Because a vector-memory instruction costs the texture-address unit the same whatever its execution mask
(scripts/ubench/vmemmask.hip), a fallback load for ONE failing lane costs what the load for all 64 costs: the scheme
pays only where the condition holds for a whole wave.  This script measures, on pairs of bench.py's stream (and on a
sub-pixel variant), the per-lane and the wave-uniform rates for the level-0 flow the oracle computes.

    python scripts/flow_coherence.py [--pairs 3] [--height 1080 --width 1920]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rates(fl):
    h, w = fl.shape[:2]
    xs, ys = np.arange(w)[None, :], np.arange(h)[:, None]
    x1 = np.floor(xs + fl[..., 0]).astype(int)
    y1 = np.floor(ys + fl[..., 1]).astype(int)
    lane_ok = np.zeros((h, w), bool)
    lane_ok[:, :-1] = (x1[:, 1:] == x1[:, :-1] + 1) & (y1[:, 1:] == y1[:, :-1])
    row_ok = np.zeros((h, w), bool)
    row_ok[1:] = (y1[1:] == y1[:-1] + 1) & (x1[1:] == x1[:-1])
    W = (w // 64) * 64
    lane_wave = lane_ok[:, :W].reshape(h, -1, 64)[:, :, :63].all(-1)   # lane 63 has no neighbour in its wave
    row_wave = row_ok[:, :W].reshape(h, -1, 64).all(-1)
    return lane_ok.mean(), row_ok.mean(), lane_wave.mean(), row_wave.mean()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    args = ap.parse_args()
    import torch
    import bench
    import oracle
    fr = bench.make_stream(torch, "cpu", args.pairs + 1, args.height, args.width, 1).numpy()
    print("pair  mean flow (x, y)      per lane: lane-share  row-carry   whole wave: lane-share  row-carry")
    for p in range(args.pairs):
        fl = oracle.optical_flow_rgb(fr[p], fr[p + 1])
        a, b, c, d = rates(fl)
        print("%4d  (%+.3f, %+.3f)              %.3f       %.3f                  %.3f       %.3f"
              % (p, fl[..., 0].mean(), fl[..., 1].mean(), a, b, c, d))


if __name__ == "__main__":
    main()
