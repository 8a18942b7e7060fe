#!/usr/bin/env python3
"""BASELINE config 3: shot detection (Histogram -> ShotBoundaries) over a long 1080p stream sharded
across the GPUs of a node.

    python scripts/shot_pipeline.py [--frames 10000] [--height 1080 --width 1920]            # 1 GPU
    python scripts/shot_pipeline.py --gpus 8                 # starts its own 8 ranks (one per GPU)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/shot_pipeline.py --gpus 8

Each rank owns a contiguous shard of the stream (scannertools_amd.sharding.shard_range), generates
it on its own GPU (a per-shot random texture with small per-frame noise; cuts planted at known
frames), runs the Histogram op on it in chunks, and the per-frame histograms (192 B/frame) are
gathered on rank 0 -- the only exchange on the path -- where ShotBoundaries runs on the host as
in the reference.  Prints one JSON line: frames/s of the histogram stage (all ranks, max-over-ranks
time), the boundaries found and whether every planted cut is among them.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def planted_cuts(n, k):
    return sorted({int(n * (i + 1) / (k + 1)) for i in range(k)})


def shard_histograms(torch, ctx, dev, a, b, h, w, bins, chunk, cuts, seed=99):
    """Frames [a, b) of the synthetic stream (a per-shot random texture with its own colour statistics
    and +-3 grey levels of per-frame noise; shots change at `cuts`), generated on `dev` chunk by chunk
    and run through the Histogram op.  Returns (int32 (b-a, 3, bins) on the device, seconds spent in
    the Histogram calls -- frame generation is not timed)."""
    def shot_of(i):
        return int(np.searchsorted(cuts, i, side="right"))

    def texture(shot):
        # every shot has its own colour statistics (per-channel gain / offset), like a real cut
        g = torch.Generator(device=dev).manual_seed(1234 + shot)
        rs = np.random.default_rng(shot)
        gain = torch.tensor(rs.uniform(0.25, 1.0, 3), device=dev, dtype=torch.float32)
        off = torch.tensor(rs.uniform(0.0, 60.0, 3), device=dev, dtype=torch.float32)
        t = torch.rand((h, w, 3), device=dev, generator=g) * 255.0 * gain + off
        return t.clamp_(0, 255).to(torch.int16)

    hist = torch.empty((b - a, 3, bins), dtype=torch.int32, device=dev)
    buf = torch.empty((min(chunk, max(b - a, 1)), h, w, 3), dtype=torch.uint8, device=dev)
    gn = torch.Generator(device=dev).manual_seed(seed)
    t_hist = 0.0
    cur_shot, base = -1, None
    for c0 in range(a, b, chunk):
        c1 = min(b, c0 + chunk)
        for i in range(c0, c1):
            s = shot_of(i)
            if s != cur_shot:
                cur_shot, base = s, texture(s)
            noise = torch.randint(-3, 4, (h, w, 3), dtype=torch.int16, device=dev, generator=gn)
            buf[i - c0] = (base + noise).clamp_(0, 255).to(torch.uint8)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ctx.histogram(buf[:c1 - c0], bins, out=hist[c0 - a:c1 - a])
        torch.cuda.synchronize(dev)
        t_hist += time.perf_counter() - t0
    return hist, t_hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--chunk", type=int, default=250)
    ap.add_argument("--bins", type=int, default=16)
    ap.add_argument("--cuts", type=int, default=8)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU self-test of the multi-rank plumbing (gloo): every rank fabricates its shard's "
                         "per-frame histogram rows instead of running the kernel; shard -> gather -> ShotBoundaries")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # one child per GPU, started before this process touches the GPU (it never does)
        from scannertools_amd.sharding import spawn_ranks
        return spawn_ranks(__file__, sys.argv[1:], args.gpus, args.master_port)
    if int(env_world or "1") != args.gpus:
        sys.stderr.write("shot_pipeline.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, env_world))
        return 2
    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist
    from scannertools_amd.hip import HipContext
    from scannertools_amd.sharding import gather_rows, shard_range
    from scannertools_amd.shot_detection import shot_boundaries

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ST_BENCH_SHARE_GPU=1: self-test of the multi-rank path on a box with fewer GPUs than ranks (ranks share the visible
    # GPUs, gloo instead of RCCL, the histogram rows gathered through host memory); never a measurement
    share = os.environ.get("ST_BENCH_SHARE_GPU") == "1"
    if share:
        local = local % max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n, h, w = args.frames, args.height, args.width
    cuts = planted_cuts(n, args.cuts)
    a, b = shard_range(n, rank, world)
    ctx = HipContext(local)
    hist, t_hist = shard_histograms(torch, ctx, dev, a, b, h, w, args.bins, args.chunk, cuts, seed=99 + rank)
    t_hist_rank = t_hist
    if world > 1:
        t = torch.tensor([t_hist], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_hist = float(t.item())
    from scannertools_amd.sharding import device_id_string, rank_table
    table = rank_table(device_id_string(torch, dev), b - a, t_hist_rank * 1e3, "shot_pipeline.py", require_distinct=world > 1 and not share)
    full = gather_rows(hist.cpu() if share and world > 1 else hist, n, dst=0)
    if full is not None and not full.is_cuda:
        full = full.to(dev)
    if rank == 0:
        # ShotBoundaries on the gathered rows: on the device (st_shot_boundaries, the same decisions bit for bit) and,
        # beside it, the reference's host formulation
        from scannertools_amd.shot_detection import shot_boundaries_device
        shot_boundaries_device(ctx, full)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res_dev = shot_boundaries_device(ctx, full)
        t_sb_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        res = shot_boundaries(None, list(full.cpu().numpy()))
        t_sb = time.perf_counter() - t0
        assert res_dev[0] == res[0], "device and host ShotBoundaries disagree"
        print(json.dumps({"frames": n, "resolution": [w, h], "n_gpus": world, "bins": args.bins,
                          "histogram_frames_per_s": n / t_hist, "shot_boundaries_s": t_sb_dev, "shot_boundaries_host_s": t_sb,
                          "pipeline_frames_per_s": n / (t_hist + t_sb_dev),
                          "ranks": table, "distinct_devices": len({t_["device"] for t_ in table}),
                          "boundaries": res[0], "planted": cuts,
                          # the detector is a 2.5-sigma outlier test over +-500 frames: windows without a
                          # cut also flag noise peaks (so does the reference); every planted cut must be found
                          "planted_found": all(c in res[0] for c in cuts)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def dry_run(args):
    """No GPU, no kernel: each rank makes up the histogram rows of its shard (a per-shot base
    histogram plus small noise), then the real sharding / gather / ShotBoundaries code runs."""
    import torch
    import torch.distributed as dist
    from scannertools_amd.sharding import gather_rows, shard_range
    from scannertools_amd.shot_detection import shot_boundaries
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = args.frames
    cuts = planted_cuts(n, args.cuts)
    a, b = shard_range(n, rank, world)
    px = args.height * args.width
    rows = np.empty((b - a, 3, args.bins), np.int32)
    for i in range(a, b):
        shot = int(np.searchsorted(cuts, i, side="right"))
        base = np.random.default_rng(shot).multinomial(px, np.random.default_rng(100 + shot).dirichlet(np.ones(args.bins)), 3)
        rows[i - a] = base + np.random.default_rng(10000 + i).integers(-2, 3, (3, args.bins))
    from scannertools_amd.sharding import rank_table
    table = rank_table("cpu:%d" % rank, b - a, 0.0, "shot_pipeline.py --dry-run", require_distinct=True)
    full = gather_rows(torch.from_numpy(rows), n, dst=0)
    if rank == 0:
        res = shot_boundaries(None, list(full.numpy()))
        print(json.dumps({"dry_run": True, "frames": n, "n_gpus": world, "bins": args.bins, "ranks": table, "boundaries": res[0],
                          "planted": cuts, "planted_found": all(c in res[0] for c in cuts)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
