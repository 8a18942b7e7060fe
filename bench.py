#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): the Farneback
OpticalFlow op over a device-resident 1080p frame stream (stencil {0,1}: one flow field per
consecutive frame pair) together with the per-channel Histogram op on the same frames.  One
"step" = one batch of B = 256 frames per GPU (257 resident frames, 256
pairs): Histogram on B frames (256 bins) + OpticalFlow on the B pairs formed with one halo frame.  value = frames/s through both ops,
whole job (all ranks), inputs already in HBM when the timed region starts.

Multi-GPU: frames are independent (pairs need one halo frame), so every rank processes its own
contiguous shard with no data-path collective ("weak" scaling: per-GPU batch fixed);
torch.distributed (RCCL) is used only for the barriers and the max-over-ranks of the time.

The JSON line also carries
  roofline     : for the dominant kernel (k_flow_iter: UpdateMatrices + 15x15 box blur + 2x2
                 solve, one launch per Farneback iteration), algorithmic bytes / HIP-event time
                 measured live over the timed region on the stream the kernels run on;
                 peak = 8 TB/s HBM3E spec.
  histogram    : frames/s and roofline of the Histogram kernel alone (same run, own timed loop).
  cpu_baseline : the CPU oracle (oracle/oracle.c, a port of the OpenCV algorithms the reference
                 calls) timed on this host's cores on a bounded sample of the same stream.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBS = 6290.0  # measured float4 copy ceiling of the same guide (report both fractions)


def make_stream(torch, device, n, h, w, seed):
    """n RGB frames on the device: a smooth random texture under an integer random-walk
    translation plus +-2 grey levels of per-frame noise (non-trivial flow, well-spread
    histograms; never zeros)."""
    g = torch.Generator(device=device).manual_seed(seed)
    m = 32
    low = torch.rand((1, 3, (h + 2 * m) // 8 + 2, (w + 2 * m) // 8 + 2), device=device, generator=g)
    tex = torch.nn.functional.interpolate(low, size=(h + 2 * m, w + 2 * m), mode="bicubic", align_corners=False)[0]
    tex = (tex - tex.amin()) / (tex.amax() - tex.amin()) * 235.0 + 10.0
    frames = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
    rng = np.random.default_rng(seed)
    pos = np.zeros(2, int)
    for i in range(n):
        oy, ox = m - pos[1], m - pos[0]
        crop = tex[:, oy:oy + h, ox:ox + w].permute(1, 2, 0)
        noise = torch.randint(-2, 3, (h, w, 3), device=device, generator=g)
        frames[i] = (crop + noise).clamp_(0, 255).to(torch.uint8)
        pos = np.clip(pos + rng.integers(-3, 4, 2), -m, m)
    return frames


def fb_geometry(h, w):
    from scannertools_amd.hip import fb_levels, fb_level_geom
    levels = fb_levels(h, w)
    return [fb_level_geom(h, w, k)[:2] for k in range(levels + 1)]


def cpu_baseline(frames_np, threads, pairs_per_thread):
    """T independent oracle instances over disjoint contiguous shards (Scanner's
    pipeline_instances_per_node model), histogram + flow per frame."""
    import oracle
    oracle.lib()
    n_pairs = threads * pairs_per_thread
    assert len(frames_np) >= 2

    def work(t):
        for j in range(pairs_per_thread):
            i = (t * pairs_per_thread + j) % (len(frames_np) - 1)
            oracle.hist_u8c3(frames_np[i], 256)
            oracle.optical_flow_rgb(frames_np[i], frames_np[i + 1])

    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return n_pairs / dt, n_pairs, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256,
                    help="frames (= flow pairs) per step per GPU; 256 pairs over 257 resident frames = SURVEY.md 8d config 2")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs-per-thread", type=int, default=3)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from scannertools_amd import _native
    from scannertools_amd.hip import HipContext
    ctx = HipContext(local_rank)

    B, h, w = args.batch, args.height, args.width
    # each rank's shard of the stream: a few distinct batches of B+1 frames (B frames + 1 halo)
    n_batches = 3
    batches = [make_stream(torch, device, B + 1, h, w, seed=1000 * rank + b) for b in range(n_batches)]
    flow_out = torch.empty((B, h, w, 2), dtype=torch.float32, device=device)
    hist_out = torch.empty((B, 3, args.bins), dtype=torch.int32, device=device)

    def step(i):
        fr = batches[i % n_batches]
        ctx.histogram(fr[:B], args.bins, out=hist_out)
        ctx.optical_flow(fr, out=flow_out)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    barrier()
    ctx.timing_enable([_native.K_BLUR_UPDATE])
    ctx.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    blur_launches, blur_ms = ctx.timing_read(_native.K_BLUR_UPDATE)
    ctx.timing_enable([])

    frames_total = B * args.steps * world
    fps = frames_total / dt

    # dominant-kernel roofline.  k_flow_iter runs numIters times per level and covers the stages
    # UpdateMatrices (60 B/px + 8 B/px of coarse flow on the finer levels), the two fused
    # blur+UpdateMatrices passes (80 B/px each) and the final blur (28 B/px) of the
    # stream-amortised model of SURVEY.md 8d: 248*sum(P_k) + 8*(sum(P_k) - P_0) bytes per pair.
    # (The kernel itself moves less: M is never materialised -- see DESIGN.md.)
    geom = fb_geometry(h, w)
    sum_p = sum(lh * lw for lh, lw in geom)
    p0 = geom[0][0] * geom[0][1]
    blur_bytes_per_step = (248 * sum_p + 8 * (sum_p - p0)) * B
    blur_gbs = blur_bytes_per_step * args.steps / (blur_ms * 1e-3) / 1e9 if blur_ms > 0 else 0.0
    flow_model_bytes = 284 * sum_p  # stream-amortised algorithmic bytes per flow frame

    # Histogram kernel alone (same data), its own timed loop
    hist_steps = max(args.steps, 10)
    ctx.timing_enable([_native.K_HIST])
    ctx.timing_reset()
    barrier()
    th0 = time.perf_counter()
    for i in range(hist_steps):
        ctx.histogram(batches[i % n_batches][:B], args.bins, out=hist_out)
    torch.cuda.synchronize(device)
    th = time.perf_counter() - th0
    hist_launches, hist_ms = ctx.timing_read(_native.K_HIST)
    ctx.timing_enable([])
    hist_bytes = (3 * h * w + 3 * args.bins * 4) * B
    hist_gbs = hist_bytes * hist_steps / (hist_ms * 1e-3) / 1e9 if hist_ms > 0 else 0.0
    # same kernel on i.i.d. uniform bytes (SURVEY.md 8d "random u8" input): the LDS-atomic-heavy case
    g = torch.Generator(device=device).manual_seed(7 + rank)
    rnd = torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8, device=device, generator=g)
    ctx.histogram(rnd, args.bins, out=hist_out)
    ctx.timing_enable([_native.K_HIST])
    ctx.timing_reset()
    for i in range(hist_steps):
        ctx.histogram(rnd, args.bins, out=hist_out)
    rnd_launches, rnd_ms = ctx.timing_read(_native.K_HIST)
    ctx.timing_enable([])
    del rnd
    rnd_gbs = hist_bytes * hist_steps / (rnd_ms * 1e-3) / 1e9 if rnd_ms > 0 else 0.0

    # HBM bytes per k_flow_iter launch from the PMC counters of the committed profile of this same
    # command (scripts/profile_round.sh -> scripts/pmc_traffic.py -> profiles/traffic.json):
    # 128-B read requests + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes.  None when the
    # profile is missing or was taken at another batch size / resolution.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and (B, h, w) == (256, 1080, 1920):
        try:
            traffic = float(json.load(open(tpath))["k_flow_iter"]["hbm_bytes_per_launch"])
        except Exception:
            traffic = None

    result = None
    if rank == 0:
        result = {
            "metric": "1080p frames/sec (optical-flow + histogram ops)",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (u8 in; f64 running sums)",
            "data": "synthetic",
            "config": {
                "workload": "Farneback OpticalFlow op (3,0.5,false,15,3,5,1.2,0), %dx%d pair stream, stencil {0,1}, "
                            "+ per-channel %d-bin Histogram op on the same frames" % (w, h, args.bins),
                "frames_per_step_per_gpu": B,
                "resolution": [w, h],
                "sharding": "contiguous frame shards per GPU + 1 halo frame, no collective",
            },
            "roofline": {
                "kernel": "k_flow_iter",
                "bound": "hbm",
                "achieved": blur_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": blur_gbs / HBM_PEAK_GBS,
                "frac_of_copy_ceiling": blur_gbs / HBM_COPY_CEILING_GBS,
                "traffic": traffic,
                "traffic_GBs": (traffic / (blur_ms / max(blur_launches, 1) * 1e-3) / 1e9) if traffic else None,
                "traffic_note": "HBM bytes per launch from rocprofv3 PMC passes of this command (profiles/traffic.json)",
                "algorithmic_bytes_per_launch": blur_bytes_per_step * args.steps / max(blur_launches, 1),
                "launches": blur_launches,
                "avg_launch_ms": blur_ms / max(blur_launches, 1),
                "algorithmic_bytes_per_step": blur_bytes_per_step,
            },
            "flow_whole_path": {
                "algorithmic_bytes_per_frame": flow_model_bytes,
                "achieved_GBs": fps / world * flow_model_bytes / 1e9,
                "frac_of_peak": fps / world * flow_model_bytes / 1e9 / HBM_PEAK_GBS,
            },
            "histogram": {
                "frames_per_s": B * hist_steps / th,
                "kernel": "k_hist_u8c3_v2",
                "achieved": hist_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": hist_gbs / HBM_PEAK_GBS,
                "frac_of_copy_ceiling": hist_gbs / HBM_COPY_CEILING_GBS,
                "launches": hist_launches,
                "avg_launch_ms": hist_ms / max(hist_launches, 1),
                "uniform_random_frames": {"achieved": rnd_gbs, "frac": rnd_gbs / HBM_PEAK_GBS,
                                          "frames_per_s": B * hist_steps / (rnd_ms * 1e-3) if rnd_ms > 0 else 0.0},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            threads = min(cores, 64)
            sample = batches[0][:min(B + 1, 9)].cpu().numpy()
            v, n_pairs, secs = cpu_baseline(sample, threads, args.cpu_pairs_per_thread)
            v1, n1, secs1 = cpu_baseline(sample, 1, 2)  # one Scanner kernel instance (1 thread and all cores)
            result["cpu_baseline"] = {
                "value": v,
                "unit": "frames/s",
                "cores": threads,
                "kind": "port",
                "sample": "%d pairs of the same %dx%d stream (histogram + Farneback per frame), %d oracle "
                          "instances on %d threads, %.1f s wall" % (n_pairs, w, h, threads, threads, secs),
                "host_cores": cores,
                "single_thread": {"value": v1, "unit": "frames/s", "cores": 1,
                                  "sample": "%d pairs, %.1f s wall" % (n1, secs1)},
                "gpu_over_cpu": {"all_cores": fps / v if v > 0 else None, "single_thread": fps / v1 if v1 > 0 else None},
            }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
