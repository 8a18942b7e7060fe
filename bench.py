#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): the Farneback
OpticalFlow op over a device-resident 1080p frame stream (stencil {0,1}: one flow field per
consecutive frame pair) together with the per-channel Histogram op on the same frames.  One
"step" = one batch of B = 256 frames per GPU (257 resident frames, 256
pairs): Histogram on B frames (256 bins) + OpticalFlow on the B pairs formed with one halo frame.
value = frames/s through both ops, whole job (all ranks), inputs already in HBM when the timed
region starts.

Multi-GPU: frames are independent (pairs need one halo frame), so every rank processes its own
contiguous shard with no data-path collective ("weak" scaling: per-GPU batch fixed);
torch.distributed (RCCL) is used only for the barriers and the max-over-ranks of the time.
`python bench.py --gpus N` starts its own N ranks (one child process per GPU, spawned before the
parent touches the GPU; the parent relays rank 0's line and fails if any rank fails); under an
external launcher (torch.distributed.run) WORLD_SIZE must equal --gpus.

With more than one rank (N > 1) the line carries the headline fields, `roofline`, `flow_whole_path`, `histogram` and the multi-GPU
evidence fields (`rccl_world_size`, `devices`, `distinct_devices`, `ms_per_step_by_rank`) only: `cpu_baseline`, `parity` and `extra`
are single-GPU records (rank 0 at N = 1), so that a scaling run times nothing but the sharded step.

The JSON line also carries
  roofline     : for the dominant kernel (k_flow_iter3: UpdateMatrices + 15x15 box blur + 2x2
                 solve, one launch per Farneback iteration): `achieved` = the compulsory bytes of the
                 launches AS BUILT (iter_as_built_bytes: R0 + R1 + flow in / out; M is never in memory)
                 / HIP-event time measured live over the timed region on the stream the kernels run
                 on, `frac` = that / 8 TB/s (<= 1 by construction); `frac_model_8d` = SURVEY.md 8d's
                 stage-by-stage model (M priced as if stored: can exceed 1) over the same time;
                 `traffic` = HBM bytes per launch from the PMC counters of the committed profile,
                 `frac_traffic` = those bytes / the live launch time / peak (what the memory system
                 really moved).  Steps of fewer than 32 pairs run without the event brackets (they cost
                 ~8 us per launch on the stream) and report no kernel roofline.
  parity       : the bench configuration's own output against the oracle (a 256-pair step taken
                 outside the timed region; first pairs' flows, first frames' histograms).
  histogram    : frames/s and roofline of the Histogram kernel alone (same run, own timed loop).
  cpu_baseline : the CPU oracle (oracle/oracle.c, a port of the OpenCV algorithms the reference
                 calls) on all host cores, median of >= 3 repetitions on a bounded sample.
  extra        : own timed loops after the headline: config 3 (10 000-frame shot detection on this GPU),
                 config 4 (4K, batch 32), config 5 (pose network), the host-fed (PCIe-inclusive) rates
                 through the DeviceType::CPU kernel classes (Histogram over 1000 host frames = config 1),
                 Histogram at small batches, OpticalFlow at 1 / 2 / 4 / 8 pairs per call (the reference's
                 unbatched calling pattern; C ABI back to back and the kernel class with its per-call
                 synchronisation), the legacy flow-histogram pipeline at 426x240 and Farneback at 640x480.
                 None of these is `value`.
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


# ------------------------------------------------------------------------------------------------
# workload
# ------------------------------------------------------------------------------------------------
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


FLOW_CONTENT_KINDS = ("integer", "subpixel", "zoom", "rotate", "occlusion", "jump", "static", "noise")


def make_stream_kind(torch, device, n, h, w, seed, kind):
    """n RGB frames on the device under one KIND of motion (extra.flow_content; the headline stream is "integer" =
    make_stream).  The reference op runs on decoded video (scannertools/tests/test_all.py:162-177): sub-pixel, zooming,
    rotating, occluding motion -- every kind resamples one smooth texture through a per-frame backward map
    (grid_sample, bicubic, reflected borders), adds the headline stream's +-2 grey levels of noise and rounds to uint8.
      integer    whole-pixel random walk, steps in [-3, 3] (= make_stream)
      subpixel   the same walk with real-valued steps
      zoom       scale 1 + 0.1 sin(2 pi i / 64) about the centre (up to 1 % per frame)
      rotate     angle 6 deg x sin(2 pi i / 128) about the centre (up to 0.29 deg = 5.6 px at the corners per frame)
      occlusion  a 480 x 640 rectangle of another texture moving (+5.3, +2.2) px per frame (wrapping) over a background
                 under a real-valued walk with steps in [-1.5, 1.5]
      jump       whole-pixel walk with steps in [-24, 24] (wrapping texture)
      static     one view, noise only
      noise      independent uniform random bytes
    """
    assert kind in FLOW_CONTENT_KINDS, kind
    if kind == "integer":
        return make_stream(torch, device, n, h, w, seed)
    g = torch.Generator(device=device).manual_seed(seed)
    frames = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
    if kind == "noise":
        for i in range(n):
            frames[i] = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device=device, generator=g)
        return frames
    F = torch.nn.functional
    m = 32
    H, W = h + 2 * m, w + 2 * m

    def texture():
        low = torch.rand((1, 3, H // 8 + 2, W // 8 + 2), device=device, generator=g)
        t = F.interpolate(low, size=(H, W), mode="bicubic", align_corners=False)
        return (t - t.amin()) / (t.amax() - t.amin()) * 235.0 + 10.0

    tex = texture()
    fg = texture() if kind == "occlusion" else None
    rng = np.random.default_rng(seed)
    yy, xx = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                            torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
    cy, cx = (h - 1) / 2.0, (w - 1) / 2.0

    def sample(t, sy, sx):   # t (1,3,H,W) at texture pixel coordinates (sy, sx)
        grid = torch.stack([sx / (W - 1) * 2 - 1, sy / (H - 1) * 2 - 1], -1)[None]
        return F.grid_sample(t, grid, mode="bicubic", padding_mode="reflection", align_corners=True)[0].permute(1, 2, 0)

    pos = np.zeros(2)
    fpos = np.array([w * 0.3, h * 0.25])
    for i in range(n):
        if kind == "jump":
            img = torch.roll(tex[0], (int(pos[1]), int(pos[0])), (1, 2))[:, m:m + h, m:m + w].permute(1, 2, 0)
            pos = pos + rng.integers(-24, 25, 2)
        elif kind == "static":
            img = tex[0, :, m:m + h, m:m + w].permute(1, 2, 0)
        elif kind == "subpixel":
            img = sample(tex, yy + (m - pos[1]), xx + (m - pos[0]))
            pos = np.clip(pos + rng.uniform(-3, 3, 2), -m, m)
        elif kind == "zoom":
            sc = 1.0 + 0.1 * np.sin(2 * np.pi * i / 64)
            img = sample(tex, cy + (yy - cy) / sc + m, cx + (xx - cx) / sc + m)
        elif kind == "rotate":
            t = np.deg2rad(6.0) * np.sin(2 * np.pi * i / 128)
            c_, s_ = float(np.cos(t)), float(np.sin(t))
            img = sample(tex, cy + (yy - cy) * c_ - (xx - cx) * s_ + m, cx + (yy - cy) * s_ + (xx - cx) * c_ + m)
        else:  # occlusion
            img = sample(tex, yy + (m - pos[1]), xx + (m - pos[0]))
            pos = np.clip(pos + rng.uniform(-1.5, 1.5, 2), -m, m)
            ry, rx = (yy - fpos[1]) % h, (xx - fpos[0]) % w          # rectangle-local coordinates (wrapping)
            inside = (ry < 480) & (rx < 640)
            img = torch.where(inside[..., None], sample(fg, ry + m, rx + m), img)
            fpos = fpos + np.array([5.3, 2.2])
        noise = torch.randint(-2, 3, (h, w, 3), device=device, generator=g)
        frames[i] = (img + noise).round_().clamp_(0, 255).to(torch.uint8)
    return frames


def fill_stream(torch, frames, seed, chunk=250):
    """Fills `frames` (n,h,w,3 uint8, on the device) with ONE continuous stream of n distinct frames: make_stream's texture
    under its integer random walk (the walk continues across chunks; every frame has its own noise)."""
    n, h, w, _ = frames.shape
    device = frames.device
    g = torch.Generator(device=device).manual_seed(seed)
    m = 32
    low = torch.rand((1, 3, (h + 2 * m) // 8 + 2, (w + 2 * m) // 8 + 2), device=device, generator=g)
    tex = torch.nn.functional.interpolate(low, size=(h + 2 * m, w + 2 * m), mode="bicubic", align_corners=False)[0]
    tex = (tex - tex.amin()) / (tex.amax() - tex.amin()) * 235.0 + 10.0
    rng = np.random.default_rng(seed)
    pos = np.zeros(2, int)
    for i in range(n):
        oy, ox = m - pos[1], m - pos[0]
        noise = torch.randint(-2, 3, (h, w, 3), device=device, generator=g)
        frames[i] = (tex[:, oy:oy + h, ox:ox + w].permute(1, 2, 0) + noise).clamp_(0, 255).to(torch.uint8)
        pos = np.clip(pos + rng.integers(-3, 4, 2), -m, m)
    return frames


def native_build_record(_native):
    """Which library this run loaded: the hash of the sources it was built from (st_build_info) beside the tree's, the machine
    that compiled it and this machine -- the .so files are not shipped to the GPU box, so the two hosts are the same there."""
    import socket
    try:
        info = _native.build_info()
        rec = {"library_source_hash": info["src"], "tree_source_hash": _native.source_hash(), "compiled_on": info["host"],
               "compiled_at": info["at"], "this_host": socket.gethostname(), "built_on_this_host": info["host"] == socket.gethostname()}
        path = os.path.join(os.path.dirname(_native.LIB_PATH), "build_record.json")
        if os.path.exists(path):
            b = json.load(open(path))
            if b.get("host") == info["host"]:
                rec["build_seconds"] = b.get("seconds")
        return rec
    except Exception as e:  # a record, not a requirement
        return {"error": repr(e)}


def fb_geometry(h, w):
    from scannertools_amd.hip import fb_levels, fb_level_geom
    levels = fb_levels(h, w)
    return [fb_level_geom(h, w, k)[:2] for k in range(levels + 1)]


def iter_model_bytes(h, w, pairs):
    """Algorithmic bytes of the stages k_flow_iter3 covers, per step of `pairs` pairs: UpdateMatrices
    (60 B/px + 8 B/px of coarse flow on the finer levels), the two fused blur+UpdateMatrices passes
    (80 B/px each) and the final blur (28 B/px) of the stream-amortised model of SURVEY.md 8d:
    248*sum(P_k) + 8*(sum(P_k) - P_0) per pair.  (The kernel itself moves less: M is never
    materialised -- DESIGN.md 4.5.)"""
    geom = fb_geometry(h, w)
    sum_p = sum(lh * lw for lh, lw in geom)
    p0 = geom[0][0] * geom[0][1]
    return (248 * sum_p + 8 * (sum_p - p0)) * pairs, 284 * sum_p


def iter_as_built_bytes(h, w, pairs):
    """Compulsory HBM bytes of the flow-iteration launches AS BUILT, per step of `pairs` pairs: every
    launch reads R0 (20 B/px) and gathers R1 (20 B/px) and writes a flow field (8 B/px); it reads the previous
    iteration's field (8 B/px) on iterations 2 and 3, the coarser level's field (8 B per coarse pixel) on the first
    iteration of a finer level, nothing on the coarsest level's first.  M is never in memory, so it is not priced.
    3 iterations per level: sum_k (3*48 + 2*8) P_k + 8 (sum P - P_0)."""
    geom = fb_geometry(h, w)
    sum_p = sum(lh * lw for lh, lw in geom)
    p0 = geom[0][0] * geom[0][1]
    return (160 * sum_p + 8 * (sum_p - p0)) * pairs


def host_threads():
    """Threads the CPU baseline may use: the cores this process may run on, bounded so that the
    oracle instances (about 0.2 GB each at 1080p) stay inside half of the available memory."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    avail = None
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(path).read().strip()
            if v.isdigit():
                avail = min(avail, int(v)) if avail else int(v)
        except OSError:
            pass
    return cores, avail


def cpu_baseline(frames_np, threads, pairs_per_thread, reps, bins, keep=0):
    """T independent oracle instances over disjoint contiguous shards (Scanner's
    pipeline_instances_per_node model), histogram + flow per frame; `reps` repetitions, median
    rate.  Returns (median frames/s, per-rep seconds, pairs per rep, {pair index: (hist, flow)} for
    the first `keep` pairs -- the parity block compares them with the GPU's output)."""
    import oracle
    oracle.lib()
    n_pairs = threads * pairs_per_thread
    assert len(frames_np) >= 2
    kept = {}

    def work(t):
        for j in range(pairs_per_thread):
            i = (t * pairs_per_thread + j) % (len(frames_np) - 1)
            hst = oracle.hist_u8c3(frames_np[i], bins)
            fl = oracle.optical_flow_rgb(frames_np[i], frames_np[i + 1])
            if i < keep and i not in kept:
                kept[i] = (hst, fl)

    secs = []
    for _ in range(reps):
        ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        secs.append(time.perf_counter() - t0)
    return n_pairs / float(np.median(secs)), secs, n_pairs, kept


def timed_flow_hist(torch, ctx, _native, batches, B, bins, steps, warmup, barrier, flow_out, hist_out):
    """W untimed + K timed steps of (Histogram on B frames, OpticalFlow on B pairs); returns the wall
    time and the HIP-event time / launch count of the flow-iteration kernel inside the timed region."""
    nb = len(batches)

    def step(i):
        fr = batches[i % nb]
        ctx.histogram(fr[:B], bins, out=hist_out)
        ctx.optical_flow(fr, out=flow_out)

    for i in range(warmup):
        step(i)
    barrier()
    # The HIP-event brackets around every flow-iteration launch cost ~8 us each on the stream (two barrier packets): below
    # 32 pairs per step that is a tenth to a quarter of the step (1 pair: 500 us with, 395 us without), so small steps run
    # without them and carry no kernel roofline (ST_BENCH_NO_KERNEL_TIMING=1 forces that for any size).
    kernel_timing = B >= 32 and not os.environ.get("ST_BENCH_NO_KERNEL_TIMING")
    ctx.timing_enable([_native.K_BLUR_UPDATE] if kernel_timing else [])
    ctx.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    launches, ms = ctx.timing_read(_native.K_BLUR_UPDATE)
    ctx.timing_enable([])
    return dt, launches, ms


def concurrent_instances(torch, device, frames, h, w, bins, ks=(1, 2, 4, 8), pairs=(1, 8), reps=3):
    """Aggregate frames/s of K concurrent kernel instances (K contexts, K streams, K host threads) each making
    Histogram + OpticalFlow calls of b pairs, one stream sync per call.  Every cell is measured `reps` times (fresh
    threads, the same contexts); the record carries the median and the range -- these figures move from run to run
    (round 5: 8 pairs per call, K = 2: 5 951 in one run, 7 757 in another)."""
    from scannertools_amd.hip import HipContext
    rec = {"what": "K instances x (Histogram + OpticalFlow of b pairs per call, st_ctx_sync after every call: the kernel classes' "
                   "execute()), K host threads, one context and one stream each, 1080p device frames; frames/s over all instances; "
                   "median of %d repetitions, [min, max] beside it" % reps}
    nfr = len(frames)
    for b in pairs:
        if b + 1 > nfr:
            continue
        calls = max(12, 96 // b)
        by_k = {}
        for K in ks:
            ctxs = [HipContext(device.index) for _ in range(K)]
            flows = [torch.empty((b, h, w, 2), dtype=torch.float32, device=device) for _ in range(K)]
            hists = [torch.empty((b, 3, bins), dtype=torch.int32, device=device) for _ in range(K)]
            rates, errors = [], []
            for rep in range(reps):
                barrier, done = threading.Barrier(K + 1), [0.0] * K

                def worker(k):
                    try:
                        stream = torch.cuda.Stream(device)
                        with torch.cuda.stream(stream):
                            for i in range(calls + 2):
                                if i == 2:          # two warm-up calls (scratch allocation), then all instances start together
                                    barrier.wait()
                                j = (i * b + 7 * k + 13 * rep) % (nfr - b)
                                ctxs[k].histogram(frames[j:j + b], bins, out=hists[k])
                                ctxs[k].optical_flow(frames[j:j + b + 1], out=flows[k])
                                ctxs[k].sync()
                        done[k] = time.perf_counter()
                    except BaseException as e:  # noqa: BLE001
                        errors.append(repr(e))
                        barrier.abort()

                th = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
                for t in th:
                    t.start()
                try:
                    barrier.wait()
                except threading.BrokenBarrierError:
                    pass
                t0 = time.perf_counter()
                for t in th:
                    t.join()
                if errors:
                    break
                rates.append(K * calls * b / (max(done) - t0))
            for c in ctxs:
                c.close()
            del flows, hists
            if errors:
                by_k["K_%d" % K] = {"error": errors[0]}
                break
            med = float(np.median(rates))
            by_k["K_%d" % K] = {"frames_per_s": med, "frames_per_s_range": [min(rates), max(rates)],
                                "ms_per_call": K * b / med * 1e3}
        rec["pairs_per_call_%d" % b] = by_k
    return rec


def flow_content(torch, ctx, _native, device, args, hist_out, steps=3):
    """The headline step (Histogram on B frames + OpticalFlow on the B pairs) on streams of other KINDS of motion than the
    headline's whole-pixel walk: frames/s, the flow-iteration kernel's time per launch, and what the estimated field looks
    like (mean |flow|, the share of 64-pixel row runs whose gather stays in one source row).  One batch per kind, `steps`
    timed steps after one warm-up."""
    B, h, w, bins = args.batch, args.height, args.width, args.bins
    rec = {"what": "256-pair step of the headline (Histogram + OpticalFlow, %dx%d) per kind of motion (bench.py: make_stream_kind); "
                   "iter_avg_launch_ms = HIP-event time per k_flow_iter3 launch; parity per kind: tests/test_flow_motion_gpu.py" % (w, h)}
    flow_out = torch.empty((B, h, w, 2), dtype=torch.float32, device=device)
    sync = lambda: torch.cuda.synchronize(device)  # noqa: E731
    for kind in FLOW_CONTENT_KINDS:
        try:
            fr = make_stream_kind(torch, device, B + 1, h, w, 7000 + FLOW_CONTENT_KINDS.index(kind), kind)
            dt, launches, ms = timed_flow_hist(torch, ctx, _native, [fr], B, bins, steps, 1, sync, flow_out, hist_out)
            fl = flow_out[:: max(B // 8, 1)]
            mag = torch.linalg.vector_norm(fl, dim=-1)
            # how coherent the R1 gather is: a wave covers 64 consecutive pixels of a row; its gather touches one source
            # row pair when floor(y + fy) - y is the same for all of them
            dy = torch.floor(fl[..., 1])[:, :, : (w // 64) * 64].reshape(fl.shape[0], h, w // 64, 64)
            dx = torch.floor(fl[..., 0])[:, :, : (w // 64) * 64].reshape(fl.shape[0], h, w // 64, 64)
            one_row = (dy.amax(-1) == dy.amin(-1)).float().mean()
            one_col = (dx.amax(-1) == dx.amin(-1)).float().mean()
            rec[kind] = {"frames_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3,
                         "iter_avg_launch_ms": ms / max(launches, 1), "launches": launches,
                         "mean_abs_flow_px": float(mag.mean()), "p99_abs_flow_px": float(torch.quantile(mag.flatten()[::97], 0.99)),
                         "waves_gathering_one_row": float(one_row), "waves_gathering_one_column_offset": float(one_col)}
            del fr, fl, mag, dy, dx
        except Exception as e:  # auxiliary record
            rec[kind] = {"error": repr(e)}
    del flow_out
    torch.cuda.empty_cache()
    return rec


def run_shard(torch, ctx, frames, rows, lo, n_total, B, bins, flow_ring, hists, events=None):
    """One rank's work on its shard of a stream -- the code path `--gpus N` shards with (scannertools_amd.sharding):
    `rows` = [a, b) are the output rows the rank owns, `frames` its resident frames (rows + the halo frame of the
    OpticalFlow stencil {0,1}), frame `lo` being frames[0].  Calls of B rows: Histogram on the rows' frames, OpticalFlow
    on the rows' (clamped) pairs; flow fields are overwritten in a ring.  Returns the number of calls."""
    from scannertools_amd.sharding import local_pairs
    a, b = rows
    calls = 0
    for s0 in range(a, b, B):
        s1 = min(s0 + B, b)
        pairs = local_pairs((s0, s1), (lo, lo + len(frames)), n_total)          # indices into `frames`
        f0, f1 = int(pairs.min()), int(pairs.max()) + 1
        ctx.histogram(frames[s0 - lo:s1 - lo], bins, out=hists[s0 - a:s1 - a])
        ctx.optical_flow(frames[f0:f1], pairs=pairs - f0, out=flow_ring[calls % len(flow_ring)][:s1 - s0])
        if events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            events.append((s1 - a, ev))
        calls += 1
    return calls


def stream_10k(torch, ctx, device, args, n=10000):
    """The north star's own stream, once: n DISTINCT 1080p frames resident in HBM (62 GB), Histogram on all of them and
    OpticalFlow on every row of the stream (Scanner's stencil {0,1}: n rows, the last one's window clamped to the edge
    frame) in calls of B rows through the shard code of the multi-GPU path; frames/s overall and per ~1 000 frames
    (drift), and the same for ONE rank's shard of an 8-GPU split (1 250 rows + 1 halo frame)."""
    from scannertools_amd.sharding import flow_shard
    B, h, w, bins = args.batch, args.height, args.width, args.bins
    free, _ = torch.cuda.mem_get_info(device)
    need = n * h * w * 3 + 2 * B * h * w * 8 + (24 << 30)
    if free < need:
        return {"skipped": "needs %.0f GB of free device memory, %.0f GB free" % (need / 1e9, free / 1e9)}
    t0 = time.perf_counter()
    frames = fill_stream(torch, torch.empty((n, h, w, 3), dtype=torch.uint8, device=device), 99)
    torch.cuda.synchronize(device)
    t_gen = time.perf_counter() - t0
    ring = [torch.empty((B, h, w, 2), dtype=torch.float32, device=device) for _ in range(2)]
    hists = torch.empty((n, 3, bins), dtype=torch.int32, device=device)
    out = {"what": "%d distinct %dx%d frames resident (%.1f GB); Histogram (%d bins) on every frame + OpticalFlow on every row "
                   "(stencil {0,1}, last window clamped) in calls of %d rows through sharding.flow_shard / local_pairs; flow "
                   "fields overwritten in a ring of 2; inputs resident, no host transfer in the timed region" % (n, w, h, n * h * w * 3 / 1e9, bins, B),
           "generation_s": t_gen}
    for name, world in (("whole_stream_1_rank", 1), ("one_shard_of_8_ranks", 8)):
        rows, fr = flow_shard(n, 0, world)
        local = frames[fr[0]:fr[1]]
        run_shard(torch, ctx, local[:min(len(local), B + 1)], (rows[0], min(rows[1], rows[0] + B)), fr[0], n, B, bins, ring, hists)  # warm-up
        torch.cuda.synchronize(device)
        best = None
        for _rep in range(2):
            events = []
            start = torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            start.record()
            calls = run_shard(torch, ctx, local, rows, fr[0], n, B, bins, ring, hists, events)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, calls, [(k, start.elapsed_time(ev)) for k, ev in events])
        dt, calls, marks = best
        # frames/s per ~1 000 rows: between the call boundaries nearest to every multiple of 1 000
        per, last_k, last_ms = [], 0, 0.0
        for k, ms in marks:
            if k - last_k >= 1000 or k == marks[-1][0]:
                per.append(round((k - last_k) / max(ms - last_ms, 1e-9) * 1e3, 1))
                last_k, last_ms = k, ms
        out[name] = {"rows": rows[1] - rows[0], "resident_frames": fr[1] - fr[0], "calls": calls, "seconds": dt,
                     "frames_per_s": (rows[1] - rows[0]) / dt, "frames_per_s_per_1000_rows": per,
                     "frames_per_s_by_event_clock": (rows[1] - rows[0]) / (marks[-1][1] * 1e-3)}
    # the stream's own content check: distinct frames (no two consecutive histograms equal), the last row's clamped pair gives a zero field
    hs = hists.cpu().numpy()
    out["distinct_consecutive_histograms"] = bool((np.abs(np.diff(hs.astype(np.int64), axis=0)).sum((1, 2)) > 0).all())
    del frames, ring, hists
    torch.cuda.empty_cache()
    return out


def extras(torch, ctx, _native, device, args, batches, hist_out):
    """Records that are not `value`: each has its own timed loop, run after the headline."""
    out = {}
    sync = lambda: torch.cuda.synchronize(device)  # noqa: E731
    # (iii) Histogram at the small batches the reference pipelines pass (old/histograms.py:10-15)
    h, w, bins = args.height, args.width, args.bins
    small = {}
    for nb in (32, 64):
        if nb > args.batch:
            continue
        fr = batches[0][:nb]
        for _ in range(3):
            ctx.histogram(fr, bins, out=hist_out[:nb])
        ctx.timing_enable([_native.K_HIST])
        ctx.timing_reset()
        reps = 40
        for _ in range(reps):
            ctx.histogram(fr, bins, out=hist_out[:nb])
        n, ms = ctx.timing_read(_native.K_HIST)
        ctx.timing_enable([])
        gbs = (3 * h * w + 3 * bins * 4) * nb * reps / (ms * 1e-3) / 1e9
        small["batch_%d" % nb] = {"frames_per_s": nb * reps / (ms * 1e-3), "achieved": gbs, "unit": "GB/s",
                                  "frac": gbs / HBM_PEAK_GBS, "avg_launch_ms": ms / max(n, 1)}
        # The Histogram launch carries its timing events itself (hipExtLaunchKernelGGL: timestamps of the dispatch's own
        # completion signal); marker events before and after a 35-70 us kernel read 5-6 us more than it runs.  The
        # kernel-trace duration of the same launch is committed under profiles/ and quoted beside the live figure while
        # st_hist.hip is the file it was measured on.
        try:
            import hashlib
            tr = json.load(open(os.path.join(ROOT, "profiles", "hist_small_trace.json")))
            key = "bins_%d" % bins
            cur = hashlib.sha256(open(os.path.join(ROOT, "scannertools_amd", "csrc", "st_hist.hip"), "rb").read()).hexdigest()
            if (h, w) == (1080, 1920) and tr.get("sha256_st_hist_hip") == cur and key in tr.get("batch_%d" % nb, {}):
                us = tr["batch_%d" % nb][key]["median_us"]
                tgb = (3 * h * w + 3 * bins * 4) * nb / (us * 1e-6) / 1e9
                small["batch_%d" % nb]["kernel_trace"] = {"median_us": us, "achieved": tgb, "frac": tgb / HBM_PEAK_GBS,
                                                          "source": tr.get("source")}
        except Exception:
            pass
    small["what"] = ("frac / avg_launch_ms: HIP events attached to each dispatch (hipExtLaunchKernelGGL), measured live; kernel_trace: "
                     "the same launch's duration in the committed rocprofv3 kernel trace (marker events around the launch, as "
                     "rounds 1-3 measured it, read 5-6 us more)")
    out["histogram_small_batches"] = small

    # (iii-b) the headline step on other kinds of motion than the whole-pixel walk (round-5 verdict, item 2)
    if not args.no_content:
        try:
            out["flow_content"] = flow_content(torch, ctx, _native, device, args, hist_out)
        except Exception as e:  # auxiliary record
            out["flow_content"] = {"error": repr(e)}

    # (iv) OpticalFlow at the batch sizes a drop-in graph uses: the reference creates the op with no batch= (one pair per
    # execute(): tests/test_all.py:166, old/histograms.py:70-72).  Two ways of calling, no HIP-event brackets in either:
    #   stream        the C ABI back to back on one stream, one sync at the end (how bench.py's own step calls it), Histogram
    #                 on the same frames in the step as in the headline
    #   kernel_class  sc.ops.OpticalFlow(frame, device=GPU[, batch=b]) through the Scanner kernel class: every execute()
    #                 ends with st_ctx_sync, the rate is rows / seconds inside the execute() calls after the first
    try:
        from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
        small_flow = {}
        nfr = min(129, args.batch + 1)
        sc1 = Client(device_id=device.index)
        sc1.ingest_frames("sb", batches[0][:nfr])
        sc1.ingest_frames("sb_short", batches[0][:min(65, nfr)])
        frame1 = sc1.io.Input([NamedVideoStream(sc1, "sb")])
        frame1s = sc1.io.Input([NamedVideoStream(sc1, "sb_short")])
        for b in (1, 2, 4, 8):
            if b > args.batch:
                continue
            fr, fo, ho = batches[0][:b + 1], torch.empty((b, h, w, 2), dtype=torch.float32, device=device), hist_out[:b]
            reps = max(24, 192 // b)
            dt_s = None
            for _rep in range(2):   # the better of two loops (the first one of a size also grows the scratch arena)
                for k in range(reps + 3):
                    if k == 3:
                        sync()
                        t0 = time.perf_counter()
                    ctx.histogram(fr[:b], bins, out=ho)
                    ctx.optical_flow(fr, out=fo)
                sync()
                dt_s = min(dt_s or 1e30, time.perf_counter() - t0)
            rec = {"stream_frames_per_s": b * reps / dt_s, "stream_ms_per_call": dt_s / reps * 1e3}
            o = NamedStream(sc1, "sb_out_%d" % b)
            sc1.execute_seconds, sc1.steady_seconds, sc1.steady_rows = 0.0, 0.0, 0
            sc1.run(sc1.io.Output(sc1.ops.OpticalFlow(frame=frame1s if b <= 2 else frame1, device=DeviceType.GPU, **({} if b == 1 else {"batch": b})), [o]),
                    PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
            if sc1.steady_rows:
                rec["kernel_class_frames_per_s"] = sc1.steady_rows / sc1.steady_seconds
                rec["kernel_class_ms_per_execute"] = sc1.steady_seconds / (sc1.steady_rows / b) * 1e3
            small_flow["pairs_per_call_%d" % b] = rec
            del fo
        out["optical_flow_small_batches"] = {
            "what": "1080p device frames, b pairs per call: 'stream' = st_farneback_pairs + st_hist back to back on one stream "
                    "(no sync between calls); 'kernel_class' = OpticalFlowKernelHIP::execute (sync per call), a 65-frame stream (b <= 2) or a "
                    "%d-frame one; b = 1 is the reference's own calling pattern (no batch= on the op)" % nfr, **small_flow}
        del sc1
        ctx.release_workspace()
        torch.cuda.empty_cache()
    except Exception as e:  # auxiliary record
        out["optical_flow_small_batches"] = {"error": repr(e)}

    # (iv-b) what an UNCHANGED reference graph gets with pipeline_instances_per_node = K (scannertools/tests/test_all.py:45,231):
    # K kernel instances in one process, each with its own context and stream on its own host thread, each call ending in
    # a stream sync as OpticalFlowKernelHIP::execute / HistogramKernelHIP::execute do.  1 pair per call is the reference's
    # own pattern (no batch= on the op), 8 pairs a modest batch=.
    try:
        out["concurrent_instances"] = concurrent_instances(torch, device, batches[0], h, w, bins)
        torch.cuda.empty_cache()
    except Exception as e:  # auxiliary record
        out["concurrent_instances"] = {"error": repr(e)}

    try:
        # (i) config 4: OpticalFlow at 4K, batch 32 (BASELINE.json configs[3])
        if not args.no_4k:
            H4, W4, B4 = 2160, 3840, 32
            fr4 = [make_stream(torch, device, B4 + 1, H4, W4, seed=4000 + b) for b in range(2)]
            fo4 = torch.empty((B4, H4, W4, 2), dtype=torch.float32, device=device)
            ho4 = torch.empty((B4, 3, bins), dtype=torch.int32, device=device)
            dt, launches, ms = timed_flow_hist(torch, ctx, _native, fr4, B4, bins, 4, 1, sync, fo4, ho4)
            it_bytes, frame_bytes = iter_model_bytes(H4, W4, B4)
            gbs_8d = it_bytes * 4 / (ms * 1e-3) / 1e9
            gbs = iter_as_built_bytes(H4, W4, B4) * 4 / (ms * 1e-3) / 1e9
            out["config4_4k_batch32"] = {
                "workload": "OpticalFlow + %d-bin Histogram, 3840x2160, 32 pairs per call (33 resident frames)" % bins,
                "frames_per_s": B4 * 4 / dt, "ms_per_step": dt / 4 * 1e3, "steps": 4,
                "roofline": {"kernel": "k_flow_iter3", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbs / HBM_PEAK_GBS, "frac_model_8d": gbs_8d / HBM_PEAK_GBS, "launches": launches,
                             "avg_launch_ms": ms / max(launches, 1)},
                "flow_whole_path_frac_of_peak": B4 * 4 / dt * frame_bytes / 1e9 / HBM_PEAK_GBS}
            del fr4, fo4, ho4
            ctx.release_workspace()
            torch.cuda.empty_cache()
    except Exception as e:  # an auxiliary record must not take the headline down
        out['config4_4k_batch32'] = {"error": repr(e)}

    try:
        # config 3 on this one GPU: Histogram (the reference's 16 bins) over a 10 000-frame 1080p stream with 8
        # planted cuts, generated chunk by chunk on the device, then ShotBoundaries on the host
        if not args.no_shots:
            import importlib.util
            spec = importlib.util.spec_from_file_location("shot_pipeline", os.path.join(ROOT, "scripts", "shot_pipeline.py"))
            sp = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(sp)
            from scannertools_amd.shot_detection import shot_boundaries
            n3 = 10000
            cuts = sp.planted_cuts(n3, 8)
            hist3, t_hist = sp.shard_histograms(torch, ctx, device, 0, n3, 1080, 1920, 16, 250, cuts)
            from scannertools_amd.shot_detection import shot_boundaries_device
            shot_boundaries_device(ctx, hist3)   # warm-up
            sync()
            t0 = time.perf_counter()
            res_dev = shot_boundaries_device(ctx, hist3)
            t_sb_dev = time.perf_counter() - t0
            t0 = time.perf_counter()
            res = shot_boundaries(None, list(hist3.cpu().numpy()))
            t_sb = time.perf_counter() - t0
            out["config3_shot_detection_10k"] = {
                "workload": "Histogram (16 bins) on 10 000 x 1080p frames in chunks of 250 + ShotBoundaries on the device "
                            "(st_shot_boundaries; the host op timed beside it); 1 GPU holds the whole stream (8 GPUs: "
                            "scripts/shot_pipeline.py --gpus 8)",
                "histogram_frames_per_s": n3 / t_hist, "histogram_s": t_hist, "shot_boundaries_s": t_sb_dev,
                "shot_boundaries_host_s": t_sb, "pipeline_frames_per_s": n3 / (t_hist + t_sb_dev),
                "pipeline_frames_per_s_with_host_op": n3 / (t_hist + t_sb), "device_equals_host": res_dev[0] == res[0],
                "planted_cuts": cuts, "planted_found": all(c in res[0] for c in cuts), "boundaries_reported": len(res[0])}
            del hist3
            torch.cuda.empty_cache()
    except Exception as e:  # an auxiliary record must not take the headline down
        out['config3_shot_detection_10k'] = {"error": repr(e)}

    try:
        # config 5: 1080p frames -> CPM2Input -> the pose network's convolution stack (random weights, float32 like
        # the reference's Caffe pass) on the matrix cores; roofline = the f32 MFMA peak
        if not args.no_pose:
            from scannertools_amd import pose_net
            from scannertools_amd.hip import cpm2_geometry
            nb5, sc5 = 32, 368 / 1080.
            net = pose_net.PoseNet(ctx, seed=1)
            fr5 = batches[0][:nb5]
            _, _, nh5, nw5 = cpm2_geometry(h, w, sc5)
            fl5 = pose_net.flops(nh5, nw5)

            def pose_step():
                # CPM2Input -> CPM2 (network, `resize`, `nms`) -> the GPU half of CPM2Output (limb candidate scores)
                maps, joints = net.detect(ctx.cpm2_input(fr5, sc5))
                return maps, joints, ctx.cpm2_limb_scores(maps, joints)

            pose_step()
            sync()
            ids5 = [_native.K_CONV, _native.K_CPM2_INPUT, _native.K_CPM2_RESIZE, _native.K_CPM2_NMS, _native.K_CPM2_LIMBS]
            ctx.timing_enable(ids5)
            ctx.timing_reset()
            t0 = time.perf_counter()
            for _ in range(2):
                maps, joints, limb = pose_step()
            sync()
            dt5 = (time.perf_counter() - t0) / 2
            kms5 = {k: ctx.timing_read(k) for k in ids5}
            nl5, ms5 = kms5[_native.K_CONV]
            ctx.timing_enable([])
            # parity of the same code on a small input against the float32 torch network (CPU)
            g5 = torch.Generator().manual_seed(4)
            xs5 = torch.rand((1, 3, 48, 80), generator=g5) - 0.5
            got5 = net.forward(xs5.to(device)).permute(0, 3, 1, 2).cpu()
            ref5 = net.reference_forward(xs5, device="cpu")
            tf5 = nb5 * fl5 / (ms5 / 2 * 1e-3) / 1e12
            out["config5_pose_conv_stack"] = {
                "workload": "%d x %dx%d frames -> CPM2Input (scale %.4f -> %dx%d) -> CPM2 = OpenPose COCO body network (92 convolutions "
                            "+ 3 poolings, random float32 weights) + x8 bicubic `resize` + `nms` -> CPM2Output's limb scores; the assembly "
                            "of people on the host is not in this loop" % (nb5, w, h, sc5, nw5, nh5),
                "dtype": "f32 (v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulation)",
                "frames_per_s": nb5 / dt5, "ms_per_batch": dt5 * 1e3, "gflop_per_frame": fl5 / 1e9,
                "roofline": {"kernel": "k_conv_tile_f32 (3x3 / 7x7 layers with 128-channel output blocks) + k_conv_nhwc_f32 (the rest)", "bound": "mfma", "achieved": tf5, "peak": 157.3, "unit": "TFLOP/s",
                             "frac": tf5 / 157.3, "launches": nl5, "kernel_ms_per_batch": ms5 / 2},
                "other_kernels_ms_per_batch": {"cpm2_input": kms5[_native.K_CPM2_INPUT][1] / 2, "resize_maps": kms5[_native.K_CPM2_RESIZE][1] / 2,
                                               "nms": kms5[_native.K_CPM2_NMS][1] / 2, "limb_scores": kms5[_native.K_CPM2_LIMBS][1] / 2},
                "resize_maps_GBs": nb5 * 57 * nh5 * nw5 * 4 / (kms5[_native.K_CPM2_RESIZE][1] / 2 * 1e-3) / 1e9,
                "parity": {"what": "same network on a 1x3x48x80 input vs torch float32 on the CPU",
                           "max_abs": float((got5 - ref5).abs().max()), "ref_max_abs": float(ref5.abs().max())}}
            # the same chain with the opt-in split-bf16 arithmetic (bf16x3: float32-grade accuracy on the bf16 matrix
            # pipe, six bf16 MFMAs per float32-equivalent product block); never the config's headline figure
            try:
                net3 = pose_net.PoseNet(ctx, seed=1, math="bf16x3")

                def pose_step3():
                    maps3, joints3 = net3.detect(ctx.cpm2_input(fr5, sc5))
                    return ctx.cpm2_limb_scores(maps3, joints3)

                pose_step3()
                sync()
                ctx.timing_enable([_native.K_CONV])
                ctx.timing_reset()
                t0 = time.perf_counter()
                for _ in range(2):
                    pose_step3()
                sync()
                dt3 = (time.perf_counter() - t0) / 2
                nl3, ms3 = ctx.timing_read(_native.K_CONV)
                ctx.timing_enable([])
                got3 = net3.forward(xs5.to(device)).permute(0, 3, 1, 2).cpu()
                tf3 = nb5 * fl5 / (ms3 / 2 * 1e-3) / 1e12
                out["config5_pose_conv_stack"]["bf16x3"] = {
                    "what": "the same chain with every convolution's operands split into three bf16 terms (opt-in: PoseNet(math='bf16x3') / "
                            "SCANNERTOOLS_POSE_MATH=bf16x3); useful flops counted once, the matrix pipe executes six bf16 products per "
                            "float32-equivalent one",
                    "dtype": "bf16x3 (v_mfma_f32_32x32x16_bf16 on hi/mid/lo terms, f32 accumulation)",
                    "frames_per_s": nb5 / dt3, "ms_per_batch": dt3 * 1e3,
                    "roofline": {"kernel": "k_conv_tile_bf16x3 (3x3 / 7x7 layers with 128-channel output blocks) + k_conv_nhwc_bf16x3 (the rest)", "bound": "mfma", "achieved_useful": tf3, "achieved_executed": 6 * tf3,
                                 "peak": 2500.0, "unit": "TFLOP/s", "frac": 6 * tf3 / 2500.0, "frac_of_f32_matrix_peak": tf3 / 157.3,
                                 "kernel_ms_per_batch": ms3 / 2},
                    "parity": {"what": "same network on a 1x3x48x80 input vs torch float32 on the CPU",
                               "max_abs": float((got3 - ref5).abs().max()), "ref_max_abs": float(ref5.abs().max()),
                               "max_abs_vs_f32_kernels": float((got3 - got5).abs().max())}}
                # the chain at the call sizes a drop-in graph uses (the reference's test: batch=5; a single frame), both arithmetics
                small5 = {}
                for nb_s in (1, 5):
                    fr_s = fr5[:nb_s]
                    for tag, nt in (("f32", net), ("bf16x3", net3)):
                        def step_s():
                            m_, j_ = nt.detect(ctx.cpm2_input(fr_s, sc5))
                            return ctx.cpm2_limb_scores(m_, j_)
                        step_s()
                        sync()
                        t0 = time.perf_counter()
                        for _ in range(4):
                            step_s()
                        sync()
                        small5["%s_frames_per_call_%d" % (tag, nb_s)] = nb_s * 4 / (time.perf_counter() - t0)
                small5["what"] = ("frames/s of the same chain at 1 and 5 frames per call (the reference's test runs the op with batch=5): "
                                  "the library then runs smaller tile instances and pairs the two branches of a stage in one launch; "
                                  "a frame's maps are the same bits at every call size")
                out["config5_pose_conv_stack"]["small_calls"] = small5
                del net3
            except Exception as e:  # auxiliary record
                out["config5_pose_conv_stack"]["bf16x3"] = {"error": repr(e)}
            # the same work through the reference's user-facing op: sc.ops.OpenPose on device frames (kernel class: transform,
            # network, merge, nms, limb scores, assembly of people on the host, element formatting), one scale, batch 32;
            # the model file holds the same random weights
            try:
                import shutil
                import tempfile
                from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
                mdir = tempfile.mkdtemp(prefix="st_openpose_")
                os.makedirs(os.path.join(mdir, "pose", "coco"))
                pose_net.write_caffemodel(os.path.join(mdir, "pose", "coco", "pose_iter_440000.caffemodel"), net.weights)
                sc5c = Client(device_id=device.index)
                sc5c.ingest_frames("v5", batches[0][:2 * nb5])
                node5 = sc5c.ops.OpenPose(frame=sc5c.io.Input([NamedVideoStream(sc5c, "v5")]), model_directory=mdir, device=DeviceType.GPU, batch=nb5)
                sc5c.execute_seconds, sc5c.steady_seconds, sc5c.steady_rows = 0.0, 0.0, 0
                o5 = NamedStream(sc5c, "pose5")
                sc5c.run(sc5c.io.Output(node5, [o5]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
                out["config5_pose_conv_stack"]["openpose_op"] = {
                    "what": "sc.ops.OpenPose(frame, pose_num_scales=1, device=GPU, batch=%d) on %d device frames: time inside the second "
                            "execute() call (the first one allocates the instance's device buffers); the random network fills every "
                            "frame with the maximum of candidate joints and people, the worst case for the host-side assembly" % (nb5, 2 * nb5),
                    "frames_per_s": sc5c.steady_rows / sc5c.steady_seconds, "first_call_frames_per_s": nb5 / (sc5c.execute_seconds - sc5c.steady_seconds),
                    "people_in_first_frame": len(list(o5.load())[0])}
                # the reference's own calling pattern: batch=5 (scannertools_caffe/tests/test_all.py:20)
                node5b = sc5c.ops.OpenPose(frame=sc5c.io.Input([NamedVideoStream(sc5c, "v5")]), model_directory=mdir, device=DeviceType.GPU, batch=5)
                sc5c.execute_seconds, sc5c.steady_seconds, sc5c.steady_rows = 0.0, 0.0, 0
                sc5c.run(sc5c.io.Output(node5b, [NamedStream(sc5c, "pose5b")]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
                out["config5_pose_conv_stack"]["openpose_op"]["batch_5_frames_per_s"] = sc5c.steady_rows / sc5c.steady_seconds
                shutil.rmtree(mdir, ignore_errors=True)
            except Exception as e:  # auxiliary record
                out["config5_pose_conv_stack"]["openpose_op"] = {"error": repr(e)}
            del net, maps, joints, limb
            torch.cuda.empty_cache()
    except Exception as e:  # an auxiliary record must not take the headline down
        out['config5_pose_conv_stack'] = {"error": repr(e)}

    # (v) the pipeline the reference's authors ran, at its own geometry (old/histograms.py:63-78): Resize(426x240) -> OpticalFlow
    # -> FlowHistogram, device-resident through the C ABI and host-fed through the kernel classes; Farneback at 640x480
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_legacy", os.path.join(ROOT, "scripts", "bench_legacy.py"))
        bl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bl)
        nl = min(args.batch, 256)
        rec = bl.measure(torch, ctx, device, batches[0][:nl + 1], batches=tuple(b for b in (64, 256) if b <= nl), steps=6)
        from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
        nh = min(args.batch, 192)
        host = batches[0][:nh].cpu().pin_memory()
        scl = Client(device_id=device.index)
        scl.ingest_frames("lv", host.numpy())
        fed = {"frames": nh}

        def run_op(node, name):
            # twice, the better: the first run's execute() calls also fill the shim's page-locked output pool
            # (a hipHostMalloc per batch of results), which a job of thousands of batches does once
            o, best = NamedStream(scl, "legacy_" + name), 0.0
            for _ in range(2):
                scl.execute_seconds, scl.steady_seconds, scl.steady_rows = 0.0, 0.0, 0
                scl.run(scl.io.Output(node, [o]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
                best = max(best, scl.steady_rows / scl.steady_seconds if scl.steady_rows else nh / scl.execute_seconds)
            return o, best

        small_s, r_resize = run_op(scl.ops.Resize(frame=scl.io.Input([NamedVideoStream(scl, "lv")]), width=426, height=240,
                                                  device=DeviceType.CPU, batch=64), "resize")
        scl.ingest_frames("lsmall", np.stack(list(small_s.load())))
        flow_s, r_flow = run_op(scl.ops.OpticalFlow(frame=scl.io.Input([NamedVideoStream(scl, "lsmall")]), device=DeviceType.CPU, batch=64), "flow")
        fed.update({"Resize_frames_per_s": r_resize, "Resize_input_GBs": r_resize * 3 * h * w / 1e9,
                    "OpticalFlow_426x240_frames_per_s": r_flow,
                    "pipelined_frames_per_s": min(r_resize, r_flow),
                    "what": "DeviceType::CPU registrations (host frames in, host results out), batch 64, rows / seconds inside the execute() "
                            "calls after the first; the stages run as separate graph ops, the pipelined rate is the slowest stage's: the "
                            "1080p upload of Resize (6.2 MB per frame in, 0.3 MB out), i.e. the H2D bound"})
        rec["host_fed"] = fed
        out["legacy_flow_hist"] = rec
        del scl, host
        ctx.release_workspace()
        torch.cuda.empty_cache()
    except Exception as e:  # auxiliary record
        out["legacy_flow_hist"] = {"error": repr(e)}

    # (ii) host-fed: frames in (page-locked) host memory -> results in host memory through the
    # DeviceType::CPU-registered kernel classes; time inside execute() (PCIe-inclusive)
    try:
        from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
        # config 1 of BASELINE.json: 1000 host-resident 1080p frames through the Histogram op (the frames of the three
        # resident batches, repeated); OpticalFlow keeps 192 frames (its 16.6 MB flow fields come back to the host)
        n = 192
        nh = 1000 if (h, w) == (1080, 1920) and args.batch >= 192 else n
        host_all = torch.cat([b_[:-1].cpu() for b_ in batches] * (nh // (len(batches) * (args.batch)) + 1))[:nh].pin_memory() if nh > n else None
        host = batches[0][:n].cpu().pin_memory()
        # measured H2D rate of this host for the same bytes (pinned -> device)
        dst = torch.empty_like(batches[0][:n])
        dst.copy_(host, non_blocking=True)
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(host, non_blocking=True)
        sync()
        h2d = 3 * host.numel() / (time.perf_counter() - t0) / 1e9
        sc = Client(device_id=device.index)
        sc.ingest_frames("v", host.numpy())
        frame = sc.io.Input([NamedVideoStream(sc, "v")])
        frame_h = frame
        if host_all is not None:
            sc.ingest_frames("vh", host_all.numpy())
            frame_h = sc.io.Input([NamedVideoStream(sc, "vh")])
        fed = {"h2d_GBs_pinned_copy": h2d, "frames": n, "histogram_frames": nh}
        for name, mk, nrep, n in (("Histogram", lambda: sc.ops.Histogram(frame=frame_h, device=DeviceType.CPU, batch=64, bins=bins), 3, nh),
                                  ("OpticalFlow", lambda: sc.ops.OpticalFlow(frame=frame, device=DeviceType.CPU, batch=32), 2, n)):
            best, steady = None, 0.0
            for _ in range(nrep):
                o = NamedStream(sc, "hf_" + name)
                sc.execute_seconds, sc.steady_seconds, sc.steady_rows = 0.0, 0.0, 0
                sc.run(sc.io.Output(mk(), [o]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
                best = sc.execute_seconds if best is None else min(best, sc.execute_seconds)
                if sc.steady_rows:
                    steady = max(steady, sc.steady_rows / sc.steady_seconds)
            # frames_per_s: a fresh kernel instance over the whole stream (its first execute() allocates the device scratch);
            # steady_frames_per_s: the execute() calls after the first, i.e. an instance that lives for a whole job
            fed[name] = {"frames_per_s": n / best, "input_GBs": n * 3 * h * w / best / 1e9,
                         "frac_of_h2d": n * 3 * h * w / best / 1e9 / h2d, "steady_frames_per_s": steady,
                         "steady_frac_of_h2d": steady * 3 * h * w / 1e9 / h2d}
        # the same Histogram graph fed from PAGEABLE host memory (the kernel bounces it through its
        # page-locked ring)
        n = 192
        del host_all
        sc.ingest_frames("vp", np.array(host.numpy(), copy=True))
        framep = sc.io.Input([NamedVideoStream(sc, "vp")])
        best = None
        for _ in range(2):
            o = NamedStream(sc, "hf_pageable")
            sc.execute_seconds = 0.0
            sc.run(sc.io.Output(sc.ops.Histogram(frame=framep, device=DeviceType.CPU, batch=64, bins=bins), [o]),
                   PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
            best = sc.execute_seconds if best is None else min(best, sc.execute_seconds)
        fed["Histogram_pageable_source"] = {"frames_per_s": n / best, "input_GBs": n * 3 * h * w / best / 1e9}
        out["host_fed"] = fed
    except Exception as e:  # the headline must not die on an auxiliary record
        out["host_fed"] = {"error": repr(e)}
    # the north star's own stream: 10 000 distinct 1080p frames through both ops once (round-5 verdict, item 3)
    if not args.no_stream10k:
        try:
            ctx.release_workspace()
            torch.cuda.empty_cache()
            out["stream_10k"] = stream_10k(torch, ctx, device, args)
        except Exception as e:  # auxiliary record
            out["stream_10k"] = {"error": repr(e)}
    return out


def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.dry_run:
        # launcher self-test (CPU, gloo): rendezvous, barriers and the max-over-ranks reduction of the
        # real path around a dummy step; no kernels run and the line is marked as such
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.001 * (rank + 1))
        dt = time.perf_counter() - t0
        my_ms = dt / max(args.steps, 1) * 1e3
        coll_world, devices, rank_ms = 1, ["cpu:0"], [my_ms]
        if world > 1:
            dist.barrier()
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            # the same evidence fields as the real line: the collective's own count of ranks, one device per rank, every
            # rank's time
            from scannertools_amd.sharding import rank_table
            table = rank_table("cpu:%d" % rank, args.batch * args.steps, my_ms, "bench.py --dry-run", require_distinct=True)
            devices, rank_ms = [t_["device"] for t_ in table], [t_["ms"] for t_ in table]
            one = torch.ones(1, dtype=torch.int32)
            ones = [torch.zeros_like(one) for _ in range(world)]
            dist.all_gather(ones, one)
            coll_world = int(sum(int(o_.item()) for o_ in ones))
        if rank == 0:
            print(json.dumps({"metric": "dry run (launcher self-test, no kernels)", "dry_run": True, "value": 0.0,
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": dt / max(args.steps, 1) * 1e3, "rccl_world_size": coll_world,
                              "collective_backend": "gloo (dry run)", "devices": devices, "distinct_devices": len(set(devices)),
                              "ms_per_step_by_rank": rank_ms}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    # ST_BENCH_SHARE_GPU=1 (self-test of the multi-rank path on a box with fewer GPUs than ranks): ranks share the
    # visible GPUs round-robin and synchronise over gloo (RCCL refuses two ranks on one device); never a measurement
    share = os.environ.get("ST_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
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

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()

    dt, blur_launches, blur_ms = timed_flow_hist(torch, ctx, _native, batches, B, args.bins, args.steps, args.warmup,
                                                 barrier, flow_out, hist_out)
    dt_rank = dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    frames_total = B * args.steps * world
    fps = frames_total / dt

    blur_bytes_per_step, flow_model_bytes = iter_model_bytes(h, w, B)
    blur_gbs = blur_bytes_per_step * args.steps / (blur_ms * 1e-3) / 1e9 if blur_ms > 0 else 0.0
    built_bytes_per_step = iter_as_built_bytes(h, w, B)
    built_gbs = built_bytes_per_step * args.steps / (blur_ms * 1e-3) / 1e9 if blur_ms > 0 else 0.0

    # multi-GPU evidence: what the collective library itself saw (an all_gather of one int per rank), the PCI bus id of
    # every rank's device, and every rank's own time per step
    from scannertools_amd.sharding import device_id_string, rank_table
    my_dev, my_ms = device_id_string(torch, device), dt_rank / args.steps * 1e3
    rccl_world, devices, rank_ms, rank_frames = 1, [my_dev], [my_ms], [B * args.steps]
    if world > 1:
        # rank 0 prints the table on stderr; N ranks on fewer than N devices is an error on every rank unless the run is
        # the declared self-test (ST_BENCH_SHARE_GPU=1)
        table = rank_table(my_dev, B * args.steps, my_ms, "bench.py", require_distinct=not share)
        devices, rank_ms, rank_frames = [t_["device"] for t_ in table], [t_["ms"] for t_ in table], [t_["frames"] for t_ in table]
        one = torch.ones(1, dtype=torch.int32, device="cpu" if share else device)
        ones = [torch.zeros_like(one) for _ in range(world)]
        dist.all_gather(ones, one)
        rccl_world = int(sum(int(o_.item()) for o_ in ones))

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

    # HBM bytes per k_flow_iter3 launch from the PMC counters of the committed profile of this same
    # command (scripts/profile_round.sh -> scripts/pmc_traffic.py -> profiles/traffic.json):
    # 128-B read requests + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes.  None when the
    # profile is missing or was taken at another batch size / resolution.
    traffic = l2_hit = None
    traffic_note = ("HBM bytes per launch from rocprofv3 PMC passes of this command (profiles/traffic.json); "
                    "frac_traffic is what the memory system really moved per second / peak")
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and (B, h, w) == (256, 1080, 1920):
        try:
            import hashlib
            tj_all = json.load(open(tpath))
            tj = tj_all.get("k_flow_iter3") or tj_all["k_flow_iter"]
            # the counters belong to the kernel source they were taken with: a profile of an older kernel is not
            # divided by this run's launch time
            # (the kernel's own file, the shared header that sets its address spaces, and the compiler flags)
            guard = ("st_farneback.hip", "st_internal.h", "Makefile")
            want = [tj_all.get("_meta", {}).get("source_sha256", {}).get(f) for f in guard]
            have = [hashlib.sha256(open(os.path.join(ROOT, "scannertools_amd", "csrc", f), "rb").read()).hexdigest() for f in guard]
            if want == have:
                traffic = float(tj["hbm_bytes_per_launch"])
                l2_hit = tj.get("L2_hit_rate")
            else:
                traffic_note = ("profiles/traffic.json was taken with another version of st_farneback.hip / st_internal.h / the Makefile: traffic withheld "
                                "(re-run scripts/profile_round.sh)")
        except Exception:
            traffic = None
    avg_launch_s = blur_ms / max(blur_launches, 1) * 1e-3
    traffic_gbs = (traffic / avg_launch_s / 1e9) if (traffic and avg_launch_s > 0) else None

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
            **({"ranks_share_gpus": True} if share else {}),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (u8 frames in; the box filter's running sums in f64, re-anchored every 32 rows / 8 columns, column sums "
                     "handed to the row pass as f32)",
            "rccl_world_size": rccl_world,
            "collective_backend": ("gloo (ranks share GPUs: self-test)" if share else "nccl (RCCL)") if world > 1 else "none (1 rank)",
            "devices": devices,
            "distinct_devices": len(set(devices)),
            "ms_per_step_by_rank": rank_ms,
            "frames_by_rank": rank_frames,
            "flow_whole_path_frac_of_peak": fps / world * flow_model_bytes / 1e9 / HBM_PEAK_GBS,
            "data": "synthetic",
            "native_build": native_build_record(_native),
            "config": {
                "workload": "Farneback OpticalFlow op (3,0.5,false,15,3,5,1.2,0), %dx%d pair stream, stencil {0,1}, "
                            "+ per-channel %d-bin Histogram op on the same frames" % (w, h, args.bins),
                "frames_per_step_per_gpu": B,
                "resolution": [w, h],
                "sharding": "contiguous frame shards per GPU + 1 halo frame, no collective",
            },
            "roofline": {
                "kernel": "k_flow_iter3",
                "bound": "hbm",
                "achieved": built_gbs,
                "achieved_is": "compulsory bytes of the launches as built (R0 20 + R1 20 + flow out 8 B/px per launch, flow in 8 B/px "
                               "on iterations 2-3, 8 B per coarse pixel at level transitions; M is never in memory) / HIP-event time "
                               "of the launches on their own stream",
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": built_gbs / HBM_PEAK_GBS,
                "frac_is": "as-built bytes / launch time / 8 TB/s: at most 1 by construction.  frac_model_8d prices the stages the launch "
                           "replaces with SURVEY 8d's stage-by-stage model (M written and re-read), which this kernel does not move "
                           "and which can therefore exceed 1; frac_traffic is what the PMC counters saw (L2 serves the expansion two "
                           "consecutive pairs share)",
                "as_built_bytes_per_launch": built_bytes_per_step * args.steps / max(blur_launches, 1),
                "achieved_model_8d": blur_gbs,
                "frac_model_8d": blur_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_GBs": traffic_gbs,
                "frac_traffic": (traffic_gbs / HBM_PEAK_GBS) if traffic_gbs else None,
                "frac_traffic_of_copy_ceiling": (traffic_gbs / HBM_COPY_CEILING_GBS) if traffic_gbs else None,
                "l2_hit_rate": l2_hit,
                "traffic_note": traffic_note,
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
            # one more 256-pair step of the bench configuration outside the timed region: its output
            # is what the oracle results computed by the baseline leg are compared with
            ctx.histogram(batches[0][:B], args.bins, out=hist_out)
            ctx.optical_flow(batches[0], out=flow_out)
            torch.cuda.synchronize(device)
            keep = min(B, 8)
            gpu_flow = flow_out[:keep].cpu().numpy()
            gpu_hist = hist_out[:keep].cpu().numpy()
            cores, avail = host_threads()
            threads = cores
            note = ""
            if avail:
                cap = max(1, int(avail * 0.5 / 0.25e9))
                if cap < threads:
                    threads, note = cap, " (capped by host memory: %.0f GB available)" % (avail / 1e9)
            sample = batches[0][:min(B + 1, 9)].cpu().numpy()
            v, secs, n_pairs, kept = cpu_baseline(sample, threads, args.cpu_pairs_per_thread, args.cpu_reps, args.bins, keep)
            v1, secs1, n1, _ = cpu_baseline(sample, 1, 1, max(args.cpu_reps, 3), args.bins)
            result["cpu_baseline"] = {
                "value": v,
                "unit": "frames/s",
                "cores": threads,
                "kind": "port",
                "sample": "%d pairs per repetition of the same %dx%d stream (histogram + Farneback per frame), %d oracle "
                          "instances on %d threads%s, median of %d repetitions (%s s)" % (
                              n_pairs, w, h, threads, threads, note, len(secs), ", ".join("%.1f" % s for s in secs)),
                "host_cores": cores,
                "single_thread": {"value": v1, "unit": "frames/s", "cores": 1,
                                  "sample": "%d pair per repetition, median of %d (%s s)" % (
                                      n1, len(secs1), ", ".join("%.2f" % s for s in secs1))},
                "gpu_over_cpu": {"all_cores": fps / v if v > 0 else None, "single_thread": fps / v1 if v1 > 0 else None},
            }
            if kept:
                idx = sorted(kept)
                ref = np.stack([kept[i][1] for i in idx])
                got = gpu_flow[idx]
                result["parity"] = {
                    "what": "GPU output of one %d-pair step of this configuration vs the oracle, pairs %s" % (B, idx),
                    "flow_rel_l2": float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)),
                    "flow_max_abs": float(np.abs(got - ref).max()),
                    "flow_tolerance": {"rel_l2": 1e-4, "max_abs_px": 5e-3},
                    "hist_bit_exact": bool(all((gpu_hist[i] == kept[i][0]).all() for i in idx)),
                }
        if world == 1 and not args.no_extras:
            del flow_out
            ctx.release_workspace()
            torch.cuda.empty_cache()
            result["extra"] = extras(torch, ctx, _native, device, args, batches, hist_out)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 0


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
    ap.add_argument("--no-extras", action="store_true", help="skip the extra records (4K, host-fed, small histogram batches)")
    ap.add_argument("--no-4k", action="store_true")
    ap.add_argument("--no-pose", action="store_true", help="skip the pose-network extra (config 5)")
    ap.add_argument("--no-shots", action="store_true", help="skip the 10 000-frame shot-detection extra (config 3)")
    ap.add_argument("--no-content", action="store_true", help="skip extra.flow_content (the step on other kinds of motion)")
    ap.add_argument("--no-stream10k", action="store_true", help="skip extra.stream_10k (10 000 distinct resident frames: 62 GB)")
    ap.add_argument("--cpu-pairs-per-thread", type=int, default=1)
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test on CPU (gloo): rendezvous + barriers + reduction around a dummy step")
    args = ap.parse_args()

    # The native libraries are not shipped to the GPU box (.gpurunignore): whoever runs first there compiles them.  In a child
    # process, so that this one (which may still have to spawn its ranks) loads nothing; ranks under an external launcher
    # leave it to local rank 0 and wait for the file lock.
    if not args.dry_run:
        import fcntl
        import subprocess
        with open(os.path.join(ROOT, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.ensure_built()"], cwd=ROOT)

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # one child per GPU, started before this process touches the GPU (it never does)
        from scannertools_amd.sharding import spawn_ranks
        return spawn_ranks(__file__, sys.argv[1:], args.gpus, args.master_port)
    if int(env_world or "1") != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s; start it as `python bench.py --gpus N` or under a "
                         "launcher whose world size equals --gpus\n" % (args.gpus, env_world))
        return 2
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
