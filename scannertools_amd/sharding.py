"""Frame-stream sharding across the GPUs of one node.

The path shards embarrassingly: every frame's histogram and every pair's flow field is
independent, so rank g of G owns the contiguous rows [g*N/G, (g+1)*N/G) and -- for the
OpticalFlow op's stencil {0,1} -- additionally reads ONE halo frame past its last row.  No
collective touches the data path.  The only exchange is the optional gather of the per-frame
histograms (<= 3 KB per frame) onto rank 0 for the stream-global ShotBoundaries op; it runs over
torch.distributed ("nccl" = RCCL on GPU tensors, "gloo" on CPU tensors).
"""
import os
import socket
import subprocess
import sys

import numpy as np


def spawn_ranks(script, argv, n, port=0, timeout=None, grace=5.0):
    """One process per GPU without an external launcher: start `n` children of `script argv` with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relay rank 0's stdout, return 0
    only if every rank exited 0.  The caller must not have touched the GPU (on this pool a process
    that has initialised HIP must not be replaced, and the children own the devices); this function
    imports neither torch nor the HIP library.

    All children are watched together: the first non-zero exit (a rank that runs out of memory or finds
    no device) terminates the others -- they would otherwise sit in the rendezvous or a barrier until
    torch's own timeout of ten minutes or more -- and so does `timeout` seconds overall (default: the
    environment's ST_SPAWN_TIMEOUT, else 3600) or a SIGTERM / SIGINT delivered to this process.  Rank 0's
    stdout is collected by a reader thread while it runs, so a full pipe never blocks it."""
    import signal
    import threading
    import time
    if timeout is None:
        timeout = float(os.environ.get("ST_SPAWN_TIMEOUT", "3600"))
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    procs, chunks = [], []

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + grace
        for p in procs:
            try:
                p.wait(max(0.0, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    class _Stop(Exception):
        pass

    def on_signal(signum, frame):
        raise _Stop("signal %d" % signum)

    in_main = threading.current_thread() is threading.main_thread()
    saved = {}
    late = []   # signals that arrive while the ranks are being stopped: recorded, never raised out of the clean-up
    reader, why = None, None
    try:
        if in_main:   # installed inside the try: a signal between installation and the loop is caught like any other
            for sig in (signal.SIGTERM, signal.SIGINT):
                saved[sig] = signal.signal(sig, on_signal)
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        t_end = time.monotonic() + timeout
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                why = "rank %d exited with code %d" % next((r, c) for r, c in enumerate(codes) if c not in (None, 0))
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > t_end:
                why = "no result within %.0f s" % timeout
                break
            time.sleep(0.05)
    except _Stop as e:
        why = str(e)
    finally:
        try:
            for sig in saved:   # a second Ctrl-C during the grace wait must not skip the kill escalation below
                signal.signal(sig, lambda signum, frame: late.append(signum))
            stop_all()
        finally:
            for sig, h in saved.items():
                signal.signal(sig, h)
        if late and not why:
            why = "signal %d" % late[0]
    if reader is not None:
        reader.join(grace)
    out0 = "".join(c for c in chunks if c)
    codes = [p.returncode for p in procs]
    if why or any(codes):
        sys.stderr.write("%s: %s; rank exit codes %s\n" % (os.path.basename(script), why or "failed", codes))
        sys.stderr.write(out0)
        return 1
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


def shard_range(n, rank, world):
    """Contiguous row range [start, end) of `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def flow_shard(n, rank, world, stencil=(0, 1)):
    """(rows, frames): the output rows `rank` computes and the input frame range it must hold
    (rows + halo).  Windows that reach outside [0, n) clamp to the edge frame, as in Scanner."""
    a, b = shard_range(n, rank, world)
    if a == b:
        return (a, b), (a, a)
    lo = max(0, a + min(min(stencil), 0))
    hi = min(n, b + max(max(stencil), 0))
    return (a, b), (lo, hi)


def local_pairs(rows, frames, n, stencil=(0, 1)):
    """(p,2) indices into the rank's LOCAL frame array for each of its rows."""
    a, b = rows
    lo, _ = frames
    out = np.empty((b - a, 2), np.int32)
    for i, r in enumerate(range(a, b)):
        out[i, 0] = min(max(r + stencil[0], 0), n - 1) - lo
        out[i, 1] = min(max(r + stencil[1], 0), n - 1) - lo
    return out


def gather_rows(local, n_total, dst=0, group=None):
    """Concatenate every rank's rows (first dimension) in rank order on `dst`.

    local: tensor (rows_of_this_rank, ...) with identical trailing shape on all ranks.  Returns the
    (n_total, ...) tensor on `dst`, None elsewhere.  Shards may be uneven (padded to the largest).
    Without an initialised process group (single GPU) returns `local`."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    biggest = max(e - s for s, e in sizes)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    if dist.get_backend(group) == "nccl":
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        if rank != dst:
            return None
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst, group=group)
        if rank != dst:
            return None
    return torch.cat([bufs[r][: sizes[r][1] - sizes[r][0]] for r in range(world)], dim=0)


def device_id_string(torch, dev):
    """PCI bus id of a torch device (domain:bus:device), or its uuid where the properties carry no bus id."""
    pr = torch.cuda.get_device_properties(dev)
    bus = getattr(pr, "pci_bus_id", None)
    if bus is not None:
        return "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), bus, getattr(pr, "pci_device_id", 0))
    return str(getattr(pr, "uuid", dev))


def rank_table(device, frames, ms, what="", require_distinct=False, stream=None):
    """The per-rank evidence of a multi-GPU run: every rank contributes (its device id, the frames it processed, its own
    time in ms); returns the list of {"rank", "device", "frames", "ms"} in rank order ON EVERY RANK, prints it as a table on
    rank 0 (stderr by default, so that a launcher's single JSON line stays alone on stdout), and -- when
    `require_distinct` -- raises RuntimeError on every rank if two ranks ran on the same device (a mis-set
    LOCAL_RANK / visible-device mask would otherwise produce a plausible-looking "N GPU" number from fewer GPUs).
    Works without a process group (one row)."""
    import torch.distributed as dist
    rank, world = 0, 1
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    # bus ids repeat from host to host: the device id is qualified by the host name, so that a valid multi-node launch
    # (same domain:bus:device on two machines) is not mistaken for two ranks on one GPU
    dev = str(device)
    if "@" not in dev:
        dev = "%s@%s" % (dev, socket.gethostname())
    mine = (rank, dev, int(frames), float(ms))
    rows = [mine]
    if world > 1:
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        rows.sort()
    table = [{"rank": r, "device": d, "frames": f, "ms": m} for r, d, f, m in rows]
    if rank == 0:
        out = stream or sys.stderr
        out.write("%s%d rank(s)\n  rank  device                            frames        ms\n" % (what + ": " if what else "", world))
        for t in table:
            out.write("  %4d  %-32s %7d  %9.3f\n" % (t["rank"], t["device"], t["frames"], t["ms"]))
        out.flush()
    distinct = len({t["device"] for t in table})
    if require_distinct and distinct != world:
        raise RuntimeError("%d ranks ran on %d distinct device(s): %s" % (world, distinct, [t["device"] for t in table]))
    return table
