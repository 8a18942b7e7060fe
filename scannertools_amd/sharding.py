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


def spawn_ranks(script, argv, n, port=0):
    """One process per GPU without an external launcher: start `n` children of `script argv` with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relay rank 0's stdout, return 0
    only if every rank exited 0.  The caller must not have touched the GPU (on this pool a process
    that has initialised HIP must not be replaced, and the children own the devices); this function
    imports neither torch nor the HIP library."""
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(codes):
        sys.stderr.write("%s: rank exit codes %s\n" % (os.path.basename(script), codes))
        sys.stderr.write(out0 or "")
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
