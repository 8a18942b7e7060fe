"""Worker for tests/test_sharding.py: world_size-2 gloo job that shards a histogram stream,
gathers it on rank 0 and runs ShotBoundaries there."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scannertools_amd.sharding import gather_rows, rank_table, shard_range  # noqa: E402
from scannertools_amd.shot_detection import shot_boundaries  # noqa: E402


def main():
    out_path, n = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "shot_golden.npz"))
    hist = g["s0_n1000_b16__hist"][:n]
    a, b = shard_range(n, rank, world)
    local = torch.from_numpy(hist[a:b].copy())      # what this rank's Histogram op produced
    full = gather_rows(local, n, dst=0)
    if rank == 0:
        assert full.shape[0] == n and torch.equal(full, torch.from_numpy(hist))
        res = shot_boundaries(None, list(full.numpy()))
        np.save(out_path, np.array(res[0], np.int64))
    else:
        assert full is None
    # the per-rank evidence table: one row per rank in rank order on every rank; ranks that report the same device are an
    # error on EVERY rank (no rank may go on to print a plausible "2 GPU" line)
    table = rank_table("dev:%d" % rank, b - a, 1.5 * (rank + 1), "worker", require_distinct=True)
    assert [t["rank"] for t in table] == list(range(world)) and [t["frames"] for t in table] == [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    try:
        rank_table("dev:0", b - a, 1.0, "worker (same device)", require_distinct=True)
        raise AssertionError("two ranks on one device went unnoticed")
    except RuntimeError as e:
        assert "distinct" in str(e)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
