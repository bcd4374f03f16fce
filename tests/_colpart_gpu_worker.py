"""Worker for the multi-PROCESS column-partition GPU test: several ranks share cuda:0 (RCCL
refuses that, so the two exchanges are staged through the host with gloo -- a test set-up), each
rank drives its own shard through the real C ABI (mi355x_shard_*)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import lp_amd  # noqa: E402


def main():
    out_dir, n, m, seed, max_pivots = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lp = lp_amd()
    cp = __import__("importlib").import_module("linear-programming_amd.colpart")
    compact = len(sys.argv) > 6 and sys.argv[6] == "compact"
    shards = cp.synthetic_shards(torch, n, m, seed, [rank], world, 0, compact=compact)
    # multi-process runs use the blocked form (the default of bench.py --workload colpart)
    tab = cp.ColumnPartitionedTableau(shards, cp.DistComm(dist, stage_through_host=True), cp.HipBackend(),
                                      block=16)
    st, npiv = tab.solve(max_pivots=max_pivots, check_every=8)
    M, b = cp.download_shard(shards[0])
    cols = cp.shard_columns(shards[0]) if compact else np.arange(shards[0].col_begin, shards[0].col_end)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), M=M, basis=b, status=st, npiv=npiv, cols=cols)
    cp.destroy_shards(shards)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
