"""Worker for tests/test_colpart_gloo.py: one rank of a world_size-N gloo job running the
column-partition protocol on its shard and writing the result for the parent to check."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.colpart_fake import FakeHandle, OracleShardBackend  # noqa: E402
from tests.helpers import lp_amd  # noqa: E402


def main():
    out_dir, n, m, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    max_pivots = int(sys.argv[5])
    block = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lp = lp_amd()
    cp = lp.colpart if hasattr(lp, "colpart") else __import__("importlib").import_module(
        "linear-programming_amd.colpart")
    M, basis = lp.synth.tableau(n, m, seed)
    b, e = cp.partition(n + m, world)[rank]
    local = np.concatenate([M[:, b:e], M[:, -1:]], axis=1)
    sh = cp.Shard(torch, FakeHandle(local, basis), b, e, m + 1, world, torch.device("cpu"))
    tab = cp.ColumnPartitionedTableau([sh], cp.DistComm(dist), OracleShardBackend(), block=block)
    st, npiv = tab.solve(max_pivots=max_pivots, check_every=16)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), M=sh.handle.M, basis=sh.handle.basis,
             trace=np.array(sh.handle.trace, dtype=np.int64).reshape(-1, 2), status=st, npiv=npiv,
             begin=b, end=e)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
