"""Worker for the multi-PROCESS test of exchange mode 2 (P2P push): several OS processes share
cuda:0, every rank owns one shard behind mi355x_colpart_* WITHOUT any communicator, the ranks map
each other's fine-grained exchange buffers through IPC handles (gathered here over gloo), and the
whole solve then runs in the library -- the shards' kernels of the different processes run
concurrently on the GPU and meet only through the granules they write into each other's buffers."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import lp_amd  # noqa: E402


def main():
    out_dir, n, m, seed, max_pivots = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lp = lp_amd()
    cp = __import__("importlib").import_module("linear-programming_amd.colpart")
    L = lp.capi.lib()
    mode = int(sys.argv[6]) if len(sys.argv) > 6 else 2          # 2: two launches per step where it applies, 3: four
    split = int(sys.argv[7]) if len(sys.argv) > 7 else 0          # 2: the multi-workgroup look-ahead step at any size
    la_block = int(sys.argv[8]) if len(sys.argv) > 8 else 1       # 0: the persistent block launch (k_shard_la_block), 1: never
    L.mi355x_tune_set_colpart_exchange(mode)
    L.mi355x_tune_set_shard_la_split(split)
    L.mi355x_tune_set_shard_la_block(la_block)
    tab = cp.NativeColumnPartition.synthetic_rank(n, m, seed, world, rank, 0, None)
    L.mi355x_tune_set_colpart_exchange(0)
    L.mi355x_tune_set_shard_la_block(0)
    handles = [None] * world
    dist.all_gather_object(handles, tab.p2p_handle())
    tab.p2p_connect(b"".join(handles))
    dist.barrier()
    st, k = tab.solve(max_pivots=max_pivots)
    _, basis, _, last_col = tab.download(matrix=False, last_row=False)
    trace = tab.trace(max(k, 1))
    stats = tab.la_stats()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), status=st, npiv=k, basis=basis, last_col=last_col, trace=trace,
             la_blocks=stats["blocks"], la_losses=stats["losses"])
    dist.barrier()                       # nobody unmaps a buffer a peer may still be writing to
    tab.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
