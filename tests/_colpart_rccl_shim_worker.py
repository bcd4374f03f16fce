"""Worker of tests/test_colpart_rccl_shim.py: ONE rank of the library's one-process-per-GPU column
partition (mi355x_colpart_create_synthetic_rank -> ncclCommInitRank -> the C++ per-pivot loop with its
two collectives per pivot), several such processes sharing cuda:0.  The collectives come from the
stand-in tests/rccl_shim.c, loaded here under the name the library looks for BEFORE the library
resolves RCCL; torch (which would bring its own librccl into the process) is deliberately not imported.
Rendezvous of the 128-byte id: a file."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir, shim, rank, world = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    n, m, seed, cap, exchange = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])
    shim_lib = ctypes.CDLL(shim, mode=ctypes.RTLD_GLOBAL)
    assert "torch" not in sys.modules
    from tests.helpers import lp_amd
    lp = lp_amd()
    cp = __import__("importlib").import_module("linear-programming_amd.colpart")
    assert "torch" not in sys.modules
    L = lp.capi.lib()
    id_path = os.path.join(out_dir, "unique_id")
    if rank == 0:
        uid = cp.NativeColumnPartition.rccl_unique_id()
        assert uid[:10] == b"MI355XSHIM", "the library resolved another librccl than the stand-in"
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(id_path + ".tmp", id_path)
    else:
        t0 = time.time()
        while not os.path.exists(id_path):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no unique id after 120 s" % rank)
            time.sleep(0.01)
        uid = open(id_path, "rb").read()
    L.mi355x_tune_set_colpart_exchange(exchange)
    tab = cp.NativeColumnPartition.synthetic_rank(n, m, seed, world, rank, 0, uid)
    L.mi355x_tune_set_colpart_exchange(0)
    info = tab.info()
    assert info["uses_rccl"] and info["n_shards"] == world, info
    st, k = tab.solve(max_pivots=cap)
    # a second, capped call carries on where the first stopped (the ranks agree on every chunk)
    st2, k2 = (st, 0) if st != lp.capi.MI_MAX_PIVOTS else tab.solve(max_pivots=0)
    _, basis, _, last_col = tab.download(matrix=False, last_row=False)
    trace = tab.trace(max(k + k2, 1))
    stats = np.zeros(4, dtype=np.int64)
    shim_lib.rccl_shim_stats(stats.ctypes.data_as(ctypes.c_void_p))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), status=st, npiv=k, status2=st2, npiv2=k2, basis=basis,
             last_col=last_col, trace=trace, stats=stats, block=tab.block_size())
    tab.close()


if __name__ == "__main__":
    main()
