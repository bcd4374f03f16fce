"""What ONE shard of the 8-GPU column partition of config 5 costs per pivot on its own GPU: a
tableau with config 5's 32768 constraints and an eighth of its 65536 structural columns is exactly
the local work of one of eight shards (same rows, same slice width), run here as a single shard
through mi355x_colpart_* -- with device-local exchanges, over a one-rank RCCL communicator
(all-gather + all-reduce of the real sizes, no wire), and in the P2P mode (push / poll kernels on
the shard's own buffer).  The 8-GPU rate is bounded by this plus the wire time of the exchanges.
    python tools/shard_step_cost.py [pivots]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
lp = importlib.import_module("linear-programming_amd")
cp = importlib.import_module("linear-programming_amd.colpart")
L = lp.capi.lib()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 336                      # (a multiple of 16, 24 and 28)
if len(sys.argv) > 2:
    print("pivots per sweep: %d" % L.mi355x_tune_set_block(int(sys.argv[2])), flush=True)
ONLY = sys.argv[3] if len(sys.argv) > 3 else ""                          # substring of the mode's name: that mode only
n, m = 65536 // 8, 32768
seed = lp.synth.seed_for(5)
# (name, MI355X_COLPART_FORCE_RCCL, exchange mode, persistent block launch: 0 on / 1 off, exchange A against the own buffer)
for name, force, mode, la_off, hop in (
        ("device-local exchanges", "0", 0, 1, 0), ("one-rank RCCL: all-gather + int64 all-reduce", "1", 0, 1, 0),
        ("one-rank RCCL: all-gather + rooted broadcast", "1", 1, 1, 0),
        ("P2P push, four launches per step", "1", 3, 1, 0), ("P2P push, two launches per step", "1", 2, 1, 0),
        ("P2P push, ONE persistent launch per block", "1", 2, 0, 1),
        ("the same without exchange A (a lone shard)", "1", 2, 0, 0)):
    if ONLY and ONLY not in name:
        continue
    os.environ["MI355X_COLPART_FORCE_RCCL"] = force
    L.mi355x_tune_set_colpart_exchange(mode)
    L.mi355x_tune_set_shard_la_block(la_off)
    L.mi355x_tune_set_shard_self_hop(hop)
    tab = cp.NativeColumnPartition.synthetic(n, m, seed, 1)
    L.mi355x_tune_set_colpart_exchange(0)
    L.mi355x_tune_set_shard_la_block(0)
    L.mi355x_tune_set_shard_self_hop(0)
    tab.solve_async(336, reset=True); tab.sync()
    best = 1e30
    for rep in range(3):
        t0 = time.perf_counter()
        tab.solve_async(K); st, done = tab.sync()
        best = min(best, time.perf_counter() - t0)
    stats = tab.la_stats()
    print("%-48s %7.1f us per pivot (%d pivots, best of 3, status %d; persistent blocks %d, lost %d)"
          % (name, best / K * 1e6, K, st, stats["blocks"], stats["losses"]), flush=True)
    tab.close()
