"""One tableau column-partitioned across GPUs (BASELINE config 5).

Every shard (= rank = GPU) owns a contiguous block of the var_count non-RHS columns of ALL rows
plus its own copy of the RHS column, which it updates redundantly.  One pivot of
n-solve-tableau (src/simplex.lisp:453-461) then needs exactly two exchanges:

    price local slice --(A: all-gather 16 B/shard)--> same global entering column everywhere
    owner contributes the column --(B: int64 SUM all-reduce of rows values == broadcast)-->
    ratio test (redundant, identical) + normalise own row slice + rank-1 update of own slice

A row partition would need the (3x longer) pivot row broadcast AND a reduction for the ratio
test.  Exchange B is an integer all-reduce of bit patterns (owner's bits + zeros) instead of a
rooted broadcast so that no rank needs to know the owner on the host: the whole loop is enqueued
on the stream without a single host synchronisation; status is read back every `check_every`
pivots only.  The global arg-min is the lexicographic (value, global column) minimum, i.e. the
sequential lowest-index strict minimum, so the pivot sequence is bit-identical to the
single-GPU solver's (tested with N logical shards on one device and with gloo on CPU).

The protocol (this file) is shared by three set-ups:
  * torch.distributed, one shard per rank, NCCL(=RCCL over xGMI) on GPUs -- production;
  * N logical shards inside one process on one GPU (collectives = local tensor ops);
  * torch.distributed with gloo on CPU tensors and a test-supplied compute backend
    (tests/test_colpart_gloo.py) -- protocol coverage without GPUs.
"""
import ctypes
import json
import time

import numpy as np

from . import capi, synth


def partition(var_count, n_shards):
    """Contiguous column blocks [begin, end), sizes differing by at most one."""
    base, extra = divmod(var_count, n_shards)
    if base == 0:
        raise ValueError("more shards (%d) than columns (%d)" % (n_shards, var_count))
    out, b = [], 0
    for r in range(n_shards):
        e = b + base + (1 if r < extra else 0)
        out.append((b, e))
        b = e
    return out


class Shard:
    """Local state of one shard: the compute handle plus the exchange buffers (torch tensors on
    the shard's device)."""

    def __init__(self, torch, handle, col_begin, col_end, rows, n_shards, device):
        self.handle = handle
        self.col_begin, self.col_end = int(col_begin), int(col_end)
        self.rows, self.n_shards = int(rows), int(n_shards)
        self.send = torch.zeros(2, dtype=torch.float64, device=device)
        self.gathered = torch.zeros(2 * n_shards, dtype=torch.float64, device=device)
        self.bits = torch.zeros(rows, dtype=torch.int64, device=device)
        self.ec = torch.full((1,), -1, dtype=torch.int64, device=device)


class HipBackend:
    """The three local steps on the GPU through the C ABI (mi355x_shard_*)."""

    def __init__(self, is_max=True, fp_factor=1024.0):
        self.is_max, self.f = int(bool(is_max)), float(fp_factor)
        self.L = capi.lib()

    def price(self, sh):
        capi.check(self.L.mi355x_shard_price(sh.handle, self.is_max, sh.col_begin,
                                             ctypes.c_void_p(sh.send.data_ptr())), "mi355x_shard_price")

    def contribute(self, sh):
        capi.check(self.L.mi355x_shard_contribute(
            sh.handle, ctypes.c_void_p(sh.gathered.data_ptr()), sh.n_shards, sh.col_begin, self.f,
            ctypes.c_void_p(sh.bits.data_ptr()), ctypes.c_void_p(sh.ec.data_ptr())),
            "mi355x_shard_contribute")

    def pivot(self, sh):
        capi.check(self.L.mi355x_shard_pivot(sh.handle, ctypes.c_void_p(sh.bits.data_ptr()),
                                             ctypes.c_void_p(sh.ec.data_ptr()), self.f),
                   "mi355x_shard_pivot")

    # blocked form: step j of a block does not touch the shard's slice of the
    # tableau; sweep() applies the pending pivots in one pass
    def la_contribute(self, sh, j):
        capi.check(self.L.mi355x_shard_la_contribute(
            sh.handle, int(j), ctypes.c_void_p(sh.gathered.data_ptr()), sh.n_shards, sh.col_begin, self.f,
            ctypes.c_void_p(sh.bits.data_ptr()), ctypes.c_void_p(sh.ec.data_ptr())),
            "mi355x_shard_la_contribute")

    def la_pivot(self, sh, j):
        capi.check(self.L.mi355x_shard_la_pivot(sh.handle, int(j), ctypes.c_void_p(sh.bits.data_ptr()),
                                                ctypes.c_void_p(sh.ec.data_ptr()), self.f),
                   "mi355x_shard_la_pivot")

    def sweep(self, sh):
        capi.check(self.L.mi355x_shard_sweep(sh.handle), "mi355x_shard_sweep")

    def reset(self, sh, max_pivots=0):
        capi.check(self.L.mi355x_tab_reset(sh.handle, int(max_pivots)), "mi355x_tab_reset")

    def status(self, sh):
        """(status, n_pivots) -- synchronises the shard's stream."""
        n = ctypes.c_int64(0)
        rc = capi.check(self.L.mi355x_tab_sync(sh.handle, ctypes.byref(n)), "mi355x_tab_sync")
        return rc, int(n.value)


class DistComm:
    """Exchanges over torch.distributed (nccl == RCCL on ROCm, or gloo on CPU): one local shard.

    stage_through_host=True is a test set-up only (several ranks sharing one GPU, where RCCL
    refuses to run): device buffers are copied to the host, exchanged with gloo and copied back."""

    def __init__(self, dist, group=None, stage_through_host=False):
        self.dist, self.group, self.stage = dist, group, stage_through_host

    def gather(self, shards):
        (sh,) = shards
        if self.stage:
            out = sh.gathered.cpu()
            self.dist.all_gather_into_tensor(out, sh.send.cpu(), group=self.group)
            sh.gathered.copy_(out)
        else:
            self.dist.all_gather_into_tensor(sh.gathered, sh.send, group=self.group)

    def reduce(self, shards):
        (sh,) = shards
        if self.stage:
            buf = sh.bits.cpu()
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
            sh.bits.copy_(buf)
        else:
            self.dist.all_reduce(sh.bits, op=self.dist.ReduceOp.SUM, group=self.group)


class LocalComm:
    """All shards live in this process (N logical shards on one device): the exchanges are
    plain tensor ops with exactly the collectives' semantics."""

    def __init__(self, torch):
        self.torch = torch

    def gather(self, shards):
        g = self.torch.cat([sh.send for sh in shards])
        for sh in shards:
            sh.gathered.copy_(g)

    def reduce(self, shards):
        total = self.torch.stack([sh.bits for sh in shards]).sum(dim=0)
        for sh in shards:
            sh.bits.copy_(total)


class ColumnPartitionedTableau:
    """The driver: `shards` are this process's shards (one under torch.distributed).

    block > 1: blocked pivoting -- the same two exchanges per pivot, but the shards' slices of the
    tableau are swept once per `block` pivots (<= 16) instead of updated after every pivot; the
    pending pivots are chained through on what a step reads.  Same pivots, same bits."""

    MAX_BLOCK = 16

    def __init__(self, shards, comm, backend, block=1):
        self.shards, self.comm, self.backend = list(shards), comm, backend
        self.block = max(1, min(int(block), self.MAX_BLOCK))
        self._j = 0                                  # steps of the current block already enqueued

    def step(self):
        """Enqueue one pivot; no host synchronisation."""
        if self.block > 1:
            return self._step_blocked()
        for sh in self.shards:
            self.backend.price(sh)
        self.comm.gather(self.shards)
        for sh in self.shards:
            self.backend.contribute(sh)
        self.comm.reduce(self.shards)
        for sh in self.shards:
            self.backend.pivot(sh)

    def _step_blocked(self):
        j = self._j
        for sh in self.shards:
            self.backend.price(sh)
        self.comm.gather(self.shards)
        for sh in self.shards:
            self.backend.la_contribute(sh, j)
        self.comm.reduce(self.shards)
        for sh in self.shards:
            self.backend.la_pivot(sh, j)
        self._j = j + 1
        if self._j == self.block:
            self.flush()

    def flush(self):
        """Apply the pending pivots of an unfinished block (no-op when there are none)."""
        if self.block > 1 and self._j > 0:
            for sh in self.shards:
                self.backend.sweep(sh)
            self._j = 0

    def reset(self, max_pivots=0):
        self.flush()
        for sh in self.shards:
            self.backend.reset(sh, max_pivots)

    def run(self, n_iterations):
        """Enqueue n_iterations iterations (an iteration after termination is a no-op on the
        device, so enqueueing too many is harmless)."""
        for _ in range(n_iterations):
            self.step()

    def status(self):
        self.flush()                                 # the tableau is whole whenever the host looks
        sts = [self.backend.status(sh) for sh in self.shards]
        assert all(s == sts[0] for s in sts), "shards disagree: %r" % (sts,)
        return sts[0]

    def solve(self, max_pivots=0, check_every=64):
        """n-solve-tableau: iterate until optimal / unbounded / the pivot cap (enforced on the
        device).  Returns (status, n_pivots)."""
        self.reset(max_pivots)
        while True:
            self.run(check_every)
            st, n = self.status()
            if st != capi.MI_RUNNING:
                return st, n


# ------------------------------------------------------------------ construction helpers (GPU)
def synthetic_shards(torch, n_vars, n_cons, seed, shard_ids, n_shards, device_index, compact=False):
    """Shards `shard_ids` of the synthetic LP, generated in HBM on cuda:device_index.

    compact=False: every shard holds a fixed block of ALL var_count logical columns.
    compact=True : only the non-basic columns are distributed (initially the n_vars structural
    columns; the slack columns start basic and are stored nowhere) -- 1/3 less memory and
    traffic per shard at n_vars = 2 n_cons."""
    L = capi.lib()
    parts = partition(n_vars if compact else n_vars + n_cons, n_shards)
    dev = torch.device("cuda", device_index)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = []
    for r in shard_ids:
        b, e = parts[r]
        h = ctypes.c_void_p()
        capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n_vars, n_cons, seed, b, e,
                                                 device_index), "mi355x_tab_create_synthetic")
        capi.check(L.mi355x_tab_set_stream(h, ctypes.c_void_p(stream), 0), "mi355x_tab_set_stream")
        if compact:
            cols = np.arange(b, e, dtype=np.int64)
            capi.check(L.mi355x_shard_set_compact(h, n_vars + n_cons,
                                                  cols.ctypes.data_as(ctypes.c_void_p)),
                       "mi355x_shard_set_compact")
        out.append(Shard(torch, h, b, e, n_cons + 1, n_shards, dev))
    return out


def shard_columns(sh):
    """Global logical column currently stored in every local slot of a compact shard."""
    L = capi.lib()
    cols = ctypes.c_int64(0)
    capi.check(L.mi355x_tab_shape(sh.handle, None, ctypes.byref(cols), None), "shape")
    out = np.empty(cols.value - 1, dtype=np.int64)
    capi.check(L.mi355x_shard_columns(sh.handle, out.ctypes.data_as(ctypes.c_void_p)), "columns")
    return out


def assemble_compact(shards, var_count):
    """Rebuild the dense logical tableau from compact shards: stored columns go to their logical
    positions, basic columns are the unit vectors the basis says they are."""
    parts = [download_shard(sh) for sh in shards]
    rows = parts[0][0].shape[0]
    M = np.zeros((rows, var_count + 1))
    basis = parts[0][1]
    for i, bcol in enumerate(basis):
        M[i, bcol] = 1.0
    for sh, (P, _) in zip(shards, parts):
        M[:, shard_columns(sh)] = P[:, :-1]
    M[:, -1] = parts[0][0][:, -1]
    return M, basis


def download_shard(sh):
    """(matrix rows x (local cols + 1), basis) of a GPU shard."""
    L = capi.lib()
    rows, cols = ctypes.c_int64(0), ctypes.c_int64(0)
    capi.check(L.mi355x_tab_shape(sh.handle, ctypes.byref(rows), ctypes.byref(cols), None), "shape")
    M = np.empty((rows.value, cols.value))
    b = np.empty(rows.value - 1, dtype=np.int64)
    capi.check(L.mi355x_tab_download(sh.handle, M.ctypes.data_as(ctypes.c_void_p),
                                     b.ctypes.data_as(ctypes.c_void_p), None, None), "download")
    return M, b


def destroy_shards(shards):
    for sh in shards:
        if sh.handle:
            capi.lib().mi355x_tab_destroy(sh.handle)
            sh.handle = None


# ------------------------------------------------------------------ the C-ABI driver
class NativeColumnPartition:
    """mi355x_colpart_*: the same protocol driven from C++ inside the library (the per-pivot loop,
    the RCCL communicators and both exchanges live there; what the Lisp host reaches with
    `:devices n`).  One process: n_devices shards, one GPU each when that many are visible
    (RCCL, one host thread per shard), logical shards on device 0 otherwise."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_arrays(cls, matrix, basis, n_devices):
        M = np.ascontiguousarray(matrix, dtype=np.float64)
        b = np.ascontiguousarray(basis, dtype=np.int64)
        h = ctypes.c_void_p()
        capi.check(capi.lib().mi355x_colpart_create(ctypes.byref(h), M.shape[0], M.shape[1],
                                                    M.ctypes.data_as(ctypes.c_void_p),
                                                    b.ctypes.data_as(ctypes.c_void_p), int(n_devices)),
                   "mi355x_colpart_create")
        obj = cls(h)
        obj.rows, obj.cols = M.shape
        return obj

    @classmethod
    def synthetic(cls, n_vars, n_cons, seed, n_devices):
        h = ctypes.c_void_p()
        capi.check(capi.lib().mi355x_colpart_create_synthetic(ctypes.byref(h), n_vars, n_cons, seed,
                                                              int(n_devices)), "mi355x_colpart_create_synthetic")
        obj = cls(h)
        obj.rows, obj.cols = n_cons + 1, n_vars + n_cons + 1
        return obj

    @classmethod
    def synthetic_rank(cls, n_vars, n_cons, seed, world, rank, device, unique_id):
        """One process per GPU: `unique_id` = the 128 bytes rank 0 got from rccl_unique_id(), or
        None in exchange mode 2 (no communicator: connect the ranks with p2p_handle / p2p_connect)."""
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
        capi.check(capi.lib().mi355x_colpart_create_synthetic_rank(ctypes.byref(h), n_vars, n_cons, seed,
                                                                   int(world), int(rank), int(device), buf),
                   "mi355x_colpart_create_synthetic_rank")
        obj = cls(h)
        obj.rows, obj.cols = n_cons + 1, n_vars + n_cons + 1
        return obj

    def block_size(self):
        """Pivots per sweep of a shard's slice (16, or a wide block of 24 / 28 for large shards)."""
        return int(capi.lib().mi355x_colpart_block_size(self._h))

    def is_compact(self):
        """True: compact shards (non-basic columns only); False: dense shards (every logical column)."""
        return bool(capi.lib().mi355x_colpart_is_compact(self._h))

    def la_stats(self):
        """Exchange mode 2, blocked: blocks enqueued through the persistent block launch (k_shard_la_block),
        exchanges it lost, demoted right now, next block persistent."""
        out = (ctypes.c_int64 * 4)()
        capi.check(capi.lib().mi355x_colpart_la_stats(self._h, out), "mi355x_colpart_la_stats")
        return {"blocks": int(out[0]), "losses": int(out[1]), "demoted": bool(out[2]), "live": bool(out[3])}

    def p2p_handle(self):
        """Exchange mode 2, one process per GPU: the 64-byte IPC handle of this rank's exchange buffer."""
        buf = ctypes.create_string_buffer(64)
        capi.check(capi.lib().mi355x_colpart_p2p_handle(self._h, buf), "mi355x_colpart_p2p_handle")
        return buf.raw

    def p2p_connect(self, handles):
        """`handles`: every rank's 64-byte handle, concatenated in rank order."""
        buf = ctypes.create_string_buffer(bytes(handles), len(handles))
        capi.check(capi.lib().mi355x_colpart_p2p_connect(self._h, buf), "mi355x_colpart_p2p_connect")

    @staticmethod
    def rccl_unique_id():
        buf = ctypes.create_string_buffer(128)
        capi.check(capi.lib().mi355x_rccl_unique_id(buf), "mi355x_rccl_unique_id")
        return buf.raw

    def info(self):
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        capi.check(capi.lib().mi355x_colpart_info(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "info")
        return {"n_shards": a.value, "n_devices_used": b.value, "uses_rccl": bool(c.value)}

    def solve(self, is_max=True, fp_tolerance=1024, max_pivots=0):
        n = ctypes.c_int64(0)
        rc = capi.check(capi.lib().mi355x_colpart_solve(self._h, int(bool(is_max)), float(fp_tolerance),
                                                        int(max_pivots), ctypes.byref(n)), "mi355x_colpart_solve")
        return rc, int(n.value)

    def solve_two_phase(self, main_objective_row, main_is_max=True, fp_tolerance=1024):
        """n-solve-tableau's two-phase branch with THIS handle as the artificial tableau
        (mi355x_colpart_solve_two_phase).  Returns (status, (phase-1 pivots, phase-2 pivots), main)
        with `main` the solved main tableau as a new NativeColumnPartition (None when phase 1 did
        not end in a feasible basis)."""
        obj = np.ascontiguousarray(main_objective_row, dtype=np.float64)
        npv = (ctypes.c_int64 * 2)()
        h = ctypes.c_void_p()
        rc = capi.check(capi.lib().mi355x_colpart_solve_two_phase(
            self._h, int(obj.shape[0]), obj.ctypes.data_as(ctypes.c_void_p), int(bool(main_is_max)),
            float(fp_tolerance), npv, ctypes.byref(h)), "mi355x_colpart_solve_two_phase")
        main = None
        if h:
            main = NativeColumnPartition(h)
            main.rows, main.cols = self.rows, int(obj.shape[0])
        return rc, (int(npv[0]), int(npv[1])), main

    def solve_async(self, n_pivots, is_max=True, fp_tolerance=1024, reset=False):
        capi.check(capi.lib().mi355x_colpart_solve_async(self._h, int(bool(is_max)), float(fp_tolerance),
                                                         int(n_pivots), int(bool(reset))), "mi355x_colpart_solve_async")

    def exchange_timing(self, stride, max_samples=256):
        """HIP-event brackets around the two collectives of every stride-th pivot (RCCL shards)."""
        capi.check(capi.lib().mi355x_colpart_exchange_timing_enable(self._h, int(stride), int(max_samples)),
                   "mi355x_colpart_exchange_timing_enable")

    def exchange_timing_read(self):
        n, ag, ar = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        capi.check(capi.lib().mi355x_colpart_exchange_timing_read(
            self._h, ctypes.byref(n), ctypes.byref(ag), ctypes.byref(ar)), "mi355x_colpart_exchange_timing_read")
        return n.value, ag.value, ar.value

    def sync(self):
        n = ctypes.c_int64(0)
        rc = capi.check(capi.lib().mi355x_colpart_sync(self._h, ctypes.byref(n)), "mi355x_colpart_sync")
        return rc, int(n.value)

    def download(self, matrix=True, last_row=True):
        """(matrix, basis, objective row, RHS column); a rank of the one-process-per-GPU form holds
        one shard only: matrix=False, last_row=False there (basis and RHS column are replicated)."""
        M = np.empty((self.rows, self.cols)) if matrix else None
        b = np.empty(self.rows - 1, dtype=np.int64)
        lr, last_col = (np.empty(self.cols) if last_row else None), np.empty(self.rows)
        capi.check(capi.lib().mi355x_colpart_download(
            self._h, M.ctypes.data_as(ctypes.c_void_p) if matrix else None, b.ctypes.data_as(ctypes.c_void_p),
            lr.ctypes.data_as(ctypes.c_void_p) if last_row else None, last_col.ctypes.data_as(ctypes.c_void_p)),
            "mi355x_colpart_download")
        return M, b, lr, last_col

    def trace(self, cap):
        ec = np.empty(max(cap, 1), dtype=np.int64); cr = np.empty(max(cap, 1), dtype=np.int64)
        n = ctypes.c_int64(0)
        capi.check(capi.lib().mi355x_colpart_trace(self._h, ec.ctypes.data_as(ctypes.c_void_p),
                                                   cr.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n)), "trace")
        k = min(int(n.value), cap)
        return np.stack([ec[:k], cr[:k]], axis=1)

    def close(self):
        h, self._h = self._h, None
        if h:
            capi.lib().mi355x_colpart_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ bench.py --workload colpart
def _timed_pivots(tab, k, torch, dist, world, sync_ranks=True, red_dev="cuda"):
    """k pivots through the library's loop, bracketed as bench.py's contract asks: barrier +
    device synchronisation on both sides, MAX over ranks."""
    if world > 1 and sync_ranks:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tab.solve_async(k)
    st, done = tab.sync()
    torch.cuda.synchronize()
    if world > 1 and sync_ranks:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1 and sync_ranks:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, st, done


def _state_digest(tab):
    """Digest of what every rank holds of the solved state (basis + RHS column, both replicated):
    equal digests <=> the same pivots were made on the same numbers."""
    import hashlib
    _, basis, _, last_col = tab.download(matrix=False, last_row=False)
    return hashlib.sha256(basis.tobytes() + last_col.tobytes()).hexdigest()[:16]


def one_shard_baseline(n, m, seed, device, steps, warmup, steady_pivots=0):
    """The SAME tableau as ONE shard on ONE GPU through the same driver (mi355x_colpart_*, device-
    local exchanges): the denominator of the north star's '>= 6 x pivots/sec at 8 GPUs vs 1'."""
    import torch
    tab = NativeColumnPartition.synthetic_rank(n, m, seed, 1, 0, device, bytes(128))
    try:
        blk = tab.block_size()
        if steps == "blocks":                     # whole blocks of this handle: 1 warm-up block, 4 timed
            warmup, steps = blk, 4 * blk
        if steady_pivots == "auto":
            steady_pivots = 8 * blk if steps < 4 * blk else 0
        tab.solve_async(warmup, reset=True)
        tab.sync()
        dt, st, done = _timed_pivots(tab, steps, torch, None, 1)
        if st != capi.MI_RUNNING:
            raise SystemExit("colpart baseline: LP terminated early (status %d after %d pivots)" % (st, done))
        rec = {"what": "the same %d x %d tableau as ONE shard on one GPU, same driver (mi355x_colpart_*), "
                       "same %d warm-up and %d timed pivots" % (m + 1, n + m + 1, warmup, steps),
               "value": steps / dt, "unit": "pivots/s", "us_per_pivot": dt / steps * 1e6,
               "pivots_per_sweep": blk, "timed_pivots": steps}
        if steady_pivots:
            dt2, st, done = _timed_pivots(tab, steady_pivots, torch, None, 1)
            rec["steady_state_pivots_per_s"] = steady_pivots / dt2
            rec["steady_state_pivots"] = steady_pivots
    finally:
        tab.close()
    torch.cuda.empty_cache()
    return rec


def bench(args, rank, local_rank, world, progress=None):
    """ONE dense LP (BASELINE config 5: 65536 vars x 32768 constraints, 32769 x 98305 f64 =
    25.8 GB) column-partitioned over `world` ranks, strong scaling: K pivots timed with both
    per-pivot exchanges in the timed region.  The per-pivot loop runs in the library
    (mi355x_colpart_*: RCCL collectives issued from C++ on the shard's stream); this function only
    distributes the RCCL id, starts the K pivots and waits.  What one record holds beyond `value`
    (so that it answers the north star's '>= 6 x at 8 GPUs vs 1' by itself, whatever the driver's
    N = 1 line measured):
      * `one_gpu_same_workload` -- the same tableau as one shard on rank 0's GPU, same run, before
        the collective leg -- and `speedup_vs_one_gpu`;
      * `steady_state_pivots_per_s` -- further FULL blocks timed right after the timed region when
        the requested steps are fewer than four blocks (a 20-step region is one block of 16 plus one
        of 4, each paying a full sweep);
      * `exchange_modes` -- the same K pivots with the entering column travelling as a rooted
        ncclBroadcast instead of the int64 all-reduce (mi355x_tune_set_colpart_exchange).
    Test set-up (ranks sharing one GPU, where RCCL refuses to run: BENCH_DIST_BACKEND=gloo): the
    Python protocol driver above with the exchanges staged through the host."""
    import os
    import torch
    import torch.distributed as dist
    n, m = 65536, 32768
    if getattr(args, "colpart_vars", None):
        n, m = args.colpart_vars, args.colpart_vars // 2
    seed = synth.seed_for(5)
    staged = world > 1 and dist.get_backend() != "nccl"          # test hook: ranks share one GPU (no RCCL between them)
    red_dev = "cpu" if staged else "cuda"
    dense = getattr(args, "colpart_dense", False)
    block = getattr(args, "colpart_block", 0) or ColumnPartitionedTableau.MAX_BLOCK
    # the library's own loop whenever the shards are the default ones; ranks that share a GPU run it
    # in exchange mode 2 without a communicator (IPC handles gathered here), everything else the
    # Python protocol driver with the exchanges staged through the host
    native = not dense and block == ColumnPartitionedTableau.MAX_BLOCK
    # BENCH_COLPART_RANK_ENTRY=1 (test hook): also a single rank goes through the entry N > 1 uses --
    # mi355x_rccl_unique_id -> mi355x_colpart_create_synthetic_rank -> ncclCommInitRank (with
    # MI355X_COLPART_FORCE_RCCL=1 over a one-rank communicator)
    rank_entry = world > 1 or os.environ.get("BENCH_COLPART_RANK_ENTRY") == "1"
    steady_pivots = "auto" if native else (8 * block if args.steps < 4 * block else 0)   # (native: by the handle's block size)
    L = capi.lib()
    baseline = None
    exchange_modes = None

    def make_native(exchange, la_block=0):
        # (exchange 2: la_block 0 = the look-ahead of a block as ONE persistent launch per device -- k_shard_la_block,
        # the default of that mode since round 6 --, 1 = the step kernels of round 5)
        L.mi355x_tune_set_shard_la_block(la_block)
        try:
            return _make_native(exchange)
        finally:
            L.mi355x_tune_set_shard_la_block(0)

    def _make_native(exchange):
        if staged:                                    # exchange mode 2, host-connected, RCCL never touched
            L.mi355x_tune_set_colpart_exchange(2)
            try:
                t = NativeColumnPartition.synthetic_rank(n, m, seed, world, rank, local_rank, None)
            finally:
                L.mi355x_tune_set_colpart_exchange(0)
            handles = [None] * world
            dist.all_gather_object(handles, t.p2p_handle())
            t.p2p_connect(b"".join(handles))
            dist.barrier()
            return t
        L.mi355x_tune_set_colpart_exchange(exchange)
        try:
            if rank_entry:
                box = [NativeColumnPartition.rccl_unique_id() if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(box, src=0)
                return NativeColumnPartition.synthetic_rank(n, m, seed, world, rank, local_rank, box[0])
            return NativeColumnPartition.synthetic(n, m, seed, 1)
        finally:
            L.mi355x_tune_set_colpart_exchange(0)

    if native:
        if world > 1 and not getattr(args, "no_colpart_baseline", False):
            if rank == 0:
                baseline = one_shard_baseline(n, m, seed, local_rank, args.steps, args.warmup, steady_pivots)
            dist.barrier()
        tab = make_native(0)
        info = tab.info()
        block = tab.block_size()                  # pivots per sweep of this handle's shards (16 / 24 / 28)
        steady_pivots = 8 * block if args.steps < 4 * block else 0
        tab.solve_async(args.warmup, reset=True)
        st, done = tab.sync()
        if info["uses_rccl"] and not staged:
            tab.exchange_timing(max(1, args.steps // 64), 128)
    else:
        shards = synthetic_shards(torch, n, m, seed, [rank], world, local_rank, compact=not dense)
        comm = DistComm(dist, stage_through_host=staged) if world > 1 else LocalComm(torch)
        tab = ColumnPartitionedTableau(shards, comm, HipBackend(), block=block)
        info = {"n_shards": world, "n_devices_used": 1 if staged else world, "uses_rccl": world > 1 and not staged}
        tab.reset()
        tab.run(args.warmup)
        st, done = tab.status()
    steady = None
    if native:
        elapsed, st, done = _timed_pivots(tab, args.steps, torch, dist, world, red_dev=red_dev)
    else:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tab.run(args.steps)
        st, done = tab.status()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if staged else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
    if st != capi.MI_RUNNING or done != args.warmup + args.steps:
        raise SystemExit("colpart: LP terminated early (status %d after %d pivots)" % (st, done))
    digest = _state_digest(tab) if native else None
    R, C = m + 1, n + m + 1
    value = args.steps / elapsed
    exchange = None
    if native and info["uses_rccl"] and not staged:
        ns, ag_us, ar_us = tab.exchange_timing_read()
        tab.exchange_timing(0, 0)
        if ns:
            exchange = {"samples": ns, "allgather_us": ag_us, "allreduce_us": ar_us,
                        "us_per_pivot": ag_us + ar_us,
                        "what": "HIP events on rank 0's stream around ncclAllGather (16 B/rank) and "
                                "ncclAllReduce (int64 x %d rows) of sampled pivots; includes waiting "
                                "for the slowest rank" % R}
    if native and steady_pivots:
        dt2, st, done2 = _timed_pivots(tab, steady_pivots, torch, dist, world, red_dev=red_dev)
        if st == capi.MI_RUNNING:
            steady = steady_pivots / dt2
    stored_bytes = 2.0 * R * ((n + m if dense else n) + world) * 8 / block   # per pivot, all shards
    rec = {
        "metric": "simplex pivots/sec, one column-partitioned dense tableau",
        "value": value, "unit": "pivots/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "BASELINE config 5: ONE dense LP %d vars x %d constraints, %dx%d f64 "
                               "tableau (%.1f GB) column-partitioned over %d GPU(s), %s shards"
                               % (n, m, R, C, R * C * 8 / 1e9, world, "dense" if dense else "compact"),
                   "parallelism": "column partition, per-pivot all-gather(16 B/rank) + int64 "
                                  "all-reduce(%d B) over RCCL; shards swept once per %d pivots"
                                  % (R * 8, block),
                   "driver": ("mi355x_colpart_* (C++ loop in the library; entry: %s)"
                              % ("mi355x_colpart_create_synthetic_rank, exchange mode 2 (P2P push), no communicator: IPC "
                                 "handles gathered by the host" if staged else
                                 "mi355x_colpart_create_synthetic_rank / ncclCommInitRank" if rank_entry
                                 else "mi355x_colpart_create_synthetic")) if native
                             else "Python protocol driver (torch.distributed)"},
        "rccl_ranks": world if (info["uses_rccl"] and not staged) else 0,
        "exchange_bytes_per_pivot_per_rank": 16 * world + 8 * R,
        "exchange": exchange,
        "exchange_modes": exchange_modes,
        "steady_state_pivots_per_s": steady,
        "steady_state_what": ("%d further pivots (full blocks) timed right after the timed region, same "
                              "bracketing" % steady_pivots) if steady else None,
        "one_gpu_same_workload": baseline,
        "speedup_vs_one_gpu": (value / baseline["value"]) if baseline else None,
        "steady_state_speedup_vs_one_gpu": (steady / baseline["steady_state_pivots_per_s"])
                                           if baseline and steady and baseline.get("steady_state_pivots_per_s") else None,
        "us_per_pivot": elapsed / args.steps * 1e6,
        "per_gpu_physical_GBps": stored_bytes * value / 1e9 / world,
        "aggregate_GBps": stored_bytes * value / 1e9,
        "dense_equivalent_GBps": 2.0 * R * C * 8 * value / 1e9,
        "roofline": {"bound": "hbm", "achieved": stored_bytes * value / 1e9 / world,
                     "peak": 8000.0, "unit": "GB/s",
                     "frac": stored_bytes * value / 1e9 / world / 8000.0, "traffic": None,
                     "note": "whole-iteration rate per GPU (exchanges included), not kernel-only"},
    }
    if progress is not None:
        progress["rec"] = rec                     # (a watchdog that fires during the legs below prints this)
    if native and info["uses_rccl"] and not staged and not getattr(args, "no_colpart_ab", False):
        # A/B of the exchanges on fresh handles of the same tableau, same K pivots each: rooted
        # broadcast (one host synchronisation per pivot for the root) and the collective-free P2P
        # push against the sync-free int64 all-reduce above
        exchange_modes = rec["exchange_modes"] = {
            "int64_sum_allreduce": {"value": value, "unit": "pivots/s", "steady_state_pivots_per_s": steady,
                                    "state_digest": digest,
                                    "what": "owner's bit patterns + zeros, ncclAllReduce(int64, SUM): no host synchronisation"}}
        for mode, name, what in ((1, "rooted_broadcast", "ncclBroadcast from the owner; the root is read back from the "
                                                         "all-gathered pricing winners: one stream synchronisation per pivot"),
                                 (2, "p2p_push", "no collective: every shard writes its pricing pair, the owner the entering "
                                                 "column, straight into the peers' fine-grained buffers (peer access / IPC "
                                                 "over xGMI) as self-validating granules; consumers poll their own memory -- "
                                                 "inside ONE persistent launch per block and device (k_shard_la_block) where "
                                                 "the shard's block fits it (la_stats says)"),
                                 (12, "p2p_push_step_kernels", "the same exchanges with round 5's step kernels (two launches "
                                                               "per step): what the persistent launch is measured against")):
            tab.close()
            entry = {"value": None, "unit": "pivots/s", "what": what}
            try:
                tab = make_native(mode % 10, la_block=1 if mode >= 10 else 0)
                entry["persistent_block_launch"] = tab.la_stats()
                tab.solve_async(args.warmup, reset=True)
                tab.sync()
                dtb, stb, doneb = _timed_pivots(tab, args.steps, torch, dist, world)
                if stb == capi.MI_RUNNING:
                    entry["value"] = args.steps / dtb
                    # same start, same pivots, deterministic arithmetic: the state after the K pivots
                    # must be the default mode's bit for bit, or the mode does not count
                    entry["state_digest"] = _state_digest(tab)
                    entry["identical_to_default_mode"] = entry["state_digest"] == digest
                    if steady_pivots:
                        dt3, st3, _ = _timed_pivots(tab, steady_pivots, torch, dist, world)
                        if st3 == capi.MI_RUNNING:
                            entry["steady_state_pivots_per_s"] = steady_pivots / dt3
            except capi.Mi355xError as e:                     # e.g. no peer access / IPC on this box
                entry["error"] = str(e)
                tab = make_native(0)
            exchange_modes[name] = entry
        # `value` stays what mi355x_colpart_solve / (solve-problem :devices n) runs out of the box -- the
        # library's default exchange (int64 all-reduce).  The fastest mode whose state after the K
        # pivots is bit-identical to the default mode's is reported NEXT to it, never instead of it.
        best = max((nm for nm, en in exchange_modes.items()
                    if en.get("value") and (nm == "int64_sum_allreduce" or en.get("identical_to_default_mode"))),
                   key=lambda nm: exchange_modes[nm]["value"])
        rec["value_mode"] = "int64_sum_allreduce"
        rec["default_mode"] = {"mode": "int64_sum_allreduce", "value": value, "steady_state_pivots_per_s": steady}
        bv, bs = exchange_modes[best]["value"], exchange_modes[best].get("steady_state_pivots_per_s")
        rec["best_mode"] = {
            "mode": best, "value": bv, "steady_state_pivots_per_s": bs, "us_per_pivot": 1e6 / bv,
            "opt_in": None if best == "int64_sum_allreduce" else
                      "mi355x_tune_set_colpart_exchange(%d) before the handle is created%s"
                      % ((1 if best == "rooted_broadcast" else 2), " (+ mi355x_tune_set_shard_la_block(1))" if best == "p2p_push_step_kernels" else ""),
            "speedup_vs_one_gpu": (bv / baseline["value"]) if baseline else None,
            "steady_state_speedup_vs_one_gpu": (bs / baseline["steady_state_pivots_per_s"])
                                               if baseline and bs and baseline.get("steady_state_pivots_per_s") else None}
    if native:
        tab.close()
    else:
        destroy_shards(shards)
    return rec


if __name__ == "__main__":
    print(json.dumps({"partition_example": partition(98304, 8)}))
