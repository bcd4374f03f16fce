"""Test-only compute backend for linear-programming_amd/colpart.py: the three per-shard steps
on CPU tensors in numpy, built from the oracle's arithmetic rules.  It lets the exchange
protocol (ordering and payload of the two collectives, owner-free column hand-off, status
agreement) run under torch.distributed/gloo without a GPU.  Never imported by the product."""
import numpy as np

EPS = 1.1102230246251568e-16
RUNNING, OPTIMAL, UNBOUNDED, MAXPIV = 100, 0, 1, 3


class FakeHandle:
    def __init__(self, M, basis):
        self.M = np.ascontiguousarray(M, dtype=np.float64)     # rows x (local cols + RHS)
        self.basis = basis.copy()
        self.status, self.n_pivots, self.max_pivots = RUNNING, 0, 0
        self.trace = []
        self.pending = []          # blocked form: (pivot row, exchanged column, local slice of prow)


class OracleShardBackend:
    def __init__(self, is_max=True, fp_factor=1024.0):
        self.sgn = 1.0 if is_max else -1.0
        self.f = float(fp_factor)

    def contribute(self, sh):
        g = sh.gathered.numpy().reshape(-1, 2)
        best = None
        for v, c in g:
            if c < 0:
                continue
            if best is None or v < best[0] or (v == best[0] and c < best[1]):
                best = (v, c)
        ec = int(best[1]) if best is not None and best[0] < 0.0 - (self.f / 8.0) * EPS else -1
        lc = ec - sh.col_begin
        mine = ec >= 0 and 0 <= lc < sh.handle.M.shape[1] - 1
        bits = sh.bits.numpy()
        bits[:] = sh.handle.M[:, lc].view(np.int64) if mine else 0
        sh.ec[0] = ec

    def pivot(self, sh):
        h = sh.handle
        if h.status != RUNNING:
            return
        ec = int(sh.ec[0])
        if ec < 0:
            h.status = OPTIMAL
            return
        if h.max_pivots > 0 and h.n_pivots >= h.max_pivots:
            h.status = MAXPIV
            return
        col = sh.bits.numpy().view(np.float64).copy()
        M = h.M
        m = M.shape[0] - 1
        thr = 0.0 + (self.f / 2.0) * EPS
        cr, bestq = -1, 0.0
        for i in range(m):
            if thr < col[i]:
                q = M[i, -1] / col[i]
                if cr < 0 or q < bestq:
                    cr, bestq = i, q
        if cr < 0:
            h.status = UNBOUNDED
            return
        prow = M[cr, :] / col[cr]
        prod = col[:, None] * prow[None, :]        # rounded products ...
        M -= prod                                  # ... then rounded differences
        M[cr, :] = prow
        h.basis[cr] = ec
        h.trace.append((ec, cr))
        h.n_pivots += 1

    # ---- blocked form: the tableau slice is only touched by sweep(); what a step reads goes
    # through the chain of the pending pivots (rounded product, rounded difference, pivot order)
    @staticmethod
    def _chain_vec(x, pending, prow_entry):
        """x: one column (all rows) -> as it is after the pending pivots."""
        x = x.copy()
        for cr, col, prow in pending:
            p = prow_entry(prow)
            prod = col * p
            y = x - prod
            y[cr] = p
            x = y
        return x

    @staticmethod
    def _chain_row(x, r, pending):
        """x: row r of the local slice -> as it is after the pending pivots."""
        x = x.copy()
        for cr, col, prow in pending:
            if r == cr:
                x = prow.copy()
            else:
                prod = col[r] * prow
                x = x - prod
        return x

    def price(self, sh):
        h = sh.handle
        obj = self._chain_row(h.M[-1, :], h.M.shape[0] - 1, h.pending)
        key = obj[:-1] * self.sgn
        j = int(np.argmin(key))
        sh.send[0], sh.send[1] = float(key[j]), float(j + sh.col_begin)

    def la_contribute(self, sh, j):
        h = sh.handle
        assert j == len(h.pending) or h.status != RUNNING
        g = sh.gathered.numpy().reshape(-1, 2)
        best = None
        for v, c in g:
            if c < 0:
                continue
            if best is None or v < best[0] or (v == best[0] and c < best[1]):
                best = (v, c)
        ec = int(best[1]) if best is not None and best[0] < 0.0 - (self.f / 8.0) * EPS else -1
        if h.status != RUNNING:
            ec = -1
        lc = ec - sh.col_begin
        mine = ec >= 0 and 0 <= lc < h.M.shape[1] - 1
        bits = sh.bits.numpy()
        if mine:
            bits[:] = self._chain_vec(h.M[:, lc], h.pending, lambda prow: prow[lc]).view(np.int64)
        else:
            bits[:] = 0
        sh.ec[0] = ec

    def la_pivot(self, sh, j):
        h = sh.handle
        if h.status != RUNNING:
            return
        ec = int(sh.ec[0])
        if ec < 0:
            h.status = OPTIMAL
            return
        if h.max_pivots > 0 and h.n_pivots >= h.max_pivots:
            h.status = MAXPIV
            return
        col = sh.bits.numpy().view(np.float64).copy()
        M = h.M
        m = M.shape[0] - 1
        rhs = self._chain_vec(M[:, -1], h.pending, lambda prow: prow[-1])
        thr = 0.0 + (self.f / 2.0) * EPS
        cr, bestq = -1, 0.0
        for i in range(m):
            if thr < col[i]:
                q = rhs[i] / col[i]
                if cr < 0 or q < bestq:
                    cr, bestq = i, q
        if cr < 0:
            h.status = UNBOUNDED
            return
        prow = self._chain_row(M[cr, :], cr, h.pending) / col[cr]
        h.pending.append((cr, col, prow))
        h.basis[cr] = ec
        h.trace.append((ec, cr))
        h.n_pivots += 1

    def sweep(self, sh):
        h = sh.handle
        M = h.M
        for cr, col, prow in h.pending:
            prod = col[:, None] * prow[None, :]
            M -= prod
            M[cr, :] = prow
        h.pending = []

    def reset(self, sh, max_pivots=0):
        sh.handle.status, sh.handle.n_pivots, sh.handle.max_pivots = RUNNING, 0, int(max_pivots)

    def status(self, sh):
        return sh.handle.status, sh.handle.n_pivots
