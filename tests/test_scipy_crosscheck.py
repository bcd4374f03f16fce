"""an INDEPENDENT answer-level cross-check of the oracle on the synthetic
benchmark LPs -- scipy's HiGHS (a different algorithm, a different code base) must find the same
optimal objective value to 1e-9 relative.  CPU only; the GPU path is then held to the oracle bit
for bit by the -m gpu tests."""
import numpy as np
import pytest

import oracle
from tests.helpers import lp_amd

lp = lp_amd()
linprog = pytest.importorskip("scipy.optimize").linprog


@pytest.mark.parametrize("n,m,cfg", [(1024, 512, 2), (512, 256, 4), (200, 100, 3)])
def test_oracle_optimum_equals_highs(n, m, cfg):
    seed = lp.synth.seed_for(cfg)
    M, b = lp.synth.tableau(n, m, seed)
    A, rhs, c = M[:m, :n].copy(), M[:m, -1].copy(), -M[m, :n].copy()
    st, npiv, _ = oracle.solve(M, b, omp=True)
    assert st == oracle.OPTIMAL and npiv > 0
    ours = M[m, -1]                                              # tableau-objective-value
    res = linprog(-c, A_ub=A, b_ub=rhs, bounds=(0, None), method="highs")
    assert res.status == 0
    assert abs(ours - (-res.fun)) <= 1e-9 * abs(ours)
    # and the oracle's basic solution is feasible for the original data
    x = np.zeros(n + m)
    x[b] = M[:m, -1]
    assert (A @ x[:n] - rhs).max() <= 1e-9 * np.abs(rhs).max() and x.min() >= 0.0
    assert abs(c @ x[:n] - ours) <= 1e-10 * abs(ours)
