"""Synthetic dense LPs for the benchmark and the parity tests.

    max c'x   s.t.  Ax <= b,  x >= 0        A: m x n dense, strictly positive

splitmix64 stream: element k = mix(seed + (k+1)*gamma), u = (z >> 11) * 2^-53 in [0,1).
Stream layout: A row-major, then b, then c.
    A[i][j] = 0.05 + u      b[i] = n * (0.25 + 0.5 u)      c[j] = 0.5 + u
The origin is feasible (single phase) and the feasible region is bounded.  The tableau is
assembled exactly as build-tableau (src/simplex.lisp:214-283) would for variable order
x0..x(n-1):  [A | I | b ; -c | 0 | 0],  basis n..n+m-1,  is_max = 1.

The same generator exists as a HIP kernel (k_synth_fill, mi355x_tab_create_synthetic) that
writes the tableau straight into HBM; both produce bit-identical doubles (tested).
"""
import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
BASE_SEED = 0x9E3779B97F4A7C15
DATE_SALT = 20260928


def seed_for(config_id, lp_index=0):
    """seed = 0x9E3779B97F4A7C15 ^ (20260928 + config_id*1000 + lp_index)"""
    return (BASE_SEED ^ (DATE_SALT + config_id * 1000 + lp_index)) & 0xFFFFFFFFFFFFFFFF


def splitmix_u01(seed, start, count):
    """u01 values of stream positions start .. start+count-1 (vectorised, wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def lp_data(n, m, seed):
    """(A, b, c) of the synthetic LP."""
    A = 0.05 + splitmix_u01(seed, 0, n * m).reshape(m, n)
    b = float(n) * (0.25 + 0.5 * splitmix_u01(seed, n * m, m))
    c = 0.5 + splitmix_u01(seed, n * m + m, n)
    return A, b, c


def tableau(n, m, seed):
    """(matrix (m+1) x (n+m+1) float64, basis int64[m]) of the synthetic LP."""
    A, b, c = lp_data(n, m, seed)
    M = np.zeros((m + 1, n + m + 1), dtype=np.float64)
    M[:m, :n] = A
    M[np.arange(m), n + np.arange(m)] = 1.0
    M[:m, n + m] = b
    M[m, :n] = -c
    return M, np.arange(n, n + m, dtype=np.int64)
