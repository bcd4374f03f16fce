/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * A plain-C, scalar, CPU restatement of the double-float hot path of the
 * reference's dense simplex solver (neil-lindquist/linear-programming v2.3.0,
 * src/simplex.lisp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this file's shared object, and only as the
 * checker / the reported CPU baseline.  libmi355x_simplex.so never links,
 * loads or calls anything in this directory.
 *
 * Parity status: PINNED against the reference's own known-answer tests
 * (tests/golden/reference_cases.json, transcribed from t/simplex.lisp,
 * t/solver.lisp, t/integration.lisp and README.md) -- see
 * tests/test_oracle_golden.py.  The reference itself (Common Lisp) cannot be
 * compiled or run in this image: no CL implementation is installed and its
 * dependencies (alexandria, iterate) are not vendored, so there is no
 * oracle/_ref build.  The reference's golden vectors are rational / single
 * float; the f64 path restated here reproduces them exactly where the values
 * are dyadic and to <= 1e-15 relative otherwise.
 *
 * Arithmetic discipline (must match what SBCL computes on IEEE doubles):
 *   - every product and every difference is rounded separately: build with
 *     -ffp-contract=off so no FMA is ever formed (src/simplex.lisp:357 is
 *     (decf a (* s b)), two generic-arithmetic calls);
 *   - true division, never multiplication by a reciprocal (simplex.lisp:348,388);
 *   - epsilon is Common Lisp's DOUBLE-FLOAT-EPSILON = 2^-53 (1 + 2^-52),
 *     which is about HALF of C's DBL_EPSILON (src/utils.lisp:84-124);
 *   - argmin / argmax use strict comparison, first index wins
 *     (iterate's `finding ... minimizing`, pinned by t/simplex.lisp:196-237).
 *
 * Layout: matrix is row-major with leading dimension `ld` (>= cols), rows =
 * constraint_count + 1 (last row = objective row), cols = var_count + 1 (last
 * column = right-hand side).  (src/simplex.lisp:48-58, 74-78, 214-221.)
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define ORC_OPTIMAL      0
#define ORC_UNBOUNDED    1
#define ORC_INFEASIBLE   2
#define ORC_MAX_PIVOTS   3
#define ORC_ART_NONZERO  4   /* "Artificial variable ~S still non-zero"            simplex.lisp:423-424 */
#define ORC_ART_STUCK    5   /* "Artificial variable still in basis and cannot..." simplex.lisp:432-433 */
#define ORC_BAD_ARG     -1

/* CL double-float-epsilon, 0x1.0000000000001p-53 (src/utils.lisp:92) */
static const double CL_DOUBLE_FLOAT_EPSILON = 1.1102230246251568e-16;

double orc_epsilon(void) { return CL_DOUBLE_FLOAT_EPSILON; }

/* threads orc_pivot_omp will use (reported as cpu_baseline.cores by bench.py) */
#ifdef _OPENMP
#include <omp.h>
int orc_omp_threads(void) { return omp_get_max_threads(); }
/* the caller knows how many cores the container may really use (a cgroup CPU quota is invisible to
 * OpenMP: 128 threads on a 16-core quota both mislabel and handicap the baseline) */
int orc_set_omp_threads(int n) { if (n >= 1) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int orc_omp_threads(void) { return 1; }
int orc_set_omp_threads(int n) { (void)n; return 1; }
#endif

/* find-entering-column, src/simplex.lisp:362-379.
 * max problem: lowest-index strict argmin of the objective row over
 * [0, var_count); returned iff (fp< v 0 factor/8)  <=>  v < 0 - (factor/8)*eps.
 * min problem: argmax, returned iff (fp> v 0 factor/8) <=> v > 0 + (factor/8)*eps.
 * Returns -1 for NIL (tableau optimal). */
int64_t orc_price(const double *M, int64_t ld, int64_t m, int64_t vc,
                  int is_max, double factor)
{
    const double *obj = M + m * ld;
    const double tol = (factor / 8.0) * CL_DOUBLE_FLOAT_EPSILON;
    int64_t best = 0;
    if (vc <= 0) return -1;
    if (is_max) {
        for (int64_t i = 1; i < vc; ++i)
            if (obj[i] < obj[best]) best = i;
        return (obj[best] < 0.0 - tol) ? best : -1;
    } else {
        for (int64_t i = 1; i < vc; ++i)
            if (obj[i] > obj[best]) best = i;
        return (obj[best] > 0.0 + tol) ? best : -1;
    }
}

/* find-pivoting-row, src/simplex.lisp:382-389.
 * Eligible rows: (fp< 0 M[i][ec] factor/2), which the compiler macro at
 * src/utils.lisp:110-119 turns into (< (+ 0 (* factor/2 eps)) M[i][ec]).
 * Among eligible rows: lowest-index strict argmin of rhs/M[i][ec].
 * Returns -1 for NIL (unbounded). */
int64_t orc_ratio(const double *M, int64_t ld, int64_t m, int64_t vc,
                  int64_t ec, double factor)
{
    const double thr = 0.0 + (factor / 2.0) * CL_DOUBLE_FLOAT_EPSILON;
    int64_t row = -1;
    double bestq = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        const double a = M[i * ld + ec];
        if (thr < a) {
            const double q = M[i * ld + vc] / a;
            if (row < 0 || q < bestq) { row = i; bestq = q; }
        }
    }
    return row;
}

/* n-pivot-row, src/simplex.lisp:337-359. */
void orc_pivot(double *M, int64_t ld, int64_t rows, int64_t cols,
               int64_t *basis, int64_t ec, int64_t cr)
{
    double *prow = M + cr * ld;
    const double row_scale = prow[ec];                 /* :343 */
    for (int64_t c = 0; c < cols; ++c)                 /* :344-348 */
        prow[c] = prow[c] / row_scale;
    for (int64_t r = 0; r < rows; ++r) {               /* :349-357, objective row included */
        if (r == cr) continue;
        double *row = M + r * ld;
        const double scale = row[ec];                  /* :353, read once */
        for (int64_t c = 0; c < cols; ++c) {
            const double prod = scale * prow[c];       /* rounded product ... */
            row[c] = row[c] - prod;                    /* ... then rounded difference */
        }
    }
    if (basis) basis[cr] = ec;                         /* :358 */
}

/* Same arithmetic, rows distributed over OpenMP threads.  Every element
 * update is independent, so the result is bit-identical to orc_pivot; used
 * only as the all-cores CPU baseline in bench.py. */
void orc_pivot_omp(double *M, int64_t ld, int64_t rows, int64_t cols,
                   int64_t *basis, int64_t ec, int64_t cr)
{
    double *prow = M + cr * ld;
    const double row_scale = prow[ec];
    for (int64_t c = 0; c < cols; ++c)
        prow[c] = prow[c] / row_scale;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        if (r == cr) continue;
        double *row = M + r * ld;
        const double scale = row[ec];
        for (int64_t c = 0; c < cols; ++c) {
            const double prod = scale * prow[c];
            row[c] = row[c] - prod;
        }
    }
    if (basis) basis[cr] = ec;
}

/* n-solve-tableau, single-phase branch, src/simplex.lisp:453-461.
 * max_pivots == 0 means no cap (the reference has none).  If trace_ec /
 * trace_cr are non-NULL the first trace_cap (entering column, pivot row)
 * pairs are recorded.  use_omp selects orc_pivot_omp. */
int orc_solve(double *M, int64_t ld, int64_t rows, int64_t cols,
              int64_t *basis, int is_max, double factor, int64_t max_pivots,
              int64_t *n_pivots_out, int64_t *trace_ec, int64_t *trace_cr,
              int64_t trace_cap, int use_omp)
{
    const int64_t m = rows - 1, vc = cols - 1;
    int64_t n = 0;
    int status = ORC_OPTIMAL;
    if (rows < 1 || cols < 1 || ld < cols) return ORC_BAD_ARG;
    for (;;) {
        const int64_t ec = orc_price(M, ld, m, vc, is_max, factor);
        if (ec < 0) { status = ORC_OPTIMAL; break; }
        if (max_pivots > 0 && n >= max_pivots) { status = ORC_MAX_PIVOTS; break; }
        const int64_t cr = orc_ratio(M, ld, m, vc, ec, factor);
        if (cr < 0) { status = ORC_UNBOUNDED; break; }
        if (trace_ec && n < trace_cap) trace_ec[n] = ec;
        if (trace_cr && n < trace_cap) trace_cr[n] = cr;
        if (use_omp) orc_pivot_omp(M, ld, rows, cols, basis, ec, cr);
        else         orc_pivot(M, ld, rows, cols, basis, ec, cr);
        ++n;
    }
    if (n_pivots_out) *n_pivots_out = n;
    return status;
}

static double orc_fabs(double x) { return x < 0.0 ? -x : x; }

/* n-solve-tableau, two-phase branch, src/simplex.lisp:402-452.
 * art: rows x art_cols artificial tableau (a `min` problem, :300-325),
 * mainM: rows x main_cols main tableau; both are modified in place.
 * n_pivots_out[0] = phase-1 pivots (including drive-out pivots),
 * n_pivots_out[1] = phase-2 pivots. */
int orc_solve_two_phase(double *art, int64_t art_ld, int64_t rows, int64_t art_cols,
                        int64_t *art_basis,
                        double *mainM, int64_t main_ld, int64_t main_cols,
                        int64_t *main_basis, int main_is_max, double factor,
                        int64_t *n_pivots_out)
{
    const int64_t m = rows - 1;
    const int64_t num_vars = main_cols - 1;         /* (tableau-var-count main-tab)       :412 */
    const int64_t num_art_vars = art_cols - 1;      /* (tableau-var-count solved-art-tab) :413 */
    int64_t n1 = 0, n2 = 0;
    int st = orc_solve(art, art_ld, rows, art_cols, art_basis, /*is_max=*/0, factor,
                       0, &n1, NULL, NULL, 0, 0);   /* :403 */
    if (n_pivots_out) { n_pivots_out[0] = n1; n_pivots_out[1] = 0; }
    if (st != ORC_OPTIMAL) return st;
    /* (fp= 0 objective factor): |0 - obj| <= factor*eps          :405-407 */
    if (!(orc_fabs(0.0 - art[m * art_ld + num_art_vars]) <= factor * CL_DOUBLE_FLOAT_EPSILON))
        return ORC_INFEASIBLE;
    /* drive degenerate artificials out of the basis               :419-434 */
    for (int64_t i = 0; i < m; ++i) {
        if (art_basis[i] >= num_vars) {
            if (art[i * art_ld + num_art_vars] != 0.0) return ORC_ART_NONZERO;
            int64_t new_col = -1;
            for (int64_t j = 0; j < num_vars; ++j) {
                if (art[i * art_ld + j] != 0.0) {
                    int in_basis = 0;
                    for (int64_t k = 0; k < m; ++k)
                        if (art_basis[k] == j) { in_basis = 1; break; }
                    if (!in_basis) { new_col = j; break; }
                }
            }
            if (new_col < 0) return ORC_ART_STUCK;
            orc_pivot(art, art_ld, rows, art_cols, art_basis, new_col, i);
            ++n1;
        }
    }
    /* copy coefficients and rhs                                    :437-441 */
    for (int64_t r = 0; r < m; ++r) {
        for (int64_t c = 0; c < num_vars; ++c)
            mainM[r * main_ld + c] = art[r * art_ld + c];
        mainM[r * main_ld + num_vars] = art[r * art_ld + num_art_vars];
    }
    /* basis + re-eliminate the objective row                       :444-451 */
    for (int64_t i = 0; i < m; ++i) {
        const int64_t bc = art_basis[i];
        main_basis[i] = bc;
        const double scale = mainM[m * main_ld + bc];
        if (scale != 0.0) {
            for (int64_t c = 0; c <= num_vars; ++c) {
                const double prod = scale * mainM[i * main_ld + c];
                mainM[m * main_ld + c] = mainM[m * main_ld + c] - prod;
            }
        }
    }
    st = orc_solve(mainM, main_ld, rows, main_cols, main_basis, main_is_max, factor,
                   0, &n2, NULL, NULL, 0, 0);       /* :452 */
    if (n_pivots_out) { n_pivots_out[0] = n1; n_pivots_out[1] = n2; }
    return st;
}

/* fp= / fp< / fp> on doubles, src/utils.lisp:84-124 (function forms, used by
 * tests/test_oracle_golden.py against the tables in t/utils.lisp:74-157). */
int orc_fp_eq(double a, double b, double factor)
{ return orc_fabs(a - b) <= factor * CL_DOUBLE_FLOAT_EPSILON; }
int orc_fp_lt(double a, double b, double factor)
{ return a < (b - factor * CL_DOUBLE_FLOAT_EPSILON); }
int orc_fp_gt(double a, double b, double factor)
{ return a > (b + factor * CL_DOUBLE_FLOAT_EPSILON); }
int orc_fp_le(double a, double b, double factor)
{ return a <= (b + factor * CL_DOUBLE_FLOAT_EPSILON); }
int orc_fp_ge(double a, double b, double factor)
{ return a >= (b - factor * CL_DOUBLE_FLOAT_EPSILON); }
