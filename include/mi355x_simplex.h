/*
 * mi355x_simplex.h -- C ABI of libmi355x_simplex.so
 *
 * MI355X (gfx950) dense-simplex backend for the Common Lisp library
 * neil-lindquist/linear-programming (v2.3.0).  This is the drop-in boundary:
 * the entry points below are what a CFFI binding behind the library's
 * `linear-programming:*solver*` hook (src/solver.lisp:39-56) calls in place of
 * the Lisp loops of src/simplex.lisp.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no callbacks, no ownership transfer: the caller owns every host
 *     buffer, the library owns the device memory behind a handle;
 *   - a tableau is the reference's `tableau-matrix` (src/simplex.lisp:48-58):
 *     rows = constraint_count + 1 (last row = objective row),
 *     cols = var_count + 1       (last column = right-hand side),
 *     row-major, tightly packed doubles on the host side;
 *   - every function returns a status (>= 0: outcome, < 0: error);
 *     mi355x_last_error() gives the thread-local message of the last error;
 *   - there is NO CPU fallback: without a gfx950 device every compute entry
 *     point fails with MI_NO_DEVICE;
 *   - a handle may be used from one thread at a time.
 *
 * Arithmetic contract (what makes results bit-identical to the reference's
 * double-float path): IEEE binary64, product and difference rounded separately
 * (no FMA) in the rank-1 update, true division for the pivot-row scaling and
 * the ratio test, tolerances factor/8, factor/2 (and factor for the phase-1
 * feasibility test) times Common Lisp's DOUBLE-FLOAT-EPSILON
 * 1.1102230246251568e-16, strict comparison with lowest index winning ties.
 */
#ifndef MI355X_SIMPLEX_H
#define MI355X_SIMPLEX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_SIMPLEX_ABI_VERSION 1

/* outcomes (returned by the solve entry points) */
#define MI_OK            0
#define MI_OPTIMAL       0   /* find-entering-column returned NIL            simplex.lisp:455-456 */
#define MI_UNBOUNDED     1   /* -> unbounded-problem-error                   simplex.lisp:458-459 */
#define MI_INFEASIBLE    2   /* -> infeasible-problem-error                  simplex.lisp:405-407 */
#define MI_MAX_PIVOTS    3   /* pivot cap reached (backend-specific; the reference has no cap) */
#define MI_ART_NONZERO   4   /* "Artificial variable ~S still non-zero"       simplex.lisp:423-424 */
#define MI_ART_STUCK     5   /* "Artificial variable still in basis and ..."  simplex.lisp:432-433 */
#define MI_NONFINITE     6   /* compact column shards only: the entering column holds an inf / NaN;
                                the reference would spread NaNs over basic columns, which are not
                                stored there (single tableaux and batches switch to the dense
                                tableau by themselves and continue) */
#define MI_CANCELLED     7   /* mi355x_*_cancel was called from another thread: the solve stopped between
                                two chunks of launches; whole pivots only, the tableau is consistent and a
                                later solve call carries on from it */
#define MI_RUNNING     100   /* asynchronous use only: the iterations enqueued so far have not
                                terminated the solve (mi355x_tab_sync) */
/* errors */
#define MI_BAD_ARG      -1
#define MI_HIP_ERROR    -2
#define MI_RCCL_ERROR   -3
#define MI_NO_DEVICE    -4
#define MI_NO_MEMORY    -5
#define MI_UNSUPPORTED  -6   /* -> unsupported-constraint-error (integer / binary variables) */

typedef struct mi355x_tab   mi355x_tab;     /* one tableau resident in HBM              */
typedef struct mi355x_batch mi355x_batch;   /* a batch of same-shape tableaux in HBM    */
typedef struct mi355x_problem  mi355x_problem;   /* a parsed LP, src/problem.lisp:45-53 (host)  */
typedef struct mi355x_solution mi355x_solution;  /* what the read-back needs of a solved tableau */
typedef struct mi355x_solve    mi355x_solve;     /* a problem on its way to a solution (resumable)  */
typedef struct mi355x_solve_many mi355x_solve_many; /* a LIST of problems on their way to solutions   */

/* ---- library / device ------------------------------------------------------------- */
int         mi355x_abi_version(void);
int         mi355x_device_count(void);             /* number of gfx950 devices visible, 0 if none */
const char *mi355x_last_error(void);               /* thread-local, never NULL */
double      mi355x_epsilon(void);                  /* CL double-float-epsilon used by the tolerances */
/* Optional: pay the one-off costs now instead of inside the first solve (the HIP context of
 * `device`, the library's code object, the pinned staging buffers of the streamed uploads) -- a host
 * calls it once when it loads the backend.  Idempotent; solves a 1-pivot LP on the device. */
int         mi355x_init(int device);

/* ---- tableau lifecycle  (tableau struct, src/simplex.lisp:48-58) ------------------- */
/* Upload a tableau.  host_basis has rows-1 entries (tableau-basis-columns) or is NULL. */
int  mi355x_tab_create(mi355x_tab **out, int64_t rows, int64_t cols,
                       const double *host_matrix, const int64_t *host_basis, int device);
/* Upload a tableau in COMPACT form: only the n_stored = var_count - (rows-1) non-basic columns
 * plus the RHS column (host_stored: rows x (n_stored+1), row-major; stored_cols[j] = logical column
 * of stored column j); the basic columns are BY DEFINITION the unit vectors host_basis describes
 * (what build-tableau produces for the slack columns of a single-phase problem).  The handle
 * behaves exactly like one made by mi355x_tab_create from the equivalent dense tableau -- the
 * dense logical form is materialised only if an entry point needs it -- but a third less data
 * crosses PCIe and no dense buffer is allocated for a plain solve + light read-back. */
int  mi355x_tab_create_compact(mi355x_tab **out, int64_t rows, int64_t var_count, int64_t n_stored,
                               const double *host_stored, const int64_t *stored_cols,
                               const int64_t *host_basis, int device);
/* Replace the contents of an existing handle (same shape). */
int  mi355x_tab_upload(mi355x_tab *t, const double *host_matrix, const int64_t *host_basis);
/* copy-tableau (src/simplex.lisp:61-71): device-to-device deep copy of matrix + basis. */
int  mi355x_tab_copy(mi355x_tab **out, const mi355x_tab *src);
/* Build the bench/test LP  max c'x, Ax <= b, x >= 0  directly in HBM
 * (splitmix64 stream; identical doubles to linear-programming_amd/synth.py):
 * tableau [A | I | b ; -c | 0 | 0], basis n_vars .. n_vars+n_cons-1.
 * col_begin/col_end select a column slice [col_begin, col_end) of the
 * var_count = n_vars+n_cons non-RHS columns (column-partitioned shards keep
 * the RHS column as their last column); pass 0, -1 for the whole tableau. */
int  mi355x_tab_create_synthetic(mi355x_tab **out, int64_t n_vars, int64_t n_cons,
                                 uint64_t seed, int64_t col_begin, int64_t col_end, int device);
void mi355x_tab_destroy(mi355x_tab *t);
int  mi355x_tab_shape(const mi355x_tab *t, int64_t *rows, int64_t *cols, int64_t *ld);

/* Which representation currently holds the tableau in HBM.  The solve entry points run on the
 * COMPACT representation [non-basic columns | RHS] (rows x (var_count - constraint_count + 1))
 * whenever the basis columns are exact unit vectors -- they are for every tableau
 * build-tableau produces, and pivoting keeps them so -- because basic columns never change
 * under a pivot; every other entry point sees the dense logical tableau (rebuilt on demand),
 * so the representation is invisible except through this query.  *stored_cols includes the
 * RHS column. */
int  mi355x_tab_layout(const mi355x_tab *t, int *compact, int64_t *stored_cols, int64_t *stored_ld);

/* ---- the hot path ------------------------------------------------------------------ */
/* n-pivot-row (src/simplex.lisp:337-359): normalise row `pivot_row` by its entry in
 * `entering_col`, eliminate that column from every other row (objective row included),
 * basis[pivot_row] = entering_col. */
int  mi355x_tab_pivot(mi355x_tab *t, int64_t entering_col, int64_t pivot_row);
/* find-entering-column (src/simplex.lisp:362-379): *col = column or -1 for NIL. */
int  mi355x_tab_price(mi355x_tab *t, int is_max, double fp_factor, int64_t *col);
/* find-pivoting-row (src/simplex.lisp:382-389): *row = row or -1 for NIL. */
int  mi355x_tab_ratio(mi355x_tab *t, int64_t entering_col, double fp_factor, int64_t *row);
/* n-solve-tableau, single-phase branch (src/simplex.lisp:453-461), entirely on the
 * device.  max_pivots = 0: no cap (as the reference).  Returns MI_OPTIMAL,
 * MI_UNBOUNDED or MI_MAX_PIVOTS; *n_pivots = pivots performed by this call. */
int  mi355x_tab_solve(mi355x_tab *t, int is_max, double fp_factor, int64_t max_pivots,
                      int64_t *n_pivots);
/* A way out of a blocking solve.  The reference's loop (src/simplex.lisp:453-461) has no iteration
 * cap and no anti-cycling rule; in Lisp a cycling LP can be interrupted, a blocking foreign call
 * cannot.  mi355x_tab_cancel may be called from ANY thread while another thread is inside
 * mi355x_tab_solve / mi355x_solve_two_phase on the same handle (the one exception to "one thread
 * at a time"): the solve returns MI_CANCELLED after the chunk of launches in flight -- at most 64
 * blocks of 16 pivots, 512 per-pivot iterations or one resident launch of 65536 pivots, i.e. well
 * under a second at every BASELINE size -- with *n_pivots = the pivots done and the tableau whole
 * (download / trace / a further solve call all work).  The request is sticky: cancelling a handle
 * nobody is solving cancels its next solve.  For mi355x_solve_two_phase cancel either (or both)
 * of the two handles.  The other way to keep a host responsive is max_pivots: solve in bounded
 * chunks and look around in between (what the Lisp glue does). */
int  mi355x_tab_cancel(mi355x_tab *t);
/* n-solve-tableau, two-phase branch (src/simplex.lisp:402-452): `art` is the
 * artificial tableau (a min problem), `main_tab` the main tableau with the same
 * row count; both are modified.  n_pivots[0] = phase 1 (incl. drive-out pivots),
 * n_pivots[1] = phase 2.  Returns MI_OPTIMAL / MI_UNBOUNDED / MI_INFEASIBLE /
 * MI_ART_NONZERO / MI_ART_STUCK. */
int  mi355x_solve_two_phase(mi355x_tab *art, mi355x_tab *main_tab, int main_is_max,
                            double fp_factor, int64_t *n_pivots);
/* The step BETWEEN the phases on its own (src/simplex.lisp:405-451), for a caller that drives the
 * phases itself in bounded chunks (mi355x_tab_solve(art, 0, f, cap, ..) until it is no longer
 * MI_MAX_PIVOTS, this, then mi355x_tab_solve(main_tab, ..) likewise -- what the Lisp glue and
 * mi355x_simplex_solver_step do): `art` holds the optimal artificial tableau; the feasibility test
 * (fp= 0 objective), the drive-out pivots of artificial variables still basic, rows and basis into
 * `main_tab`, its objective row re-eliminated.  *n_driveout = drive-out pivots made.  Returns MI_OK
 * (main_tab is ready for phase 2), MI_INFEASIBLE, MI_ART_NONZERO or MI_ART_STUCK. */
int  mi355x_two_phase_handover(mi355x_tab *art, mi355x_tab *main_tab, double fp_factor,
                               int64_t *n_driveout);

/* ---- read-back (src/simplex.lisp:74-120 needs last row, last column, basis) -------- */
/* Any pointer may be NULL.  host_matrix: rows*cols, host_basis: rows-1,
 * last_row: cols (objective row), last_col: rows (RHS column). */
int  mi355x_tab_download(mi355x_tab *t, double *host_matrix, int64_t *host_basis,
                         double *last_row, double *last_col);
/* A sub-block of the logical tableau -- (aref matrix r c) for r in [row0, row0+n_rows),
 * c in [col0, col0+n_cols) -- into a tightly packed row-major host array of n_rows*n_cols
 * doubles: single columns / rows / entries of tableaux too large to download whole. */
int  mi355x_tab_download_block(mi355x_tab *t, int64_t row0, int64_t n_rows, int64_t col0,
                               int64_t n_cols, double *host_block);
/* (entering column, pivot row) of the pivots made by solve calls since the last
 * upload / trace reset, oldest first; at most cap pairs are written, *n = number
 * of pivots recorded on the device (tracing holds up to 2^20 pairs). */
int  mi355x_tab_trace(mi355x_tab *t, int64_t *entering_cols, int64_t *pivot_rows,
                      int64_t cap, int64_t *n);

/* ---- asynchronous / measurement plumbing ------------------------------------------- */
/* Run the handle's kernels on an existing HIP stream (hipStream_t as void*), e.g.
 * torch.cuda.current_stream().cuda_stream -- NULL is HIP's default (null) stream, which is what
 * torch uses unless told otherwise.  use_own != 0: back to the handle's private stream. */
int  mi355x_tab_set_stream(mi355x_tab *t, void *hip_stream, int use_own);
/* Enqueue up to n_pivots iterations of price -> ratio -> pivot without any host
 * synchronisation (kernels turn into no-ops once the tableau is optimal/unbounded).
 * reset != 0 restarts the device-side pivot counter/status first. */
int  mi355x_tab_solve_async(mi355x_tab *t, int is_max, double fp_factor, int64_t n_pivots,
                            int reset);
/* Restart the device-side status / pivot counter of a handle and set its pivot cap
 * (0 = none) without touching the tableau. */
int  mi355x_tab_reset(mi355x_tab *t, int64_t max_pivots);
/* Wait for the stream; returns the device-side status (MI_OPTIMAL, MI_UNBOUNDED,
 * MI_MAX_PIVOTS, or MI_RUNNING when the enqueued iterations ran out first) and the number
 * of pivots done since the last reset.  That number can be smaller than what was enqueued even
 * with MI_RUNNING: after a non-finite entering column (the handle moved to the dense tableau) or
 * on a GPU shared with other work (the handle moved to the look-ahead form that needs no
 * co-resident workgroups) the rest of the request was dropped -- enqueue the difference again. */
int  mi355x_tab_sync(mi355x_tab *t, int64_t *n_pivots);
/* Per-launch HIP-event timing of the rank-1 update kernel (the bandwidth kernel):
 * enable = k > 0 brackets every k-th update launch with an event pair on the launch stream
 * (up to 4096 timed launches kept); 0 switches it off.  mi355x_tab_timing_read waits for the stream and returns
 * the number of timed launches, their summed and minimum duration in milliseconds,
 * then clears the record. */
int  mi355x_tab_timing_enable(mi355x_tab *t, int enable);
int  mi355x_tab_timing_read(mi355x_tab *t, int64_t *n_launches, double *sum_ms, double *min_ms);
/* Name of the rank-1 update kernel as it appears in rocprofv3 kernel traces. */
const char *mi355x_update_kernel_name(void);
/* Pivots one tableau-update launch of this handle applies in its current representation: > 1
 * when the solve loops run blocked (up to 16 pivots selected ahead, then applied by one k_sweep
 * launch; the timing above then brackets the sweeps), 1 for the plain k_update path.
 * Results do not depend on it. */
int  mi355x_tab_block_size(mi355x_tab *t);

/* ---- native host side of the hook: problem -> tableau -> solution ------------------- */
/* The reference's parsed `problem` struct (src/problem.lisp:45-53), variables identified by
 * their index in problem-vars.  A variable without a bounds entry is >= 0 (`positive`
 * mapping with offset 0); set_bounds(var, 0,_, 0,_) makes it free (`signed`, two columns).
 * op: 0 `<=`, 1 `>=`, 2 `=`; as after parsing, `<=` / `>=` rows carry rhs >= 0. */
int  mi355x_problem_create(mi355x_problem **out, int is_max, int64_t n_vars);
int  mi355x_problem_set_objective(mi355x_problem *p, const int64_t *var, const double *coef,
                                  int64_t nnz);
int  mi355x_problem_set_bounds(mi355x_problem *p, int64_t var, int has_lb, double lb, int has_ub,
                               double ub);
int  mi355x_problem_set_integer(mi355x_problem *p, int64_t var);
int  mi355x_problem_add_constraint(mi355x_problem *p, int op, const int64_t *var,
                                   const double *coef, int64_t nnz, double rhs);
void mi355x_problem_destroy(mi355x_problem *p);
/* The parsed problem as JSON text (inspection / tests).  Returns the length needed, writes at
 * most cap-1 characters plus a NUL. */
int64_t mi355x_problem_to_json(const mi355x_problem *p, char *buf, int64_t cap);
/* read-mps (src/external-formats.lisp:78-348): fixed-width MPS text -> problem.
 * default_is_max: 1 max, 0 min, -1 = the file's OBJSENSE section must say; rhs_id: name of the
 * RHS vector to use, NULL = the first one in the file; read_case: 0 upcase, 1 downcase,
 * 2 preserve, 3 invert (the reference's :read-case).  Variable names of the problem most
 * recently read by the calling thread, in problem-vars order: mi355x_mps_var_name. */
int  mi355x_problem_read_mps(const char *text, int64_t len, int default_is_max, const char *rhs_id,
                             int read_case, mi355x_problem **out);
/* The same with options.  flags = 0 is mi355x_problem_read_mps: single-variable rows are folded into
 * bounds exactly as the reference's loop does it (src/external-formats.lisp:312-323, quirks included:
 * a `<=` row writes (lb-max ub bound) into the upper bound, a `>=` row turns the variable integer, the
 * coefficient's sign is ignored, the constraint after a folded row is skipped).
 * MI_MPS_SINGLE_VARIABLE_ROWS_AS_MEANT: fold them the way the rows read instead (`<=` tightens the
 * upper bound, `>=` the lower bound, the sense flips for a negative coefficient, `=` fixes). */
#define MI_MPS_SINGLE_VARIABLE_ROWS_AS_MEANT 1
/* The default spelled out (flags = 0): the reference's loop, quirks included.  Since round 5 this is what
 * mi355x_problem_read_mps does (rounds 1-4 folded such rows "as meant": callers that want that behaviour back
 * pass MI_MPS_SINGLE_VARIABLE_ROWS_AS_MEANT to mi355x_problem_read_mps_ex).  Whenever the reference's loop
 * changes what a row means -- a `>=` / `=` row turns its variable INTEGER, a `<=` row takes the LARGER bound, the
 * constraint behind a folded row is skipped (its negative right-hand side not flipped) -- the call still
 * returns MI_OK and mi355x_last_error() holds a note naming the rows ("mps note: ..."; empty otherwise). */
#define MI_MPS_REFERENCE_COMPATIBLE 0
int  mi355x_problem_read_mps_ex(const char *text, int64_t len, int default_is_max, const char *rhs_id,
                                int read_case, int flags, mi355x_problem **out);
int64_t     mi355x_mps_var_count(void);
const char *mi355x_mps_var_name(int64_t i);
const char *mi355x_mps_objective_name(void);
/* build-tableau (src/simplex.lisp:142-328) in double-float, on the host.  which = 0: the main
 * tableau, 1: the artificial tableau (two-phase problems only).  Call once with NULL arrays for
 * the shape, again to fill matrix (rows*cols) and basis (rows-1).  Returns MI_UNBOUNDED for the
 * unbounded no-constraint special case (:170,:174). */
int  mi355x_build_tableau(const mi355x_problem *p, int which, int64_t *rows, int64_t *cols,
                          double *matrix, int64_t *basis, int *two_phase);
/* var-mapping entry of a variable (src/simplex.lisp:44-46): kind 0 positive, 1 negative,
 * 2 signed; first column; offset. */
int  mi355x_var_mapping(const mi355x_problem *p, int64_t var, int *kind, int64_t *col,
                        double *offset);
/* simplex-solver for LPs (src/simplex.lisp:506-542 without branch-and-bound): build-tableau,
 * upload, n-solve-tableau on the device (single- or two-phase), read back the objective row,
 * the RHS column and the basis only.  Returns MI_OPTIMAL and a solution, or MI_UNBOUNDED /
 * MI_INFEASIBLE / MI_UNSUPPORTED (integer variables) / an error. */
int  mi355x_simplex_solver(const mi355x_problem *p, double fp_tolerance, int device,
                           mi355x_solution **out);
/* The same solve as a resumable job, for a host that must never sit in an unbounded foreign call
 * (the reference's loop has no pivot cap and no anti-cycling rule, src/simplex.lisp:453-461; a Lisp
 * thread inside a foreign call cannot serve an interrupt).  begin: build-tableau + upload (declines
 * integer problems with MI_UNSUPPORTED, reports the unbounded no-constraint case as MI_UNBOUNDED).
 * step: at most max_pivots pivots (0 = no cap) of n-solve-tableau, across the phases of a two-phase
 * problem; *n_pivots = pivots of this call.  Returns MI_MAX_PIVOTS while the solve is still running
 * (call again: the continuation takes exactly the pivots one long call would), otherwise the final
 * status -- MI_OPTIMAL / MI_UNBOUNDED / MI_INFEASIBLE / MI_ART_NONZERO / MI_ART_STUCK, MI_CANCELLED
 * after mi355x_simplex_solver_cancel from another thread (a further step carries on).  finish: the
 * light read-back into a solution object (only after MI_OPTIMAL; MI_BAD_ARG otherwise) -- it always
 * consumes the job, as does abandon.  mi355x_simplex_solver = begin, step(0), finish. */
int  mi355x_simplex_solver_begin(const mi355x_problem *p, double fp_tolerance, int device,
                                 mi355x_solve **out);
int  mi355x_simplex_solver_step(mi355x_solve *job, int64_t max_pivots, int64_t *n_pivots);
int  mi355x_simplex_solver_cancel(mi355x_solve *job);
int  mi355x_simplex_solver_finish(mi355x_solve *job, mi355x_solution **out);
void mi355x_simplex_solver_abandon(mi355x_solve *job);
/* A LIST of problems (the hook takes one per call, src/solver.lisp:53-56; N small LPs one after the
 * other leave the GPU almost empty -- BASELINE config 4): what mi355x_simplex_solver returns for every
 * member, with the members solved side by side.  begin: build-tableau per member in C++, members of one
 * tableau shape and sense packed into ONE batch over n_devices GPUs (device_ids NULL = 0 .. n_devices-1;
 * two-phase members: a pair of batches), a member alone in its group as an ordinary job on the first
 * device; integer members are declined (MI_UNSUPPORTED) and unbounded no-constraint members decided here.
 * step: at most max_pivots pivots per member and phase (0 = no cap); returns MI_MAX_PIVOTS while some
 * member is still running (call again), MI_OK when every member has its final status.  status (n
 * entries, may be NULL): MI_RUNNING while undecided, then MI_OPTIMAL / MI_UNBOUNDED / MI_INFEASIBLE /
 * MI_ART_NONZERO / MI_ART_STUCK / MI_UNSUPPORTED.  finish: out[k] = the solution of member k (light
 * read-back) or NULL for a member without one; consumes the job, as does abandon. */
int  mi355x_simplex_solver_many_begin(const mi355x_problem *const *problems, int64_t n, double fp_tolerance,
                                      int n_devices, const int *device_ids, mi355x_solve_many **out);
int  mi355x_simplex_solver_many_step(mi355x_solve_many *job, int64_t max_pivots, int32_t *status);
int  mi355x_simplex_solver_many_finish(mi355x_solve_many *job, int32_t *status, mi355x_solution **out);
void mi355x_simplex_solver_many_abandon(mi355x_solve_many *job);
/* tableau-objective-value / tableau-variable / tableau-reduced-cost (src/simplex.lisp:74-120).
 * reduced_cost fails with MI_BAD_ARG for a variable without a lower bound, as the reference. */
int  mi355x_solution_objective_value(const mi355x_solution *s, double *out);
int  mi355x_solution_variable(const mi355x_solution *s, int64_t var, double *out);
int  mi355x_solution_reduced_cost(const mi355x_solution *s, int64_t var, double *out);
int  mi355x_solution_pivots(const mi355x_solution *s, int64_t *phase1, int64_t *phase2);
void mi355x_solution_destroy(mi355x_solution *s);

/* ---- batches of independent LPs (BASELINE config 4) -------------------------------- */
/* n_lps same-shape LPs stacked in one allocation; one (select, update) launch pair advances
 * every unfinished LP by one pivot.  host_matrices: n_lps * rows * cols doubles,
 * host_bases: n_lps * (rows-1) (may be NULL). */
int  mi355x_batch_create(mi355x_batch **out, int64_t n_lps, int64_t rows, int64_t cols,
                         const double *host_matrices, const int64_t *host_bases, int device);
/* The synthetic LP of mi355x_tab_create_synthetic, one per seed (seeds: n_lps host values). */
int  mi355x_batch_create_synthetic(mi355x_batch **out, int64_t n_lps, int64_t n_vars,
                                   int64_t n_cons, const uint64_t *seeds, int device);
/* n-solve-tableau (single phase) on every LP of the batch; status[i] (MI_OPTIMAL /
 * MI_UNBOUNDED / MI_MAX_PIVOTS) and n_pivots[i] per LP (host arrays, may be NULL).
 * max_pivots = 0: no cap.  Returns MI_OK or an error. */
int  mi355x_batch_solve(mi355x_batch *b, int is_max, double fp_factor, int64_t max_pivots,
                        int32_t *status, int64_t *n_pivots);
int  mi355x_batch_download(mi355x_batch *b, int64_t lp_index, double *host_matrix,
                           int64_t *host_basis, double *last_row, double *last_col);
/* HIP-event timing of the batched update launches (see mi355x_tab_timing_*). */
/* Optional: do now what the first mi355x_batch_solve would do first -- allocate and fill the
 * representation the solve loop runs on -- so that a timed solve contains no allocation. */
int  mi355x_batch_prepare(mi355x_batch *b);
int  mi355x_batch_timing_enable(mi355x_batch *b, int enable);
int  mi355x_batch_timing_read(mi355x_batch *b, int64_t *n_launches, double *sum_ms, double *min_ms);
void mi355x_batch_destroy(mi355x_batch *b);

/* The same without blocking the caller: the batch loop is driven by the host (blind chunks of
 * launches, one status read-back per chunk), so the asynchronous form runs it on a worker thread
 * of the library; mi355x_batch_sync waits for it and delivers what mi355x_batch_solve delivers.
 * One solve at a time per batch; the batch must not be touched in between.  (How a
 * single-threaded host -- the Lisp image -- keeps the sub-batches of several GPUs going.) */
int  mi355x_batch_solve_async(mi355x_batch *b, int is_max, double fp_factor, int64_t max_pivots);
int  mi355x_batch_sync(mi355x_batch *b, int32_t *status, int64_t *n_pivots);
/* mi355x_tab_cancel for batches (any thread): the solve in flight -- blocking, or on the library's
 * worker thread -- returns MI_CANCELLED (mi355x_batch_solve / _sync / mi355x_multibatch_solve)
 * after its current chunk of launches; status[i] = MI_RUNNING for the LPs it left unfinished,
 * every tableau whole.  (The per-LP kernels end a launch after 4096 pivots per LP at the latest.) */
int  mi355x_batch_cancel(mi355x_batch *b);

/* One batch over several devices (config 4 as specified: 1024 LPs over 8 GPUs): LP k lives in
 * sub-batch k / ceil(n_lps / n_devices) (contiguous blocks, one sub-batch per device), independent
 * units, no communication.  mi355x_multibatch_solve starts every sub-batch's loop on its own worker
 * thread and waits; status / n_pivots are indexed by the GLOBAL LP index.  device_ids: one per
 * sub-batch, or NULL = devices 0 .. n_devices-1 -- with fewer visible devices than that the
 * sub-batches become logical sub-batches on device 0 (own stream each; how the path is tested on
 * one GPU).  n_devices is capped at n_lps; mi355x_multibatch_info reports what is in use. */
typedef struct mi355x_multibatch mi355x_multibatch;
int  mi355x_multibatch_create(mi355x_multibatch **out, int64_t n_lps, int64_t rows, int64_t cols,
                              const double *host_matrices, const int64_t *host_bases, int n_devices,
                              const int *device_ids);
int  mi355x_multibatch_create_synthetic(mi355x_multibatch **out, int64_t n_lps, int64_t n_vars,
                                        int64_t n_cons, const uint64_t *seeds, int n_devices,
                                        const int *device_ids);
int  mi355x_multibatch_info(const mi355x_multibatch *mb, int *n_sub_batches, int *n_devices_used);
int  mi355x_multibatch_solve(mi355x_multibatch *mb, int is_max, double fp_factor, int64_t max_pivots,
                             int32_t *status, int64_t *n_pivots);
int  mi355x_multibatch_download(mi355x_multibatch *mb, int64_t lp_index, double *host_matrix,
                                int64_t *host_basis, double *last_row, double *last_col);
/* n-solve-tableau, two-phase branch (src/simplex.lisp:402-452), member by member of two matching
 * batches: member k of `art` is the artificial tableau (a min problem) of the problem whose main
 * tableau is member k of `main_mb` (same member count, row count and sub-batch layout).  Phase 1 runs
 * as the batch loop on `art`; the feasibility test (fp= 0 objective), the drive-out pivots of
 * artificial variables still basic (419-434: row fetches and single pivots on that member inside the
 * batch) and the hand-over (437-451) run per member on the devices; phase 2 is the batch loop on
 * `main_mb`.  status[k]: MI_OPTIMAL / MI_UNBOUNDED / MI_INFEASIBLE / MI_ART_NONZERO / MI_ART_STUCK as
 * mi355x_solve_two_phase.  n_pivots: two entries per member (phase 1 incl. drive-out pivots, phase
 * 2), may be NULL.  Read results with mi355x_multibatch_download(main_mb, k, ...).
 * mi355x_multibatch_two_phase_handover is the step between the phases on its own, for a caller that
 * drives both phases in bounded chunks (mi355x_multibatch_solve with a cap, what the Lisp glue does):
 * phase1_status = the per-member statuses phase 1 ended with (NULL: all MI_OPTIMAL); status[k] = MI_OK
 * when member k of main_mb is ready for phase 2, otherwise that member's final outcome (its main
 * tableau is then neutralised so that the batch loop of phase 2 passes over it); n_driveout per
 * member, may be NULL. */
int  mi355x_multibatch_two_phase_handover(mi355x_multibatch *art, mi355x_multibatch *main_mb,
                                          double fp_factor, const int32_t *phase1_status,
                                          int32_t *status, int64_t *n_driveout);
int  mi355x_multibatch_solve_two_phase(mi355x_multibatch *art, mi355x_multibatch *main_mb, int main_is_max,
                                       double fp_factor, int32_t *status, int64_t *n_pivots);
int  mi355x_multibatch_cancel(mi355x_multibatch *mb);
void mi355x_multibatch_destroy(mi355x_multibatch *mb);

/* ---- column-partitioned tableau: per-shard steps (BASELINE config 5) --------------- */
/* One tableau whose non-RHS columns are split across shards (one shard = one handle = one
 * GPU / rank; every shard keeps its own copy of the RHS column as its last column and updates
 * it redundantly).  One pivot = three local steps and two exchanges, all asynchronous on the
 * handle's stream (see mi355x_tab_set_stream), none needing the host to know who owns the
 * entering column:
 *   1. mi355x_shard_price       local pricing winner -> dev_out2 = {key, global column}
 *      -- exchange A: all-gather of the 2 doubles of every shard
 *   2. mi355x_shard_contribute  every shard derives the same global winner (lexicographic
 *      (key, column) minimum == lowest-index strict minimum) and applies the threshold;
 *      the owner writes the entering column's bit patterns (rows int64) to dev_col_bits,
 *      everyone else zeros; dev_ec = global entering column or -1
 *      -- exchange B: integer SUM all-reduce of dev_col_bits (= broadcast from the owner)
 *   3. mi355x_shard_pivot       ratio test on the exchanged column against the own RHS copy
 *      (same result on every shard), normalise own slice of the pivot row, rank-1 update
 *      of own slice, basis[cr] = global column.
 * col_offset = global index of the shard's first column.  dev_* are raw device addresses
 * (e.g. torch tensors' data_ptr()).  Status/pivot count: mi355x_tab_sync. */
/* A COMPACT shard stores only non-basic columns (plus the RHS copy): global_cols[j] is the
 * global logical column held in local slot j (cols-1 entries, host array).  The shard that
 * owns the entering column hands its slot over to the leaving basic column at every pivot
 * (DESIGN.md 4.5), so ownership of logical columns moves between slots but a shard's size never
 * changes and basic columns are stored nowhere.  basis[] holds global column indices.
 * mi355x_shard_columns reads the current slot -> global column map back. */
/* global_cols[j] == -1: slot j is DEAD -- stored and updated like any other column, never priced
 * and owned by no logical column (what the two-phase hand-over leaves in a shard that held
 * artificial columns only). */
int  mi355x_shard_set_compact(mi355x_tab *t, int64_t global_var_count, const int64_t *global_cols);
int  mi355x_shard_columns(mi355x_tab *t, int64_t *global_cols);
int  mi355x_shard_price(mi355x_tab *t, int is_max, int64_t col_offset, double *dev_out2);
int  mi355x_shard_contribute(mi355x_tab *t, const double *dev_gathered, int n_shards,
                             int64_t col_offset, double fp_factor, int64_t *dev_col_bits,
                             int64_t *dev_ec);
int  mi355x_shard_pivot(mi355x_tab *t, const int64_t *dev_col_bits, const int64_t *dev_ec,
                        double fp_factor);
/* Blocked form of the three steps: `step` = 0 .. 15 within a block.  The
 * exchanges are the same two per pivot; the shard's slice of the tableau is NOT updated by
 * mi355x_shard_la_pivot -- the pending pivots are chained through on whatever a step reads --
 * and mi355x_shard_sweep applies all pending pivots in one pass (call it after the last step of
 * a block, at the latest after 16 steps, and before anything else touches the shard). */
int  mi355x_shard_la_contribute(mi355x_tab *t, int step, const double *dev_gathered, int n_shards,
                                int64_t col_offset, double fp_factor, int64_t *dev_col_bits,
                                int64_t *dev_ec);
int  mi355x_shard_la_pivot(mi355x_tab *t, int step, const int64_t *dev_col_bits, const int64_t *dev_ec,
                           double fp_factor);
int  mi355x_shard_sweep(mi355x_tab *t);

/* ---- column-partitioned tableau: the whole solve behind one handle (BASELINE config 5) ---- */
/* n-solve-tableau (src/simplex.lisp:453-461, single phase) on ONE tableau whose non-RHS columns
 * are distributed over n_devices GPUs: the driver of the per-shard steps above lives in the
 * library, with the two exchanges per pivot issued from C++ on the shards' streams --
 *   * one process, several GPUs (what the Lisp host uses: `:devices n`): one shard and one host
 *     thread per device, RCCL communicators from ncclCommInitAll, exchange A = ncclAllGather of
 *     16 bytes per shard, exchange B = ncclAllReduce(int64 SUM) of the entering column's bit
 *     patterns (owner's bits + zeros == a broadcast whose root no host needs to know);
 *   * fewer visible devices than shards: the shards become LOGICAL shards on device 0 and the
 *     exchanges device-local kernels with the collectives' semantics (same pivots, same bits:
 *     how the partitioned path is tested on one GPU);
 *   * one process per GPU (torch.distributed.run / mpirun): every rank creates the handle with
 *     world/rank and the 128-byte id rank 0 obtained from mi355x_rccl_unique_id, the
 *     communicator comes from ncclCommInitRank, the per-pivot loop is still entirely in here.
 * Shards are compact (only non-basic columns are distributed) whenever the basis columns of the
 * uploaded tableau are exact unit vectors, dense otherwise; pivoting is blocked (16 pivots per
 * sweep of a shard's slice).  A failing RCCL call returns MI_RCCL_ERROR; the communicators are then
 * aborted (the other ranks' pending collectives could never complete) and the handle can only be
 * destroyed.
 * n_devices >= 1; a tableau with fewer distributable columns than that gets one shard per column
 * (mi355x_colpart_info reports the number in use).  Results are bit-identical to mi355x_tab_solve
 * on one device. */
typedef struct mi355x_colpart mi355x_colpart;
int  mi355x_colpart_create(mi355x_colpart **out, int64_t rows, int64_t cols,
                           const double *host_matrix, const int64_t *host_basis, int n_devices);
/* the same on chosen devices: device_ids[0 .. n_devices) (distinct), NULL = devices 0 .. n_devices-1.
 * Logical shards (fewer visible devices than shards) all live on device_ids[0]. */
int  mi355x_colpart_create_on(mi355x_colpart **out, int64_t rows, int64_t cols,
                              const double *host_matrix, const int64_t *host_basis, int n_devices,
                              const int *device_ids);
/* the synthetic LP of mi355x_tab_create_synthetic generated shard by shard in HBM (benchmarks) */
int  mi355x_colpart_create_synthetic(mi355x_colpart **out, int64_t n_vars, int64_t n_cons,
                                     uint64_t seed, int n_devices);
/* one process per GPU: this rank's shard of a world of `world` shards, on `device` */
int  mi355x_rccl_unique_id(void *id128);
int  mi355x_colpart_create_synthetic_rank(mi355x_colpart **out, int64_t n_vars, int64_t n_cons,
                                          uint64_t seed, int world, int rank, int device,
                                          const void *id128);
/* number of shards, of distinct physical devices they live on, and whether the exchanges are
 * RCCL collectives (1) or device-local kernels (0) */
int  mi355x_colpart_info(const mi355x_colpart *p, int *n_shards, int *n_devices_used, int *uses_rccl);
/* n-solve-tableau; max_pivots = 0: no cap.  MI_OPTIMAL / MI_UNBOUNDED / MI_MAX_PIVOTS /
 * MI_NONFINITE (compact shards, see above) or an error */
int  mi355x_colpart_solve(mi355x_colpart *p, int is_max, double fp_factor, int64_t max_pivots,
                          int64_t *n_pivots);
/* n-solve-tableau, two-phase branch (src/simplex.lisp:402-452) with the artificial tableau
 * column-partitioned (`art`: made by mi355x_colpart_create[_on] from the artificial tableau of
 * build-tableau; its basis columns are unit columns, so its shards are compact).  Phase 1 = the
 * partitioned loop on `art` (a min problem); the feasibility test (fp= 0 objective), the drive-out
 * pivots of artificials still basic (the owner contributes the chosen column, one exchange, every
 * shard pivots on the given row) and the hand-over run as in mi355x_solve_two_phase.  The main
 * tableau is never uploaded: its constraint rows ARE the artificial tableau's (:437-441), so every
 * shard keeps the main problem's columns among its slots (gathered on its own device, nothing moves
 * between devices; a shard left with artificial columns only keeps one as a dead slot that is
 * never priced), and its objective row -- main_objective_row: the last row of the main tableau,
 * main_cols doubles -- is re-eliminated over the basic rows on the devices (:444-451).
 * *main_out (also set on MI_UNBOUNDED) is the main tableau, solved by phase 2: read it with
 * mi355x_colpart_download (rows x main_cols), destroy it like any other handle.  `art` keeps the
 * artificial tableau as phase 1 and the drive-out pivots left it (download / destroy only: its
 * communicators now belong to *main_out).  One-process forms only (logical shards, or one shard
 * per visible device).  n_pivots[0] = phase 1 incl. drive-out pivots, n_pivots[1] = phase 2.
 * Returns what mi355x_solve_two_phase returns, or MI_UNSUPPORTED when the partition cannot follow
 * the reference bit for bit: an artificial tableau whose basis is not a set of unit columns, or a
 * drive-out pivot on a NEGATIVE element (it leaves -0.0 in basic columns, which compact shards do
 * not store) -- solve the caller's (untouched) tableaux with mi355x_solve_two_phase then. */
int  mi355x_colpart_solve_two_phase(mi355x_colpart *art, int64_t main_cols,
                                    const double *main_objective_row, int main_is_max,
                                    double fp_factor, int64_t *n_pivots, mi355x_colpart **main_out);
/* benchmarks: enqueue exactly n_pivots iterations (reset != 0: restart the pivot count), then
 * mi355x_colpart_sync waits, applies pending pivots and reports like mi355x_tab_sync */
int  mi355x_colpart_solve_async(mi355x_colpart *p, int is_max, double fp_factor, int64_t n_pivots,
                                int reset);
int  mi355x_colpart_sync(mi355x_colpart *p, int64_t *n_pivots);
/* mi355x_tab_cancel for mi355x_colpart_solve (any thread; the one-process forms -- logical shards or
 * one shard per visible device): MI_CANCELLED after the chunk of 64 .. 256 pivots in flight, every
 * shard swept.  One process per GPU: MI_UNSUPPORTED (the ranks would have to agree on the chunk, or
 * the others' collectives never complete) -- bound such solves with max_pivots. */
int  mi355x_colpart_cancel(mi355x_colpart *p);
/* read-back as mi355x_tab_download (any pointer may be NULL).  host_matrix (the whole logical
 * tableau) needs every shard in this process; basis, last row and last column are complete in
 * the one-process modes, and basis / last column also on every rank of the multi-process mode. */
int  mi355x_colpart_download(mi355x_colpart *p, double *host_matrix, int64_t *host_basis,
                             double *last_row, double *last_col);
int  mi355x_colpart_trace(mi355x_colpart *p, int64_t *entering_cols, int64_t *pivot_rows,
                          int64_t cap, int64_t *n);
void mi355x_colpart_destroy(mi355x_colpart *p);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_SIMPLEX_H */
