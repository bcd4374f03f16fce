// simplex_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the dense-simplex hot path.
//
// What the reference does per iteration (src/simplex.lisp:453-461):
//     find-entering-column (362-379) -> find-pivoting-row (382-389) -> n-pivot-row (337-359)
// as three scalar loops over a boxed (simple-array real 2).  Here the tableau stays resident in
// HBM and one iteration is a SELECT step followed by an UPDATE launch, enqueued blind by the
// host (every kernel tests the device-side status word first and is a no-op once the solve has
// terminated, so there is no per-pivot host synchronisation).
//
//   select  (latency-bound, ~R + 2C doubles of traffic)
//       price:  lowest-index strict arg-min (max problems) / arg-max (min problems) of the
//               objective row, threshold factor/8 * eps -- normally from the per-wave partial
//               winners the previous k_update left behind when it wrote the objective row
//       gather: snapshot col[r] = M[r][ec] of the entering column (the only strided access)
//       ratio:  lowest-index strict arg-min of rhs/col over rows with col > factor/2 * eps
//       scale:  prow[c] = M[cr][c] / M[cr][ec]   (true division)
//       bookkeeping: (ec, cr, status, basis, trace) in the device-side control block
//     k_select                          one 1024-thread workgroup (small tableaux: one launch)
//     k_select_gather + k_select_scale  many workgroups (large tableaux: a strided gather costs a
//                                       64-byte line per double and one workgroup's memory
//                                       pipeline moves only ~10 B/clk: 29.6 us -> 5 + 5 us at
//                                       4097 rows)
//
//   k_update  (the bandwidth kernel: every stored element is read once and written once)
//       M[r][c] = M[r][c] - col[r]*prow[c]   for r != cr   (product and difference rounded
//       M[cr][c] = prow[c]                                   separately: built with
//                                                            -ffp-contract=off, no FMA)
//       A workgroup owns a strip of <= 256 column pairs x 4 rows: the thread's two prow entries
//       live in registers, col[r] is wave-uniform (scalar loads), 16-byte coalesced
//       loads/stores, 4 rows in flight.  Small tiles dispatched x-fastest make the resident
//       workgroups cover one contiguous window that sweeps through memory once per launch
//       (6.2-6.4 TB/s on the 403 MB dense config-3 tableau, vs 5.1 TB/s with 32-row tiles).
//
// The snapshots col[]/prow[] remove the in-place hazard of the reference's loop order (rows read
// M[r][ec] and M[cr][c] while other rows are being overwritten) without changing a single
// rounding: every element sees exactly the operands the sequential loop would have used, every
// reduction is an exact comparison on (value, index) pairs, so results are bit-identical to the
// reference algorithm for any parallel schedule.
//
// That is the PER-PIVOT form (dense fall-back, step-wise entry points, lockstep batches).  The
// solve loops themselves run BLOCKED (section "blocked pivoting" below): up to 16 pivots are
// selected ahead of the tableau -- the objective row, one column, the RHS column and one row are
// evaluated as they WOULD be after the pending pivots -- and then applied to every stored element
// in one pass, with the same operands and roundings as 16 k_update launches:
//     k_la_block                         the look-ahead of a whole block as ONE launch of <= 32
//                                        persistent workgroups that exchange their reduction
//                                        candidates through memory (release/acquire records)
//     k_la_gather<j> + k_la_scale<j>     the same as two launches per step (tableaux too large
//                                        for the persistent form)
//     k_sweep                            applies the pending pivots (col values in SGPRs, prow in
//                                        registers: 2 v_mul_f64 + 2 v_add_f64 per pair and pivot)
//     k_batch_block                      all of it inside one workgroup per LP of a batch (LDS)
//     k_shard_la_contribute/_la_prepare  the look-ahead step of a column shard (the two
//                                        exchanges per pivot are the per-pivot path's)
//
// Further down: the compact representation [non-basic columns | RHS] the solve loops run on
// (basic columns never change under a pivot), the per-pivot batch kernel k_batch_solve, the
// per-pivot column-shard steps, the two-phase hand-over, and the synthetic-LP generator.
#include "simplex_kernels.h"
#include <type_traits>

namespace mi355x {

// ------------------------------------------------------------------ small helpers
struct ValIdx {
    double  v;
    int64_t i;   // < 0 : empty
    int64_t s;   // payload that travels with the winner (never compared): the physical slot of a
                 // pricing candidate / the bit pattern of the pivot element of a ratio candidate
};

// lexicographic (value, index) minimum; an empty slot loses against anything
// (branch-free on purpose: written with early returns the compiler produced a chain of divergent
// branches per call, and a workgroup reduction -- ten of these in sequence on a lone wave -- took
// 1.3 us)
__device__ __forceinline__ ValIdx vi_min(ValIdx a, ValIdx b)
{
    const bool a_empty = a.i < 0, b_empty = b.i < 0;
    const bool better  = (b.v < a.v) | ((b.v == a.v) & (b.i < a.i));
    const bool take_b  = a_empty | (!b_empty & better);
    ValIdx r;
    r.v = take_b ? b.v : a.v;
    r.i = take_b ? b.i : a.i;
    r.s = take_b ? b.s : a.s;
    return r;
}

// A pricing candidate (key = objective-row entry in key space, LOGICAL column, physical slot).
// find-entering-column (src/simplex.lisp:362-379) starts from column 0 and replaces the incumbent
// only by a strictly smaller entry.  With NaNs in the objective row (inf - inf after an overflow)
// that is: a NaN entry never replaces anything -- it is no candidate at all --, and a NaN in
// column 0 is never replaced: the winner is then column 0, (fp< NaN 0) fails, and the tableau
// counts as optimal.  Every reduction below is order-independent under exactly these rules: a NaN
// key leaves the candidate empty, except in logical column 0, where it becomes the unbeatable
// candidate (-inf, 0) whose payload is kNanColumn0; whoever consumes a pricing winner tests
// price_says_optimal().
constexpr int64_t kNanColumn0 = 0xffffffffll;       // (fits the 32 payload bits of an exchange record)

// (bias: a dense column shard numbers its columns from 0; its first GLOBAL column is t.col_bias)
__device__ __forceinline__ ValIdx price_cand(double key, int64_t logical, int64_t slot, int64_t bias = 0)
{
    ValIdx c; c.v = key; c.i = logical; c.s = slot;
    if (key != key) {
        if (logical + bias == 0) { c.v = -__builtin_inf(); c.s = kNanColumn0; }
        else              c.i = -1;
    }
    return c;
}

// (fp< v 0 factor/8) on the winner: nothing to enter <=> the tableau is optimal
__device__ __forceinline__ bool price_says_optimal(const ValIdx &e, double price_tol)
{
    return e.i < 0 || e.s == kNanColumn0 || !(e.v < 0.0 - price_tol);
}

// Wave-wide lexicographic minimum, result valid in lane 0.  A fixed binary tree -- lane l takes
// lane l+32, then l+16, l+8, ...  (NaN keys never get here: price_cand and the ratio tests drop them
// and apply the reference's scan-order rules for them separately, so the reduction is a plain
// associative minimum and its shape does not matter for the result.)  The two steps that cross rows of 16 lanes use
// __shfl_down (ds_bpermute: an LDS round trip per 32-bit word); the four steps inside a row use
// DPP row shifts, a few cycles each.  (All six as ds_bpermute made one reduction 1.3 us.)
template <int CTRL>
__device__ __forceinline__ long long dpp64(long long x)
{
    // old = own value: a lane whose partner lies outside its row (or is switched off) combines
    // with itself, exactly as __shfl_down past the end of the wave does
    const int xl = (int)(x & 0xffffffffll), xh = (int)((unsigned long long)x >> 32);
    const int lo = __builtin_amdgcn_update_dpp(xl, xl, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(xh, xh, CTRL, 0xf, 0xf, false);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ ValIdx dpp_validx(ValIdx x)
{
    ValIdx y;
    y.v = __longlong_as_double(dpp64<CTRL>(__double_as_longlong(x.v)));
    y.i = dpp64<CTRL>((long long)x.i);
    y.s = dpp64<CTRL>((long long)x.s);
    return y;
}
__device__ __forceinline__ ValIdx shfl_down_validx(ValIdx x, int off)
{
    ValIdx y;
    y.v = __shfl_down(x.v, off, 64);
    y.i = __shfl_down((long long)x.i, off, 64);
    y.s = __shfl_down((long long)x.s, off, 64);
    return y;
}
__device__ __forceinline__ ValIdx wave_reduce_min(ValIdx x)
{
    x = vi_min(x, shfl_down_validx(x, 32));
    x = vi_min(x, shfl_down_validx(x, 16));
    x = vi_min(x, dpp_validx<0x108>(x));          // row_shl:8  (lane l <- lane l+8 of its row)
    x = vi_min(x, dpp_validx<0x104>(x));          // row_shl:4
    x = vi_min(x, dpp_validx<0x102>(x));          // row_shl:2
    x = vi_min(x, dpp_validx<0x101>(x));          // row_shl:1
    return x;
}

constexpr int kSelThreads = 1024;
constexpr int kSelWaves   = kSelThreads / 64;

// Block-wide lexicographic arg-min over THREADS threads; result valid in every thread.
template <int THREADS = kSelThreads>
__device__ __forceinline__ ValIdx block_reduce_min(ValIdx x, double *s_v, long long *s_i)
{
    __shared__ long long s_s[THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    x = wave_reduce_min(x);
    __syncthreads();                       // protects s_v/s_i/s_s reuse across calls
    if (lane == 0) { s_v[wave] = x.v; s_i[wave] = x.i; s_s[wave] = x.s; }
    __syncthreads();
    ValIdx r;
    r.v = s_v[0]; r.i = s_i[0]; r.s = s_s[0];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w) {
        ValIdx y; y.v = s_v[w]; y.i = s_i[w]; y.s = s_s[w];
        r = vi_min(r, y);
    }
    return r;
}

// find-entering-column (src/simplex.lisp:362-379) over columns [0, ncols) of the objective
// row.  sgn = +1 for max problems (arg-min), -1 for min problems (arg-max of v == arg-min of
// -v: negation is exact and order reversing).  Returns the lowest-index strict extremum in
// key space; the caller applies the threshold.
// All of a thread's loads are issued before the first use (kBatch independent 16-byte loads
// in flight per thread): this kernel is one workgroup, so its time is the number of
// serialised memory round trips, not bandwidth.
constexpr int kBatch = 8;

template <int THREADS = kSelThreads>
__device__ __forceinline__ ValIdx block_price(const double *__restrict__ obj, int64_t ncols,
                                              double sgn, double *s_v, long long *s_i,
                                              const int64_t *__restrict__ p2l = nullptr,
                                              const int64_t bias = 0)
{
    // p2l != nullptr (compact representation): physical slot -> logical column; the winner is
    // the lexicographic (key, LOGICAL column) minimum, i.e. still the reference's lowest-index
    // strict minimum, whatever order the columns are stored in.
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    const int64_t npair = ncols >> 1;                 // obj is 128-byte aligned (row start)
    const double2 *obj2 = reinterpret_cast<const double2 *>(obj);
    for (int64_t base = 0; base < npair; base += (int64_t)kBatch * THREADS) {
        double2 v[kBatch];
#pragma unroll
        for (int g = 0; g < kBatch; ++g) {
            const int64_t p = base + (int64_t)g * THREADS + threadIdx.x;
            v[g] = p < npair ? obj2[p] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int g = 0; g < kBatch; ++g) {
            const int64_t p = base + (int64_t)g * THREADS + threadIdx.x;
            if (p < npair) {
                const ValIdx c0 = price_cand(v[g].x * sgn, p2l ? p2l[2 * p] : 2 * p, 2 * p, p2l ? 0 : bias);
                const ValIdx c1 = price_cand(v[g].y * sgn, p2l ? p2l[2 * p + 1] : 2 * p + 1, 2 * p + 1, p2l ? 0 : bias);
                best = vi_min(vi_min(best, c0), c1);
            }
        }
    }
    if ((ncols & 1) && threadIdx.x == 0) {          // odd tail element
        ValIdx t = price_cand(obj[ncols - 1] * sgn, p2l ? p2l[ncols - 1] : ncols - 1, ncols - 1, p2l ? 0 : bias);
        best = vi_min(best, t);
    }
    return block_reduce_min<THREADS>(best, s_v, s_i);
}

// Same result from the per-wave partial winners that k_update left behind when it wrote the
// objective row (key space already, lowest index per wave): n_part (value, index) pairs.
template <int THREADS = kSelThreads>
__device__ __forceinline__ ValIdx block_price_partials(const double *__restrict__ pv,
                                                       const int64_t *__restrict__ pi,
                                                       const int64_t *__restrict__ ps,
                                                       int n_part, double *s_v, long long *s_i)
{
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    for (int k = threadIdx.x; k < n_part; k += THREADS) {
        ValIdx t; t.v = pv[k]; t.i = pi[k]; t.s = ps[k];
        best = vi_min(best, t);
    }
    return block_reduce_min<THREADS>(best, s_v, s_i);
}

// Gather the entering column and run find-pivoting-row (src/simplex.lisp:382-389).
// col_src == nullptr: read M[r][ec] (and snapshot it into t.col); otherwise the column was
// supplied by another shard and is read from col_src (and copied into t.col).
template <int THREADS = kSelThreads>
__device__ __forceinline__ ValIdx block_gather_ratio(const TabView &t, int64_t ec,
                                                     const double *__restrict__ col_src,
                                                     double ratio_thr, double *s_v, long long *s_i,
                                                     const double *__restrict__ rhs_src = nullptr,
                                                     int *nonfinite = nullptr)
{
    const int64_t m = t.rows - 1, vc = t.cols - 1;
    int bad = 0, nanq = 0;
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    ValIdx first; first.v = 0.0; first.i = -1; first.s = 0;   // this thread's first eligible row (all keys equal: lowest row wins)
    for (int64_t base = 0; base < t.rows; base += (int64_t)kBatch * THREADS) {
        double a[kBatch], b[kBatch];
#pragma unroll
        for (int g = 0; g < kBatch; ++g) {           // the strided gathers: all in flight at once
            const int64_t r = base + (int64_t)g * THREADS + threadIdx.x;
            a[g] = r < t.rows ? (col_src ? col_src[r] : t.M[r * t.ld + ec]) : 0.0;
            b[g] = r < m ? (rhs_src ? rhs_src[r] : t.M[r * t.ld + vc]) : 0.0;
        }
#pragma unroll
        for (int g = 0; g < kBatch; ++g) {
            const int64_t r = base + (int64_t)g * THREADS + threadIdx.x;
            if (r < t.rows) { t.col[r] = a[g]; bad |= !(fabs(a[g]) <= 1.7976931348623157e308); }
            if (r < m && ratio_thr < a[g]) {         // (fp< 0 a factor/2) -> (< (+ 0 thr) a)
                const double q = b[g] / a[g];
                if (first.i < 0) { first.i = r; first.s = __double_as_longlong(a[g]); }
                if (q != q) nanq = 1;              // no candidate (unless it is the FIRST eligible row: below)
                else if (best.i < 0 || q < best.v) { best.v = q; best.i = r; best.s = __double_as_longlong(a[g]); }
            }
        }
    }
    if (nonfinite) *nonfinite = bad;               // this thread's entries only
    best = block_reduce_min<THREADS>(best, s_v, s_i);
    // find-pivoting-row takes the first eligible row and replaces it only by a strictly smaller
    // quotient: a NaN quotient (inf / inf, NaN / x after an overflow) wins iff its row is the
    // FIRST eligible one, and is no candidate otherwise.
    if (__syncthreads_or(nanq)) {
        first = block_reduce_min<THREADS>(first, s_v, s_i);
        if (first.i >= 0) {
            const double a0 = __longlong_as_double(first.s);
            const double b0 = rhs_src ? rhs_src[first.i] : t.M[first.i * t.ld + vc];
            const double q0 = b0 / a0;
            if (q0 != q0) { best.v = q0; best.i = first.i; best.s = first.s; }
        }
    }
    return best;
}

// The same decision from a contiguous snapshot of the entering column (col[r], r < rows) -- what the
// split select falls back to when one of its workgroups met a NaN quotient (see above).
template <int THREADS>
__device__ __forceinline__ ValIdx block_ratio_from_snapshot(const TabView &t, const double *__restrict__ col,
                                                            double ratio_thr, double *s_v, long long *s_i)
{
    const int64_t m = t.rows - 1, vc = t.cols - 1;
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    ValIdx first; first.v = 0.0; first.i = -1; first.s = 0;
    for (int64_t r = threadIdx.x; r < m; r += THREADS) {
        const double a = col[r];
        if (ratio_thr < a) {
            const double q = t.M[r * t.ld + vc] / a;
            if (first.i < 0) { first.i = r; first.s = __double_as_longlong(a); }
            if (q == q && (best.i < 0 || q < best.v)) { best.v = q; best.i = r; best.s = __double_as_longlong(a); }
        }
    }
    best  = block_reduce_min<THREADS>(best, s_v, s_i);
    first = block_reduce_min<THREADS>(first, s_v, s_i);
    if (first.i >= 0) {
        const double q0 = t.M[first.i * t.ld + vc] / __longlong_as_double(first.s);
        if (q0 != q0) { best.v = q0; best.i = first.i; best.s = first.s; }
    }
    return best;
}

// prow[c] = M[cr][c] / M[cr][ec]  (src/simplex.lisp:343-348), padding columns zeroed.
// unit_slot >= 0 (compact representation): that physical column is about to hold the LEAVING
// basic column, whose pre-pivot content is the unit vector e_cr, so its pivot-row entry is 1.0.
__device__ __forceinline__ double2 scale_pair(const TabView &t, int64_t p, double2 v,
                                              double row_scale, int64_t unit_slot)
{
    if (2 * p == unit_slot)     v.x = 1.0;
    if (2 * p + 1 == unit_slot) v.y = 1.0;
    double2 o;
    o.x = (2 * p     < t.cols) ? v.x / row_scale : 0.0;
    o.y = (2 * p + 1 < t.cols) ? v.y / row_scale : 0.0;
    return o;
}

template <int THREADS = kSelThreads>
__device__ __forceinline__ void block_scale_row(const TabView &t, int64_t cr, double row_scale,
                                                int64_t unit_slot = -1)
{
    const double2 *__restrict__ src = reinterpret_cast<const double2 *>(t.M + cr * t.ld);
    double2 *dst = reinterpret_cast<double2 *>(t.prow);
    const int64_t npair = t.ld >> 1;
    for (int64_t base = 0; base < npair; base += (int64_t)kBatch * THREADS) {
        double2 v[kBatch];
#pragma unroll
        for (int g = 0; g < kBatch; ++g) {
            const int64_t p = base + (int64_t)g * THREADS + threadIdx.x;
            v[g] = p < npair ? src[p] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int g = 0; g < kBatch; ++g) {
            const int64_t p = base + (int64_t)g * THREADS + threadIdx.x;
            if (p < npair) dst[p] = scale_pair(t, p, v[g], row_scale, unit_slot);
        }
    }
}

// Compact representation, bookkeeping of one pivot (ONE thread): the entering logical column
// becomes basic in row cr and gives up its physical slot to the leaving basic column.
__device__ __forceinline__ void swap_columns(const TabView &t, int64_t ec_log, int64_t cr,
                                             int64_t slot)
{
    const int64_t leaving = t.basis[cr];            // read BEFORE record_pivot overwrites it
    t.p2l[slot]    = leaving;
    t.l2p[leaving] = slot;
    t.l2p[ec_log]  = -1;
}

// `c0` is the control block as loaded at kernel entry: the bookkeeping at the END of a select
// kernel is then a handful of independent stores instead of a chain of dependent
// load-modify-store round trips on one thread (which used to be the kernel's tail).
__device__ __forceinline__ void record_pivot(const TabView &t, const Ctl &c0, int64_t ec, int64_t cr)
{
    Ctl *ctl = t.ctl;
    ctl->ec = ec;
    ctl->cr = cr;
    if (t.basis) t.basis[cr] = ec;                  // src/simplex.lisp:358
    if (t.trace_ec && c0.trace_n < t.trace_cap) {
        t.trace_ec[c0.trace_n] = ec;
        t.trace_cr[c0.trace_n] = cr;
    }
    ctl->trace_n  = c0.trace_n + 1;
    ctl->n_pivots = c0.n_pivots + 1;
}

// A batch of same-shape LPs is one TabView plus per-LP element strides; grid.z = LP index
// (all strides are zero for a single tableau, where grid.z == 1).
__device__ __forceinline__ TabView lp_slice_at(TabView t, const int64_t z)
{
    t.M      += z * t.zs_M;
    t.basis  += z * t.zs_basis;
    t.col    += z * t.zs_col;
    t.prow   += z * t.zs_prow;
    t.part_v += z * t.zs_part;
    t.part_i += z * t.zs_part;
    t.part_s += z * t.zs_part;
    t.ctl    += z;
    if (t.p2l) { t.p2l += z * t.zs_p2l; t.l2p += z * t.zs_l2p; }
    if (t.blk && t.n_lps > 1) {                     // per-LP block state of a batch
        t.blk += z;
        t.bk_col += z * t.zs_bk; t.bk_prow += z * t.zs_bkp;
        t.bk_rmask += z * t.zs_rm; t.bk_smask += z * t.zs_sm;
    }
    return t;
}
__device__ __forceinline__ TabView lp_slice(TabView t) { return lp_slice_at(t, blockIdx.z); }

// ------------------------------------------------------------------ select kernels
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_select(TabView t, double sgn, double price_tol,
                                                   double ratio_thr, int n_part)
{
    __shared__ double    s_v[THREADS / 64];
    __shared__ long long s_i[THREADS / 64];
    t = lp_slice(t);
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;                            // in flight together with the pricing inputs
    const int64_t m = t.rows - 1, vc = t.cols - 1;

    // n_part > 0: the preceding k_update of this tableau priced the new objective row
    const ValIdx e = n_part > 0 ? block_price_partials<THREADS>(t.part_v, t.part_i, t.part_s, n_part, s_v, s_i)
                                : block_price<THREADS>(t.M + m * t.ld, vc, sgn, s_v, s_i, t.p2l, t.col_bias);
    if (c0.status != kRunning) return;
    // (fp< v 0 factor/8): v < 0 - tol ; min problems: (fp> v 0 factor/8) <=> -v < 0 - tol
    if (price_says_optimal(e, price_tol)) {
        if (threadIdx.x == 0) ctl->status = 0;      // MI_OPTIMAL
        return;
    }
    if (c0.max_pivots > 0 && c0.n_pivots >= c0.max_pivots) {
        __syncthreads();
        if (threadIdx.x == 0) ctl->status = 3;      // MI_MAX_PIVOTS
        return;
    }
    const int64_t ec   = e.i;                       // LOGICAL column
    const int64_t slot = e.s;                       // where it is stored (travels with the winner)
    int bad = 0;
    const ValIdx q = block_gather_ratio<THREADS>(t, slot, nullptr, ratio_thr, s_v, s_i, nullptr, &bad);
    if (t.p2l && __syncthreads_or(bad)) {           // see kNeedDense
        if (threadIdx.x == 0) ctl->status = kNeedDense;
        return;
    }
    if (q.i < 0) {
        if (threadIdx.x == 0) ctl->status = 1;      // MI_UNBOUNDED
        return;
    }
    const int64_t cr = q.i;
    const double row_scale = __longlong_as_double(q.s);   // M[cr][slot], carried by the winner
    block_scale_row<THREADS>(t, cr, row_scale, t.p2l ? slot : -1);
    if (t.p2l) {                                    // the slot now holds the leaving column: e_cr
        for (int64_t r = threadIdx.x; r < t.rows; r += THREADS)
            t.M[r * t.ld + slot] = (r == cr) ? 1.0 : 0.0;
        if (threadIdx.x == 0) swap_columns(t, ec, cr, slot);
    }
    if (threadIdx.x == 0) record_pivot(t, c0, ec, cr);
}

// ---- the same select, split over many workgroups (large tableaux) -------------------------
// One workgroup's memory pipeline moves ~10 bytes/cycle; the strided column gather costs a
// full 64-byte line per useful double, so for thousands of rows a single workgroup needs tens
// of microseconds.  Split: k_select_gather (ceil(rows/128) workgroups: every workgroup derives
// the entering column on its own from the same inputs, gathers 128 rows of it and of the RHS
// column, leaves a ratio-test partial) then k_select_scale (ceil(ld/512) workgroups: every
// workgroup reduces the partials to the same pivot row and normalises its slice of that row).
constexpr int kGatherThreads = 128;
constexpr int kScaleThreads  = 256;

__global__ __launch_bounds__(kGatherThreads) void k_select_gather(TabView t, double sgn,
                                                                  double price_tol,
                                                                  double ratio_thr, int n_part)
{
    __shared__ double    s_v[kGatherThreads / 64];
    __shared__ long long s_i[kGatherThreads / 64];
    t = lp_slice(t);
    double  *rp_v = t.part_v + t.part_cap / 2;      // ratio partials: upper half of the buffers
    int64_t *rp_i = t.part_i + t.part_cap / 2;
    int64_t *rp_s = t.part_s + t.part_cap / 2;
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;                            // one load of the whole control block ...
    const int64_t m = t.rows - 1, vc = t.cols - 1;
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    // ... in flight together with the pricing inputs: the status is only tested afterwards (this
    // kernel is a chain of dependent memory round trips; a launch after termination merely
    // reads a few values it does not use)
    const ValIdx e = n_part > 0
        ? block_price_partials<kGatherThreads>(t.part_v, t.part_i, t.part_s, n_part, s_v, s_i)
        : block_price<kGatherThreads>(t.M + m * t.ld, vc, sgn, s_v, s_i, t.p2l, t.col_bias);
    if (c0.status != kRunning) return;
    if (price_says_optimal(e, price_tol)) {
        if (leader) ctl->status = 0;                // MI_OPTIMAL
        return;
    }
    if (c0.max_pivots > 0 && c0.n_pivots >= c0.max_pivots) {
        if (leader) ctl->status = 3;                // MI_MAX_PIVOTS
        return;
    }
    const int64_t ec   = e.i;                       // LOGICAL column
    const int64_t slot = e.s;                       // its physical column
    const int64_t r = (int64_t)blockIdx.x * kGatherThreads + threadIdx.x;
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    if (r < t.rows) {
        const double a = t.M[r * t.ld + slot];
        const double b = r < m ? t.M[r * t.ld + vc] : 0.0;
        t.col[r] = a;
        if (t.p2l && !(fabs(a) <= 1.7976931348623157e308)) atomicOr(&ctl->poison, 1);   // see kNeedDense
        if (r < m && ratio_thr < a) {
            const double q = b / a;
            if (q != q) atomicOr(&ctl->poison, 2);   // a NaN quotient: k_select_scale decides from the snapshot
            else { best.v = q; best.i = r; best.s = __double_as_longlong(a); }
        }
    }
    best = block_reduce_min<kGatherThreads>(best, s_v, s_i);
    if (threadIdx.x == 0) { rp_v[blockIdx.x] = best.v; rp_i[blockIdx.x] = best.i; rp_s[blockIdx.x] = best.s; }
    if (leader) { ctl->ec = ec; ctl->slot = slot; }
}

__global__ __launch_bounds__(kScaleThreads) void k_select_scale(TabView t, int n_rp, double ratio_thr)
{
    __shared__ double    s_v[kScaleThreads / 64];
    __shared__ long long s_i[kScaleThreads / 64];
    t = lp_slice(t);
    const double  *rp_v = t.part_v + t.part_cap / 2;
    const int64_t *rp_i = t.part_i + t.part_cap / 2;
    const int64_t *rp_s = t.part_s + t.part_cap / 2;
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;                            // in flight together with the ratio partials
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    ValIdx q = block_price_partials<kScaleThreads>(rp_v, rp_i, rp_s, n_rp, s_v, s_i);
    if (c0.status != kRunning) return;
    if (c0.poison && t.p2l) {                       // the gather met an inf / NaN: see kNeedDense
        if (leader) ctl->status = kNeedDense;
        return;
    }
    // dense tableau, a NaN quotient somewhere (now or earlier in this solve: the flag stays up):
    // every workgroup takes the decision again from the snapshot of the column, with the
    // reference's first-eligible-row rule (block_gather_ratio)
    if (c0.poison & 2) q = block_ratio_from_snapshot<kScaleThreads>(t, t.col, ratio_thr, s_v, s_i);
    if (q.i < 0) {
        if (leader) ctl->status = 1;                // MI_UNBOUNDED
        return;
    }
    const int64_t cr = q.i;
    const double row_scale = __longlong_as_double(q.s);   // M[cr][slot], carried by the winner
    const int64_t ec   = c0.ec;                     // LOGICAL column (written by the gather)
    const int64_t slot = t.p2l ? c0.slot : -1;      // compact: the slot the leaving column takes
    const int64_t npair = t.ld >> 1;
    const int64_t p = (int64_t)blockIdx.x * kScaleThreads + threadIdx.x;
    if (p < npair) {
        const double2 v = reinterpret_cast<const double2 *>(t.M + cr * t.ld)[p];
        reinterpret_cast<double2 *>(t.prow)[p] = scale_pair(t, p, v, row_scale, slot);
    }
    if (t.p2l) {
        // the slot now holds the leaving basic column, whose pre-pivot content is e_cr; the
        // only reader of M[cr][slot] above substitutes 1.0, so the overwrite cannot race
        for (int64_t r = p; r < t.rows; r += (int64_t)gridDim.x * kScaleThreads)
            t.M[r * t.ld + slot] = (r == cr) ? 1.0 : 0.0;
    }
    if (leader) {
        if (t.p2l) swap_columns(t, ec, cr, slot);
        record_pivot(t, c0, ec, cr);
    }
}

// find-entering-column only: ctl->ec = column or -1.  With out2 (shard pricing) the local
// best key (v*sgn) and its GLOBAL column (col_offset + local, as a double; -1 = none) go to a
// device buffer instead, without the threshold (applied once on the global best later).
// system-scope one-granule store / load (exchange mode 2: another GPU, or another process, is at the other end)
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kSelThreads) void k_price_only(TabView t, double sgn, double price_tol,
                                                           int64_t col_offset, double *out2,
                                                           int n_part, P2pArgs x)
{
    __shared__ double    s_v[kSelWaves];
    __shared__ long long s_i[kSelWaves];
    const int64_t m = t.rows - 1, vc = t.cols - 1;
    const ValIdx e = n_part > 0 ? block_price_partials(t.part_v, t.part_i, t.part_s, n_part, s_v, s_i)
                                : block_price(t.M + m * t.ld, vc, sgn, s_v, s_i, t.p2l, t.col_bias);
    // compact shards price GLOBAL columns already
    // (-2: the objective entry of GLOBAL column 0 is a NaN -- see price_cand: everybody stops)
    const double pk = e.i < 0 ? 0.0 : e.v;
    const double pc = e.i < 0 ? -1.0 : (e.s == kNanColumn0 ? -2.0 : (double)(e.i + (t.p2l ? 0 : col_offset)));
    if (threadIdx.x == 0) {
        if (out2) { out2[0] = pk; out2[1] = pc; }
        else      t.ctl->ec = price_says_optimal(e, price_tol) ? -1 : e.i;
    }
    // exchange A, producer: my pair into slot `rank` of EVERY shard -- only while the solve is
    // running: the iterations a host enqueues blind behind a terminating pivot e must stay silent,
    // or their tags e+2, e+4 ... would overwrite the pair of pivot e in a peer that has not polled
    // it yet (the consumers of those iterations do not wait, so nothing paces this shard any more)
    if (x.peers && (int)threadIdx.x < x.lay.world && t.ctl->status == kRunning) {
        const unsigned long long kb = (unsigned long long)__double_as_longlong(pk), cb = (unsigned long long)__double_as_longlong(pc);
        unsigned long long *dst = x.peers[threadIdx.x] + x.lay.pair_off(x.epoch & 1u, x.rank);
        const unsigned long long tg = (unsigned long long)x.epoch << 32;
        st_sys(dst + 0, tg | (kb & 0xffffffffull));
        st_sys(dst + 1, tg | (kb >> 32));
        st_sys(dst + 2, tg | (cb & 0xffffffffull));
        st_sys(dst + 3, tg | (cb >> 32));
    }
}

// find-pivoting-row only (given ec): ctl->cr = row or -1.  Does not touch the tableau.
__global__ __launch_bounds__(kSelThreads) void k_ratio_only(TabView t, int64_t ec, double ratio_thr)
{
    __shared__ double    s_v[kSelWaves];
    __shared__ long long s_i[kSelWaves];
    const ValIdx q = block_gather_ratio(t, ec, nullptr, ratio_thr, s_v, s_i);
    if (threadIdx.x == 0) t.ctl->cr = q.i;
}

// Forced pivot (n-pivot-row with caller-chosen ec, cr): snapshot column + scaled row.
__global__ __launch_bounds__(kSelThreads) void k_prepare_pivot(TabView t, int64_t ec, int64_t cr)
{
    for (int64_t r = threadIdx.x; r < t.rows; r += kSelThreads)
        t.col[r] = t.M[r * t.ld + ec];
    block_scale_row(t, cr, t.M[cr * t.ld + ec]);
    if (threadIdx.x == 0) {
        Ctl c0 = *t.ctl;
        t.ctl->status = kRunning;
        record_pivot(t, c0, ec, cr);
    }
}

// Column-partitioned tableau, exchange step.  `gathered` holds (key, global column) of every
// shard's local pricing winner (all-gathered, 2 doubles per shard).  Every shard derives the
// same global winner -- lexicographic (key, column) minimum = the sequential lowest-index strict
// minimum -- applies the pricing threshold, and contributes to the column exchange: the owner
// of the entering column writes the BIT PATTERNS of its entries, everyone else zeros, so an
// integer sum all-reduce delivers the owner's column to every shard exactly (no float addition,
// signed zeros preserved) without any shard needing to know the owner on the host.
__global__ __launch_bounds__(kSelThreads) void k_shard_contribute(TabView t, const double *gathered,
                                                                 int n_shards, int64_t col_offset,
                                                                 double price_tol,
                                                                 long long *bits_out, int64_t *ec_out)
{
    // Iterations enqueued past termination must stay no-ops: the pricing they were given is
    // that of launches which did nothing (stale or never-written partials), so neither trust it
    // nor index with it.
    const bool running = t.ctl->status == kRunning;
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    bool nan0 = false;                              // some shard holds a NaN in global column 0's objective entry
    for (int k = 0; k < n_shards; ++k) {
        ValIdx c; c.v = gathered[2 * k]; c.i = (int64_t)gathered[2 * k + 1]; c.s = 0;
        nan0 |= gathered[2 * k + 1] == -2.0;
        best = vi_min(best, c);
    }
    const int64_t ec = (running && !nan0 && !price_says_optimal(best, price_tol)) ? best.i : -1;
    // dense shard: a fixed block of logical columns; compact shard: whatever non-basic columns
    // currently live in its slots (l2p: global logical column -> local slot, -1 = not here)
    const int64_t lc = ec < 0 ? -1 : (t.l2p ? t.l2p[ec] : ec - col_offset);
    const bool mine = ec >= 0 && lc >= 0 && lc < t.cols - 1;
    // The strided gathers (entering column on the owner, RHS column on everyone: snapshotted
    // contiguously for the ratio test of k_shard_prepare) run here, spread over many workgroups.
    for (int64_t r = blockIdx.x * (int64_t)kSelThreads + threadIdx.x; r < t.rows;
         r += (int64_t)gridDim.x * kSelThreads) {
        bits_out[r] = mine ? __double_as_longlong(t.M[r * t.ld + lc]) : 0ll;
        if (ec >= 0) t.rhs[r] = t.M[r * t.ld + (t.cols - 1)];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *ec_out = ec;
}

// The same contribution for a column the CALLER chose (drive-out pivots of the two-phase
// hand-over): the owner writes the column's bit patterns, everyone else zeros.
__global__ __launch_bounds__(kSelThreads) void k_shard_forced_contribute(TabView t, int64_t ec, int64_t col_offset,
                                                                        long long *bits_out, int64_t *ec_out)
{
    const int64_t lc = t.l2p ? t.l2p[ec] : ec - col_offset;
    const bool mine = lc >= 0 && lc < t.cols - 1;
    for (int64_t r = blockIdx.x * (int64_t)kSelThreads + threadIdx.x; r < t.rows;
         r += (int64_t)gridDim.x * kSelThreads) {
        bits_out[r] = mine ? __double_as_longlong(t.M[r * t.ld + lc]) : 0ll;
        t.rhs[r] = t.M[r * t.ld + (t.cols - 1)];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *ec_out = ec;
}

// Column-partitioned tableau, local step after the exchange: ratio test on the (now global)
// entering column against the shard's own RHS copy (identical on every shard => identical
// pivot row everywhere, no further exchange), row scale = col[cr] (== M[cr][ec] bit for bit),
// normalise the local slice of row cr.  *ec_dev < 0: the tableau is optimal.
// forced_cr >= 0 (the drive-out pivots of the two-phase hand-over, src/simplex.lisp:419-434: the caller
// chose column AND row): no ratio test, the exchanged column is only snapshotted.
__global__ __launch_bounds__(kSelThreads) void k_shard_prepare(TabView t, const double *col_src,
                                                              const int64_t *ec_dev,
                                                              double ratio_thr, int64_t forced_cr)
{
    __shared__ double    s_v[kSelWaves];
    __shared__ long long s_i[kSelWaves];
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;                            // one load of the whole control block
    if (c0.status != kRunning) return;
    const int64_t global_ec = *ec_dev;
    if (global_ec < 0) {
        if (threadIdx.x == 0) ctl->status = 0;      // MI_OPTIMAL
        return;
    }
    if (c0.max_pivots > 0 && c0.n_pivots >= c0.max_pivots) {
        if (threadIdx.x == 0) ctl->status = 3;      // MI_MAX_PIVOTS
        return;
    }
    int bad = 0;
    ValIdx q; q.v = 0.0; q.i = forced_cr; q.s = 0;
    if (forced_cr < 0) {
        q = block_gather_ratio(t, 0, col_src, ratio_thr, s_v, s_i, t.rhs, &bad);
    } else {
        for (int64_t r = threadIdx.x; r < t.rows; r += kSelThreads) {
            const double a = col_src[r];
            t.col[r] = a;
            bad |= !(fabs(a) <= 1.7976931348623157e308);
        }
    }
    if (t.p2l && __syncthreads_or(bad)) {           // compact shard: cannot follow the reference
        if (threadIdx.x == 0) ctl->status = 6;      // MI_NONFINITE
        return;
    }
    if (q.i < 0) {
        if (threadIdx.x == 0) ctl->status = 1;      // MI_UNBOUNDED
        return;
    }
    const int64_t cr = q.i;
    // compact shard that owns the entering column: its slot is taken over by the leaving basic
    // column (pre-pivot content e_cr); the other shards only update what they store
    const int64_t slot = t.p2l ? t.l2p[global_ec] : -1;
    block_scale_row(t, cr, col_src[cr], slot);
    if (slot >= 0) {
        for (int64_t r = threadIdx.x; r < t.rows; r += kSelThreads)
            t.M[r * t.ld + slot] = (r == cr) ? 1.0 : 0.0;
        if (threadIdx.x == 0) swap_columns(t, global_ec, cr, slot);
    }
    if (threadIdx.x == 0) record_pivot(t, c0, global_ec, cr);   // basis holds GLOBAL column indices
}

// ------------------------------------------------------------------ the bandwidth kernel
typedef double vec2d __attribute__((ext_vector_type(2)));   // one global_load/store_dwordx4
//
// Tile = (strip_pairs <= BLOCK 16-byte column pairs) x (tr rows).  A thread owns ONE column pair:
// its two entries of prow sit in VGPRs for the whole tile; per row it issues one
// global_load_dwordx4, 2 v_mul_f64 + 2 v_add_f64 (never fused: -ffp-contract=off) and one
// global_store_dwordx4.  U rows are in flight per thread before the first use.  col[r] is
// uniform across the workgroup -> scalar loads.  tr and strip_pairs are chosen by the launcher
// so that the whole grid is resident in ONE balanced round (no tail wave of workgroups).
// The workgroups that write the objective row also price it for the next iteration: every
// wave leaves its lowest-index arg-min (in key space v*sgn) in part_v/part_i.
template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void k_update(TabView t, const int tr, const int strip_pairs,
                                                  const double sgn, const int price,
                                                  const int reverse)
{
    t = lp_slice(t);
    double *__restrict__ M = t.M;
    const int64_t ld = t.ld, rows = t.rows, vc = t.cols - 1;
    const double *__restrict__ col  = t.col;
    const double *__restrict__ prow = t.prow;
    const Ctl *__restrict__ ctl = t.ctl;
    double  *__restrict__ part_v = price ? t.part_v : nullptr;
    int64_t *__restrict__ part_i = t.part_i;
    int64_t *__restrict__ part_s = t.part_s;
    if (ctl->status != kRunning) return;
    const int64_t cr   = ctl->cr;
    const int64_t ldv  = ld >> 1;                              // row length in 16-byte pairs
    const int64_t pair = (int64_t)blockIdx.x * strip_pairs + threadIdx.x;
    const bool active  = (int)threadIdx.x < strip_pairs && pair < ldv;
    // Consecutive launches sweep the tableau in opposite directions (reverse flips the row-band
    // order): whatever part of the tableau the previous sweep left in the 256 MiB Infinity Cache
    // is what the next sweep touches first.
    const int64_t band = reverse ? (int64_t)gridDim.y - 1 - blockIdx.y : (int64_t)blockIdx.y;
    const int64_t r0 = band * tr;
    const int64_t r1 = (r0 + tr < rows) ? r0 + tr : rows;
    const bool prices = (r1 == rows) && part_v != nullptr;     // this tile holds the objective row
    if (!active && !prices) return;                            // no workgroup barrier below

    vec2d *Mp = reinterpret_cast<vec2d *>(M) + pair;
    vec2d last; last.x = 0.0; last.y = 0.0;
    if (active) {
        const vec2d p = reinterpret_cast<const vec2d *>(prow)[pair];
        auto ld2 = [&](int64_t r) -> vec2d {
            if constexpr (NT) return __builtin_nontemporal_load(Mp + r * ldv);
            else              return Mp[r * ldv];
        };
        auto st2 = [&](int64_t r, vec2d v) {
            if constexpr (NT) __builtin_nontemporal_store(v, Mp + r * ldv);
            else              Mp[r * ldv] = v;
        };
        int64_t r = r0;
        for (; r + U <= r1; r += U) {
            vec2d v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld2(r + u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double s = col[r + u];
                const double m0 = s * p.x;                     // rounded product
                const double m1 = s * p.y;
                vec2d o;
                o.x = v[u].x - m0;                             // rounded difference
                o.y = v[u].y - m1;
                if (r + u == cr) o = p;                        // the pivot row itself
                st2(r + u, o);
                last = o;
            }
        }
        for (; r < r1; ++r) {                                  // row tail of the tile
            const vec2d x = ld2(r);
            const double s = col[r];
            const double m0 = s * p.x;
            const double m1 = s * p.y;
            vec2d o;
            o.x = x.x - m0;
            o.y = x.y - m1;
            if (r == cr) o = p;
            st2(r, o);
            last = o;
        }
    }
    if (prices) {                                              // `last` = new objective-row entries
        ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
        const int64_t c0 = 2 * pair;
        if (active && c0 < vc) {
            ValIdx c = price_cand(last.x * sgn, t.p2l ? t.p2l[c0] : c0, c0, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        if (active && c0 + 1 < vc) {
            ValIdx c = price_cand(last.y * sgn, t.p2l ? t.p2l[c0 + 1] : c0 + 1, c0 + 1, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        best = wave_reduce_min(best);
        if ((threadIdx.x & 63) == 0) {
            const int slot = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
            part_v[slot] = best.v;
            part_i[slot] = best.i;
            part_s[slot] = best.s;
        }
    }
}

// ------------------------------------------------------------------ blocked pivoting
// k_update moves every stored element through HBM once per pivot.  Nothing in the algorithm needs
// the whole tableau between two pivots: find-entering-column reads the objective row,
// find-pivoting-row one column and the RHS column, and the normalisation one row.  Those four
// pieces can be evaluated AS THEY WOULD BE after the pivots selected so far in the block -- every
// pending pivot i is the map  x -> x - col_i[r]*prow_i[c]  (x -> prow_i[c] on its own pivot row),
// applied in pivot order with the operands the sequential loop would have used -- so pivot j+1
// is selected before pivot j has touched the tableau, and one k_sweep launch then applies all k
// pending pivots to every element while it is in registers: 2 x 8 bytes of HBM traffic per
// element per BLOCK instead of per pivot.  Per element the operation sequence (k rounded
// products, k rounded differences, in pivot order) is exactly that of k k_update launches, so the
// results stay bit-identical to the reference's n-pivot-row loop (src/simplex.lisp:337-359).
//
//   k_la_gather<J> / k_la_scale<J>   look-ahead step J of a block (the split select, plus the
//                                    pending chain on what it reads; J is a template parameter so
//                                    that all the chain operands are loaded up front)
//   k_la_select<J>                   the same in one workgroup (small tableaux)
//   k_sweep                          applies blk->n_pending pivots; runs even when the solve has
//                                    just terminated (the pivots selected before the terminating
//                                    step must still be applied)
//
// Compact representation only: pivot i's entering column gives its slot to the leaving basic
// column, whose pre-pivot content is e_cr -- in the chain that is a RESET of the element to
// (r == cr_i ? 1 : 0) before pivot i is applied (k_select_scale stores e_cr into the slot for
// the same reason).

// pending pivot i applied to element x of (row r, column c): colv = col_i[r], prowv = prow_i[c]
__device__ __forceinline__ double pend(double x, bool is_slot, bool is_cr, double colv, double prowv)
{
    if (is_slot) x = is_cr ? 1.0 : 0.0;
    const double prod = colv * prowv;                          // rounded product
    const double d = x - prod;                                 // rounded difference
    return is_cr ? prowv : d;
}

// wave-uniform 64-bit value held in VGPRs -> SGPRs (so that addresses built from it are scalar)
__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffll));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// The wave-uniform operands of the pending chain (one value per pending pivot) are fetched by
// ONE vector load -- lane i fetches the value of pivot i -- and handed out with v_readlane: a
// single round trip whatever the number of pending pivots (as scalar loads the compiler issued
// them one after the other, each waiting for the previous one).
__device__ __forceinline__ double lane_value(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(b & 0xffffffffll), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int64_t lane_value(int64_t v, int lane)
{
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xffffffffll), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)((uint64_t)v >> 32), lane);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Each of the two kernels is: final reduction of the partials the previous launch left ->
// operands that depend on its result -> chain -> partials for the next launch.  Everything that
// does NOT depend on the reduction result is requested before it, so that it travels together
// with the partials (one memory round trip) and only the few dependent operands form the second.
template <int J>
__global__ __launch_bounds__(kGatherThreads) void k_la_gather(TabView t, double sgn, double price_tol,
                                                              double ratio_thr, int n_part)
{
    __shared__ double    s_v[kGatherThreads / 64];
    __shared__ long long s_i[kGatherThreads / 64];
    constexpr int JJ = J > 0 ? J : 1;
    double  *rp_v = t.part_v + t.part_cap / 2;
    int64_t *rp_i = t.part_i + t.part_cap / 2;
    int64_t *rp_s = t.part_s + t.part_cap / 2;
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    BlockCtl *blk = t.blk;
    const int64_t m = t.rows - 1, vc = t.cols - 1, ld = t.ld;
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    const int64_t r = (int64_t)blockIdx.x * kGatherThreads + threadIdx.x;
    // independent of the entering column: RHS entry, col snapshots and pivot rows of the pending
    // pivots, their RHS entries
    double  b = (r < m) ? t.M[r * ld + vc] : 0.0;
    double  ci[JJ];
    const int lane = threadIdx.x & 63;
    const bool  lj = lane < J;
    const double  v_pb = lj ? t.bk_prow[(int64_t)lane * ld + vc] : 0.0;
    const int64_t v_cr = lj ? blk->cr[lane] : -1;
    const int64_t v_sl = lj ? blk->slot[lane] : -1;
#pragma unroll
    for (int i = 0; i < J; ++i)
        ci[i] = (r < t.rows) ? t.bk_col[(int64_t)i * t.bk_stride + r] : 0.0;
    if (J == 0) {                                   // a new block starts (whatever the status)
        if (leader) blk->n_pending = 0;
        const int64_t n = t.bk_stride > (ld >> 1) ? t.bk_stride : (ld >> 1);
        for (int64_t idx = r; idx < n; idx += (int64_t)gridDim.x * kGatherThreads) {
            if (idx < t.bk_stride) t.bk_rmask[idx] = 0u;
            if (idx < (ld >> 1))   t.bk_smask[idx] = 0u;
        }
    }
    // J > 0: the partials were left by k_la_scale<J-1> (objective row after pivot J-1)
    ValIdx e;
    if (J == 0 && n_part <= 0) e = block_price<kGatherThreads>(t.M + m * ld, vc, sgn, s_v, s_i, t.p2l, t.col_bias);
    else e = block_price_partials<kGatherThreads>(t.part_v, t.part_i, t.part_s, n_part, s_v, s_i);
    if (c0.status != kRunning) return;
    if (price_says_optimal(e, price_tol)) {
        if (leader) ctl->status = 0;                // MI_OPTIMAL
        return;
    }
    if (c0.max_pivots > 0 && c0.n_pivots >= c0.max_pivots) {
        if (leader) ctl->status = 3;                // MI_MAX_PIVOTS
        return;
    }
    const int64_t ec = e.i, slot = uniform64(e.s);
    const double v_pa = lj ? t.bk_prow[(int64_t)lane * ld + slot] : 0.0;
    double a = (r < t.rows) ? t.M[r * ld + slot] : 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) {                   // (all lanes: v_readlane ignores exec)
        const bool is_cr = r == lane_value(v_cr, i);
        a = pend(a, slot == lane_value(v_sl, i), is_cr, ci[i], lane_value(v_pa, i));
        b = pend(b, false, is_cr, ci[i], lane_value(v_pb, i));
    }
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    if (r < t.rows) {
        t.bk_col[(int64_t)J * t.bk_stride + r] = a;
        if (!(fabs(a) <= 1.7976931348623157e308)) atomicOr(&ctl->poison, 1);   // see kNeedDense
        if (r < m && ratio_thr < a) {
            const double q = b / a;
            if (q != q) atomicOr(&ctl->poison, 1);   // a NaN quotient: decided on the dense path (block_gather_ratio)
            else { best.v = q; best.i = r; best.s = __double_as_longlong(a); }
        }
    }
    best = block_reduce_min<kGatherThreads>(best, s_v, s_i);
    if (threadIdx.x == 0) { rp_v[blockIdx.x] = best.v; rp_i[blockIdx.x] = best.i; rp_s[blockIdx.x] = best.s; }
    if (leader) { ctl->ec = ec; ctl->slot = slot; }
}

// Pivot row cr and the objective row, both through the pending chain; the objective row
// additionally through pivot J itself, priced on the way out (per-wave partials for step J+1 or
// for nobody if this was the last step: the sweep prices again).  The thread that owns the slot
// does the bookkeeping: it is the only one that needs the leaving column's index, and the only
// p2l entry that changes is the one no other thread reads.
template <int J>
__global__ __launch_bounds__(kScaleThreads) void k_la_scale(TabView t, int n_rp, double sgn)
{
    __shared__ double    s_v[kScaleThreads / 64];
    __shared__ long long s_i[kScaleThreads / 64];
    constexpr int JJ = J > 0 ? J : 1;
    const double  *rp_v = t.part_v + t.part_cap / 2;
    const int64_t *rp_i = t.part_i + t.part_cap / 2;
    const int64_t *rp_s = t.part_s + t.part_cap / 2;
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    BlockCtl *blk = t.blk;
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    const int64_t m = t.rows - 1, vc = t.cols - 1, ldv = t.ld >> 1;
    const int64_t p = (int64_t)blockIdx.x * kScaleThreads + threadIdx.x;
    const bool in = p < ldv;
    const double2 *M2 = reinterpret_cast<const double2 *>(t.M);
    double2 *P2 = reinterpret_cast<double2 *>(t.bk_prow);
    // independent of the pivot row: the objective row, the pending pivots' rows / slots / prow
    // entries / objective-row col entries, this step's objective-row col entry, the column maps
    double2 z = in ? M2[m * ldv + p] : make_double2(0.0, 0.0);
    double2 pi[JJ];
    const int lane = threadIdx.x & 63;
    const bool  lj = lane < J;
    const double  v_cm = lj ? t.bk_col[(int64_t)lane * t.bk_stride + m] : 0.0;
    const int64_t v_cr = lj ? blk->cr[lane] : -1;
    const int64_t v_sl = lj ? blk->slot[lane] : -1;
#pragma unroll
    for (int i = 0; i < J; ++i)
        pi[i] = in ? P2[(int64_t)i * ldv + p] : make_double2(0.0, 0.0);
    const double cmj = t.bk_col[(int64_t)J * t.bk_stride + m];
    const int64_t c0i = 2 * p;
    const int64_t l0 = (in && c0i < vc) ? t.p2l[c0i] : -1;
    const int64_t l1 = (in && c0i + 1 < vc) ? t.p2l[c0i + 1] : -1;
    const ValIdx q = block_price_partials<kScaleThreads>(rp_v, rp_i, rp_s, n_rp, s_v, s_i);
    if (c0.status != kRunning) return;
    if (c0.poison) {
        if (leader) ctl->status = kNeedDense;
        return;
    }
    if (q.i < 0) {
        if (leader) ctl->status = 1;                // MI_UNBOUNDED
        return;
    }
    const int64_t cr = uniform64(q.i);
    const double piv = __longlong_as_double(q.s);
    const int64_t ec = c0.ec, slot = c0.slot;
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    const double v_ccr = lj ? t.bk_col[(int64_t)lane * t.bk_stride + cr] : 0.0;
    double2 y = in ? M2[cr * ldv + p] : make_double2(0.0, 0.0);   // row cr
#pragma unroll
    for (int i = 0; i < J; ++i) {                   // (all lanes: v_readlane ignores exec)
        const bool    is_cr = cr == lane_value(v_cr, i);
        const int64_t sl = lane_value(v_sl, i);
        const double  ccr = lane_value(v_ccr, i), cm = lane_value(v_cm, i);
        y.x = pend(y.x, 2 * p     == sl, is_cr, ccr, pi[i].x);
        y.y = pend(y.y, 2 * p + 1 == sl, is_cr, ccr, pi[i].y);
        z.x = pend(z.x, 2 * p     == sl, false, cm, pi[i].x);
        z.y = pend(z.y, 2 * p + 1 == sl, false, cm, pi[i].y);
    }
    if (in) {
        const bool own = (p == (slot >> 1));
        const int64_t leaving = own ? t.basis[cr] : -1;
        const double2 pr = scale_pair(t, p, y, piv, slot);
        P2[(int64_t)J * ldv + p] = pr;
        z.x = pend(z.x, 2 * p     == slot, false, cmj, pr.x);
        z.y = pend(z.y, 2 * p + 1 == slot, false, cmj, pr.y);
        if (c0i < vc) {
            ValIdx c = price_cand(z.x * sgn, (c0i == slot) ? leaving : l0, c0i);
            best = vi_min(best, c);
        }
        if (c0i + 1 < vc) {
            ValIdx c = price_cand(z.y * sgn, (c0i + 1 == slot) ? leaving : l1, c0i + 1);
            best = vi_min(best, c);
        }
        if (own) {
            swap_columns(t, ec, cr, slot);
            record_pivot(t, c0, ec, cr);
            blk->cr[J] = cr;
            blk->slot[J] = slot;
            blk->n_pending = J + 1;
            t.bk_rmask[cr] |= 1u << J;              // this thread is the only writer in the launch
            t.bk_smask[slot >> 1] |= 1u << (J + 16 * (int)(slot & 1));
        }
    }
    best = wave_reduce_min(best);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (kScaleThreads / 64) + (threadIdx.x >> 6);
        t.part_v[w] = best.v;
        t.part_i[w] = best.i;
        t.part_s[w] = best.s;
    }
}

// ---- column shards, blocked: the per-pivot exchanges stay what they are -- one
// 16-byte all-gather and one column all-reduce per pivot -- but the shard's slice of the tableau is
// swept once per block.  Step j of a block on every shard: price (k_price_only, from the partials
// the previous step left) -> exchange -> k_shard_la_contribute -> exchange -> k_shard_la_prepare;
// after the last step k_sweep.  The chain of the pending pivots needs col_i (every shard has the
// whole exchanged column), prow_i on the shard's own columns and on its RHS copy (local), and the
// pivot rows (identical everywhere), so it needs no exchange of its own.
__global__ __launch_bounds__(256) void k_shard_la_contribute(TabView t, int j, const double *gathered,
                                                            int n_shards, int64_t col_offset,
                                                            double price_tol, long long *bits_out,
                                                            int64_t *ec_out, P2pArgs x)
{
    __shared__ double s_g[2 * 64];
    const bool running = t.ctl->status == kRunning;
    BlockCtl *blk = t.blk;
    const int64_t ldv = t.ld >> 1, vcl = t.cols - 1;
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    if (x.peers) {
        // exchange A, consumer (every workgroup for itself): wait for every shard's pair of this pivot
        int lost = 0;
        if (running && (int)threadIdx.x < n_shards) {
            const unsigned long long *src = x.mine + x.lay.pair_off(x.epoch & 1u, (int)threadIdx.x);
            unsigned long long g[4];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) { g[k] = ld_sys(src + k); ok &= (unsigned)(g[k] >> 32) == x.epoch; }
                if (ok) break;
                if (spins > x.max_spins) { lost = 1; break; }
            }
            s_g[2 * threadIdx.x]     = __longlong_as_double((long long)(((g[1] & 0xffffffffull) << 32) | (g[0] & 0xffffffffull)));
            s_g[2 * threadIdx.x + 1] = __longlong_as_double((long long)(((g[3] & 0xffffffffull) << 32) | (g[2] & 0xffffffffull)));
        }
        if (__syncthreads_or(lost)) {
            if (threadIdx.x == 0) t.ctl->status = kExchangeLost;
            if (gid == 0) *ec_out = -1;
            return;
        }
        gathered = s_g;
    }
    if (j == 0) {                                   // a new block starts (whatever the status)
        if (gid == 0) blk->n_pending = 0;
        const int64_t n = t.bk_stride > ldv ? t.bk_stride : ldv;
        for (int64_t idx = gid; idx < n; idx += gsz) {
            if (idx < t.bk_stride) t.bk_rmask[idx] = 0u;
            if (idx < ldv)         t.bk_smask[idx] = 0u;
        }
    }
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    bool nan0 = false;                              // some shard holds a NaN in global column 0's objective entry
    for (int k = 0; k < n_shards; ++k) {
        ValIdx c; c.v = gathered[2 * k]; c.i = (int64_t)gathered[2 * k + 1]; c.s = 0;
        nan0 |= gathered[2 * k + 1] == -2.0;
        best = vi_min(best, c);
    }
    const int64_t ec = (running && !nan0 && !price_says_optimal(best, price_tol)) ? best.i : -1;
    const int64_t lc = ec < 0 ? -1 : (t.l2p ? t.l2p[ec] : ec - col_offset);
    const bool mine = ec >= 0 && lc >= 0 && lc < vcl;
    // The RHS copy is carried from step to step: t.rhs holds it through the pending pivots 0 .. j-2
    // (step j - 1 left it there), so this step adds ONE link -- the same operations in the same
    // order as chaining all j links from the tableau, a j-th of the loads (every shard does this
    // on all its rows at every step; only the owner chains a column).
    const int64_t crp = j > 0 ? blk->cr[j - 1] : -1;
    const double  prp = j > 0 ? t.bk_prow[(int64_t)(j - 1) * t.ld + vcl] : 0.0;
    for (int64_t r = gid; r < t.rows; r += gsz) {
        double a = mine ? t.M[r * t.ld + lc] : 0.0;
        if (ec >= 0) {
            double b = j == 0 ? t.M[r * t.ld + vcl] : t.rhs[r];
            if (j > 0) b = pend(b, false, r == crp, t.bk_col[(int64_t)(j - 1) * t.bk_stride + r], prp);
            if (mine)                                   // four links' operands requested together: the chain is
                for (int i0 = 0; i0 < j; i0 += 4) {     // j dependent subtractions, not j dependent round trips
                    double ci[4], pi[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = i0 + k < j ? i0 + k : j - 1;
                        ci[k] = t.bk_col[(int64_t)i * t.bk_stride + r];
                        pi[k] = t.bk_prow[(int64_t)i * t.ld + lc];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (i0 + k < j) a = pend(a, lc == blk->slot[i0 + k], r == blk->cr[i0 + k], ci[k], pi[k]);
                }
            t.rhs[r] = b;
        }
        if (!x.peers) bits_out[r] = mine ? __double_as_longlong(a) : 0ll;
        else if (mine) {
            // exchange B, producer: the owner writes the column's granules straight into EVERY shard's buffer
            const unsigned long long vb = (unsigned long long)__double_as_longlong(a), tg = (unsigned long long)x.epoch << 32;
            const int64_t off = x.lay.col_off(x.epoch & 1u) + 2 * r;
            for (int q = 0; q < x.lay.world; ++q) {
                st_sys(x.peers[q] + off, tg | (vb & 0xffffffffull));
                st_sys(x.peers[q] + off + 1, tg | (vb >> 32));
            }
        }
    }
    if (gid == 0) *ec_out = ec;
}

__global__ __launch_bounds__(kSelThreads) void k_shard_la_prepare(TabView t, int j, const double *col_src,
                                                                 const int64_t *ec_dev, double ratio_thr,
                                                                 double sgn)
{
    __shared__ double    s_v[kSelWaves];
    __shared__ long long s_i[kSelWaves];
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    BlockCtl *blk = t.blk;
    if (c0.status != kRunning) return;
    const int64_t global_ec = *ec_dev;
    if (global_ec < 0) {
        if (threadIdx.x == 0) ctl->status = 0;      // MI_OPTIMAL
        return;
    }
    if (c0.max_pivots > 0 && c0.n_pivots >= c0.max_pivots) {
        if (threadIdx.x == 0) ctl->status = 3;      // MI_MAX_PIVOTS
        return;
    }
    int bad = 0;
    const ValIdx q = block_gather_ratio(t, 0, col_src, ratio_thr, s_v, s_i, t.rhs, &bad);
    if (t.p2l && __syncthreads_or(bad)) {           // compact shard: cannot follow the reference
        if (threadIdx.x == 0) ctl->status = 6;      // MI_NONFINITE
        return;
    }
    if (q.i < 0) {
        if (threadIdx.x == 0) ctl->status = 1;      // MI_UNBOUNDED
        return;
    }
    const int64_t cr = q.i, m = t.rows - 1, vcl = t.cols - 1, ldv = t.ld >> 1;
    const double piv = col_src[cr];
    const int64_t slot = t.p2l ? t.l2p[global_ec] : -1;   // compact owner: the slot the leaving column takes
    for (int64_t r = threadIdx.x; r < t.rows; r += kSelThreads)
        t.bk_col[(int64_t)j * t.bk_stride + r] = col_src[r];
    const double cmj = col_src[m];
    const double2 *M2 = reinterpret_cast<const double2 *>(t.M);
    double2 *P2 = reinterpret_cast<double2 *>(t.bk_prow);
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    int64_t leaving = -1;
    if (slot >= 0) leaving = t.basis[cr];           // (read by everyone before thread 0 overwrites it below)
    __syncthreads();
    for (int64_t p = threadIdx.x; p < ldv; p += kSelThreads) {
        double2 y = M2[cr * ldv + p], z = M2[m * ldv + p];
        for (int i = 0; i < j; ++i) {
            const int64_t cri = blk->cr[i], sli = blk->slot[i];
            const double  ccr = t.bk_col[(int64_t)i * t.bk_stride + cr], cm = t.bk_col[(int64_t)i * t.bk_stride + m];
            const double2 pi = P2[(int64_t)i * ldv + p];
            y.x = pend(y.x, 2 * p     == sli, cr == cri, ccr, pi.x);
            y.y = pend(y.y, 2 * p + 1 == sli, cr == cri, ccr, pi.y);
            z.x = pend(z.x, 2 * p     == sli, false, cm, pi.x);
            z.y = pend(z.y, 2 * p + 1 == sli, false, cm, pi.y);
        }
        const double2 pr = scale_pair(t, p, y, piv, slot);
        P2[(int64_t)j * ldv + p] = pr;
        z.x = pend(z.x, 2 * p     == slot, false, cmj, pr.x);
        z.y = pend(z.y, 2 * p + 1 == slot, false, cmj, pr.y);
        const int64_t c0i = 2 * p;
        if (c0i < vcl) {
            ValIdx c = price_cand(z.x * sgn, (c0i == slot) ? leaving : (t.p2l ? t.p2l[c0i] : c0i), c0i, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        if (c0i + 1 < vcl) {
            ValIdx c = price_cand(z.y * sgn, (c0i + 1 == slot) ? leaving : (t.p2l ? t.p2l[c0i + 1] : c0i + 1), c0i + 1, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
    }
    best = wave_reduce_min(best);                    // per-wave pricing partials for the next step
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        t.part_v[w] = best.v;
        t.part_i[w] = best.i;
        t.part_s[w] = best.s;
    }
    __syncthreads();                                 // every p2l read above precedes the swap
    if (threadIdx.x == 0) {
        if (slot >= 0) {
            swap_columns(t, global_ec, cr, slot);
            t.bk_smask[slot >> 1] |= 1u << (j + 16 * (int)(slot & 1));
        }
        record_pivot(t, c0, global_ec, cr);          // basis holds GLOBAL column indices
        blk->cr[j] = cr;
        blk->slot[j] = slot;
        blk->n_pending = j + 1;
        t.bk_rmask[cr] |= 1u << j;
    }
}

// ---- the same local step spread over many workgroups (large shards) -----------------------------
// k_shard_la_prepare is ONE workgroup: fine for a shard of a few thousand rows / column pairs, a
// hundred microseconds for a shard that holds tens of thousands (config 5 on few GPUs).  The split
// form is the k_la_gather / k_la_scale pair of the single-tableau path with the shard's inputs:
//   k_shard_la_ratio     the exchanged column is col_J: store it, ratio-test it against the
//                        shard's RHS copy (brought up to date by k_shard_la_contribute), one
//                        partial per workgroup;
//   k_shard_la_scale<J>  every workgroup reduces the partials to the same pivot row, chains its
//                        slice of that row and of the objective row through the J pending pivots
//                        (J a template parameter: all operands requested up front), prow_J, the
//                        objective row through pivot J priced on the way out (per-wave partials).
// Exchange mode 2 where a shard has its device (or at least its stream and the GPU's scheduler) to
// itself: pricing pair out, everybody's pairs in, the entering column (chained by its owner and
// pushed to the peers, or waited for) and the ratio-test partials as ONE launch -- what
// k_price_only + k_shard_la_contribute + k_shard_la_ratio do in three.  A consumer and the
// producer it waits for are then the same kernel on different shards, so this form needs the
// shards' launches to run CONCURRENTLY (one device / process each); logical shards that share a
// stream keep the three launches, where every producer of an exchange is enqueued before its
// consumers.  Nobody waits in a cycle: a shard's pair is pushed by its first workgroup before that
// workgroup waits for anything, and the owner of the column waits for nothing after the pairs.
// n_part: pricing partials the previous step left (> 0; the first pivot after an upload goes
// through the three launches).  One ratio partial per workgroup.
__global__ __launch_bounds__(256) void k_shard_p2p_step(TabView t, int j, int n_part, int n_shards,
                                                       int64_t col_offset, double price_tol, double ratio_thr,
                                                       int64_t *ec_out, P2pArgs x)
{
    __shared__ double    s_v[256 / 64];
    __shared__ long long s_i[256 / 64];
    __shared__ double    s_g[2 * 64];
    double  *rp_v = t.part_v + t.part_cap / 2;      // ratio partials: upper half of the buffers
    int64_t *rp_i = t.part_i + t.part_cap / 2;
    int64_t *rp_s = t.part_s + t.part_cap / 2;
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    const bool running = c0.status == kRunning;
    BlockCtl *blk = t.blk;
    const int64_t ldv = t.ld >> 1, vcl = t.cols - 1, m = t.rows - 1;
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    const unsigned long long tg = (unsigned long long)x.epoch << 32;
    // ---- exchange A, producer (the first workgroup): this shard's pricing pair into slot `rank` of EVERY shard
    // (only while running: a finished shard goes silent, see k_price_only)
    if (blockIdx.x == 0 && running) {
        const ValIdx e = block_price_partials<256>(t.part_v, t.part_i, t.part_s, n_part, s_v, s_i);
        const double pk = e.i < 0 ? 0.0 : e.v;
        const double pc = e.i < 0 ? -1.0 : (e.s == kNanColumn0 ? -2.0 : (double)(e.i + (t.p2l ? 0 : col_offset)));
        if ((int)threadIdx.x < x.lay.world) {
            const unsigned long long kb = (unsigned long long)__double_as_longlong(pk), cb = (unsigned long long)__double_as_longlong(pc);
            unsigned long long *dst = x.peers[threadIdx.x] + x.lay.pair_off(x.epoch & 1u, x.rank);
            st_sys(dst + 0, tg | (kb & 0xffffffffull));
            st_sys(dst + 1, tg | (kb >> 32));
            st_sys(dst + 2, tg | (cb & 0xffffffffull));
            st_sys(dst + 3, tg | (cb >> 32));
        }
    }
    // ---- exchange A, consumer (every workgroup for itself): every shard's pair of this pivot
    {
        int lost = 0;
        if (running && (int)threadIdx.x < n_shards) {
            const unsigned long long *src = x.mine + x.lay.pair_off(x.epoch & 1u, (int)threadIdx.x);
            unsigned long long g[4];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) { g[k] = ld_sys(src + k); ok &= (unsigned)(g[k] >> 32) == x.epoch; }
                if (ok) break;
                if (spins > x.max_spins) { lost = 1; break; }
            }
            s_g[2 * threadIdx.x]     = __longlong_as_double((long long)(((g[1] & 0xffffffffull) << 32) | (g[0] & 0xffffffffull)));
            s_g[2 * threadIdx.x + 1] = __longlong_as_double((long long)(((g[3] & 0xffffffffull) << 32) | (g[2] & 0xffffffffull)));
        }
        if (__syncthreads_or(lost)) {
            if (threadIdx.x == 0) ctl->status = kExchangeLost;
            if (gid == 0) *ec_out = -1;
            if (threadIdx.x == 0) { rp_v[blockIdx.x] = 0.0; rp_i[blockIdx.x] = -1; rp_s[blockIdx.x] = 0; }
            return;
        }
    }
    if (j == 0) {                                   // a new block starts (whatever the status)
        if (gid == 0) blk->n_pending = 0;
        const int64_t n = t.bk_stride > ldv ? t.bk_stride : ldv;
        for (int64_t idx = gid; idx < n; idx += gsz) {
            if (idx < t.bk_stride) t.bk_rmask[idx] = 0u;
            if (idx < ldv)         t.bk_smask[idx] = 0u;
        }
    }
    ValIdx win; win.v = 0.0; win.i = -1; win.s = 0;
    bool nan0 = false;                              // some shard holds a NaN in global column 0's objective entry
    for (int k = 0; k < n_shards; ++k) {
        ValIdx c; c.v = s_g[2 * k]; c.i = (int64_t)s_g[2 * k + 1]; c.s = 0;
        nan0 |= s_g[2 * k + 1] == -2.0;
        win = vi_min(win, c);
    }
    const int64_t ec = (running && !nan0 && !price_says_optimal(win, price_tol)) ? win.i : -1;
    const int64_t lc = ec < 0 ? -1 : (t.l2p ? t.l2p[ec] : ec - col_offset);
    const bool mine = ec >= 0 && lc >= 0 && lc < vcl;
    const bool live = ec >= 0 && !(c0.max_pivots > 0 && c0.n_pivots >= c0.max_pivots);
    // (k_shard_la_contribute: the RHS copy is carried from step to step, one link per step)
    const int64_t crp = j > 0 ? blk->cr[j - 1] : -1;
    const double  prp = j > 0 ? t.bk_prow[(int64_t)(j - 1) * t.ld + vcl] : 0.0;
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    for (int64_t r = gid; r < t.rows; r += gsz) {
        if (ec < 0) break;
        double a = mine ? t.M[r * t.ld + lc] : 0.0;
        double b = j == 0 ? t.M[r * t.ld + vcl] : t.rhs[r];
        if (j > 0) b = pend(b, false, r == crp, t.bk_col[(int64_t)(j - 1) * t.bk_stride + r], prp);
        t.rhs[r] = b;
        if (mine) {
            for (int i0 = 0; i0 < j; i0 += 4) {     // four links' operands requested together
                double ci[4], pi[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k < j ? i0 + k : j - 1;
                    ci[k] = t.bk_col[(int64_t)i * t.bk_stride + r];
                    pi[k] = t.bk_prow[(int64_t)i * t.ld + lc];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + k < j) a = pend(a, lc == blk->slot[i0 + k], r == blk->cr[i0 + k], ci[k], pi[k]);
            }
            // exchange B, producer: the owner writes the column's granules straight into the OTHER shards' buffers
            const unsigned long long vb = (unsigned long long)__double_as_longlong(a);
            const int64_t off = x.lay.col_off(x.epoch & 1u) + 2 * r;
            for (int q = 0; q < x.lay.world; ++q)
                if (q != x.rank) {
                    st_sys(x.peers[q] + off, tg | (vb & 0xffffffffull));
                    st_sys(x.peers[q] + off + 1, tg | (vb >> 32));
                }
        } else if (live) {
            // exchange B, consumer: my row's two granules of this pivot's column, in my own buffer
            const unsigned long long *src = x.mine + x.lay.col_off(x.epoch & 1u) + 2 * r;
            unsigned long long lo, hi;
            for (unsigned spins = 0;; ++spins) {
                lo = ld_sys(src);
                hi = ld_sys(src + 1);
                if ((unsigned)(lo >> 32) == x.epoch && (unsigned)(hi >> 32) == x.epoch) break;
                if (spins > x.max_spins) { ctl->status = kExchangeLost; lo = hi = 0ull; break; }
            }
            a = __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
        }
        if (live) {                                 // (k_shard_la_ratio)
            t.bk_col[(int64_t)j * t.bk_stride + r] = a;
            if (t.p2l && !(fabs(a) <= 1.7976931348623157e308)) atomicOr(&ctl->poison, 1);   // compact shard: MI_NONFINITE
            if (r < m && ratio_thr < a) {
                const double q = b / a;
                if (q != q) atomicOr(&ctl->poison, 1);
                else { ValIdx c; c.v = q; c.i = r; c.s = __double_as_longlong(a); best = vi_min(best, c); }
            }
        }
    }
    best = block_reduce_min<256>(best, s_v, s_i);
    if (threadIdx.x == 0) { rp_v[blockIdx.x] = best.v; rp_i[blockIdx.x] = best.i; rp_s[blockIdx.x] = best.s; }
    if (gid == 0) *ec_out = ec;
}

__global__ __launch_bounds__(kGatherThreads) void k_shard_la_ratio(TabView t, int j, const double *col_src,
                                                                   const int64_t *ec_dev, double ratio_thr, P2pArgs x)
{
    __shared__ double    s_v[kGatherThreads / 64];
    __shared__ long long s_i[kGatherThreads / 64];
    double  *rp_v = t.part_v + t.part_cap / 2;      // ratio partials: upper half of the buffers
    int64_t *rp_i = t.part_i + t.part_cap / 2;
    int64_t *rp_s = t.part_s + t.part_cap / 2;
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    const int64_t m = t.rows - 1;
    const int64_t r = (int64_t)blockIdx.x * kGatherThreads + threadIdx.x;
    const bool live = c0.status == kRunning && *ec_dev >= 0 &&
                      !(c0.max_pivots > 0 && c0.n_pivots >= c0.max_pivots);
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    if (live && r < t.rows) {
        double a;
        if (x.peers) {
            // exchange B, consumer: my row's two granules of this pivot's column, in my own buffer
            const unsigned long long *src = x.mine + x.lay.col_off(x.epoch & 1u) + 2 * r;
            unsigned long long lo, hi;
            for (unsigned spins = 0;; ++spins) {
                lo = ld_sys(src);
                hi = ld_sys(src + 1);
                if ((unsigned)(lo >> 32) == x.epoch && (unsigned)(hi >> 32) == x.epoch) break;
                if (spins > x.max_spins) { ctl->status = kExchangeLost; lo = hi = 0ull; break; }
            }
            a = __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
        } else {
            a = col_src[r];
        }
        const double b = r < m ? t.rhs[r] : 0.0;
        t.bk_col[(int64_t)j * t.bk_stride + r] = a;
        if (t.p2l && !(fabs(a) <= 1.7976931348623157e308)) atomicOr(&ctl->poison, 1);   // compact shard: MI_NONFINITE
        if (r < m && ratio_thr < a) {
            const double q = b / a;
            if (q != q) atomicOr(&ctl->poison, 1);   // a NaN quotient: MI_NONFINITE as well (the first-eligible-row rule is not reproduced here)
            else { best.v = q; best.i = r; best.s = __double_as_longlong(a); }
        }
    }
    best = block_reduce_min<kGatherThreads>(best, s_v, s_i);
    if (threadIdx.x == 0) { rp_v[blockIdx.x] = best.v; rp_i[blockIdx.x] = best.i; rp_s[blockIdx.x] = best.s; }
}

template <int J>
__global__ __launch_bounds__(kScaleThreads) void k_shard_la_scale(TabView t, int n_rp, const int64_t *ec_dev,
                                                                 double sgn)
{
    __shared__ double    s_v[kScaleThreads / 64];
    __shared__ long long s_i[kScaleThreads / 64];
    constexpr int JJ = J > 0 ? J : 1;
    const double  *rp_v = t.part_v + t.part_cap / 2;
    const int64_t *rp_i = t.part_i + t.part_cap / 2;
    const int64_t *rp_s = t.part_s + t.part_cap / 2;
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    BlockCtl *blk = t.blk;
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    const int64_t m = t.rows - 1, vcl = t.cols - 1, ldv = t.ld >> 1;
    const int64_t p = (int64_t)blockIdx.x * kScaleThreads + threadIdx.x;
    const bool in = p < ldv;
    const double2 *M2 = reinterpret_cast<const double2 *>(t.M);
    double2 *P2 = reinterpret_cast<double2 *>(t.bk_prow);
    // independent of the pivot row: the objective row, the pending pivots' rows / slots / prow
    // entries / objective-row col entries, this step's objective-row col entry, the column map
    double2 z = in ? M2[m * ldv + p] : make_double2(0.0, 0.0);
    double2 pi[JJ];
    const int lane = threadIdx.x & 63;
    const bool  lj = lane < J;
    const double  v_cm = lj ? t.bk_col[(int64_t)lane * t.bk_stride + m] : 0.0;
    const int64_t v_cr = lj ? blk->cr[lane] : -1;
    const int64_t v_sl = lj ? blk->slot[lane] : -1;
#pragma unroll
    for (int i = 0; i < J; ++i)
        pi[i] = in ? P2[(int64_t)i * ldv + p] : make_double2(0.0, 0.0);
    const int64_t global_ec = *ec_dev;
    const int64_t c0i = 2 * p;
    const int64_t l0 = (in && c0i < vcl) ? (t.p2l ? t.p2l[c0i] : c0i) : -1;
    const int64_t l1 = (in && c0i + 1 < vcl) ? (t.p2l ? t.p2l[c0i + 1] : c0i + 1) : -1;
    const int64_t slot = (t.p2l && global_ec >= 0) ? t.l2p[global_ec] : -1;   // compact owner: the slot the leaving column takes
    const ValIdx q = block_price_partials<kScaleThreads>(rp_v, rp_i, rp_s, n_rp, s_v, s_i);
    if (c0.status != kRunning) return;
    if (global_ec < 0) {
        if (leader) ctl->status = 0;                // MI_OPTIMAL
        return;
    }
    if (c0.max_pivots > 0 && c0.n_pivots >= c0.max_pivots) {
        if (leader) ctl->status = 3;                // MI_MAX_PIVOTS
        return;
    }
    if (c0.poison) {
        if (leader) ctl->status = 6;                // MI_NONFINITE (compact shard)
        return;
    }
    if (q.i < 0) {
        if (leader) ctl->status = 1;                // MI_UNBOUNDED
        return;
    }
    const int64_t cr = uniform64(q.i);
    const double piv = __longlong_as_double(q.s);   // == col_J[cr], carried by the winner
    const double cmj = t.bk_col[(int64_t)J * t.bk_stride + m];
    ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
    const double v_ccr = lj ? t.bk_col[(int64_t)lane * t.bk_stride + cr] : 0.0;
    double2 y = in ? M2[cr * ldv + p] : make_double2(0.0, 0.0);   // row cr
#pragma unroll
    for (int i = 0; i < J; ++i) {                   // (all lanes: v_readlane ignores exec)
        const bool    is_cr = cr == lane_value(v_cr, i);
        const int64_t sl = lane_value(v_sl, i);
        const double  ccr = lane_value(v_ccr, i), cm = lane_value(v_cm, i);
        y.x = pend(y.x, 2 * p     == sl, is_cr, ccr, pi[i].x);
        y.y = pend(y.y, 2 * p + 1 == sl, is_cr, ccr, pi[i].y);
        z.x = pend(z.x, 2 * p     == sl, false, cm, pi[i].x);
        z.y = pend(z.y, 2 * p + 1 == sl, false, cm, pi[i].y);
    }
    if (in) {
        const bool own = slot >= 0 && (p == (slot >> 1));
        const int64_t leaving = own ? t.basis[cr] : -1;
        const double2 pr = scale_pair(t, p, y, piv, slot);
        P2[(int64_t)J * ldv + p] = pr;
        z.x = pend(z.x, 2 * p     == slot, false, cmj, pr.x);
        z.y = pend(z.y, 2 * p + 1 == slot, false, cmj, pr.y);
        if (c0i < vcl) {
            ValIdx c = price_cand(z.x * sgn, (c0i == slot) ? leaving : l0, c0i, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        if (c0i + 1 < vcl) {
            ValIdx c = price_cand(z.y * sgn, (c0i + 1 == slot) ? leaving : l1, c0i + 1, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        // bookkeeping: the owner of the slot where there is one (it alone reads basis[cr] before
        // it changes), the leader otherwise -- one writer either way
        if (slot >= 0 ? own : leader) {
            if (own) {
                swap_columns(t, global_ec, cr, slot);
                t.bk_smask[slot >> 1] |= 1u << (J + 16 * (int)(slot & 1));
            }
            record_pivot(t, c0, global_ec, cr);      // basis holds GLOBAL column indices
            blk->cr[J] = cr;
            blk->slot[J] = slot;
            blk->n_pending = J + 1;
            t.bk_rmask[cr] |= 1u << J;
        }
    }
    best = wave_reduce_min(best);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (kScaleThreads / 64) + (threadIdx.x >> 6);
        t.part_v[w] = best.v;
        t.part_i[w] = best.i;
        t.part_s[w] = best.s;
    }
}

// ---- the look-ahead of a whole block as ONE launch ------------------------------------------
// Two launches per look-ahead step are two kernel boundaries (2.5 us each) plus cold caches at
// every start.  For tableaux whose rows and column pairs fit a few workgroups (config 3: 17 x 256
// threads cover 4097 rows and 4104 pairs) the steps of a block run inside one launch instead:
// thread g owns row g (entering-column / RHS side) AND column pair g (pivot-row / objective-row
// side); its col_i[r], prow_i[pair], RHS entry and objective-row pair stay in registers / LDS from
// step to step, and the two reductions of a step go through a message exchange between the
// workgroups.  All workgroups reduce the same records with the same comparisons, so they take
// every decision (entering column, pivot row, termination) identically without further
// communication.
//
// The hand-off protocol.  gfx950: a CU's vector L1 is never refreshed by another CU's stores, the
// eight XCDs' L2s are not coherent with each other, and a workgroup barrier does NOT wait for the
// other waves' outstanding stores.  Hence:
//   * a record is eight self-validating 8-byte granules {tag = epoch, 32 bits of payload}, each
//     written by ONE store and polled with L1-bypassing (sc1) loads until all tags match: no
//     fence, no release/acquire pair, no read-modify-write;
//   * what a step has just produced and the next half-step needs at once travels IN the records
//     (the pivot-row entry of the winning candidate, the RHS entry of the new pivot row, the
//     objective-row entry of the new column);
//   * everything else another workgroup reads inside the launch (col_i, prow_i of the older
//     pending pivots, basis) is stored WRITE-THROUGH (sc1: relaxed agent-scope atomic stores) and
//     read with sc1 loads, and every wave drains its stores (s_waitcnt vmcnt(0)) half a step
//     after issuing them -- before the barrier of the NEXT exchange, which is the first one whose
//     records anybody takes as evidence that those stores have landed.  (Round 1 published behind
//     a plain __syncthreads(), i.e. possibly before the other waves' col / prow stores had left
//     the CU: a reader on another XCD could chain a stale value -- the one-in-thousands
//     pivot-trace mismatch recorded in round 1's DESIGN.md.)
//   * words that only LATER launches read but that several workgroups write in turn (column
//     maps, basis) go write-through as well (two L2s holding different dirty versions of one word
//     are written back in an unspecified order); the control block, the pending list and the
//     trace have ONE writer, the leader thread; a mask word is written by its owner only.
//   * one-XCD mode (a pure speed option): the launch is 8x as wide and only every eighth block
//     takes part, which is where the dispatcher puts one XCD's blocks.  Nothing RELIES on that:
//     the first exchange of a launch (write-through, valid anywhere) carries every workgroup's
//     HW_REG_XCC_ID, and only if they all agree do the later stores become plain stores that stay
//     in the one shared L2 (where the sc1 loads find them a fabric round trip sooner).
struct LaMsg { ValIdx c; unsigned flag, same; double u, w; };

__device__ __forceinline__ double lane_value_dyn(double v, int lane)    // lane: uniform, run-time
{
    return lane_value(v, __builtin_amdgcn_readfirstlane(lane));
}
__device__ __forceinline__ int64_t lane_value_dyn(int64_t v, int lane)
{
    return lane_value(v, __builtin_amdgcn_readfirstlane(lane));
}

template <class T> __device__ __forceinline__ void st_wt(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one untorn store; local == true: all readers share this XCD's L2, the line may stay there
template <class T> __device__ __forceinline__ void st_x(T *p, T v, bool local)
{
    if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else       __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> __device__ __forceinline__ T ld_l2(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// this wave's outstanding loads and stores have completed -- inline asm: the compiler neither
// moves nor drops it
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

constexpr int kLaThreads = 256;
constexpr unsigned kEmptyIdx = 0x7fffffffu;

// a 16-byte aligned pair of granules as ONE 16-byte access (every granule is validated on its own,
// and a naturally aligned 16-byte access never tears an aligned 8-byte half)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long pair_lo(const v4u &v) { return ((unsigned long long)v.y << 32) | v.x; }
__device__ __forceinline__ unsigned long long pair_hi(const v4u &v) { return ((unsigned long long)v.w << 32) | v.z; }
// one or two whole records (64 bytes each) per lane: four 16-byte loads per record, all in flight,
// L1 bypassed (sc1: what ld_l2 compiles to), the wait inside the statement (the compiler does not
// count these loads).  Half the requests of eight 8-byte loads per record -- and a wave's poll is
// bound by its requests: the lanes are a record (a 64-byte line) apart.
__device__ __forceinline__ void load_record(unsigned long long (&g)[8], const ExchRec *r)
{
    v4u q0, q1, q2, q3;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(r) : "memory");
    g[0] = pair_lo(q0); g[1] = pair_hi(q0); g[2] = pair_lo(q1); g[3] = pair_hi(q1);
    g[4] = pair_lo(q2); g[5] = pair_hi(q2); g[6] = pair_lo(q3); g[7] = pair_hi(q3);
}
__device__ __forceinline__ void load_records2(unsigned long long (&g)[8], unsigned long long (&h)[8],
                                              const ExchRec *r, const ExchRec *r2)
{
    v4u q0, q1, q2, q3, p0, p1, p2, p3;
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %9, off sc1\n\t"
                 "global_load_dwordx4 %5, %9, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %6, %9, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %7, %9, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
                 : "v"(r), "v"(r2) : "memory");
    g[0] = pair_lo(q0); g[1] = pair_hi(q0); g[2] = pair_lo(q1); g[3] = pair_hi(q1);
    g[4] = pair_lo(q2); g[5] = pair_hi(q2); g[6] = pair_lo(q3); g[7] = pair_hi(q3);
    h[0] = pair_lo(p0); h[1] = pair_hi(p0); h[2] = pair_lo(p1); h[3] = pair_hi(p1);
    h[4] = pair_lo(p2); h[5] = pair_hi(p2); h[6] = pair_lo(p3); h[7] = pair_hi(p3);
}

__device__ __forceinline__ unsigned long long dbits(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double join_bits(unsigned long long lo, unsigned long long hi)
{
    return __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
}

// The reductions of the look-ahead move (value, 32-bit index) pairs only -- half the DPP /
// ds_bpermute traffic of a ValIdx -- and fetch the winner's payload from the lane that holds it
// afterwards (indices are unique, so that lane is).  Same decision rule, same tree as
// vi_min / wave_reduce_min.
struct Cand { double v; int i; };                                // i < 0: empty
__device__ __forceinline__ Cand cand_min(Cand a, Cand b)
{
    const bool a_empty = a.i < 0, b_empty = b.i < 0;
    const bool better  = (b.v < a.v) | ((b.v == a.v) & (b.i < a.i));
    const bool take_b  = a_empty | (!b_empty & better);
    Cand r;
    r.v = take_b ? b.v : a.v;
    r.i = take_b ? b.i : a.i;
    return r;
}
template <int CTRL> __device__ __forceinline__ Cand dpp_cand(Cand x)
{
    Cand y;
    y.v = __longlong_as_double(dpp64<CTRL>(__double_as_longlong(x.v)));
    y.i = __builtin_amdgcn_update_dpp(x.i, x.i, CTRL, 0xf, 0xf, false);
    return y;
}
__device__ __forceinline__ Cand shfl_down_cand(Cand x, int off)
{
    Cand y;
    y.v = __shfl_down(x.v, off, 64);
    y.i = __shfl_down(x.i, off, 64);
    return y;
}
// winner of the wave in EVERY lane, plus the lane that holds it (-1: all empty)
__device__ __forceinline__ Cand wave_reduce_cand(Cand x, int &src)
{
    const int mine = x.i;
    x = cand_min(x, shfl_down_cand(x, 32));
    x = cand_min(x, shfl_down_cand(x, 16));
    x = cand_min(x, dpp_cand<0x108>(x));
    x = cand_min(x, dpp_cand<0x104>(x));
    x = cand_min(x, dpp_cand<0x102>(x));
    x = cand_min(x, dpp_cand<0x101>(x));
    x.v = lane_value(x.v, 0);
    x.i = __builtin_amdgcn_readfirstlane(x.i);
    const unsigned long long m = __ballot((mine == x.i) & (x.i >= 0));
    src = m ? (int)__ffsll((long long)m) - 1 : -1;
    return x;
}
__device__ __forceinline__ int64_t lane_pick(int64_t v, int src) { return lane_value_dyn(v, src < 0 ? 0 : src); }
__device__ __forceinline__ double  lane_pick(double v, int src)  { return lane_value_dyn(v, src < 0 ? 0 : src); }
__device__ __forceinline__ int     lane_pick(int v, int src)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src < 0 ? 0 : src));
}

// The same arg-min in ~30 instead of ~140 instructions for the common case -- no NaN among the
// wave's candidates and a unique minimum: the minimum VALUE by a butterfly of v_min_f64 (gfx950's
// v_permlane32_swap / v_permlane16_swap across the rows of 16 lanes, DPP row rotations inside
// them; every lane ends up with it), then the lane that holds it by a ballot.  Without NaNs the
// lexicographic (value, index) minimum is unique and independent of the reduction order, so this
// IS the tree's winner; with a NaN candidate (vi_min is then order dependent) or an exact tie
// (lowest index decides) the tree itself runs.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double min_f64(double a, double b)     // operands are never NaN here
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double wave_allmin_f64(double x)
{
    {   // lanes l and l ^ 32: whichever half a swap puts where, {r.x, r.y} is the pair in every lane
        const long long b = __double_as_longlong(x);
        const unsigned lo = (unsigned)b, hi = (unsigned)((unsigned long long)b >> 32);
        const v2u l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const v2u h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        x = min_f64(__longlong_as_double((long long)(((unsigned long long)h2.x << 32) | l2.x)),
                    __longlong_as_double((long long)(((unsigned long long)h2.y << 32) | l2.y)));
    }
    {   // rows 0 <-> 1, 2 <-> 3
        const long long b = __double_as_longlong(x);
        const unsigned lo = (unsigned)b, hi = (unsigned)((unsigned long long)b >> 32);
        const v2u l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const v2u h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        x = min_f64(__longlong_as_double((long long)(((unsigned long long)h2.x << 32) | l2.x)),
                    __longlong_as_double((long long)(((unsigned long long)h2.y << 32) | l2.y)));
    }
    x = min_f64(x, __longlong_as_double(dpp64<0x128>(__double_as_longlong(x))));   // row_ror:8
    x = min_f64(x, __longlong_as_double(dpp64<0x124>(__double_as_longlong(x))));   // row_ror:4
    x = min_f64(x, __longlong_as_double(dpp64<0x122>(__double_as_longlong(x))));   // row_ror:2
    x = min_f64(x, __longlong_as_double(dpp64<0x121>(__double_as_longlong(x))));   // row_ror:1
    return x;
}
__device__ __forceinline__ Cand wave_argmin(Cand x, int &src)
{
    const bool valid = x.i >= 0;
    if (__any(valid & (x.v != x.v))) return wave_reduce_cand(x, src);
    const double key = valid ? x.v : __builtin_huge_val();
    const double vmin = wave_allmin_f64(key);
    const unsigned long long mask = __ballot(valid & (key == vmin));
    if (__popcll(mask) > 1) return wave_reduce_cand(x, src);
    src = mask ? (int)__ffsll((long long)mask) - 1 : -1;
    Cand r;
    r.v = lane_pick(x.v, src);
    r.i = mask ? lane_pick(x.i, src) : -1;
    if (!mask) r.v = 0.0;
    return r;
}

// Exchange of one reduction between the workgroups.  `mine` is this thread's candidate.  Every
// WAVE reduces its 64 candidates (tree of wave_reduce_min) and publishes its winner at once as
// record 4 w + wave -- together with two doubles it picks out of its lanes' registers
// (extra(winner, lane that holds it, u, x2)) -- so nothing waits for a workgroup barrier on the
// way out; the first wave of every workgroup collects all records (lane l: records l and l + 64),
// reduces them (tree again) and hands the winner to the other waves through LDS, with the first
// double of the winner's own record and the second double (PRICE) / first double (RATIO) of
// record rec_from.
// PRICE: granules v v i s u u w w   (s = slot: 32 bits)      RATIO: v v i|flag s s u u -
// false: a record did not arrive within max_spins polls.
constexpr int kLaWaves = kLaThreads / 64;
// true: every wave collects the records itself (no LDS hop, no workgroup barrier left in the
// kernel; four times the poll traffic on the one L2) -- measured 149 us per block of 16 at config 3
// against 122 us for false: the first wave of a workgroup collects and broadcasts through LDS
constexpr bool kLaEveryWavePolls = false;

template <bool PRICE>
__device__ __forceinline__ void decode_rec(const unsigned long long (&g)[8], bool valid, Cand &x, int64_t &xs,
                                           unsigned &fl, double &ru, double &rw)
{
    const unsigned iw = (unsigned)g[2];
    x.v = 0.0; x.i = -1; xs = 0; fl = 0u;
    if (valid) {
        x.v = join_bits(g[0], g[1]);
        x.i = (iw & kEmptyIdx) == kEmptyIdx ? -1 : (int)(iw & kEmptyIdx);
        xs = PRICE ? (int64_t)(g[3] & 0xffffffffull) : (int64_t)(((g[4] & 0xffffffffull) << 32) | (g[3] & 0xffffffffull));
        fl = iw >> 31;
    }
    ru = PRICE ? join_bits(g[4], g[5]) : join_bits(g[5], g[6]);
    rw = PRICE ? join_bits(g[6], g[7]) : 0.0;
}

template <bool PRICE, class Extra>
__device__ __forceinline__ bool la_exchange(ValIdx mine, unsigned myflag, ExchRec *recs, int nw, int w,
                                            unsigned tag, unsigned max_spins, bool mute, bool local,
                                            int rec_from, LaMsg *s_res, LaMsg &out, Extra extra,
                                            unsigned long long *ts, double *dbg = nullptr,
                                            bool give_up = false)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrec = nw * kLaWaves;
    {   // ---- this wave's winner -> its record
        Cand c; c.v = mine.v; c.i = (int)mine.i;
        int src;
        c = wave_argmin(c, src);
        const int64_t cs = lane_pick(mine.s, src);
        const unsigned wf = __any(myflag != 0u) ? 1u : 0u;
        double u = 0.0, x2 = 0.0;
        ValIdx win; win.v = c.v; win.i = c.i; win.s = cs;
        extra(win, src, u, x2);
        if (!mute) {
            const unsigned long long vb = dbits(c.v), sb = (unsigned long long)cs, ub = dbits(u), wb = dbits(x2);
            const unsigned iw = (c.i < 0 ? kEmptyIdx : (unsigned)c.i) | (wf ? 0x80000000u : 0u);
            const unsigned word[8] = { (unsigned)vb, (unsigned)(vb >> 32), iw, (unsigned)sb,
                                       PRICE ? (unsigned)ub : (unsigned)(sb >> 32),
                                       PRICE ? (unsigned)(ub >> 32) : (unsigned)ub,
                                       PRICE ? (unsigned)wb : (unsigned)(ub >> 32),
                                       PRICE ? (unsigned)(wb >> 32) : 0u };
            unsigned val = word[0];                              // lane k stores granule k
#pragma unroll
            for (int k = 1; k < 8; ++k) val = lane == k ? word[k] : val;
            if (lane < 8) st_x(&recs[w * kLaWaves + wave].g[lane], ((unsigned long long)tag << 32) | val, local);
        }
#ifdef MI355X_LA_TIMING
        if (dbg && lane == 0) dbg[w * kLaWaves + wave] = (double)wall_clock64();   // when this wave published
#endif
    }
#ifdef MI355X_LA_TIMING
    if (ts) ts[0] = wall_clock64();
#endif
    if (give_up) { out.flag = 2u; return false; }               // test hook (workgroup-uniform): published, then "timed out"
    if (kLaEveryWavePolls || tid < 64) {
        // ---- collect: lane l <- records l and l + 64
        const bool v0 = lane < nrec, v1 = lane + 64 < nrec;
        const ExchRec *r0 = recs + (v0 ? lane : 0), *r1 = recs + (v1 ? lane + 64 : 0);
        unsigned long long g0[8], g1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g1[k] = 0ull;
        unsigned spins = 0;
        bool fine = true;
        for (;;) {
            if (nrec > 64) load_records2(g0, g1, r0, r1);
            else           load_record(g0, r0);
            bool ok0 = true, ok1 = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) { ok0 &= (unsigned)(g0[k] >> 32) == tag; ok1 &= (unsigned)(g1[k] >> 32) == tag; }
            if (__all((ok0 | !v0) & (ok1 | !v1))) break;
            if (++spins > max_spins) { fine = false; break; }
        }
#ifdef MI355X_LA_TIMING
        if (ts) { ts[1] = wall_clock64(); ts[3] = spins; }
#endif
        Cand x0, x1;
        int64_t s0, s1;
        unsigned f0, f1;
        double u0, u1, w0, w1;
        decode_rec<PRICE>(g0, v0, x0, s0, f0, u0, w0);
        decode_rec<PRICE>(g1, v1, x1, s1, f1, u1, w1);
        // lane-local fold of the two records (= the 64-apart level of a 128-wide tree)
        const Cand x01 = cand_min(x0, x1);
        const bool hi = x01.i != x0.i;                            // record l + 64 won (indices are unique)
        int src;
        const Cand x = wave_argmin(x01, src);                     // uniform
        const unsigned fl = __any((f0 | f1) != 0u) ? 1u : 0u;
        const int64_t bs = lane_pick(hi ? s1 : s0, src);
        const double bu = lane_pick(hi ? u1 : u0, src);           // the winner's own record
        const double fsel = PRICE ? (rec_from < 64 ? w0 : w1) : (rec_from < 64 ? u0 : u1);
        const double bw = lane_value_dyn(fsel, rec_from & 63);
        // do all records carry the same first double?  (the XCC ids of the first exchange)
        const unsigned long long ref = dbits(lane_value(u0, 0));
        const unsigned same = __all(((dbits(u0) == ref) || !v0) && ((dbits(u1) == ref) || !v1)) ? 1u : 0u;
        if (kLaEveryWavePolls) {                                 // every wave has the result in registers
            out.c.v = x.v; out.c.i = x.i; out.c.s = bs;
            out.flag = fine ? fl : 2u; out.same = same; out.u = bu; out.w = bw;
        } else if (lane == 0) {
            s_res->c.v = x.v; s_res->c.i = x.i; s_res->c.s = bs;
            s_res->flag = fine ? fl : 2u; s_res->same = same; s_res->u = bu; s_res->w = bw;
        }
    }
    if (!kLaEveryWavePolls) {
        __syncthreads();
        out = *s_res;
    }
#ifdef MI355X_LA_TIMING
    if (ts) ts[2] = wall_clock64();
#endif
    return out.flag != 2u;
}

// The steps are a run-time loop (fully unrolled the kernel was 300 KB of straight-line code and
// ran at the speed of instruction-cache misses): per-thread col_i[row] / prow_i[pair] of the
// pending pivots live in LDS ([pivot][thread]: conflict-free), everything else in registers.
//
// With ONE wave per SIMD every instruction of the critical wave costs its full issue + latency,
// so the chains through the pending pivots are written for instruction count: the product of a
// link does not depend on the chained value (J independent multiplications, then J dependent
// subtractions), the two rare exceptions of a link -- the element lies on pending pivot i's row /
// in the slot it gave up -- are bit tests on masks the thread keeps anyway (my_rm, my_sm) plus a
// wave-uniform mask over the pending pivots, and a wave none of whose lanes is an exception runs
// the bare chain.  Links i >= J of a group of four are exact identities (operands 0.0:
// x - (+0.0) == x bit for bit).
// one_xcd: see above.  fault > 0 (test hook): the last workgroup stops publishing from step
// `fault - 1` on, as a workgroup that is not resident would.
template <int KMAX>
__global__ __launch_bounds__(kLaThreads) void k_la_block(TabView t, int ksteps, double sgn,
                                                        double price_tol, double ratio_thr,
                                                        unsigned epoch_base, unsigned max_spins,
                                                        int one_xcd, int fault)
{
    static_assert(KMAX % 4 == 0 && KMAX <= 16, "groups of four links; 16 + 16 bits of my_sm");
    __shared__ double    s_ci[KMAX][kLaThreads];                 // 32 KB
    __shared__ double2   s_pi[KMAX][kLaThreads];                 // 64 KB
    __shared__ LaMsg     s_res;
    int nw = gridDim.x, w = blockIdx.x;
    if (one_xcd) {                                               // only every eighth block takes part
        if (w & 7) return;
        w >>= 3; nw >>= 3;
    }
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    BlockCtl *blk = t.blk;
    const int tid = threadIdx.x;
    const bool leader = w == 0 && tid == 0;
    const int lane = tid & 63;
    const int64_t m = t.rows - 1, vc = t.cols - 1, ld = t.ld, ldv = ld >> 1;
    const int64_t g = (int64_t)w * kLaThreads + tid;
    const bool has_row = g < t.rows, has_pair = g < ldv;
    const int64_t r = g, p = g;
    const double2 *M2 = reinterpret_cast<const double2 *>(t.M);
    // thread that owns the RHS pair / the objective row: workgroup, wave, lane, record
    const int g_vc = (int)(vc >> 1), g_m = (int)m;
    const int rec_vc = g_vc / 64, rec_m = g_m / 64;              // (= 4 * workgroup + wave)
    const bool wave_has_vc = (int)(g / 64) == rec_vc, wave_has_m = (int)(g / 64) == rec_m;

    // a new block starts (whatever the status): pending list, stamp (the sweep applies the list
    // only under this launch's stamp -- had the leader's workgroup never run, the list would be
    // the previous block's) and this thread's OWN mask words, which only it ever writes
    // (not behind a launch that lost an exchange: the host's recovery reads that launch's list)
    if (leader && c0.status != kSyncLost) { blk->n_pending = 0; blk->stamp = epoch_base; }
    unsigned my_rm = 0u, my_sm = 0u;     // bit i: my row is pivot row i / bits i, 16+i: my pair's columns are slot i
    if (g < t.bk_stride) t.bk_rmask[g] = 0u;
    if (g < ldv)         t.bk_smask[g] = 0u;
    // every way out of the launch records how many steps this workgroup COMPLETED (its col_i /
    // prow_i entries stored): the sweep applies no pivot that some workgroup did not finish
    auto leave = [&](int steps) {
        if (tid == 0) st_wt(&blk->done[w], (int64_t)(((unsigned long long)epoch_base << 8) | (unsigned)steps));
    };
    if (c0.status != kRunning) return;       // (nothing pending; behind a lost exchange `done` belongs to that launch)

    double  b = (has_row && r < m) ? t.M[r * ld + vc] : 0.0;     // RHS entry of my row
    double2 z = has_pair ? M2[m * ldv + p] : make_double2(0.0, 0.0);   // objective row, my pair
    int64_t l0 = (has_pair && 2 * p < vc) ? t.p2l[2 * p] : -1;   // logical columns of my pair
    int64_t l1 = (has_pair && 2 * p + 1 < vc) ? t.p2l[2 * p + 1] : -1;
    int64_t v_cr = -1, v_sl = -1;                                // lane i: pivot row / slot of pivot i
    unsigned wave_rm = 0u, wave_sm = 0u;                         // bit i: SOME lane of my wave is on pivot row i / holds slot i
    double2 pr = make_double2(0.0, 0.0);                         // my pair of the last normalised pivot row
    bool local = false;                                          // all workgroups on one XCD (verified)
    const double my_xcc = (double)xcc_id();
    unsigned long long *ts_p = nullptr, *ts_r = nullptr;
#ifdef MI355X_LA_TIMING
    unsigned long long tsp[4] = {0, 0, 0, 0}, tsr[4] = {0, 0, 0, 0};
    if (leader) { ts_p = tsp; ts_r = tsr; }
#endif

#ifndef MI355X_TEST_HOOKS
    fault = 0;                                                   // fault injection exists in the test build only
#endif
#pragma unroll 1
    for (int J = 0; J < ksteps; ++J) {
        const unsigned e_price = epoch_base + 2 * J + 1, e_ratio = epoch_base + 2 * J + 2;
        const bool mute = fault > 0 && J >= fault - 1 && w == nw - 1;
        // fault < 0 (test hook): the last workgroup publishes its ratio record of step -fault - 1 and
        // then gives up as if its polls had run out -- while every other workgroup sees all records,
        // the leader commits that pivot, and nobody gets past the next exchange
        const bool quit = fault < 0 && J == -fault - 1 && w == nw - 1 && nw > 1;
        double *dbg_p = nullptr, *dbg_r = nullptr;
#ifdef MI355X_LA_TIMING
        unsigned long long T0 = wall_clock64(), T2, T3, T5, T6;
        dbg_p = t.rhs + 512 + (2 * J) * 72;                      // publish times of the last block, per record
        dbg_r = t.rhs + 512 + (2 * J + 1) * 72;
#endif
        // ---- pricing: my pair's candidates -> wave winner -> record -> everybody's winner.
        // The record also carries the winner's entry of prow_{J-1} and (from its owner) the RHS
        // entry of prow_{J-1}: what the chain below needs of the row that was stored last.
        ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
        if (has_pair && 2 * p < vc)     { ValIdx c = price_cand(z.x * sgn, l0, 2 * p);     best = vi_min(best, c); }
        if (has_pair && 2 * p + 1 < vc) { ValIdx c = price_cand(z.y * sgn, l1, 2 * p + 1); best = vi_min(best, c); }
        LaMsg e;
        if (!la_exchange<true>(best, 0u, t.la_px, nw, w, e_price, max_spins, mute, local, rec_vc, &s_res, e,
                [&](const ValIdx &c, int src, double &u, double &x2) {
                    if (J == 0) { u = my_xcc; return; }
                    u = lane_pick((c.s & 1) ? pr.y : pr.x, src);            // prow_{J-1}[winner's slot]
                    if (wave_has_vc) x2 = lane_value_dyn((vc & 1) ? pr.y : pr.x, g_vc & 63);
                }, ts_p, dbg_p)) {
            if (tid == 0) st_wt(&ctl->status, kSyncLost);        // same word, same value from everyone
            leave(J);
            return;
        }
        if (J == 0) local = one_xcd != 0 && e.same != 0u;        // same records, same decision everywhere
#ifdef MI355X_LA_TIMING
        T2 = wall_clock64();
#endif
        if (price_says_optimal(e.c, price_tol)) {
            if (leader) ctl->status = 0;                         // MI_OPTIMAL
            leave(J);
            return;
        }
        if (c0.max_pivots > 0 && c0.n_pivots + J >= c0.max_pivots) {
            if (leader) ctl->status = 3;                         // MI_MAX_PIVOTS
            leave(J);
            return;
        }
        const int64_t ec = e.c.i, slot = uniform64(e.c.s);
        // ---- entering column through the pending chain; RHS entry brought up to date
        double a = has_row ? t.M[r * ld + slot] : 0.0;
        double v_pa = (lane < J - 1) ? ld_l2(&t.bk_prow[(int64_t)lane * ld + slot]) : 0.0;
        drain_vmem();                  // the loads -- and what this wave stored in the previous half-step
        if (lane == J - 1) v_pa = e.u;
        if (J > 0) b = pend(b, false, (my_rm >> (J - 1)) & 1u, s_ci[J - 1][tid], e.w);
        {
            // pending pivots whose given-up slot is the entering column's slot (uniform); lanes on a pending pivot row
            const unsigned slmask = (unsigned)__ballot((lane < J) & (v_sl == slot));
            const unsigned gen = slmask | wave_rm;                         // links that need the general form
#pragma unroll
            for (int i0 = 0; i0 < KMAX; i0 += 4) {
                if (i0 < J) {
                    double prod[4], pa[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double ci = (i0 + k < J) ? s_ci[i0 + k][tid] : 0.0;
                        pa[k] = lane_value(v_pa, i0 + k);
                        prod[k] = ci * pa[k];                          // rounded product
                    }
                    if (((gen >> i0) & 0xfu) == 0u) {                  // (uniform) the bare chain: ONE branch per four links
#pragma unroll
                        for (int k = 0; k < 4; ++k) a = a - prod[k];   // rounded differences
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if ((gen >> (i0 + k)) & 1u) {              // (uniform; rare)
                                const bool is_cr = (my_rm >> (i0 + k)) & 1u;
                                if ((slmask >> (i0 + k)) & 1u) a = is_cr ? 1.0 : 0.0;
                                const double d = a - prod[k];
                                a = is_cr ? pa[k] : d;
                            } else {
                                a = a - prod[k];                       // rounded difference
                            }
                        }
                    }
                }
            }
        }
        s_ci[J][tid] = a;
        ValIdx q; q.v = 0.0; q.i = -1; q.s = 0;
        unsigned bad = 0u;
        if (has_row) {
            st_x(&t.bk_col[(int64_t)J * t.bk_stride + r], a, local);
            bad = !(fabs(a) <= 1.7976931348623157e308);
            if (r < m && ratio_thr < a) {
                const double qv = b / a;
                if (qv != qv) bad = 1u;                        // a NaN quotient: decided on the dense path (kNeedDense)
                else { q.v = qv; q.i = r; q.s = __double_as_longlong(a); }
            }
        }
#ifdef MI355X_LA_TIMING
        T3 = wall_clock64();
#endif
        // ---- ratio test; the record of the objective row's wave carries col_J[m]
        LaMsg qq;
        if (!la_exchange<false>(q, bad, t.la_rx, nw, w, e_ratio, max_spins, mute, local, rec_m, &s_res, qq,
                [&](const ValIdx &, int, double &u, double &) { if (wave_has_m) u = lane_value_dyn(a, g_m & 63); },
                ts_r, dbg_r, quit)) {
            if (tid == 0) st_wt(&ctl->status, kSyncLost);
            leave(J);
            return;
        }
#ifdef MI355X_LA_TIMING
        T5 = wall_clock64();
#endif
        if (qq.flag != 0u) {                                     // inf / NaN in the column: see kNeedDense
            if (leader) ctl->status = kNeedDense;
            leave(J);
            return;
        }
        if (qq.c.i < 0) {
            if (leader) ctl->status = 1;                         // MI_UNBOUNDED
            leave(J);
            return;
        }
        const int64_t cr = uniform64(qq.c.i);
        const double piv = __longlong_as_double(qq.c.s);
        const double cmj = qq.w;                                 // col_J[m]
        // ---- pivot row through the chain -> prow_J; objective row through pivot J
        double2 y = has_pair ? M2[cr * ldv + p] : make_double2(0.0, 0.0);
        const double v_ccr = (lane < J) ? ld_l2(&t.bk_col[(int64_t)lane * t.bk_stride + cr]) : 0.0;
        const bool own = has_pair && (p == (slot >> 1));
        const int64_t leaving = own ? ld_l2(&t.basis[cr]) : -1;
        drain_vmem();                  // the loads -- and the col_J entry stored above
        {
            // pending pivots whose pivot row is the new pivot row (uniform); lanes holding a given-up slot
            const unsigned crmask = (unsigned)__ballot((lane < J) & (v_cr == cr));
            const unsigned gen = crmask | wave_sm;                         // links that need the general form
#pragma unroll
            for (int i0 = 0; i0 < KMAX; i0 += 4) {
                if (i0 < J) {
                    double2 pii[4], prod[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        pii[k] = (i0 + k < J) ? s_pi[i0 + k][tid] : make_double2(0.0, 0.0);
                        const double ccr = lane_value(v_ccr, i0 + k);
                        prod[k].x = ccr * pii[k].x;                    // rounded products
                        prod[k].y = ccr * pii[k].y;
                    }
                    if (((gen >> i0) & 0xfu) == 0u) {                  // (uniform) the bare chain: ONE branch per four links
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            y.x = y.x - prod[k].x;
                            y.y = y.y - prod[k].y;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if ((gen >> (i0 + k)) & 1u) {              // (uniform; rare)
                                const bool is_cr = (crmask >> (i0 + k)) & 1u;
                                if ((my_sm >> (i0 + k)) & 1u)      y.x = is_cr ? 1.0 : 0.0;
                                if ((my_sm >> (16 + i0 + k)) & 1u) y.y = is_cr ? 1.0 : 0.0;
                                const double dx = y.x - prod[k].x, dy = y.y - prod[k].y;
                                y.x = is_cr ? pii[k].x : dx;
                                y.y = is_cr ? pii[k].y : dy;
                            } else {
                                y.x = y.x - prod[k].x;
                                y.y = y.y - prod[k].y;
                            }
                        }
                    }
                }
            }
        }
        pr = make_double2(0.0, 0.0);
        if (has_pair) {
            pr = scale_pair(t, p, y, piv, slot);
            st_x(&t.bk_prow[(int64_t)J * ld + 2 * p], pr.x, local);
            st_x(&t.bk_prow[(int64_t)J * ld + 2 * p + 1], pr.y, local);
            z.x = pend(z.x, 2 * p     == slot, false, cmj, pr.x);
            z.y = pend(z.y, 2 * p + 1 == slot, false, cmj, pr.y);
        }
        s_pi[J][tid] = pr;
        if (own) {                                               // the slot changes hands
            if (2 * p == slot) l0 = leaving; else l1 = leaving;
            st_x(&t.p2l[slot], leaving, local);
            st_x(&t.l2p[leaving], slot, local);
            st_x(&t.l2p[ec], (int64_t)-1, local);
            st_x(&t.basis[cr], ec, local);                       // src/simplex.lisp:358
            my_sm |= 1u << (J + 16 * (int)(slot & 1));
            t.bk_smask[p] = my_sm;
        }
        if (has_row && r == cr) {
            my_rm |= 1u << J;
            t.bk_rmask[r] = my_rm;
        }
        if (__any(has_row && r == cr)) wave_rm |= 1u << J;
        if (__any(own))                wave_sm |= 1u << J;
        if (leader) {                                            // the one writer of these words
            const int64_t tn = c0.trace_n + J;
            ctl->ec = ec;
            ctl->cr = cr;
            ctl->slot = slot;
            if (t.trace_ec && tn < t.trace_cap) { t.trace_ec[tn] = ec; t.trace_cr[tn] = cr; }
            ctl->trace_n  = tn + 1;
            ctl->n_pivots = c0.n_pivots + J + 1;
            blk->cr[J] = cr;
            blk->slot[J] = slot;
            blk->ec[J] = ec;
            blk->n_pending = J + 1;
        }
        if (lane == J) { v_cr = cr; v_sl = slot; }
#if defined(MI355X_LA_TIMING) && MI355X_LA_TIMING != 2       // (2: publish stamps only, the leader is not slowed down)
        T6 = wall_clock64();
        if (leader) {
            double *d = t.rhs + J * 24;
            d[0] += 1.0; d[1] += (double)(T2 - T0); d[2] += (double)(T3 - T2);
            d[3] += (double)(T5 - T3); d[4] += (double)(T6 - T5);
            // inside the exchanges: entry -> own record published -> all records in -> result in every thread
            d[5] += (double)(tsp[0] - T0); d[6] += (double)(tsp[1] - tsp[0]); d[7] += (double)(tsp[2] - tsp[1]); d[8] += (double)tsp[3];
            d[9] += (double)(tsr[0] - T3); d[10] += (double)(tsr[1] - tsr[0]); d[11] += (double)(tsr[2] - tsr[1]); d[12] += (double)tsr[3];
        }
#endif
    }
    leave(ksteps);
}

// Pivots of the pending list that a sweep may apply: those EVERY workgroup of the persistent
// look-ahead launch `stamp` completed (BlockCtl::done).  All of n_pending in every run in which no
// exchange was lost; one fewer when a workgroup gave up on the ratio exchange of the last step while
// the leader's workgroup still saw all records and committed the pivot.
__device__ __forceinline__ int committed_pivots(const BlockCtl *__restrict__ blk, int k, unsigned stamp, int la_nw)
{
    for (int w = 0; w < la_nw; ++w) {
        const int64_t d = blk->done[w];
        const int dw = (unsigned)((unsigned long long)d >> 8) == stamp ? (int)(d & 0xff) : 0;
        k = dw < k ? dw : k;
    }
    return k;
}

// After a lost exchange: the sweep applied committed_pivots() of the n_pending pivots the leader
// recorded; take the bookkeeping of the others (at most one) back, newest first, so that the
// handle describes the tableau as it is -- the host then continues on the two-launch look-ahead.
// (The list and its stamp are those of the launch that lost the exchange: the launches enqueued
// behind it find the status kSyncLost and leave both alone.)
__global__ void k_la_rollback(TabView t, int la_nw)
{
    BlockCtl *blk = t.blk;
    Ctl *ctl = t.ctl;
    const int n = (int)blk->n_pending;
    const unsigned stamp = (unsigned)blk->stamp;
    if (n == 0 || stamp == 0u) return;
    const int keep = committed_pivots(blk, n, stamp, la_nw);
    for (int i = n - 1; i >= keep; --i) {
        const int64_t cr = blk->cr[i], slot = blk->slot[i], ec = blk->ec[i];
        // the maps are swapped by the thread that owns the slot's column pair -- which sits in the
        // workgroup that gave up when that workgroup holds the slot: then there is nothing to undo
        if (t.basis[cr] == ec) {
            const int64_t leaving = t.p2l[slot];
            t.p2l[slot] = ec;
            t.l2p[ec] = slot;
            t.l2p[leaving] = -1;
            t.basis[cr] = leaving;
        }
        ctl->n_pivots -= 1;
        ctl->trace_n -= 1;
    }
    blk->n_pending = keep;
}

// The sweep.  A workgroup owns a strip of <= 256 column pairs x tr rows and walks it four rows
// at a time; a thread keeps its prow pairs of all pending pivots in registers (loaded once per
// tile), the col values of the four rows are wave-uniform and sit in SGPRs (s_load_dwordx8 per
// pivot, issued by hand so that all of a chunk's loads are in flight together), and v_mul_f64
// takes them straight from there: per element pair and pivot 2 v_mul_f64 + 2 v_add_f64 and
// nothing else.  The two exceptions -- the step holds the pivot row of pending pivot i, or the
// wave holds the slot column pivot i gave up -- are decided per (step, pivot) by a scalar bit
// test, so only that one link of the chain takes the general form (pend()).
typedef int    v8i __attribute__((ext_vector_type(8)));
typedef double v4d __attribute__((ext_vector_type(4)));

// CH scalar loads of 4 doubles each (rows r .. r+3 of CH consecutive pivots: base + i*off bytes)
// and the wait for them, as ONE asm statement: the outputs must not be touched (or spilled)
// before the data has landed, and the compiler cannot know that about a bare s_load.
template <int CH>
__device__ __forceinline__ void sload_chunk(v8i (&c)[CH], const double *base, const unsigned (&off)[8])
{
    if constexpr (CH == 2)
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(c[0]), "=&s"(c[1]) : "s"(base), "s"(off[1]));
    if constexpr (CH == 4)
        asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, %5\n\t"
                     "s_load_dwordx8 %2, %4, %6\n\ts_load_dwordx8 %3, %4, %7\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3])
                     : "s"(base), "s"(off[1]), "s"(off[2]), "s"(off[3]));
    if constexpr (CH == 8)
        asm volatile("s_load_dwordx8 %0, %8, 0x0\n\ts_load_dwordx8 %1, %8, %9\n\t"
                     "s_load_dwordx8 %2, %8, %10\n\ts_load_dwordx8 %3, %8, %11\n\t"
                     "s_load_dwordx8 %4, %8, %12\n\ts_load_dwordx8 %5, %8, %13\n\t"
                     "s_load_dwordx8 %6, %8, %14\n\ts_load_dwordx8 %7, %8, %15\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3]), "=&s"(c[4]), "=&s"(c[5]), "=&s"(c[6]), "=&s"(c[7])
                     : "s"(base), "s"(off[1]), "s"(off[2]), "s"(off[3]), "s"(off[4]), "s"(off[5]), "s"(off[6]), "s"(off[7]));
}

template <int BLOCK, int KMAX, bool NT>
__global__ __launch_bounds__(BLOCK) void k_sweep(TabView t, const int tr, const int strip_pairs,
                                                 const double sgn, const int price, const unsigned stamp,
                                                 const int la_nw)
{
    constexpr int U = 4;                                       // rows per step
    constexpr int CH = KMAX < 4 ? KMAX : 4;                    // pivots per SGPR chunk (32 SGPRs: more would spill)
    t = lp_slice(t);                                           // batch: grid.z = LP
    const BlockCtl *__restrict__ blk = t.blk;
    int k = (int)blk->n_pending;
    if (k == 0) return;
    if (stamp != 0u && (unsigned)blk->stamp != stamp) return;  // the list is not this block's
    if (stamp != 0u && (k = committed_pivots(blk, k, stamp, la_nw)) == 0) return;
    double *__restrict__ M = t.M;
    const int64_t ld = t.ld, rows = t.rows, vc = t.cols - 1;
    const int64_t ldv  = ld >> 1;
    const int64_t pair = (int64_t)blockIdx.x * strip_pairs + threadIdx.x;
    const bool active  = (int)threadIdx.x < strip_pairs && pair < ldv;
    const int64_t r0 = (int64_t)blockIdx.y * tr;
    const int64_t r1 = (r0 + tr < rows) ? r0 + tr : rows;
    double  *__restrict__ part_v = price ? t.part_v : nullptr;
    const bool prices = (r1 == rows) && part_v != nullptr;
    if (!active && !prices) return;                            // no workgroup barrier below

    vec2d *Mp = reinterpret_cast<vec2d *>(M) + pair;
    auto ld2 = [&](int64_t r) -> vec2d {
        if constexpr (NT) return __builtin_nontemporal_load(Mp + r * ldv);
        else              return Mp[r * ldv];
    };
    auto st2 = [&](int64_t r, vec2d v) {
        if constexpr (NT) __builtin_nontemporal_store(v, Mp + r * ldv);
        else              Mp[r * ldv] = v;
    };
    vec2d last; last.x = 0.0; last.y = 0.0;
    if (active) {
        vec2d x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                          // the first step's rows, requested first
            x[u].x = 0.0; x[u].y = 0.0;
            if (r0 + u < r1) x[u] = ld2(r0 + u);
        }
        vec2d p[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            p[i] = reinterpret_cast<const vec2d *>(t.bk_prow)[(int64_t)i * ldv + pair];
        const unsigned sm = t.bk_smask[pair];
        const unsigned sx = sm & 0xffffu, sy = sm >> 16;
        unsigned wm = sx | sy;                                 // pivots whose slot this wave holds
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wm |= __shfl_xor(wm, off, 64);
        wm = __builtin_amdgcn_readfirstlane(wm);
        const unsigned pending = (k >= 32) ? 0xffffffffu : ((1u << k) - 1u);
        unsigned off[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) off[i] = (unsigned)(i * t.bk_stride * 8);
        for (int64_t r = r0; r < r1; r += U) {
            if (r != r0) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (r + u < r1) x[u] = ld2(r + u);
            }
            const uint4 rm = *reinterpret_cast<const uint4 *>(t.bk_rmask + r);   // uniform
            const unsigned rmu[U] = { rm.x, rm.y, rm.z, rm.w };
            const unsigned general = wm | rm.x | rm.y | rm.z | rm.w;   // bit i: pivot i needs pend()
#pragma unroll
            for (int c0 = 0; c0 < KMAX; c0 += CH) {
                constexpr unsigned cmask = (1u << CH) - 1u;
                const unsigned pend_c = (pending >> c0) & cmask;       // pending pivots of this chunk
                if (pend_c == 0u) continue;
                v8i cq[CH];
                sload_chunk<CH>(cq, t.bk_col + (int64_t)c0 * t.bk_stride + r, off);
                if (pend_c == cmask && ((general >> c0) & cmask) == 0u) {
                    // the common case: CH links of the bare chain
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const v4d cv = __builtin_bit_cast(v4d, cq[i]);
                        const vec2d pi = p[c0 + i];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const double m0 = cv[u] * pi.x;
                            const double m1 = cv[u] * pi.y;
                            x[u].x = x[u].x - m0;
                            x[u].y = x[u].y - m1;
                        }
                    }
                } else {
                    // a partial chunk (last block of a solve), a pivot row in this step, or a
                    // slot column in this wave: the general form of every link
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        if ((pend_c >> i) & 1u) {
                            const v4d cv = __builtin_bit_cast(v4d, cq[i]);
                            const vec2d pi = p[c0 + i];
                            if ((general >> (c0 + i)) & 1u) {
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    const bool is_cr = (rmu[u] >> (c0 + i)) & 1u;
                                    x[u].x = pend(x[u].x, (sx >> (c0 + i)) & 1u, is_cr, cv[u], pi.x);
                                    x[u].y = pend(x[u].y, (sy >> (c0 + i)) & 1u, is_cr, cv[u], pi.y);
                                }
                            } else {
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    const double m0 = cv[u] * pi.x;
                                    const double m1 = cv[u] * pi.y;
                                    x[u].x = x[u].x - m0;
                                    x[u].y = x[u].y - m1;
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u < r1) {
                    st2(r + u, x[u]);
                    last = x[u];
                }
            }
        }
    }
    if (prices) {                                              // `last` = new objective-row entries
        ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
        const int64_t c0 = 2 * pair;
        if (active && c0 < vc) {
            ValIdx c = price_cand(last.x * sgn, t.p2l ? t.p2l[c0] : c0, c0, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        if (active && c0 + 1 < vc) {
            ValIdx c = price_cand(last.y * sgn, t.p2l ? t.p2l[c0 + 1] : c0 + 1, c0 + 1, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        best = wave_reduce_min(best);
        if ((threadIdx.x & 63) == 0) {
            const int w = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
            part_v[w]   = best.v;
            t.part_i[w] = best.i;
            t.part_s[w] = best.s;
        }
    }
}

// ---- the sweep of a FULL block of 16 pending pivots (what a solve runs all the time; k_sweep
// above keeps the partial last block, other block sizes and the step that holds a pivot row) ----
// Same tile, same operands, same roundings; written so that the compiler has nothing to spill or
// copy in the row loop:
//   * two register sets of four rows (the loads of the next step are in flight while this one
//     is computed; the step function is instantiated for (A, B) and (B, A), so there is no copy);
//   * the col values of a chunk of 4 pivots x 4 rows (32 SGPRs) are requested while the previous
//     chunk is being applied (two SGPR sets; a scalar load has the latency of an L2 hit);
//   * a lane that holds a slot column is no special case in the row loop: pivot i's entering
//     column gave its slot to the leaving basic column, whose content before pivot i is e_cr, so
//     the chain of that column STARTS at pivot i from (r == cr_i ? 1 : 0).  The lane therefore
//     replaces what it loads by that unit entry and has its prow entries of the pivots before i
//     zeroed: unit - col*0 == unit bit for bit, links 0..i-1 are identities (col is finite: the
//     look-ahead refuses non-finite entering columns on this representation);
//   * a step that contains the pivot row of a pending pivot (16 of ~1000 steps) takes pend().
constexpr int kSweepK = 16;
typedef int    v16i __attribute__((ext_vector_type(16)));
typedef double v8d  __attribute__((ext_vector_type(8)));

// One chunk of col values = 32 SGPRs = CP pending pivots x U rows (U = 4: 4 pivots, U = 8: 2):
// requested by one asm statement, waited for by another -- the registers are tied through the
// wait, so nothing reads them before the data has landed.
template <int U> struct ColChunk;
template <> struct ColChunk<4> {
    static constexpr int CP = 4;
    v8i c[4];
#ifdef MI355X_SWEEP_FAKE_COL      // measurement only (tools/sweep_fake_col.py): no col loads, every col value +0.0
    __device__ __forceinline__ void issue(const double *, unsigned) {}
    __device__ __forceinline__ void wait() {}
    __device__ __forceinline__ double col(int, int) const { return 0.0; }
#else
    __device__ __forceinline__ void issue(const double *base, unsigned o1)
    {
        const unsigned o2 = 2u * o1, o3 = 3u * o1;
        asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, %5\n\t"
                     "s_load_dwordx8 %2, %4, %6\n\ts_load_dwordx8 %3, %4, %7"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3])
                     : "s"(base), "s"(o1), "s"(o2), "s"(o3));
    }
    __device__ __forceinline__ void wait()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c[0]), "+s"(c[1]), "+s"(c[2]), "+s"(c[3]));
    }
    __device__ __forceinline__ double col(int i, int u) const { return __builtin_bit_cast(v4d, c[i])[u]; }
#endif
};
template <> struct ColChunk<8> {
    static constexpr int CP = 2;
    v16i c[2];
    __device__ __forceinline__ void issue(const double *base, unsigned o1)
    {
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, %3"
                     : "=&s"(c[0]), "=&s"(c[1]) : "s"(base), "s"(o1));
    }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c[0]), "+s"(c[1])); }
    __device__ __forceinline__ double col(int i, int u) const { return __builtin_bit_cast(v8d, c[i])[u]; }
};

template <bool NT, int U>
__global__ __launch_bounds__(256) void k_sweep16(TabView t, const int tr, const int strip_pairs,
                                                 const double sgn, const int price, const unsigned stamp,
                                                 const int la_nw)
{
    constexpr int K = kSweepK, CP = ColChunk<U>::CP, NCH = K / CP;
    t = lp_slice(t);                                           // batch: grid.z = LP
    const BlockCtl *__restrict__ blk = t.blk;
    int k = (int)blk->n_pending;
    if (k == 0) return;
    if (stamp != 0u && (unsigned)blk->stamp != stamp) return;  // the list is not this block's
    if (stamp != 0u && (k = committed_pivots(blk, k, stamp, la_nw)) == 0) return;
    double *__restrict__ M = t.M;
    const int64_t ld = t.ld, rows = t.rows, vc = t.cols - 1;
    const int64_t ldv  = ld >> 1;
    const int64_t pair = (int64_t)blockIdx.x * strip_pairs + threadIdx.x;
    const bool active  = (int)threadIdx.x < strip_pairs && pair < ldv;
    const int64_t r0 = (int64_t)blockIdx.y * tr;
    const int64_t r1 = (r0 + tr < rows) ? r0 + tr : rows;
    double  *__restrict__ part_v = price ? t.part_v : nullptr;
    const bool prices = (r1 == rows) && part_v != nullptr;
    if (!active && !prices) return;                            // no workgroup barrier below

    vec2d *Mp = reinterpret_cast<vec2d *>(M) + pair;
    auto ld2 = [&](int64_t r) -> vec2d {
        if constexpr (NT) return __builtin_nontemporal_load(Mp + r * ldv);
        else              return Mp[r * ldv];
    };
    auto st2 = [&](int64_t r, vec2d v) {
        if constexpr (NT) __builtin_nontemporal_store(v, Mp + r * ldv);
        else              Mp[r * ldv] = v;
    };
    vec2d last; last.x = 0.0; last.y = 0.0;
    if (active) {
        const unsigned sm = t.bk_smask[pair];
        if (k != K) {
            // a partial block (the look-ahead terminated inside it): one row at a time, operands
            // from memory -- runs once per solve
            const unsigned sx = sm & 0xffffu, sy = sm >> 16;
            for (int64_t r = r0; r < r1; ++r) {
                vec2d x = ld2(r);
                const unsigned rm = t.bk_rmask[r];
                for (int i = 0; i < k; ++i) {
                    const double cv = t.bk_col[(int64_t)i * t.bk_stride + r];
                    const vec2d pi = reinterpret_cast<const vec2d *>(t.bk_prow)[(int64_t)i * ldv + pair];
                    const bool is_cr = (rm >> i) & 1u;
                    x.x = pend(x.x, (sx >> i) & 1u, is_cr, cv, pi.x);
                    x.y = pend(x.y, (sy >> i) & 1u, is_cr, cv, pi.y);
                }
                st2(r, x);
                last = x;
            }
        } else {
            vec2d xa[U], xb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                      // the first step's rows, requested first
                xa[u].x = 0.0; xa[u].y = 0.0; xb[u] = xa[u];
                if (r0 + u < r1) xa[u] = ld2(r0 + u);
            }
            vec2d p[K];
#pragma unroll
            for (int i = 0; i < K; ++i)
                p[i] = reinterpret_cast<const vec2d *>(t.bk_prow)[(int64_t)i * ldv + pair];
            const unsigned sx = sm & 0xffffu, sy = sm >> 16;
            // slot columns: chain starts at the last pivot that handed the slot over
            const bool wave_slots = __any(sm != 0u);
            int64_t crx = -1, cry = -1;
            if (wave_slots) {
                const int lx = sx ? 31 - __clz((int)sx) : -1, ly = sy ? 31 - __clz((int)sy) : -1;
                if (lx >= 0) crx = blk->cr[lx];
                if (ly >= 0) cry = blk->cr[ly];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    if (i < lx) p[i].x = 0.0;
                    if (i < ly) p[i].y = 0.0;
                }
            }
            const unsigned o1 = (unsigned)(t.bk_stride * 8);
            const double *colbase = t.bk_col;
            const int64_t chunk_stride = (int64_t)CP * t.bk_stride;

            auto step = [&](vec2d (&cur)[U], vec2d (&nxt)[U], const int64_t r) {
#pragma unroll
                for (int u = 0; u < U; ++u)                   // next step's rows travel during this one
                    if (r + U + u < r1) nxt[u] = ld2(r + U + u);
                unsigned rmu[U];                               // uniform
#pragma unroll
                for (int u4 = 0; u4 < U; u4 += 4) {
                    const uint4 rm = *reinterpret_cast<const uint4 *>(t.bk_rmask + r + u4);
                    rmu[u4] = rm.x; rmu[u4 + 1] = rm.y; rmu[u4 + 2] = rm.z; rmu[u4 + 3] = rm.w;
                }
                unsigned rm_any = 0u;
#pragma unroll
                for (int u = 0; u < U; ++u) rm_any |= rmu[u];
                if (wave_slots) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (crx >= 0) cur[u].x = (r + u == crx) ? 1.0 : 0.0;
                        if (cry >= 0) cur[u].y = (r + u == cry) ? 1.0 : 0.0;
                    }
                }
                const double *cb = colbase + r;
                if (rm_any == 0u) {
                    ColChunk<U> A, B;
                    auto apply = [&](const ColChunk<U> &c, const int i0) {
#pragma unroll
                        for (int i = 0; i < CP; ++i) {
                            const vec2d pi = p[i0 + i];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const double cv = c.col(i, u);
                                const double m0 = cv * pi.x;          // rounded products
                                const double m1 = cv * pi.y;
                                cur[u].x = cur[u].x - m0;             // rounded differences
                                cur[u].y = cur[u].y - m1;
                            }
                        }
                    };
                    A.issue(cb, o1);
                    A.wait();
#pragma unroll
                    for (int c = 0; c < NCH; c += 2) {        // two SGPR sets, alternating
                        B.issue(cb + (int64_t)(c + 1) * chunk_stride, o1);
                        apply(A, c * CP);
                        B.wait();
                        if (c + 2 < NCH) A.issue(cb + (int64_t)(c + 2) * chunk_stride, o1);
                        apply(B, (c + 1) * CP);
                        if (c + 2 < NCH) A.wait();
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        ColChunk<U> C;
                        C.issue(cb + (int64_t)c * chunk_stride, o1);
                        C.wait();
#pragma unroll
                        for (int i = 0; i < CP; ++i) {
                            const int pi_idx = c * CP + i;
                            const vec2d pi = p[pi_idx];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const bool is_cr = (rmu[u] >> pi_idx) & 1u;
                                cur[u].x = pend(cur[u].x, (sx >> pi_idx) & 1u, is_cr, C.col(i, u), pi.x);
                                cur[u].y = pend(cur[u].y, (sy >> pi_idx) & 1u, is_cr, C.col(i, u), pi.y);
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (r + u < r1) {
                        st2(r + u, cur[u]);
                        last = cur[u];
                    }
                }
            };
            for (int64_t r = r0; r < r1; r += 2 * U) {
                step(xa, xb, r);
                if (r + U < r1) step(xb, xa, r + U);
            }
        }
    }
    if (prices) {                                              // `last` = new objective-row entries
        ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
        const int64_t c0 = 2 * pair;
        if (active && c0 < vc) {
            ValIdx c = price_cand(last.x * sgn, t.p2l ? t.p2l[c0] : c0, c0, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        if (active && c0 + 1 < vc) {
            ValIdx c = price_cand(last.y * sgn, t.p2l ? t.p2l[c0 + 1] : c0 + 1, c0 + 1, t.p2l ? 0 : t.col_bias);
            best = vi_min(best, c);
        }
        best = wave_reduce_min(best);
        if ((threadIdx.x & 63) == 0) {
            const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
            part_v[w]   = best.v;
            t.part_i[w] = best.i;
            t.part_s[w] = best.s;
        }
    }
}

// ------------------------------------------------------------------ batch: one workgroup per LP
// BASELINE config 4 is many SMALL independent LPs (257 x 769 doubles = 1.6 MB each).  Advancing
// them in lockstep with the launch pairs above makes every LP wait for the slowest one (78..199
// pivots per LP in the benchmark batch) and pays 2-3 launch boundaries per pivot.  Here ONE
// 1024-thread workgroup owns one LP and runs its whole n-solve-tableau loop
// (src/simplex.lisp:453-461) inside a single launch: the snapshot of the entering column and
// the normalised pivot row live in LDS, the phases are separated by workgroup barriers only,
// LPs progress independently and the hardware schedules waiting LPs onto free CUs.  Same
// arithmetic, same lexicographic reductions => same bits as the lockstep path and the oracle.
constexpr int kLpThreads = 1024;
constexpr int64_t kBatchLaunchCap = 4096;   // pivots per LP and launch of the per-LP kernels (a bounded launch: see mi355x_batch_cancel)
constexpr int kLpUnroll  = 4;   // measured at 128 / 1024 LPs of 257x513: 2 -> 1.76 / 1.49, 4 -> 1.76 / 2.13, 8 -> 1.22 / 1.33 M pivots/s

__global__ __launch_bounds__(kLpThreads) void k_batch_solve(TabView t, double sgn, double price_tol,
                                                           double ratio_thr)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double    s_v[kLpThreads / 64];
    __shared__ long long s_i[kLpThreads / 64];
    t = lp_slice(t);
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;                            // one load of the whole control block
    if (c0.status != kRunning) return;
    const int64_t rows = t.rows, m = rows - 1, vc = t.cols - 1, ld = t.ld, ldv = ld >> 1;
    double  *s_prow = lds;                          // ld doubles (16-byte aligned)
    double  *s_col  = lds + ld;                     // rows doubles
    vec2d   *M2 = reinterpret_cast<vec2d *>(t.M);
    const int64_t total = rows * ldv;               // tableau size in 16-byte pairs
    const int64_t dr = kLpThreads / ldv, dp = kLpThreads % ldv;   // flat-index stride as (row, pair)
    Ctl cl = c0;                                    // running copy (n_pivots / trace_n advance)
    int64_t n_pivots = c0.n_pivots;
    const int64_t max_pivots = c0.max_pivots;

    ValIdx e = block_price(t.M + m * ld, vc, sgn, s_v, s_i, t.p2l, t.col_bias);
    for (;;) {
        if (price_says_optimal(e, price_tol)) {
            if (threadIdx.x == 0) ctl->status = 0;  // MI_OPTIMAL
            break;
        }
        if (max_pivots > 0 && n_pivots >= max_pivots) {
            if (threadIdx.x == 0) ctl->status = 3;  // MI_MAX_PIVOTS
            break;
        }
        // a launch is bounded (the host can then honour a cancel between launches; the reference's
        // loop has no cap and no anti-cycling rule): still kRunning, the host launches again
        if (n_pivots - c0.n_pivots >= kBatchLaunchCap) break;
        const int64_t ec   = e.i;                   // LOGICAL column
        const int64_t slot = e.s;                   // its physical column
        // gather the entering column into LDS + ratio test
        ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
        ValIdx first; first.v = 0.0; first.i = -1; first.s = 0;   // this thread's first eligible row
        int bad = 0, nanq = 0;
        for (int64_t r = threadIdx.x; r < rows; r += kLpThreads) {
            const double a = t.M[r * ld + slot];
            s_col[r] = a;
            bad |= !(fabs(a) <= 1.7976931348623157e308);
            if (r < m && ratio_thr < a) {
                ValIdx c; c.v = t.M[r * ld + vc] / a; c.i = r; c.s = 0;
                if (first.i < 0) first.i = r;
                if (c.v != c.v) nanq = 1;            // no candidate, unless its row is the first eligible one
                else best = vi_min(best, c);
            }
        }
        ValIdx q = block_reduce_min(best, s_v, s_i);           // barriers: s_col complete
        if (__syncthreads_or(nanq)) {                          // the first-eligible-row rule: see block_gather_ratio
            first = block_reduce_min(first, s_v, s_i);
            if (first.i >= 0) {
                const double q0 = t.M[first.i * ld + vc] / s_col[first.i];
                if (q0 != q0) { q.v = q0; q.i = first.i; }
            }
        }
        if (t.p2l && __syncthreads_or(bad)) {                  // see kNeedDense
            if (threadIdx.x == 0) ctl->status = kNeedDense;
            break;
        }
        if (q.i < 0) {
            if (threadIdx.x == 0) ctl->status = 1;  // MI_UNBOUNDED
            break;
        }
        const int64_t cr = q.i;
        const double row_scale = s_col[cr];
        // normalised pivot row into LDS
        for (int64_t p = threadIdx.x; p < ldv; p += kLpThreads) {
            const double2 v = reinterpret_cast<const double2 *>(t.M + cr * ld)[p];
            reinterpret_cast<double2 *>(s_prow)[p] = scale_pair(t, p, v, row_scale, t.p2l ? slot : -1);
        }
        if (t.p2l) {                                // compact: the slot takes over the leaving column
            __syncthreads();                        // row cr has been read
            for (int64_t r = threadIdx.x; r < rows; r += kLpThreads)
                t.M[r * ld + slot] = (r == cr) ? 1.0 : 0.0;
            if (threadIdx.x == 0) swap_columns(t, ec, cr, slot);
        }
        __syncthreads();
        // rank-1 update of the whole tableau, kLpUnroll independent 16-byte accesses in flight per thread
        // (one workgroup streams at bytes-in-flight / latency: 8 x 16 B x 1024 threads per ~2 us);
        // the threads that write the objective row price it for the next iteration
        best.v = 0.0; best.i = -1; best.s = 0;
        int64_t idx = threadIdx.x, r = threadIdx.x / ldv, p = threadIdx.x % ldv;
        while (idx < total) {
            vec2d   v[kLpUnroll];
            int64_t ri[kLpUnroll], pi[kLpUnroll], ii[kLpUnroll];
            int     n = 0;
#pragma unroll
            for (int u = 0; u < kLpUnroll; ++u) {
                if (idx < total) {
                    ri[u] = r; pi[u] = p; ii[u] = idx;
                    v[u] = M2[idx];
                    n = u + 1;
                    idx += kLpThreads; r += dr; p += dp;
                    if (p >= ldv) { p -= ldv; r += 1; }
                }
            }
#pragma unroll
            for (int u = 0; u < kLpUnroll; ++u) {
                if (u < n) {
                    const double s = s_col[ri[u]];
                    const double2 pp = reinterpret_cast<const double2 *>(s_prow)[pi[u]];
                    const double m0 = s * pp.x;
                    const double m1 = s * pp.y;
                    vec2d o;
                    o.x = v[u].x - m0;
                    o.y = v[u].y - m1;
                    if (ri[u] == cr) { o.x = pp.x; o.y = pp.y; }
                    M2[ii[u]] = o;
                    if (ri[u] == m) {
                        const int64_t c0 = 2 * pi[u];
                        if (c0 < vc)     { ValIdx c = price_cand(o.x * sgn, t.p2l ? t.p2l[c0] : c0, c0, t.p2l ? 0 : t.col_bias);     best = vi_min(best, c); }
                        if (c0 + 1 < vc) { ValIdx c = price_cand(o.y * sgn, t.p2l ? t.p2l[c0 + 1] : c0 + 1, c0 + 1, t.p2l ? 0 : t.col_bias); best = vi_min(best, c); }
                    }
                }
            }
        }
        if (threadIdx.x == 0) record_pivot(t, cl, ec, cr);
        cl.n_pivots += 1; cl.trace_n += 1;
        n_pivots += 1;
        e = block_reduce_min(best, s_v, s_i);       // barrier: the update is complete and visible
    }
}

// ---- the same with blocked pivoting (compact representation) --------------------------------
// One workgroup streams its LP at (bytes in flight) / (memory latency) ~ 30 GB/s, so a pivot of a
// 1 MB tableau costs ~70 us however the loop is written.  Blocked as in k_la_block / k_sweep, but
// with everything inside the one workgroup: the look-ahead state of up to KB pending pivots
// (col_i, prow_i), the running objective row, RHS column and column map live in LDS, a look-ahead
// step is two memory round trips (one strided column, one row) and two workgroup reductions, and
// the tableau itself is read and written once per KB pivots.
// 512 threads: the look-ahead never has more than a few hundred elements to spread, and the sweep
// wants 256 VGPRs per thread (16 prow pairs + four rows in flight; at 1024 threads it spilled).
constexpr int kBbThreads = 512, kRowPhases = kBbThreads / 256;

// Reduction of the per-LP kernel: wave arg-min (wave_argmin: v_min_f64 butterfly, the tree when a
// NaN or a tie is involved), one LDS slot per wave, ONE barrier, then every thread folds the
// wave winners in wave order -- the order of block_reduce_min.  `buf` alternates between the
// pricing and the ratio reduction of a step, so the slots of one are never rewritten before every
// thread has passed the other's barrier.
struct BbMsg { ValIdx c; unsigned flag; unsigned pad; };
__device__ __forceinline__ ValIdx bb_reduce(ValIdx mine, unsigned myflag, BbMsg *buf, unsigned &flag)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Cand c; c.v = mine.v; c.i = (int)mine.i;
    int src;
    c = wave_argmin(c, src);
    const int64_t cs = lane_pick(mine.s, src);
    const unsigned wf = __any(myflag != 0u) ? 1u : 0u;
    if (lane == 0) { buf[wave].c.v = c.v; buf[wave].c.i = c.i; buf[wave].c.s = cs; buf[wave].flag = wf; }
    __syncthreads();
    ValIdx r = buf[0].c;
    flag = buf[0].flag;
#pragma unroll
    for (int w = 1; w < kBbThreads / 64; ++w) { r = vi_min(r, buf[w].c); flag |= buf[w].flag; }
    return r;
}

template <int KB, bool SPLIT>
__global__ __launch_bounds__(kBbThreads) void k_batch_block(TabView t, double sgn, double price_tol,
                                                           double ratio_thr)
{
    // SPLIT == false: the whole solve of the LP in this launch, look-ahead and sweeps alternating.
    // SPLIT == true: the look-ahead of ONE block only; col_i / prow_i / masks / pending list go to
    // the LP's global block state and the sweep is a separate launch over ALL LPs (k_sweep with
    // grid.z = LP), which uses every CU of the chip instead of one per LP.  (A template parameter:
    // the look-ahead-only form carries neither the code nor the registers of the in-kernel sweep.)
    constexpr bool split = SPLIT;
    static_assert(KB % 4 == 0 && KB <= 16, "chains in groups of four links; 16 + 16 mask bits per pair");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ BbMsg s_wp[kBbThreads / 64], s_wr[kBbThreads / 64];
    t = lp_slice(t);
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    if (c0.status != kRunning) {
        if (split && threadIdx.x == 0) t.blk->n_pending = 0;   // nothing for the sweep that follows
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t rows = t.rows, m = rows - 1, vc = t.cols - 1, ld = t.ld, ldv = ld >> 1;
    const int64_t rp = (rows + 1) & ~(int64_t)1;
    double    *s_prow = lds;                                   // KB x ld
    double    *s_col  = s_prow + (int64_t)KB * ld;             // KB x rp
    double    *s_z    = s_col + (int64_t)KB * rp;              // ld: objective row through all pending pivots
    double    *s_b    = s_z + ld;                              // rp: RHS column, through all but the last one
    long long *s_p2l  = reinterpret_cast<long long *>(s_b + rp);   // ld: logical column of a slot
    long long *s_bas  = s_p2l + ld;                            // rp: basis (the leaving column of a pivot)
    unsigned  *s_rm   = reinterpret_cast<unsigned *>(s_bas + rp);  // rp: row -> pending pivots whose row it is
    unsigned  *s_sm   = s_rm + rp;                             // ldv: pair -> pending pivots whose slot it holds
    vec2d *M2 = reinterpret_cast<vec2d *>(t.M);

    // Ownership, fixed for the whole solve: a thread owns the rows r = tid, tid + T, ... (their
    // s_b / s_col / s_rm entries) and the column pairs p = tid, tid + T, ... (their s_z / s_prow /
    // s_p2l / s_sm entries): what a thread writes of these only it reads before the next barrier.
    for (int64_t c = tid; c < ld; c += kBbThreads) {
        s_z[c] = t.M[m * ld + c];
        s_p2l[c] = c < vc ? t.p2l[c] : -1;
    }
    for (int64_t r = tid; r < rp; r += kBbThreads) {
        s_b[r] = r < rows ? t.M[r * ld + vc] : 0.0;
        s_bas[r] = r < m ? t.basis[r] : -1;
    }
    int64_t n_pivots = c0.n_pivots, trace_n = c0.trace_n;
    int term = -1;                                             // status that ends the solve
    __syncthreads();

    while (term < 0) {
        for (int64_t r = tid; r < rp; r += kBbThreads) s_rm[r] = 0u;
        for (int64_t p = tid; p < ldv; p += kBbThreads) s_sm[p] = 0u;
        int k = 0, b_done = 0;                                 // pending pivots; how many of them s_b has seen
        int64_t v_cr = -1, v_sl = -1;                          // lane i: pivot row / slot of pending pivot i
        for (int J = 0; J < KB && term < 0; ++J) {
#ifdef MI355X_LA_TIMING
            const bool tm = blockIdx.z == 0 && tid == 0;
            unsigned long long T0 = wall_clock64(), T1 = T0, T2 = T0, T3 = T0, T4 = T0, T5 = T0;
#endif
            // ---- find-entering-column on the running objective row (my pairs)
            ValIdx best; best.v = 0.0; best.i = -1; best.s = 0;
            for (int64_t p = tid; p < ldv; p += kBbThreads) {
                const double2 z = reinterpret_cast<const double2 *>(s_z)[p];
                if (2 * p < vc)     { ValIdx x = price_cand(z.x * sgn, s_p2l[2 * p], 2 * p);     best = vi_min(best, x); }
                if (2 * p + 1 < vc) { ValIdx x = price_cand(z.y * sgn, s_p2l[2 * p + 1], 2 * p + 1); best = vi_min(best, x); }
            }
            unsigned fl;
#ifdef MI355X_LA_TIMING
            T1 = wall_clock64();
#endif
            const ValIdx e = bb_reduce(best, 0u, s_wp, fl);     // barrier: prow_{J-1}, s_z, masks of step J-1 complete
#ifdef MI355X_LA_TIMING
            T2 = wall_clock64();
#endif
            if (price_says_optimal(e, price_tol)) { term = 0; break; }          // MI_OPTIMAL
            if (c0.max_pivots > 0 && n_pivots >= c0.max_pivots) { term = 3; break; }   // MI_MAX_PIVOTS
            const int64_t ec = e.i, slot = uniform64(e.s);
            // ---- entering column through the pending chain, ratio test.  Uniform operands of the
            // chain: lane i holds prow_i[slot]; bit i of slmask: pending pivot i gave up this slot
            const double v_pa = lane < J ? s_prow[(int64_t)lane * ld + slot] : 0.0;
            const unsigned slmask = (unsigned)__ballot((lane < J) & (v_sl == slot));
            const double pb = J > 0 ? s_prow[(int64_t)(J - 1) * ld + vc] : 0.0;   // RHS entry of prow_{J-1}
            ValIdx q; q.v = 0.0; q.i = -1; q.s = 0;
            unsigned bad = 0u;
            for (int64_t r = tid; r < rows; r += kBbThreads) {
                double a = t.M[r * ld + slot];
                const unsigned rmb = s_rm[r];
                double b = s_b[r];
                if (J > 0) {                                   // RHS entry brought up to date with pivot J-1
                    b = pend(b, false, (rmb >> (J - 1)) & 1u, s_col[(int64_t)(J - 1) * rp + r], pb);
                    s_b[r] = b;
                }
#pragma unroll
                for (int i0 = 0; i0 < KB; i0 += 4) {
                    if (i0 < J) {
                        double prod[4], pa[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const double ci = (i0 + kk < J) ? s_col[(int64_t)(i0 + kk) * rp + r] : 0.0;
                            pa[kk] = lane_value(v_pa, i0 + kk);
                            prod[kk] = ci * pa[kk];                    // rounded product
                        }
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const bool is_cr = (rmb >> (i0 + kk)) & 1u;
                            if ((slmask >> (i0 + kk)) & 1u) a = is_cr ? 1.0 : 0.0;
                            const double d = a - prod[kk];             // rounded difference
                            a = is_cr ? pa[kk] : d;
                        }
                    }
                }
                s_col[(int64_t)J * rp + r] = a;
                if (split) t.bk_col[(int64_t)J * t.bk_stride + r] = a;   // for the sweep launch
                bad |= !(fabs(a) <= 1.7976931348623157e308);
                if (r < m && ratio_thr < a) {
                    ValIdx x; x.v = b / a; x.i = r; x.s = __double_as_longlong(a);
                    if (x.v != x.v) bad = 1;         // a NaN quotient: decided by the lockstep select (kNeedDense)
                    else q = vi_min(q, x);
                }
            }
            b_done = J;
#ifdef MI355X_LA_TIMING
            T3 = wall_clock64();
#endif
            q = bb_reduce(q, bad, s_wr, fl);                   // barrier: s_col[J] complete
#ifdef MI355X_LA_TIMING
            T4 = wall_clock64();
#endif
            if (fl) { term = kNeedDense; break; }
            if (q.i < 0) { term = 1; break; }                  // MI_UNBOUNDED
            const int64_t cr = uniform64(q.i);
            const double piv = __longlong_as_double(q.s);
            // ---- pivot row through the chain -> prow_J; objective row through pivot J
            const double cmj = s_col[(int64_t)J * rp + m];
            const double v_ccr = lane < J ? s_col[(int64_t)lane * rp + cr] : 0.0;
            const unsigned crmask = (unsigned)__ballot((lane < J) & (v_cr == cr));
            for (int64_t p = tid; p < ldv; p += kBbThreads) {
                const vec2d y0 = M2[cr * ldv + p];
                double2 y = make_double2(y0.x, y0.y);
                const unsigned smb = s_sm[p];
#pragma unroll
                for (int i0 = 0; i0 < KB; i0 += 4) {
                    if (i0 < J) {
                        double2 pii[4], prod[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            pii[kk] = (i0 + kk < J) ? reinterpret_cast<const double2 *>(s_prow + (int64_t)(i0 + kk) * ld)[p]
                                                    : make_double2(0.0, 0.0);
                            const double ccr = lane_value(v_ccr, i0 + kk);
                            prod[kk].x = ccr * pii[kk].x;              // rounded products
                            prod[kk].y = ccr * pii[kk].y;
                        }
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const bool is_cr = (crmask >> (i0 + kk)) & 1u;
                            if ((smb >> (i0 + kk)) & 1u)      y.x = is_cr ? 1.0 : 0.0;
                            if ((smb >> (16 + i0 + kk)) & 1u) y.y = is_cr ? 1.0 : 0.0;
                            const double dx = y.x - prod[kk].x, dy = y.y - prod[kk].y;
                            y.x = is_cr ? pii[kk].x : dx;
                            y.y = is_cr ? pii[kk].y : dy;
                        }
                    }
                }
                const double2 pr = scale_pair(t, p, y, piv, slot);
                reinterpret_cast<double2 *>(s_prow + (int64_t)J * ld)[p] = pr;
                if (split) reinterpret_cast<double2 *>(t.bk_prow + (int64_t)J * ld)[p] = pr;
                double2 z = reinterpret_cast<double2 *>(s_z)[p];
                z.x = pend(z.x, 2 * p     == slot, false, cmj, pr.x);
                z.y = pend(z.y, 2 * p + 1 == slot, false, cmj, pr.y);
                reinterpret_cast<double2 *>(s_z)[p] = z;
                if (p == (slot >> 1)) {                        // the slot changes hands: its owner keeps the books
                    const int64_t leaving = s_bas[cr];
                    s_p2l[slot] = leaving;
                    s_sm[p] = smb | (1u << (J + 16 * (int)(slot & 1)));
                    t.p2l[slot] = leaving;
                    t.l2p[leaving] = slot;
                    t.l2p[ec] = -1;
                    t.basis[cr] = ec;                          // src/simplex.lisp:358
                    s_bas[cr] = ec;                            // (next read: after the next barrier)
                }
            }
            for (int64_t r = tid; r < rows; r += kBbThreads)
                if (r == cr) s_rm[r] |= 1u << J;
            if (tid == 0) {
                if (split) { t.blk->cr[J] = cr; t.blk->slot[J] = slot; }
                if (t.trace_ec && trace_n < t.trace_cap) { t.trace_ec[trace_n] = ec; t.trace_cr[trace_n] = cr; }
            }
            if (lane == J) { v_cr = cr; v_sl = slot; }
            n_pivots += 1; trace_n += 1;
            k = J + 1;
#ifdef MI355X_LA_TIMING
            T5 = wall_clock64();
            if (tm) {
                double *d = t.col;                              // (unused by this kernel)
                d[0] += 1.0; d[1] += (double)(T1 - T0); d[2] += (double)(T2 - T1); d[3] += (double)(T3 - T2);
                d[4] += (double)(T4 - T3); d[5] += (double)(T5 - T4);
            }
#endif
        }
        __syncthreads();                                       // prow / col / masks of the block complete
        if (split) {                                           // hand the block over to the sweep launch
            for (int64_t r = tid; r < t.bk_stride; r += kBbThreads) t.bk_rmask[r] = r < rp ? s_rm[r] : 0u;
            for (int64_t p = tid; p < ldv; p += kBbThreads) t.bk_smask[p] = s_sm[p];
            if (tid == 0) {
                t.blk->n_pending = k;
                if (term >= 0) ctl->status = term;
                ctl->n_pivots = n_pivots;
                ctl->trace_n = trace_n;
            }
            return;
        }
        if (k > b_done) {                                      // RHS column through the last pending pivot
            const double pb = s_prow[(int64_t)(k - 1) * ld + vc];
            for (int64_t r = tid; r < rows; r += kBbThreads)
                s_b[r] = pend(s_b[r], false, (s_rm[r] >> (k - 1)) & 1u, s_col[(int64_t)(k - 1) * rp + r], pb);
        }
        // ---- the sweep: the k pending pivots applied to every stored element.  Full strips of
        // 256 column pairs: a thread owns one pair (its prow entries of the pending pivots in
        // registers) and every fourth row; the row's col values come out of LDS as broadcasts.
        // The pairs left over beyond the last full strip: flat, operands from LDS.
        if (k > 0) {
            constexpr int kSU = KB >= 16 ? 2 : 4;                  // rows in flight per thread (16 pending pivots: 64 VGPRs of prow pairs)
            const int pp = tid & 255, rq = tid >> 8;               // pair within the strip, row phase
            const int64_t full = ldv & ~(int64_t)255;
            for (int64_t s0 = 0; s0 < full; s0 += 256) {
                const int64_t p = s0 + pp;
                double2 pr[KB];
#pragma unroll
                for (int i = 0; i < KB; ++i)
                    pr[i] = reinterpret_cast<const double2 *>(s_prow + (int64_t)i * ld)[p];
                const unsigned sm = s_sm[p];
                vec2d x[kSU], nx[kSU];
#pragma unroll
                for (int u = 0; u < kSU; ++u)
                    if (rq + kRowPhases * u < rows) nx[u] = M2[(rq + kRowPhases * u) * ldv + p];
                for (int64_t r = rq; r < rows; r += kRowPhases * kSU) {
#pragma unroll
                    for (int u = 0; u < kSU; ++u) x[u] = nx[u];
#pragma unroll
                    for (int u = 0; u < kSU; ++u)                  // the next rows travel during the chain
                        if (r + kRowPhases * (kSU + u) < rows) nx[u] = M2[(r + kRowPhases * (kSU + u)) * ldv + p];
#pragma unroll
                    for (int u = 0; u < kSU; ++u) {
                        const int64_t rr = r + kRowPhases * u;
                        if (rr < rows) {
                            const unsigned rm = s_rm[rr];
                            vec2d v = x[u];
                            if ((rm | sm) == 0u) {             // the bare chain
#pragma unroll
                                for (int i = 0; i < KB; ++i) {
                                    if (i < k) {
                                        const double s = s_col[(int64_t)i * rp + rr];
                                        const double m0 = s * pr[i].x, m1 = s * pr[i].y;
                                        v.x = v.x - m0;
                                        v.y = v.y - m1;
                                    }
                                }
                            } else {
#pragma unroll
                                for (int i = 0; i < KB; ++i) {
                                    if (i < k) {
                                        const double s = s_col[(int64_t)i * rp + rr];
                                        const bool is_cr = (rm >> i) & 1u;
                                        v.x = pend(v.x, (sm >> i) & 1u, is_cr, s, pr[i].x);
                                        v.y = pend(v.y, (sm >> (i + 16)) & 1u, is_cr, s, pr[i].y);
                                    }
                                }
                            }
                            M2[rr * ldv + p] = v;
                        }
                    }
                }
            }
            const int64_t rem = ldv - full, total_rem = rows * rem;   // < 256 pairs per row
            for (int64_t idx = tid; idx < total_rem; idx += kBbThreads) {
                const int64_t r = idx / rem, p = full + (idx - r * rem);
                const unsigned rm = s_rm[r], sm = s_sm[p];
                vec2d v = M2[r * ldv + p];
                for (int i = 0; i < k; ++i) {
                    const double  s = s_col[(int64_t)i * rp + r];
                    const double2 pi = reinterpret_cast<const double2 *>(s_prow + (int64_t)i * ld)[p];
                    const bool is_cr = (rm >> i) & 1u;
                    v.x = pend(v.x, (sm >> i) & 1u, is_cr, s, pi.x);
                    v.y = pend(v.y, (sm >> (i + 16)) & 1u, is_cr, s, pi.y);
                }
                M2[r * ldv + p] = v;
            }
            __syncthreads();                                   // tableau consistent before the next block reads it
        }
        // a launch is bounded (see k_batch_solve): still kRunning, the host launches again
        if (term < 0 && n_pivots - c0.n_pivots >= kBatchLaunchCap) term = kRunning;
    }
    if (tid == 0) {
        ctl->status = term;
        ctl->n_pivots = n_pivots;
        ctl->trace_n = trace_n;
    }
}


// ------------------------------------------------------------------ the resident solve
// Tableaux that fit the chip's REGISTER FILES (BASELINE config 2: 513 x 1025 stored doubles = 4.2 MB
// against 256 CUs x 512 KB of vector registers; every LP of config 4) never need to move through
// HBM inside the solve loop at all.  The stored tableau [non-basic columns | RHS] is split into
// column strips of CW columns; workgroup w of the LP keeps strip w -- ALL constraint rows of its
// CW columns -- in registers (thread t owns rows t, t + 256, ...: TR x CW doubles), plus its own
// copy of the RHS column, its part of the objective row and the basis.  One pivot of
// n-solve-tableau (src/simplex.lisp:453-461) is then
//     local pricing of the strip's objective entries -> this workgroup's best column
//     ONE exchange: every workgroup publishes (key, logical column) AND that column itself
//         (speculatively: 16 bytes per row) AND -- a moment later -- the result of
//         find-pivoting-row on that column (it has the column and, like everybody, an identical
//         copy of the RHS column); everybody reduces the G records to the same winner --
//         lexicographic (key, logical column) minimum = find-entering-column's lowest-index strict
//         minimum -- and reads the winner's column and pivot row, which are already there
//     pivot row: the strip's own entries of row cr, normalised locally; the objective entries
//         brought up to date, priced, the next candidate column computed as the update will leave
//         it and published; the ratio test on it; and only then -- at the top of the next
//         iteration, between asking for the records and looking at them -- the rank-1 update of
//         the strip in registers
// so a pivot costs one all-to-all exchange through L2 plus a few hundred cycles of arithmetic per
// phase, and no HBM traffic.  What bounds it is the serial chain
//     price -> publish -> ratio test on the candidate -> (L2) -> winner -> column + row -> pivot row
// of ONE wave per SIMD (every instruction costs its full latency), which is why the order above
// puts everything that is not on that chain (the strip update, the store traffic of the publish)
// under a wait: measured per pivot at config 2, round 3: 4.3 us with the ratio test behind the
// exchange and the update in front of it, 3.9 us in this order.  The operations on every element
// are n-pivot-row's
// (rounded product, rounded difference, true division), so pivots and bits are those of every
// other path.  The tableau is read from HBM when the launch starts and written back when it ends
// (optimal / unbounded / cap / a pivot the compact representation cannot follow).
//
// Exchange: slot (workgroup, epoch parity) = 8 record granules (4 + the pivot row in use) + 2 granules per row, every granule
// {tag = epoch, 32 bits of payload} written by one write-through store and polled with
// L1-bypassing loads until the tag matches (as k_la_block's records).  Two parities suffice: a
// workgroup can only publish epoch e + 2 after everybody has published e + 1, i.e. has finished
// reading e.  All workgroups of an LP must be co-resident: block b of the launch serves LP
// (b / 8 / G) * 8 + b % 8, so the G workgroups of an LP are dispatched together (and to one XCD);
// a workgroup that waits in vain at the FIRST exchange gives up (kSyncLost, nothing has been
// modified: the host continues on the established paths), a later one cannot happen short of a
// hung GPU and is reported as an error (kResidentStuck) instead of a wrong tableau.
constexpr int kResThreads = 256;

typedef double v16d __attribute__((ext_vector_type(16)));

// Two adjacent granules (a 16-byte aligned pair) by ONE store / ONE load.  Every granule carries its
// own tag and is validated on its own, so all that is needed is that an aligned 8-byte half is never
// torn -- which a naturally aligned 16-byte access does not do.  LOCAL: every reader shares this
// XCD's L2 (st_x).  Hand-issued: the compiler has no 16-byte access with these cache bits.
template <bool LOCAL>
__device__ __forceinline__ void st_pair(unsigned long long *p, unsigned long long g0, unsigned long long g1)
{
    const v4u v = { (unsigned)g0, (unsigned)(g0 >> 32), (unsigned)g1, (unsigned)(g1 >> 32) };
    // (s_nop: a store of more than 8 bytes reads its data registers for a few cycles after it has
    // issued; the compiler pads its own such stores, it cannot see into this one -- without the
    // padding the next pair's payload overwrote this one's in flight: records that never validate)
    if (LOCAL) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 2" :: "v"(p), "v"(v) : "memory");
    else       asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" :: "v"(p), "v"(v) : "memory");
}
// N granule pairs + one more pair + one single granule, all loads in flight together, L1 bypassed
// (sc1: what ld_l2 compiles to); the wait is part of the statement -- the compiler does not count
// these loads
template <int N> struct PairLoads;
template <> struct PairLoads<1> {
    static __device__ __forceinline__ void run(v4u (&r)[1], v4u &rm, unsigned long long &g, const unsigned long long *const (&a)[1],
                                               const unsigned long long *am, const unsigned long long *ag)
    {
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                     "global_load_dwordx4 %1, %4, off sc1\n\t"
                     "global_load_dwordx2 %2, %5, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r[0]), "=&v"(rm), "=&v"(g) : "v"(a[0]), "v"(am), "v"(ag) : "memory");
    }
};
template <> struct PairLoads<2> {
    static __device__ __forceinline__ void run(v4u (&r)[2], v4u &rm, unsigned long long &g, const unsigned long long *const (&a)[2],
                                               const unsigned long long *am, const unsigned long long *ag)
    {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                     "global_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\t"
                     "global_load_dwordx2 %3, %7, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(rm), "=&v"(g) : "v"(a[0]), "v"(a[1]), "v"(am), "v"(ag) : "memory");
    }
};
template <> struct PairLoads<4> {
    static __device__ __forceinline__ void run(v4u (&r)[4], v4u &rm, unsigned long long &g, const unsigned long long *const (&a)[4],
                                               const unsigned long long *am, const unsigned long long *ag)
    {
        asm volatile("global_load_dwordx4 %0, %6, off sc1\n\t"
                     "global_load_dwordx4 %1, %7, off sc1\n\t"
                     "global_load_dwordx4 %2, %8, off sc1\n\t"
                     "global_load_dwordx4 %3, %9, off sc1\n\t"
                     "global_load_dwordx4 %4, %10, off sc1\n\t"
                     "global_load_dwordx2 %5, %11, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(rm), "=&v"(g)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(am), "v"(ag) : "memory");
    }
};

template <int TR, int CW, bool EVERY>
__global__ __launch_bounds__(kResThreads, (TR == 1 ? 2 : 1)) void k_resident(TabView t, const ResidentArgs a)
{
    static_assert(CW % 16 == 0 && CW <= 64 && TR * CW <= 64, "strip of TR x CW doubles per thread, pricing in one wave");
    constexpr int CH = CW / 16;                                  // 16-column register chunks (indexable by a uniform value)
    constexpr int NW = kResThreads / 64;
    __shared__ double    s_wv[2][NW];                            // ratio reduction: wave winners (double-buffered by pivot parity)
    __shared__ int       s_wi[2][NW];
    __shared__ unsigned  s_wf[2][NW];
    __shared__ __attribute__((aligned(16))) double s_row[CW];    // raw pivot-row entries of the strip (owner wave only)
    __shared__ __attribute__((aligned(16))) double s_prow[2][CW + 2];   // normalised, by pivot parity; [CW] = rhs[cr] / piv
    __shared__ long long s_win[8];                               // the exchange's winner, from wave 0 to the others
    __shared__ long long s_leave[2];                             // logical column that leaves the basis (by pivot parity)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, xq = b >> 3;
    const int64_t lpi = (int64_t)(xq / a.G) * 8 + (b & 7);
    const int wg = xq % a.G;
    if (lpi >= t.n_lps) return;
    t = lp_slice_at(t, lpi);
    Ctl *ctl = t.ctl;
    const Ctl c0 = *ctl;
    if (c0.status != kRunning) return;
    const int64_t m = t.rows - 1, nnb = t.cols - 1, ld = t.ld;
    const int64_t col0 = (int64_t)wg * CW;
    const int ncl = (int)((nnb - col0) < CW ? (nnb - col0) : CW);   // valid local columns (>= 1)
    unsigned long long *xb = a.xbuf + lpi * a.xs_lp;
    unsigned long long *lostflag = xb + (int64_t)a.G * 2 * a.xs_slot;   // one word behind the slots
    const bool leader = wg == 0 && tid == 0;

    // ---- load the strip: thread t owns rows t, t + 256, ... of all CW columns
    v16d    x[TR][CH];
    double  rhs[TR];
    int64_t bas[TR];
    bool    valid[TR];
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        const int64_t r = tid + (int64_t)kResThreads * k;
        valid[k] = r < m;
        const double *row = t.M + r * ld + col0;
#pragma unroll
        for (int c = 0; c < CW; c += 2) {
            double2 v = make_double2(0.0, 0.0);
            if (valid[k] && c < ncl) v = *reinterpret_cast<const double2 *>(row + c);
            x[k][c >> 4][c & 15] = v.x;
            x[k][c >> 4][(c & 15) + 1] = (c + 1 < ncl) ? v.y : 0.0;
        }
        rhs[k] = valid[k] ? t.M[r * ld + nnb] : 0.0;
        bas[k] = valid[k] ? t.basis[r] : -1;
    }
    // objective entry / logical column of local column `lane`: replicated in every wave (all four
    // price and update them identically, so no wave waits for another to know the local best column)
    double  obj = 0.0;
    int     lidx = -1;
    if (lane < ncl) { obj = t.M[m * ld + col0 + lane]; lidx = (int)t.p2l[col0 + lane]; }
    double  objv = t.M[m * ld + nnb];
    int64_t n_piv = c0.n_pivots, tn = c0.trace_n, last_ec = c0.ec, last_cr = c0.cr;
    int32_t status = kRunning;
    bool lost = false;
    // all workgroups of the LP on one XCD (verified by the first exchange, whose records carry the
    // XCC ids): later stores may stay in the shared L2, where the polls find them sooner
    bool local = false;
    int it = 0;
#ifdef MI355X_RES_TIMING
    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RES_T(i) T##i = wall_clock64()
#else
#define RES_T(i)
#endif

    // The loop is software-pipelined around the exchange: as soon as pivot k's normalised row is known,
    // the objective entries are brought up to date, priced, and the workgroup's best column for pivot
    // k + 1 is computed AS THE UPDATE WILL LEAVE IT and published -- the rest of the strip is updated
    // while those records travel.
    int lcb = 0;                                                 // this workgroup's best local column (uniform)
    Cand cc; cc.v = 0.0; cc.i = -1;                              // ... its (key, logical column)
    bool pnan0 = false;
    auto price_local = [&]() {
        // price_cand's rules with lane == local column: a NaN entry is no candidate, except in
        // logical column 0, where it is the unbeatable one (find-entering-column never replaces it)
        cc.v = obj * a.sgn;
        cc.i = lidx;                                              // (-1 in lanes without a column)
        int n0 = 0;
        if (__any(cc.i >= 0 && cc.v != cc.v)) {                   // (rare)
            if (cc.v != cc.v) {
                if (cc.i == 0) { cc.v = -__builtin_inf(); n0 = 1; }
                else cc.i = -1;
            }
        }
        int psrc;
        cc = wave_argmin(cc, psrc);                               // psrc: the lane that holds the winner = its local column
        pnan0 = cc.i >= 0 && lane_pick(n0, psrc) != 0;
        lcb = __builtin_amdgcn_readfirstlane((cc.i < 0 || pnan0) ? 0 : psrc);
    };
    // record + column of epoch `ep`: v[] = my rows' entries of local column lcb, vobj = its objective entry
    auto publish = [&](unsigned ep, const double (&v)[TR], double vobj, bool mute) {
        unsigned long long *slot = xb + ((int64_t)wg * 2 + (ep & 1u)) * a.xs_slot;
        unsigned long long *cg = slot + 8;
        const unsigned long long tg = (unsigned long long)ep << 32;
        unsigned val = 0u;
        if (wave == 0) {
            const unsigned long long vb = dbits(cc.v);
            const unsigned iw = (cc.i < 0 ? kEmptyIdx : (unsigned)cc.i) | (pnan0 ? 0x80000000u : 0u);
            const unsigned word[4] = { (unsigned)vb, (unsigned)(vb >> 32), iw, (unsigned)lcb | (xcc_id() << 8) };
            val = word[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) val = lane == k ? word[k] : val;
        }
        const bool rec = wave == 0 && lane < 4 && !mute;
        // (ONE branch on `local`, not one per store: as st_x calls this was thirty branches per pivot)
        auto stores = [&](auto loc) {
            constexpr bool L = decltype(loc)::value;
            if (rec) st_x(&slot[lane], tg | val, L);
#pragma unroll
            for (int k = 0; k < TR; ++k)
                if (valid[k]) {
                    const unsigned long long vb = dbits(v[k]);
                    st_pair<L>(&cg[2 * (tid + kResThreads * k)], tg | (vb & 0xffffffffull), tg | (vb >> 32));
                }
            if (tid == lcb) {
                const unsigned long long vb = dbits(vobj);
                st_pair<L>(&cg[2 * m], tg | (vb & 0xffffffffull), tg | (vb >> 32));
            }
        };
        if (local) stores(std::true_type()); else stores(std::false_type());
    };
    // my rows' entries of local column c (uniform): a macro, not a lambda -- a closure that indexes x
    // dynamically makes the compiler keep the whole strip in scratch memory
#define RES_COLUMN_OF(c, v)                                                                     \
    do {                                                                                        \
        const int lh_ = (c) >> 4, lj_ = (c) & 15;                                               \
        _Pragma("unroll") for (int k = 0; k < TR; ++k) {                                        \
            (v)[k] = x[k][0][lj_];                                                              \
            _Pragma("unroll") for (int h = 1; h < CH; ++h)                                      \
                if (h == lh_) (v)[k] = x[k][h][lj_];              /* (uniform) */               \
        }                                                                                       \
    } while (0)
    double vnext[TR];                                             // my rows' entries of column lcb, as published
    // find-pivoting-row on this workgroup's OWN candidate column, ahead of the exchange: should the
    // candidate win, its pivot row is what everybody needs next -- so it travels as a second part
    // of the record (granule 4, published as soon as it is known, while the records travel) instead
    // of being worked out by every workgroup after the column has arrived.  Same data (the column
    // as published, the RHS copy that every workgroup holds identically), same result.
    // 30 low bits: the pivot row; bits 30..31: 0 = a pivot row, 1 = no eligible row (MI_UNBOUNDED),
    // 2 = inf / NaN in the column or a NaN quotient (kNeedDense)
    unsigned spec = 0u;
    // (a macro, as RES_COLUMN_OF: see there)
#define RES_RATIO_AHEAD(ep_, par_, mute_)                                                                   \
    do {                                                                                                    \
        const double vobj_ = lane_value_dyn(obj, lcb);           /* (every wave holds the objective entries) */ \
        Cand rbest_; rbest_.v = 0.0; rbest_.i = -1;                                                         \
        unsigned flags_ = 0u;                                                                               \
        if (!(fabs(vobj_) <= 1.7976931348623157e308)) flags_ = 1u;                                          \
        _Pragma("unroll") for (int k = 0; k < TR; ++k)                                                      \
            if (valid[k]) {                                                                                 \
                const double av_ = vnext[k];                                                                \
                if (!(fabs(av_) <= 1.7976931348623157e308)) flags_ = 1u;                                    \
                if (a.ratio_thr < av_) {                                                                    \
                    const double qv_ = rhs[k] / av_;                                                        \
                    if (qv_ != qv_) flags_ = 1u;                 /* a NaN quotient: decided on the dense path (kNeedDense) */ \
                    else {                                                                                  \
                        Cand c_; c_.v = qv_; c_.i = tid + kResThreads * k;                                  \
                        rbest_ = cand_min(rbest_, c_);                                                      \
                    }                                                                                       \
                }                                                                                           \
            }                                                                                               \
        {                                                                                                   \
            int src_;                                                                                       \
            const Cand w_ = wave_argmin(rbest_, src_);                                                      \
            const unsigned wf_ = __any(flags_ != 0u) ? 1u : 0u;                                             \
            if (lane == 0) { s_wv[par_][wave] = w_.v; s_wi[par_][wave] = w_.i; s_wf[par_][wave] = wf_; }    \
        }                                                                                                   \
        __syncthreads();                                         /* barrier B */                            \
        Cand q_; q_.v = s_wv[par_][0]; q_.i = s_wi[par_][0];                                                \
        unsigned allf_ = s_wf[par_][0];                                                                     \
        _Pragma("unroll") for (int w = 1; w < NW; ++w) {                                                    \
            Cand y_; y_.v = s_wv[par_][w]; y_.i = s_wi[par_][w];                                            \
            q_ = cand_min(q_, y_);                                                                          \
            allf_ |= s_wf[par_][w];                                                                         \
        }                                                                                                   \
        spec = allf_ ? (2u << 30) : (q_.i < 0 ? (1u << 30) : (unsigned)q_.i);                               \
        if (a.G > 1 && tid == 0 && !(mute_)) {                                                              \
            unsigned long long *slot_ = xb + ((int64_t)wg * 2 + ((ep_) & 1u)) * a.xs_slot;                  \
            st_x(&slot_[4], ((unsigned long long)(ep_) << 32) | spec, local);                               \
        }                                                                                                   \
    } while (0)
    price_local();
    RES_COLUMN_OF(lcb, vnext);
#ifdef MI355X_TEST_HOOKS
    const bool fault_mute = a.fault > 0 && wg == a.G - 1;       // fault injection: the test build only
#else
    constexpr bool fault_mute = false;
#endif
    if (a.G > 1) publish(a.epoch_base + 1u, vnext, obj, fault_mute);
    RES_RATIO_AHEAD(a.epoch_base + 1u, 0, fault_mute);

    // The strip update of pivot k is the first thing iteration k + 1 does -- AFTER it has asked for
    // the records of pivot k + 1 and before it looks at what came back: the update (pure register /
    // LDS work) runs while the polls travel.  What it needs of pivot k stays in these variables; the
    // normalised row stays in LDS (two buffers, by pivot parity: the owner wave of the next pivot row
    // writes the other one).
    double col[TR];                                               // my rows' entries of the entering column
    bool   is_cr[TR];
    bool   mine = false;                                          // the entering column of the pending update is in MY strip
    int    lc = 0;                                                // ... there
#pragma unroll
    for (int k = 0; k < TR; ++k) { col[k] = 0.0; is_cr[k] = false; }

#pragma unroll 1
    for (;; ++it) {
#ifdef MI355X_RES_TIMING
        unsigned long long T0 = 0, T1 = 0, T2 = 0, T3 = 0, T4 = 0, T5 = 0, T6 = 0, T7 = 0, T8 = 0, T9 = 0, T10 = 0;
#endif
        RES_T(0);
        const unsigned epoch = a.epoch_base + (unsigned)it + 1u;
        const bool go = it < a.cap;
        // ---- ask for everybody's records of this pivot.  Few workgroups per LP: EVERY wave polls the
        // (small) records itself -- no LDS hop, no workgroup barrier between the exchange and the
        // column read (batch of 512 x 256 LPs, 8 workgroups each: 9.2 -> 9.7 M pivots/s); many: wave 0
        // polls and hands the winner over through LDS (config 2, 32 workgroups: 240 k pivots/s against
        // 232 k with four times the poll traffic)
        constexpr bool every_wave = EVERY;                        // (the launcher: G <= 8)
        const bool poller = a.G > 1 && go && (every_wave || wave == 0);
        const bool have = lane < a.G;
        const unsigned long long *rp = xb + ((int64_t)(have ? lane : 0) * 2 + (epoch & 1u)) * a.xs_slot;
        unsigned long long g[4] = {0ull, 0ull, 0ull, 0ull};
        if (poller) {
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = ld_l2(&rp[k]);
        }
        // ---- rank-1 update of the strip with the PREVIOUS pivot (n-pivot-row), while those loads travel
        if (it > 0) {
            const double *pw = s_prow[(it - 1) & 1];
            if (mine) {
                const int lh = lc >> 4, lj = lc & 15;
#pragma unroll
                for (int k = 0; k < TR; ++k) {
                    const double unit = is_cr[k] ? 1.0 : 0.0;
#pragma unroll
                    for (int h = 0; h < CH; ++h)
                        if (h == lh) x[k][h][lj] = unit;          // (uniform)
                }
            }
#pragma unroll
            for (int h = 0; h < CH; ++h) {
#pragma unroll
                for (int j0 = 0; j0 < 16; j0 += 8) {
                    double p8[8];
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const double2 pp = *reinterpret_cast<const double2 *>(&pw[h * 16 + j0 + j]);
                        p8[j] = pp.x; p8[j + 1] = pp.y;
                    }
#pragma unroll
                    for (int k = 0; k < TR; ++k) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const double prod = col[k] * p8[j];   // rounded product
                            x[k][h][j0 + j] = x[k][h][j0 + j] - prod;   // rounded difference
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < TR; ++k)
                if (is_cr[k]) {                                   // ONE thread of the workgroup: the pivot row itself
#pragma unroll
                    for (int c = 0; c < CW; c += 2) {
                        const double2 pp = *reinterpret_cast<const double2 *>(&pw[c]);
                        x[k][c >> 4][c & 15] = pp.x;
                        x[k][c >> 4][(c & 15) + 1] = pp.y;
                        if ((c & 7) == 6) __builtin_amdgcn_sched_barrier(0);   // (a few loads in flight, not CW / 2: registers)
                    }
                }
        }
        if (!go) break;                                           // (the pivot cap of this launch)
        RES_T(1);
        ValIdx e;
        int owner = wg;
        lc = lcb;
        if (a.G > 1) {
            // ---- everybody's records -> the same winner everywhere
            long long win[7];
            if (poller) {
                unsigned spins = 0;
                const unsigned limit = it == 0 ? a.spins_first : a.spins;
                bool fine = true;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ok &= (unsigned)(g[k] >> 32) == epoch;
                    if (__all(ok | !have)) break;
                    if (++spins > limit) { fine = false; break; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[k] = ld_l2(&rp[k]);
                }
                Cand rc; rc.v = 0.0; rc.i = -1;
                const unsigned iw = (unsigned)g[2];
                if (have && fine && (iw & kEmptyIdx) != kEmptyIdx) { rc.v = join_bits(g[0], g[1]); rc.i = (int)(iw & kEmptyIdx); }
                int src;
                const Cand w = wave_argmin(rc, src);
                const int own = src < 0 ? 0 : src;
                const unsigned wiw = (unsigned)lane_pick((int)iw, own);
                const int wlc = lane_pick((int)((unsigned)g[3] & 0xffu), own);
                const unsigned x0 = (unsigned)lane_pick((int)((unsigned)g[3] >> 8), 0);
                const bool same = __all(!have || ((unsigned)g[3] >> 8) == x0);
                win[0] = (long long)dbits(w.v); win[1] = w.i; win[2] = own; win[3] = wlc;
                win[4] = (w.i >= 0 && (wiw >> 31)) ? 1 : 0; win[5] = fine ? 0 : 1;
                win[6] = (fine && same) ? 1 : 0;
                if (!every_wave && lane == 0) {
#pragma unroll
                    for (int k = 0; k < 7; ++k) s_win[k] = win[k];
                }
            }
            if (!every_wave) {
                __syncthreads();                                  // barrier A
#pragma unroll
                for (int k = 0; k < 7; ++k) win[k] = s_win[k];
            }
            e.v = __longlong_as_double(win[0]);
            e.i = win[1];
            owner = __builtin_amdgcn_readfirstlane((int)win[2]);
            lc = __builtin_amdgcn_readfirstlane((int)win[3]);
            e.s = win[4] ? kNanColumn0 : lc;
            if (win[5]) { lost = true; break; }
            if (it == 0) local = win[6] != 0;                     // same records, same decision everywhere
        } else {
            e.v = cc.v; e.i = cc.i; e.s = pnan0 ? kNanColumn0 : lcb;
        }
        RES_T(2);
        if (price_says_optimal(e, a.price_tol)) { status = 0; break; }                    // MI_OPTIMAL
        if (c0.max_pivots > 0 && n_piv >= c0.max_pivots) { status = 3; break; }          // MI_MAX_PIVOTS
        const int64_t ec = e.i;
        // ---- the entering column (my rows' entries and the objective row's) and what its owner found
        // on it: the pivot row
        double colm;
        unsigned p2 = spec;
        if (a.G > 1) {
            const unsigned long long *wrec = xb + ((int64_t)owner * 2 + (epoch & 1u)) * a.xs_slot;
            const unsigned long long *wsl = wrec + 8;
            v4u gp[TR], gm;
            unsigned long long g2;
            const unsigned long long *ap[TR];
#pragma unroll
            for (int k = 0; k < TR; ++k) ap[k] = &wsl[2 * (valid[k] ? tid + (int64_t)kResThreads * k : m)];
            // No bound on this wait.  Every workgroup of the LP has published its record of this epoch
            // (above), so all of them are running, and between that record and these granules their
            // owner waits for nobody: they WILL arrive, and a workgroup that the GPU takes off the CU
            // for a while in between is waited for like at any barrier.  (A bound here -- a second way
            // out of this loop next to the exit on the owner's result below -- also made the compiler
            // move the whole strip between registers in every iteration.)
            for (;;) {
                PairLoads<TR>::run(gp, gm, g2, ap, &wsl[2 * m], &wrec[4]);
                bool ok = gm.y == epoch && gm.w == epoch && (unsigned)(g2 >> 32) == epoch;
#pragma unroll
                for (int k = 0; k < TR; ++k) ok &= gp[k].y == epoch && gp[k].w == epoch;
                if (ok) break;
            }
#pragma unroll
            for (int k = 0; k < TR; ++k) col[k] = valid[k] ? join_bits(pair_lo(gp[k]), pair_hi(gp[k])) : 0.0;
            colm = join_bits(pair_lo(gm), pair_hi(gm));
            p2 = (unsigned)g2;
        } else {
#pragma unroll
            for (int k = 0; k < TR; ++k) col[k] = valid[k] ? vnext[k] : 0.0;
            colm = lane_value_dyn(obj, lc);                       // (every wave holds the objective entries)
        }
        const unsigned code = (unsigned)__builtin_amdgcn_readfirstlane((int)(p2 >> 30));
        if (code != 0u) { status = code == 2u ? kNeedDense : 1; break; }   // (1: MI_UNBOUNDED)
        const int cr = __builtin_amdgcn_readfirstlane((int)(p2 & 0x3fffffffu));
        const int otid = cr & (kResThreads - 1), okk = cr >> 8;
        mine = owner == wg;
        double *pw = s_prow[it & 1];
        RES_T(3);
        // ---- the pivot row's entries of this strip, normalised -- inside the wave that owns row cr
        if (wave == (otid >> 6)) {
            double pivl = 0.0, rhsl = 0.0;
            if (tid == otid) {
#pragma unroll
                for (int k = 0; k < TR; ++k)
                    if (k == okk) {
#pragma unroll
                        for (int c = 0; c < CW; c += 2)
                            *reinterpret_cast<double2 *>(&s_row[c]) = make_double2(x[k][c >> 4][c & 15], x[k][c >> 4][(c & 15) + 1]);
                        pivl = col[k];
                        rhsl = rhs[k];
                        s_leave[it & 1] = bas[k];
                        bas[k] = ec;                              // src/simplex.lisp:358
                    }
            }
            const double piv = lane_value_dyn(pivl, otid & 63);   // == M[cr][ec] bit for bit
            const double rhsc = lane_value_dyn(rhsl, otid & 63);
            __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the owner lane's LDS stores (same wave: in order)
            __builtin_amdgcn_wave_barrier();
            if (lane < CW) {
                double rv = s_row[lane];
                if (mine && lane == lc) rv = 1.0;                 // the slot takes over the leaving column e_cr
                pw[lane] = (lane < ncl) ? rv / piv : 0.0;
            }
            if (lane == 0) pw[CW] = rhsc / piv;
        }
        __syncthreads();                                          // barrier C
        const int64_t leaving = s_leave[it & 1];
        const double prhs = pw[CW];
        RES_T(4);
        // ---- objective entries through the pivot (every wave: its own copy), priced at once; the
        // workgroup's best column for the NEXT pivot as the update will leave it -> published
        if (lane < CW) {
            const double pr = pw[lane];
            if (mine && lane == lc) { obj = 0.0; lidx = (int)leaving; }
            const double prod = colm * pr;
            obj = obj - prod;
        }
#pragma unroll
        for (int k = 0; k < TR; ++k) is_cr[k] = valid[k] && tid + kResThreads * k == cr;
        RES_T(8);
        price_local();
        RES_T(9);
        {
            RES_COLUMN_OF(lcb, vnext);
            const double pl = pw[lcb];
            const bool slot_col = mine && lcb == lc;
#pragma unroll
            for (int k = 0; k < TR; ++k) {
                const double base = slot_col ? (is_cr[k] ? 1.0 : 0.0) : vnext[k];
                const double prod = col[k] * pl;
                const double d = base - prod;
                vnext[k] = is_cr[k] ? pl : d;
            }
        }
        const bool more = it + 1 < a.cap;
        RES_T(10);
        if (a.G > 1 && more) publish(epoch + 1u, vnext, obj, false);
        RES_T(5);
        // ---- the RHS copy and the objective value through the pivot; then the ratio test on the
        // candidate just published, and its result behind it
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            const double prod = col[k] * prhs;
            const double d = rhs[k] - prod;
            rhs[k] = is_cr[k] ? prhs : d;
        }
        {
            const double prod = colm * prhs;
            objv = objv - prod;
        }
        if (more) RES_RATIO_AHEAD(epoch + 1u, (it + 1) & 1, false);
        RES_T(6);
        // ---- bookkeeping (the logical -> slot map is rebuilt from lidx / bas at write-back: nothing but
        // the trace is written to HBM inside the loop)
        (void)leaving;
        if (leader && t.trace_ec && tn < t.trace_cap) { t.trace_ec[tn] = ec; t.trace_cr[tn] = cr; }
        last_ec = ec; last_cr = cr;
        n_piv += 1; tn += 1;
#ifdef MI355X_RES_TIMING
        if (a.G > 1) {
            RES_T(7);
            tacc[0] += 1; tacc[1] += T1 - T0; tacc[2] += T2 - T1; tacc[3] += T3 - T2; tacc[4] += T4 - T3;
            tacc[5] += T5 - T4; tacc[6] += T6 - T5; tacc[7] += T7 - T6;
            tacc[8] += T8 - T4; tacc[9] += T9 - T8; tacc[10] += T10 - T9; tacc[11] += T5 - T10;
            tacc[12] += mine ? 1 : 0;
        }
#endif
    }
#ifdef MI355X_RES_TIMING
    // (every workgroup's first thread: 16 doubles per workgroup)
    if (tid == 0 && t.rhs && 16 * (wg + 1) <= t.rows) for (int k = 0; k < 16; ++k) t.rhs[16 * wg + k] += (double)tacc[k];
#endif

    if (lost) {
        // nothing is written back: the tableau is what it was when the launch started
        if (tid == 0) {
            bool first = it == 0;
            if (first) st_wt(lostflag, (unsigned long long)a.epoch_base);
            else first = ld_l2(lostflag) == (unsigned long long)a.epoch_base;    // somebody never got past the first exchange
            st_wt(&ctl->status, first ? kSyncLost : kResidentStuck);
        }
        return;
    }
    // ---- write back
#pragma unroll
    for (int k = 0; k < TR; ++k)
        if (valid[k]) {
            const int64_t r = tid + (int64_t)kResThreads * k;
            double *row = t.M + r * ld + col0;
#pragma unroll
            for (int c = 0; c < CW; c += 2) {
                const double a0 = x[k][c >> 4][c & 15], a1 = x[k][c >> 4][(c & 15) + 1];
                if (c + 1 < ncl)  *reinterpret_cast<double2 *>(row + c) = make_double2(a0, a1);
                else if (c < ncl) row[c] = a0;
            }
            if (wg == 0) { t.M[r * ld + nnb] = rhs[k]; t.basis[r] = bas[k]; t.l2p[bas[k]] = -1; }
        }
    if (tid < ncl) {
        t.M[m * ld + col0 + tid] = obj;
        t.p2l[col0 + tid] = (int64_t)lidx;
        if (lidx >= 0) t.l2p[lidx] = col0 + tid;                  // every logical column is basic or in exactly one strip
    }
    if (leader) {
        t.M[m * ld + nnb] = objv;
        ctl->ec = last_ec; ctl->cr = last_cr;
        ctl->n_pivots = n_piv;
        ctl->trace_n = tn;
        if (status != kRunning) ctl->status = status;
    }
}

// ------------------------------------------------------------------ compact representation
// Basic columns of a consistent tableau are unit vectors and stay bit-for-bit unchanged under
// every pivot (x - s*(+0) == x, and a column that becomes basic is produced as x - x = +0 /
// rs/rs = 1 exactly), so only the var_count - m NON-basic columns and the RHS column carry
// information.  The solve loop therefore runs on P = [non-basic columns | RHS]
// (rows x (var_count - m + 1)): a pivot snapshots the entering column, overwrites its slot with
// e_cr (= the pre-pivot content of the leaving basic column, which takes the slot over) and runs
// the SAME rank-1 update on P.  One third less traffic at n = 2m, identical results.  The
// dense logical tableau is rebuilt by k_expand whenever an entry point needs it.

// flag[0] |= 1 unless column basis[i] is exactly e_i (bit patterns: +0.0 and 1.0) for every i,
// including a +0.0 in the objective row.  One workgroup per row.
__global__ __launch_bounds__(256) void k_verify_basis(TabView t, int *flag)
{
    t = lp_slice(t);
    const int64_t m = t.rows - 1;
    const unsigned long long one = 0x3FF0000000000000ull;
    for (int64_t r = blockIdx.x; r < t.rows; r += gridDim.x) {
        bool bad = false;
        for (int64_t i = threadIdx.x; i < m; i += blockDim.x) {
            const unsigned long long bits =
                (unsigned long long)__double_as_longlong(t.M[r * t.ld + t.basis[i]]);
            bad |= bits != ((r == i) ? one : 0ull);
        }
        if (bad) atomicOr(flag, 1);
    }
}

// P[r][j] = M[r][p2l[j]] (j < n_nb), P[r][n_nb] = M[r][vc]; padding zero.
__global__ __launch_bounds__(256) void k_compact(TabView d, TabView c)
{
    d = lp_slice(d);
    c = lp_slice(c);
    const int64_t n_nb = c.cols - 1, vc = d.cols - 1;
    for (int64_t r = blockIdx.y; r < d.rows; r += gridDim.y)
        for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < c.ld;
             j += (int64_t)gridDim.x * blockDim.x) {
            double v = 0.0;
            if (j < n_nb)       v = d.M[r * d.ld + c.p2l[j]];
            else if (j == n_nb) v = d.M[r * d.ld + vc];
            c.M[r * c.ld + j] = v;
        }
}

// brow[col] = row in which logical column `col` is basic, -1 otherwise.
__global__ __launch_bounds__(256) void k_basis_rows(TabView d, int64_t *brow, int phase)
{
    d = lp_slice(d);
    brow += (int64_t)blockIdx.z * (d.cols - 1);
    const int64_t vc = d.cols - 1, m = d.rows - 1;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (phase == 0) { if (i < vc) brow[i] = -1; }
    else            { if (i < m && d.basis[i] >= 0 && d.basis[i] < vc) brow[d.basis[i]] = i; }
}

// The inverse: rebuild the dense logical tableau from P, the maps and the basis.
__global__ __launch_bounds__(256) void k_expand(TabView d, TabView c, const int64_t *brow)
{
    d = lp_slice(d);
    c = lp_slice(c);
    brow += (int64_t)blockIdx.z * (d.cols - 1);
    const int64_t n_nb = c.cols - 1, vc = d.cols - 1;
    for (int64_t r = blockIdx.y; r < d.rows; r += gridDim.y)
        for (int64_t col = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; col < d.ld;
             col += (int64_t)gridDim.x * blockDim.x) {
            double v = 0.0;
            if (col < vc) {
                const int64_t slot = c.l2p[col];
                v = slot >= 0 ? c.M[r * c.ld + slot] : (brow[col] == r ? 1.0 : 0.0);
            } else if (col == vc) {
                v = c.M[r * c.ld + n_nb];
            }
            d.M[r * d.ld + col] = v;
        }
}

// ------------------------------------------------------------------ control block
__global__ void k_ctl_reset(Ctl *ctl, int64_t max_pivots, int reset_trace)
{
    ctl += blockIdx.x;                                         // one block per LP of a batch
    ctl->status     = kRunning;
    ctl->poison     = 0;
    ctl->ec         = -1;
    ctl->cr         = -1;
    ctl->n_pivots   = 0;
    ctl->max_pivots = max_pivots;
    if (reset_trace) ctl->trace_n = 0;
}

__global__ void k_ctl_resume(Ctl *ctl, int32_t from)
{
    ctl += blockIdx.x;
    if (ctl->status == from) { ctl->status = kRunning; ctl->poison = 0; }
}

// After the last enqueued iteration: a tableau that is still "running" has simply used up
// the pivots it was given.
__global__ void k_ctl_finish(Ctl *ctl)
{
    ctl += blockIdx.x;
    if (ctl->status == kRunning) ctl->status = 3;              // MI_MAX_PIVOTS
}

// ------------------------------------------------------------------ two-phase hand-over
// src/simplex.lisp:437-441: main[r][0..nv) = art[r][0..nv), main[r][nv] = art[r][nav], r < m.
__global__ __launch_bounds__(256) void k_handover_copy(TabView art, TabView mt)
{
    const int64_t nv = mt.cols - 1, nav = art.cols - 1, m = mt.rows - 1;
    for (int64_t r = blockIdx.y; r < m; r += gridDim.y)
        for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c <= nv;
             c += (int64_t)gridDim.x * blockDim.x)
            mt.M[r * mt.ld + c] = art.M[r * art.ld + (c < nv ? c : nav)];
}

// src/simplex.lisp:444-451: copy the basis and re-eliminate the objective row, one basic
// column after the other.  The row-i step reads scale = obj[basis[i]] AFTER steps 0..i-1,
// exactly as the sequential loop does, hence one workgroup and a barrier per step.
__global__ __launch_bounds__(kSelThreads) void k_handover_objective(TabView art, TabView mt)
{
    const int64_t m = mt.rows - 1, nv = mt.cols - 1;
    double *obj = mt.M + m * mt.ld;
    for (int64_t i = 0; i < m; ++i) {
        const int64_t bc = art.basis[i];
        if (threadIdx.x == 0) mt.basis[i] = bc;
        const double scale = obj[bc];
        __syncthreads();                                   // everyone has read scale
        if (scale != 0.0) {
            const double *row = mt.M + i * mt.ld;
            for (int64_t c = threadIdx.x; c <= nv; c += kSelThreads) {
                const double prod = scale * row[c];
                obj[c] = obj[c] - prod;
            }
        }
        __syncthreads();                                   // obj updated before the next scale
    }
}

// Column-parallel form of the same re-elimination, valid when the basic columns of the rows just
// copied are EXACT unit vectors (they are whenever phase 1 ran on the compact representation,
// which verifies that on entry and preserves it).  Then obj[basis[i]] reaches step i unchanged
// (every earlier step subtracts scale*(+0) from it), so all scales are known up front, and each
// column's chain  ((obj[c] - s0*row0[c]) - s1*row1[c]) - ...  -- same operations, same order as
// the sequential loop -- is independent of every other column: one thread per column, rows
// streamed coalesced across threads.  One pass over the tableau instead of m workgroup-serial
// steps (config-3 size: ~0.1 ms instead of ~16 ms).
__global__ __launch_bounds__(256) void k_handover_scales(TabView art, TabView mt)
{
    const int64_t m = mt.rows - 1;
    const double *obj = mt.M + m * mt.ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bc = art.basis[i];
        mt.basis[i] = bc;
        mt.col[i] = obj[bc];                                   // scale of step i
    }
}

__global__ __launch_bounds__(256) void k_handover_objective_columns(TabView mt)
{
    const int64_t m = mt.rows - 1, nv = mt.cols - 1;
    double *obj = mt.M + m * mt.ld;
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c > nv) return;
    double v = obj[c];
    for (int64_t i = 0; i < m; ++i) {
        const double scale = mt.col[i];                        // wave-uniform
        if (scale != 0.0) {
            const double prod = scale * mt.M[i * mt.ld + c];
            v = v - prod;
        }
    }
    obj[c] = v;
}

// The hand-over on a compact column shard (mi355x_colpart_solve_two_phase).  The artificial
// shard `art` holds non-basic columns of the artificial tableau in its slots -- columns of the main
// problem and artificial columns mixed, wherever the pivots of phase 1 left them.  The main shard
// `mt` takes over the slots keep[0 .. nk) (the main problem's columns; the host chose them from
// the slot map) and the RHS copy: rows < m are copied (src/simplex.lisp:437-441), and the objective
// row is the main tableau's own (obj0, gathered by the host: obj0[k] = objective coefficient of the
// logical column in keep[k], obj0[nk] = its constant) re-eliminated over the basic rows
// (:444-451) in the column-parallel form of k_handover_objective_columns -- the basic columns of a
// compact shard are exact unit vectors by construction, so scale_i = objective coefficient of
// basis[i] is known up front (scales[], gathered by the host) and every column's chain
// ((obj0 - s_0 row_0) - s_1 row_1) - ... runs on its own, same operations, same order.
__global__ __launch_bounds__(256) void k_shard_handover(TabView art, TabView mt, const int64_t *keep,
                                                        const double *obj0, const double *scales)
{
    const int64_t m = mt.rows - 1, nk = mt.cols - 1;
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k <= nk) {
        const int64_t src = k < nk ? keep[k] : art.cols - 1;
        double v = obj0[k];
        for (int64_t i = 0; i < m; ++i) {
            const double x = art.M[i * art.ld + src];
            mt.M[i * mt.ld + k] = x;
            const double scale = scales[i];                     // wave-uniform
            if (scale != 0.0) {
                const double prod = scale * x;
                v = v - prod;
            }
        }
        mt.M[m * mt.ld + k] = v;
    }
    for (int64_t i = k; i < m; i += (int64_t)gridDim.x * blockDim.x) mt.basis[i] = art.basis[i];
}

// ------------------------------------------------------------------ synthetic LP generator
// splitmix64 stream, element k of the stream = mix(seed + (k+1)*gamma); u = (z >> 11) * 2^-53.
// Stream layout: A row-major (n_cons x n_vars), then b (n_cons), then c (n_vars).
//   A[i][j] = 0.05 + u     b[i] = n_vars * (0.25 + 0.5 u)     c[j] = 0.5 + u
// Tableau: [A | I | b ; -c | 0 | 0]  (what build-tableau, src/simplex.lisp:214-283, produces
// for  max c'x, Ax <= b, x >= 0  with variable order x0..x(n-1)).
__device__ __forceinline__ double splitmix_u01(uint64_t seed, uint64_t k)
{
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

__global__ __launch_bounds__(256) void k_synth_fill(TabView t, int64_t n, int64_t m, uint64_t seed,
                                                    const uint64_t *seeds, int64_t col_begin,
                                                    int64_t col_end)
{
    if (seeds) seed = seeds[blockIdx.z];                       // batch: one seed per LP
    t = lp_slice(t);
    const int64_t lcols = t.cols;                              // (col_end - col_begin) + 1
    for (int64_t i = blockIdx.y; i < t.rows; i += gridDim.y) {
    for (int64_t jl = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; jl < t.ld;
         jl += (int64_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (jl < lcols) {
            const bool    rhs = (jl == lcols - 1);
            const int64_t j   = col_begin + jl;                // global column (non-RHS)
            if (i < m) {
                if (rhs)            v = (double)n * (0.25 + 0.5 * splitmix_u01(seed, (uint64_t)(n * m + i)));
                else if (j < n)     v = 0.05 + splitmix_u01(seed, (uint64_t)(i * n + j));
                else                v = (j - n == i) ? 1.0 : 0.0;
            } else {
                if (!rhs && j < n)  v = -(0.5 + splitmix_u01(seed, (uint64_t)(n * m + m + j)));
            }
        }
        t.M[i * t.ld + jl] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && i < m && t.basis) t.basis[i] = n + i;
    }
    (void)col_end;
}

// ------------------------------------------------------------------ host-side launchers
static inline double sgn_of(int is_max) { return is_max ? 1.0 : -1.0; }

void launch_select(const TabView &t, int is_max, double f, int n_part, hipStream_t s)
{
    // one workgroup per tableau; a narrow workgroup when there is little to do per phase (its
    // barriers and reductions are the cost): measured on config 2, 10.3 us with 1024 threads
    const dim3 grid(1, 1, (unsigned)t.n_lps);
    const double ptol = (f / 8.0) * kClEpsilon, rthr = 0.0 + (f / 2.0) * kClEpsilon;
    if (t.rows <= 2048 && t.ld <= 4096)
        hipLaunchKernelGGL(k_select<256>, grid, dim3(256), 0, s, t, sgn_of(is_max), ptol, rthr, n_part);
    else
        hipLaunchKernelGGL(k_select<kSelThreads>, grid, dim3(kSelThreads), 0, s, t, sgn_of(is_max),
                           ptol, rthr, n_part);
}
void launch_select_split(const TabView &t, int is_max, double f, int n_part, hipStream_t s)
{
    const int g1 = (int)((t.rows + kGatherThreads - 1) / kGatherThreads);
    const int g2 = (int)(((t.ld >> 1) + kScaleThreads - 1) / kScaleThreads);
    hipLaunchKernelGGL(k_select_gather, dim3(g1, 1, (unsigned)t.n_lps), dim3(kGatherThreads), 0, s, t,
                       sgn_of(is_max), (f / 8.0) * kClEpsilon, 0.0 + (f / 2.0) * kClEpsilon, n_part);
    hipLaunchKernelGGL(k_select_scale, dim3(g2, 1, (unsigned)t.n_lps), dim3(kScaleThreads), 0, s, t, g1,
                       0.0 + (f / 2.0) * kClEpsilon);
}
bool select_split_supported(const TabView &t)
{
    return (t.rows + kGatherThreads - 1) / kGatherThreads <= t.part_cap / 2;
}
void launch_price_only(const TabView &t, int is_max, double f, hipStream_t s)
{
    hipLaunchKernelGGL(k_price_only, dim3(1), dim3(kSelThreads), 0, s, t, sgn_of(is_max),
                       (f / 8.0) * kClEpsilon, (int64_t)0, (double *)nullptr, 0, P2pArgs());
}
void launch_ratio_only(const TabView &t, int64_t ec, double f, hipStream_t s)
{
    hipLaunchKernelGGL(k_ratio_only, dim3(1), dim3(kSelThreads), 0, s, t, ec,
                       0.0 + (f / 2.0) * kClEpsilon);
}
void launch_prepare_pivot(const TabView &t, int64_t ec, int64_t cr, hipStream_t s)
{
    hipLaunchKernelGGL(k_prepare_pivot, dim3(1), dim3(kSelThreads), 0, s, t, ec, cr);
}
void launch_shard_price(const TabView &t, int is_max, int64_t col_offset, double *out2, int n_part,
                        hipStream_t s, const P2pArgs &x)
{
    hipLaunchKernelGGL(k_price_only, dim3(1), dim3(kSelThreads), 0, s, t, sgn_of(is_max), 0.0,
                       col_offset, out2, n_part, x);
}
void launch_shard_contribute(const TabView &t, const double *gathered, int n_shards,
                             int64_t col_offset, double f, int64_t *bits_out, int64_t *ec_out,
                             hipStream_t s)
{
    int blocks = (int)((t.rows + kSelThreads - 1) / kSelThreads);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_shard_contribute, dim3(blocks), dim3(kSelThreads), 0, s, t, gathered,
                       n_shards, col_offset, (f / 8.0) * kClEpsilon, (long long *)bits_out, ec_out);
}
void launch_shard_prepare(const TabView &t, const double *col, const int64_t *ec_dev, double f,
                          hipStream_t s, int64_t forced_cr)
{
    hipLaunchKernelGGL(k_shard_prepare, dim3(1), dim3(kSelThreads), 0, s, t, col, ec_dev,
                       0.0 + (f / 2.0) * kClEpsilon, forced_cr);
}
void launch_shard_forced_contribute(const TabView &t, int64_t ec, int64_t col_offset, int64_t *bits_out,
                                    int64_t *ec_out, hipStream_t s)
{
    int blocks = (int)((t.rows + kSelThreads - 1) / kSelThreads);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_shard_forced_contribute, dim3(blocks), dim3(kSelThreads), 0, s, t, ec, col_offset,
                       (long long *)bits_out, ec_out);
}
void launch_shard_handover(const TabView &art, const TabView &mt, const int64_t *keep, const double *obj0,
                           const double *scales, hipStream_t s)
{
    hipLaunchKernelGGL(k_shard_handover, dim3((unsigned)((mt.cols + 255) / 256)), dim3(256), 0, s, art, mt,
                       keep, obj0, scales);
}
void launch_shard_la_contribute(const TabView &t, int j, const double *gathered, int n_shards,
                                int64_t col_offset, double f, int64_t *bits_out, int64_t *ec_out,
                                hipStream_t s, const P2pArgs &x)
{
    // 256-thread workgroups, one row per thread: the strided gather of the entering column (a 64-byte
    // sector per row) needs many workgroups' memory pipelines (33 x 1024 threads: 12.5 us at 32769 rows)
    int blocks = (int)((t.rows + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_shard_la_contribute, dim3(blocks), dim3(256), 0, s, t, j, gathered,
                       n_shards, col_offset, (f / 8.0) * kClEpsilon, (long long *)bits_out, ec_out, x);
}
static int g_shard_la_split = 0;                     // 0 by size, 1 always one workgroup, 2 always split
void set_shard_la_split(int mode) { g_shard_la_split = mode; }

template <int J>
static void launch_shard_la_scale_t(const TabView &t, int g2, int g1, const int64_t *ec_dev, int is_max, hipStream_t s)
{
    hipLaunchKernelGGL(k_shard_la_scale<J>, dim3(g2), dim3(kScaleThreads), 0, s, t, g1, ec_dev, sgn_of(is_max));
}

bool shard_la_split(const TabView &t)
{
    const int g1 = (int)((t.rows + kGatherThreads - 1) / kGatherThreads);
    const int g2 = (int)(((t.ld >> 1) + kScaleThreads - 1) / kScaleThreads);
    const bool fits = g1 <= t.part_cap / 2 && g2 * (kScaleThreads / 64) <= t.part_cap / 2;
    return fits && (g_shard_la_split == 2 || (g_shard_la_split == 0 && (t.rows > 4096 || t.ld > 8192)));
}

// (x.peers != nullptr is honoured by the split form only: the caller checks shard_la_split())
int launch_shard_la_prepare(const TabView &t, int j, const double *col, const int64_t *ec_dev, double f,
                            int is_max, hipStream_t s, const P2pArgs &x)
{
    // one workgroup for small shards (one launch, ~10 us), the split pair for large ones (rows or
    // column pairs in the tens of thousands: config 5 as one shard on one GPU 1 536 -> 2 287 pivots/s)
    const int g1 = (int)((t.rows + kGatherThreads - 1) / kGatherThreads);
    const int g2 = (int)(((t.ld >> 1) + kScaleThreads - 1) / kScaleThreads);
    const bool split = shard_la_split(t);
    if (!split) {
        hipLaunchKernelGGL(k_shard_la_prepare, dim3(1), dim3(kSelThreads), 0, s, t, j, col, ec_dev,
                           0.0 + (f / 2.0) * kClEpsilon, sgn_of(is_max));
        return kSelWaves;                            // pricing partials left for the next step
    }
    hipLaunchKernelGGL(k_shard_la_ratio, dim3(g1), dim3(kGatherThreads), 0, s, t, j, col, ec_dev,
                       0.0 + (f / 2.0) * kClEpsilon, x);
    switch (j) {
#define MI_SLA(J) case J: launch_shard_la_scale_t<J>(t, g2, g1, ec_dev, is_max, s); break;
        MI_SLA(0) MI_SLA(1) MI_SLA(2) MI_SLA(3) MI_SLA(4) MI_SLA(5) MI_SLA(6) MI_SLA(7)
        MI_SLA(8) MI_SLA(9) MI_SLA(10) MI_SLA(11) MI_SLA(12) MI_SLA(13) MI_SLA(14) MI_SLA(15)
#undef MI_SLA
    }
    return g2 * (kScaleThreads / 64);
}
// the look-ahead step of a large shard in exchange mode 2 as TWO launches (see k_shard_p2p_step);
// returns the pricing partials it leaves, 0 when this shard / state needs the separate launches
int launch_shard_p2p_step(const TabView &t, int j, int n_part, int n_shards, int64_t col_offset, double f,
                          int is_max, int64_t *ec_dev, hipStream_t s, const P2pArgs &x)
{
    if (!x.peers || n_part <= 0 || !shard_la_split(t)) return 0;
    int blocks = (int)((t.rows + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks > t.part_cap / 2) return 0;
    const int g2 = (int)(((t.ld >> 1) + kScaleThreads - 1) / kScaleThreads);
    hipLaunchKernelGGL(k_shard_p2p_step, dim3(blocks), dim3(256), 0, s, t, j, n_part, n_shards, col_offset,
                       (f / 8.0) * kClEpsilon, 0.0 + (f / 2.0) * kClEpsilon, ec_dev, x);
    switch (j) {
#define MI_SLA(J) case J: launch_shard_la_scale_t<J>(t, g2, blocks, ec_dev, is_max, s); break;
        MI_SLA(0) MI_SLA(1) MI_SLA(2) MI_SLA(3) MI_SLA(4) MI_SLA(5) MI_SLA(6) MI_SLA(7)
        MI_SLA(8) MI_SLA(9) MI_SLA(10) MI_SLA(11) MI_SLA(12) MI_SLA(13) MI_SLA(14) MI_SLA(15)
#undef MI_SLA
    }
    return g2 * (kScaleThreads / 64);
}
void launch_handover(const TabView &art, const TabView &mt, bool unit_basis, hipStream_t s)
{
    const int64_t m = mt.rows - 1;
    if (m > 0) {
        int bx = (int)((mt.cols + 255) / 256);
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(k_handover_copy, dim3(bx, (unsigned)(m < 32768 ? m : 32768)), dim3(256), 0, s, art, mt);
    }
    if (unit_basis && m > 0) {
        hipLaunchKernelGGL(k_handover_scales, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, art, mt);
        hipLaunchKernelGGL(k_handover_objective_columns, dim3((unsigned)((mt.cols + 255) / 256)), dim3(256),
                           0, s, mt);
    } else {
        hipLaunchKernelGGL(k_handover_objective, dim3(1), dim3(kSelThreads), 0, s, art, mt);
    }
}
// one launch solves the whole batch; returns false if an LP does not fit the LDS budget
static int g_batch_block = 0;                                  // 0 = default (16), 1 = per-pivot k_batch_solve
void set_batch_block(int k) { g_batch_block = k; }

template <int KB>
static bool launch_batch_block_t(const TabView &t, int is_max, double f, hipStream_t s, int split = 0)
{
    const int64_t rp = (t.rows + 1) & ~(int64_t)1, ldv = t.ld >> 1;
    const size_t bytes = (size_t)((int64_t)KB * (t.ld + rp) + t.ld + rp + t.ld + rp) * 8 + (size_t)(rp + ldv) * 4;
    if (bytes > 150 * 1024) return false;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch_block<KB, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch_block<KB, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipGetLastError();
        attr_set = true;
    }
    if (split)
        hipLaunchKernelGGL((k_batch_block<KB, true>), dim3(1, 1, (unsigned)t.n_lps), dim3(kBbThreads), bytes, s, t,
                           sgn_of(is_max), (f / 8.0) * kClEpsilon, 0.0 + (f / 2.0) * kClEpsilon);
    else
        hipLaunchKernelGGL((k_batch_block<KB, false>), dim3(1, 1, (unsigned)t.n_lps), dim3(kBbThreads), bytes, s, t,
                           sgn_of(is_max), (f / 8.0) * kClEpsilon, 0.0 + (f / 2.0) * kClEpsilon);
    return true;
}

bool launch_batch_solve(const TabView &t, int is_max, double f, hipStream_t s)
{
    // blocked (compact representation), largest block that fits the LDS.  Measured at 257 x 513,
    // steady state (tools/batch_blocks.py): 128 LPs 2.0 M pivots/s per-pivot, 3.36 M at 8, 3.44 M
    // at 16; 1024 LPs 2.5 M per-pivot, 7.3 M at 8, 7.9 M at 16
    int kb = g_batch_block;
    if (kb == 0) kb = 16;
    if (t.p2l && kb > 1 && t.rows >= 2 && (t.ld >> 1) >= 1) {
        if (kb >= 16 && launch_batch_block_t<16>(t, is_max, f, s)) return true;
        if (kb >= 8 && launch_batch_block_t<8>(t, is_max, f, s)) return true;
        if (launch_batch_block_t<4>(t, is_max, f, s)) return true;
    }
    const size_t lds = (size_t)(t.ld + t.rows) * sizeof(double);
    if (lds > 96 * 1024 || t.ld / 2 < 1) return false;
    hipLaunchKernelGGL(k_batch_solve, dim3(1, 1, (unsigned)t.n_lps), dim3(kLpThreads), lds, s, t,
                       sgn_of(is_max), (f / 8.0) * kClEpsilon, 0.0 + (f / 2.0) * kClEpsilon);
    return true;
}
static int g_sweep_tr = 0, g_sweep_nt = -1;                     // 0 / -1: by size
static int g_sweep_impl = 0;                                    // 0: k_sweep16 for full blocks, 1: k_sweep always
bool launch_batch_block_split(const TabView &t, int is_max, double f, hipStream_t s)
{
    if (!t.p2l || !t.blk || !t.bk_col || !t.bk_prow || t.n_lps < 2 || t.rows < 2) return false;
    if (!launch_batch_block_t<16>(t, is_max, f, s, /*split=*/1)) return false;
    // the sweep of every LP's pending pivots: k_sweep copes with any number of pending pivots per
    // LP (an LP that has finished has none), grid.z = LP
    constexpr int block = 256;
    const int64_t ldv = t.ld >> 1;
    int strips = (int)((ldv + block - 1) / block);
    int64_t sp = (ldv + strips - 1) / strips;
    sp = (sp + 7) / 8 * 8;
    if (sp > block) sp = block;
    strips = (int)((ldv + sp - 1) / sp);
    // measured (tools/batch_sweep_ab.py, 257 x 513 stored per LP): k_sweep with 16-row tiles 7.2 /
    // 11.4 M pivots/s at 128 / 1024 LPs (32 rows 6.8 / 11.3); k_sweep16, whose partial-block form is
    // a slow one and every LP ends on a partial block, 6.7 / 10.2
    int64_t tr = g_sweep_tr ? g_sweep_tr : 16;
    while (tr > 4 && ((t.rows + tr - 1) / tr) * strips * t.n_lps < 2048) tr /= 2;
    const dim3 grid((unsigned)strips, (unsigned)((t.rows + tr - 1) / tr), (unsigned)t.n_lps);
    hipLaunchKernelGGL((k_sweep<256, 16, false>), grid, dim3(256), 0, s, t, (int)tr, (int)sp, sgn_of(is_max), 0, 0u, 0);
    return true;
}
void launch_verify_basis(const TabView &t, int *flag, hipStream_t s)
{
    const unsigned g = (unsigned)(t.rows < 16384 ? t.rows : 16384);
    hipLaunchKernelGGL(k_verify_basis, dim3(g, 1, (unsigned)t.n_lps), dim3(256), 0, s, t, flag);
}
void launch_compact(const TabView &d, const TabView &c, hipStream_t s)
{
    int bx = (int)((c.ld + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_compact, dim3(bx, (unsigned)(d.rows < 32768 ? d.rows : 32768), (unsigned)d.n_lps),
                       dim3(256), 0, s, d, c);
}
void launch_expand(const TabView &d, const TabView &c, int64_t *brow, hipStream_t s)
{
    const int64_t n = (d.cols > d.rows ? d.cols : d.rows);
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_basis_rows, dim3(g, 1, (unsigned)d.n_lps), dim3(256), 0, s, d, brow, 0);
    hipLaunchKernelGGL(k_basis_rows, dim3(g, 1, (unsigned)d.n_lps), dim3(256), 0, s, d, brow, 1);
    int bx = (int)((d.ld + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_expand, dim3(bx, (unsigned)(d.rows < 32768 ? d.rows : 32768), (unsigned)d.n_lps),
                       dim3(256), 0, s, d, c, brow);
}
void launch_ctl_reset(const TabView &t, int64_t max_pivots, int reset_trace, hipStream_t s)
{
    hipLaunchKernelGGL(k_ctl_reset, dim3((unsigned)t.n_lps), dim3(1), 0, s, t.ctl, max_pivots, reset_trace);
}
void launch_ctl_resume(const TabView &t, hipStream_t s, int32_t from)
{
    hipLaunchKernelGGL(k_ctl_resume, dim3((unsigned)t.n_lps), dim3(1), 0, s, t.ctl, from);
}
void launch_ctl_finish(const TabView &t, hipStream_t s)
{
    hipLaunchKernelGGL(k_ctl_finish, dim3((unsigned)t.n_lps), dim3(1), 0, s, t.ctl);
}
void launch_synth_fill(const TabView &t, int64_t n, int64_t m, uint64_t seed,
                       const uint64_t *dev_seeds, int64_t cb, int64_t ce, hipStream_t s)
{
    int bx = (int)((t.ld + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_synth_fill, dim3(bx, (unsigned)(t.rows < 32768 ? t.rows : 32768), (unsigned)t.n_lps),
                       dim3(256), 0, s, t, n, m, seed, dev_seeds, cb, ce);
}

// ---- update-kernel variants (one is the default; the others exist for the tuning sweep)
struct UpdateVariant {
    const char *name;
    int block;            // threads per workgroup = max column pairs per strip
    int unroll;           // rows in flight per thread; rows per workgroup is a multiple of it
    int min_tr;           // never fewer rows per workgroup than this
    int rounds;           // aim for rounds * (resident workgroup slots) workgroups
    void (*launch)(const TabView &, dim3, int, int, double, int, int, hipStream_t);
};

template <int BLOCK, int U, bool NT>
static void launch_update_t(const TabView &t, dim3 grid, int tr, int strip_pairs, double sgn,
                            int price, int reverse, hipStream_t s)
{
    hipLaunchKernelGGL((k_update<BLOCK, U, NT>), grid, dim3(BLOCK), 0, s, t, tr, strip_pairs, sgn,
                       price, reverse);
}

#define MI_VARIANT(B, U, NT, MINTR, ROUNDS) \
    { "b" #B "_u" #U "_nt" #NT "_mintr" #MINTR "_x" #ROUNDS, (B), (U), (MINTR), (ROUNDS), &launch_update_t<B, U, NT> }

static const UpdateVariant kVariants[] = {
    MI_VARIANT(256, 4, true, 4, 0),     // 0: default -- 4-row tiles whatever the size; the
                                        //    launcher swaps in variant 1 (plain loads/stores)
                                        //    when the stored tableau fits the Infinity Cache
    MI_VARIANT(256, 4, false, 4, 0),
    MI_VARIANT(256, 4, true, 4, 16),
    MI_VARIANT(256, 4, true, 4, 8),
    MI_VARIANT(256, 2, true, 2, 0),
    MI_VARIANT(256, 2, false, 2, 0),
    MI_VARIANT(256, 8, true, 8, 0),
    MI_VARIANT(256, 8, false, 8, 0),
    MI_VARIANT(256, 8, true, 8, 4),
    MI_VARIANT(128, 4, false, 4, 0),
    MI_VARIANT(128, 8, true, 8, 0),
    MI_VARIANT(512, 4, true, 4, 0),
    MI_VARIANT(512, 4, false, 4, 0),
    MI_VARIANT(512, 2, false, 2, 0),
    MI_VARIANT(64, 4, false, 4, 0),
    MI_VARIANT(1024, 4, false, 4, 0),
    MI_VARIANT(256, 4, false, 8, 0),
    MI_VARIANT(256, 1, false, 1, 0),
};
static int g_variant = 0;
constexpr int kCUs = 256, kThreadsPerCU = 2048;

int         update_variant_count() { return (int)(sizeof(kVariants) / sizeof(kVariants[0])); }
const char *update_variant_name(int v) { return kVariants[v].name; }
void        set_update_variant(int v) { if (v >= 0 && v < update_variant_count()) g_variant = v; }
int         get_update_variant() { return g_variant; }
const char *update_kernel_symbol() { return "k_update"; }

// Non-temporal accesses pay off when the stored tableau is streamed from HBM every pivot
// (config 3 dense, 403 MB: 125 vs 145 us); when it (mostly) fits the 256 MiB Infinity Cache
// plain accesses are faster (config 3 compact, 269 MB: 82 vs 90 us).
constexpr double kNtThresholdBytes = 320.0 * 1024 * 1024;

static int effective_variant(const TabView &t)
{
    if (g_variant != 0) return g_variant;
    const double bytes = (double)t.rows * (double)t.ld * 8.0 * (double)t.n_lps;
    return bytes > kNtThresholdBytes ? 0 : 1;
}

UpdateShape update_shape(const TabView &t)
{
    const UpdateVariant &v = kVariants[effective_variant(t)];
    const int64_t ldv = t.ld >> 1;
    UpdateShape g;
    g.strips = (int)((ldv + v.block - 1) / v.block);
    // equal-width strips, a multiple of 8 pairs (128 bytes) wide
    int64_t sp = (ldv + g.strips - 1) / g.strips;
    sp = (sp + 7) / 8 * 8;
    if (sp > v.block) sp = v.block;
    g.strip_pairs = (int)sp;
    g.strips = (int)((ldv + sp - 1) / sp);
    // Rows per workgroup.  Measured on config 3: SMALL tiles win -- with 4-row
    // tiles dispatched x-fastest the resident workgroups cover one contiguous window of the
    // tableau that sweeps through memory once.  rounds == 0 selects that fixed small tile;
    // rounds > 0 is the older "rounds x resident slots" sizing kept for the tuning sweep.
    int64_t tr = v.min_tr;
    if (v.rounds > 0) {
        const int64_t slots = (int64_t)kCUs * (kThreadsPerCU / v.block) * v.rounds;
        int64_t by = slots / ((int64_t)g.strips * t.n_lps);
        if (by < 1) by = 1;
        tr = (t.rows + by - 1) / by;
        if (tr < v.min_tr) tr = v.min_tr;
    }
    tr = (tr + v.unroll - 1) / v.unroll * v.unroll;
    while ((t.rows + tr - 1) / tr > 65535) tr *= 2;            // grid.y limit
    g.tr = (int)tr;
    g.row_chunks = (int)((t.rows + tr - 1) / tr);
    g.waves_per_block = v.block / 64;
    g.n_partials = g.strips * g.waves_per_block;
    return g;
}

// measured: no gain (config 3 compact 82.9 vs 81.9 us, dense 131.7 vs 131.2 us) -- the Infinity
// Cache does not behave like an LRU over a slightly-too-large streamed working set.  Off.
static int g_alternate_sweep = 0;
void set_alternate_sweep(int on) { g_alternate_sweep = on ? 1 : 0; }

int launch_update(const TabView &t, double sgn, int price, hipStream_t s, int64_t launch_index)
{
    const UpdateVariant &v = kVariants[effective_variant(t)];
    const UpdateShape g = update_shape(t);
    if (price && g.n_partials > t.part_cap / 2) price = 0;
    v.launch(t, dim3((unsigned)g.strips, (unsigned)g.row_chunks, (unsigned)t.n_lps), g.tr,
             g.strip_pairs, sgn, price, (g_alternate_sweep && (launch_index & 1)) ? 1 : 0, s);
    return price ? g.n_partials : 0;
}

// ---- blocked pivoting launchers
bool block_supported(const TabView &t)
{
    return t.blk && t.bk_col && t.bk_prow && t.p2l && t.n_lps == 1 && select_split_supported(t) &&
           (int64_t)(((t.ld >> 1) + kScaleThreads - 1) / kScaleThreads) * (kScaleThreads / 64) <= t.part_cap / 2;
}

template <int J>
static int launch_lookahead_t(const TabView &t, int is_max, double f, int n_part, hipStream_t s)
{
    const int g1 = (int)((t.rows + kGatherThreads - 1) / kGatherThreads);
    const int g2 = (int)(((t.ld >> 1) + kScaleThreads - 1) / kScaleThreads);
    hipLaunchKernelGGL(k_la_gather<J>, dim3(g1), dim3(kGatherThreads), 0, s, t, sgn_of(is_max),
                       (f / 8.0) * kClEpsilon, 0.0 + (f / 2.0) * kClEpsilon, n_part);
    hipLaunchKernelGGL(k_la_scale<J>, dim3(g2), dim3(kScaleThreads), 0, s, t, g1, sgn_of(is_max));
    return g2 * (kScaleThreads / 64);
}

int launch_lookahead(const TabView &t, int j, int is_max, double f, int n_part, hipStream_t s)
{
    switch (j) {
#define MI_LA(J) case J: return launch_lookahead_t<J>(t, is_max, f, n_part, s);
        MI_LA(0) MI_LA(1) MI_LA(2) MI_LA(3) MI_LA(4) MI_LA(5) MI_LA(6) MI_LA(7)
        MI_LA(8) MI_LA(9) MI_LA(10) MI_LA(11) MI_LA(12) MI_LA(13) MI_LA(14) MI_LA(15)
#undef MI_LA
    }
    return 0;
}

bool la_block_supported(const TabView &t)
{
    if (!block_supported(t) || !t.la_px || !t.la_rx) return false;
    const int64_t need = t.rows > (t.ld >> 1) ? t.rows : (t.ld >> 1);
    return (need + kLaThreads - 1) / kLaThreads <= kMaxLaWorkgroups;
}

static int      g_la_one_xcd = 1, g_la_fault = 0;
static unsigned g_la_max_spins = 1u << 21;
void set_la_one_xcd(int on) { g_la_one_xcd = on ? 1 : 0; }
void set_la_max_spins(unsigned n) { g_la_max_spins = n ? n : (1u << 21); }
#ifdef MI355X_TEST_HOOKS
void set_la_fault(int step_plus_1) { g_la_fault = step_plus_1; }
#endif

int la_block_workgroups(const TabView &t)
{
    const int64_t need = t.rows > (t.ld >> 1) ? t.rows : (t.ld >> 1);
    return (int)((need + kLaThreads - 1) / kLaThreads);
}

void launch_la_rollback(const TabView &t, int la_nw, hipStream_t s)
{
    hipLaunchKernelGGL(k_la_rollback, dim3(1), dim3(1), 0, s, t, la_nw);
}

void launch_la_block(const TabView &t, int ksteps, int is_max, double f, unsigned epoch_base, hipStream_t s)
{
    const int nw = la_block_workgroups(t);
    // one-XCD mode: 8 x nw blocks, every eighth takes part (the kernel verifies where they run)
    const int one_xcd = g_la_one_xcd && nw > 1;
    hipLaunchKernelGGL(k_la_block<kMaxBlock>, dim3(one_xcd ? 8 * nw : nw), dim3(kLaThreads), 0, s, t, ksteps,
                       sgn_of(is_max), (f / 8.0) * kClEpsilon, 0.0 + (f / 2.0) * kClEpsilon, epoch_base,
                       g_la_max_spins, one_xcd, g_la_fault);
}


// ---- the resident solve
static int g_res_fault = 0, g_res_poll = 0;
#ifdef MI355X_TEST_HOOKS
void set_resident_fault(int on) { g_res_fault = on; }
#endif
void set_resident_poll(int mode) { g_res_poll = mode; }     // tuning: 0 / 2 every wave polls (default), 1 wave 0 polls

bool resident_plan(const TabView &c, ResidentPlan *p)
{
    if (!c.p2l || !c.l2p || c.rows < 2 || c.cols < 2) return false;
    const int64_t m = c.rows - 1, nnb = c.cols - 1;
    if (m > 1024) return false;
    const int TR = m <= 256 ? 1 : (m <= 512 ? 2 : 4);
    const int CW = 64 / TR;
    const int64_t G = (nnb + CW - 1) / CW;
    if (G > 32) return false;
    if (p) {
        p->TR = TR; p->CW = CW; p->G = (int)G;
        p->slot_granules = 8 + 2 * ((m + 1 + 7) / 8 * 8);
        p->lp_granules = G * 2 * p->slot_granules + 8;            // + the lost flag (padded)
    }
    return true;
}

size_t resident_xbuf_bytes(const TabView &c)
{
    ResidentPlan p;
    if (!resident_plan(c, &p)) return 0;
    return (size_t)c.n_lps * (size_t)p.lp_granules * sizeof(unsigned long long);
}

bool launch_resident(const TabView &c, unsigned long long *xbuf, int is_max, double f, int cap,
                     unsigned epoch_base, hipStream_t s)
{
    ResidentPlan p;
    if (!xbuf || cap < 1 || !resident_plan(c, &p)) return false;
    ResidentArgs a;
    a.sgn = sgn_of(is_max);
    a.price_tol = (f / 8.0) * kClEpsilon;
    a.ratio_thr = 0.0 + (f / 2.0) * kClEpsilon;
    a.xbuf = xbuf;
    a.xs_lp = p.lp_granules;
    a.xs_slot = p.slot_granules;
    a.G = p.G;
    a.cap = cap;
    a.epoch_base = epoch_base;
    a.spins_first = g_la_max_spins;
    a.spins = 1u << 27;
    a.fault = g_res_fault;
    const unsigned groups = (unsigned)((c.n_lps + 7) / 8);
    const dim3 grid(groups * 8u * (unsigned)p.G);
    // who polls the records: every wave for itself (no LDS hop, no workgroup barrier behind the
    // exchange) -- measured round 3, final loop: config 2 (32 workgroups) 257 k pivots/s against 247 k
    // with wave 0 polling for the workgroup, 128-LP batch 10.3 against 9.8 M
    const bool every = g_res_poll != 1;
#define MI_RES(TR_, CW_)                                                                                       \
    do {                                                                                                       \
        if (every) hipLaunchKernelGGL((k_resident<TR_, CW_, true>),  grid, dim3(kResThreads), 0, s, c, a);    \
        else       hipLaunchKernelGGL((k_resident<TR_, CW_, false>), grid, dim3(kResThreads), 0, s, c, a);    \
    } while (0)
    if (p.TR == 1)      MI_RES(1, 64);
    else if (p.TR == 2) MI_RES(2, 32);
    else                MI_RES(4, 16);
#undef MI_RES
    return true;
}

static int g_sweep_u = 4;                                       // rows per step of k_sweep16: 4, or 8 (measured
                                                                // slower: 199 VGPRs, 2 waves per SIMD, 125 vs 103 us)
void set_sweep_shape(int tr, int nt) { g_sweep_tr = tr >= 4 ? tr / 4 * 4 : 0; g_sweep_nt = nt; }
void set_sweep_impl(int impl) { g_sweep_impl = impl == 1 ? 1 : 0; if (impl == 4 || impl == 8) g_sweep_u = impl; }

template <int KMAX>
static void launch_sweep_t(const TabView &t, dim3 grid, int tr, int sp, double sgn, bool nt, unsigned stamp,
                           int la_nw, hipStream_t s)
{
    if (nt) hipLaunchKernelGGL((k_sweep<256, KMAX, true>),  grid, dim3(256), 0, s, t, tr, sp, sgn, 1, stamp, la_nw);
    else    hipLaunchKernelGGL((k_sweep<256, KMAX, false>), grid, dim3(256), 0, s, t, tr, sp, sgn, 1, stamp, la_nw);
}

// applies up to kmax pending pivots; returns the number of pricing partials it leaves
int launch_sweep(const TabView &t, int kmax, double sgn, hipStream_t s, unsigned stamp, int la_nw)
{
    constexpr int block = 256;
    const int64_t ldv = t.ld >> 1;
    int strips = (int)((ldv + block - 1) / block);
    int64_t sp = (ldv + strips - 1) / strips;
    sp = (sp + 7) / 8 * 8;
    if (sp > block) sp = block;
    strips = (int)((ldv + sp - 1) / sp);
    // rows per workgroup: 32 (config 3, measured: 8 rows 116 us, 16 rows 105, 32 rows 103, 64 rows
    // 114), fewer when the tableau would otherwise not fill the chip with workgroups
    int64_t tr = g_sweep_tr;
    if (tr == 0) {
        // (a short block leaves most of the registers free: more, smaller tiles in flight --
        // config 3, 4 pending pivots: 8 rows 87 us, 16 rows 89, 32 rows 97, 64 rows 95)
        tr = kmax <= 4 ? 8 : kmax <= 8 ? 16 : 32;
        while (tr > 4 && ((t.rows + tr - 1) / tr) * strips < 2048) tr /= 2;
    }
    while ((t.rows + tr - 1) / tr > 65535) tr *= 2;            // grid.y limit
    const dim3 grid((unsigned)strips, (unsigned)((t.rows + tr - 1) / tr));
    const double bytes = (double)t.rows * (double)t.ld * 8.0;
    const bool nt = g_sweep_nt < 0 ? bytes > kNtThresholdBytes : g_sweep_nt != 0;
    if (kmax == kSweepK && g_sweep_impl == 0) {
        // rows in flight per thread and step: 8 when the tile is a multiple of 8 rows (bk_rmask is
        // padded to a multiple of 16 rows, so the uint4 mask loads of the last tile stay inside)
        const bool u8 = g_sweep_u == 8 && tr % 8 == 0;
        if (u8) {
            if (nt) hipLaunchKernelGGL((k_sweep16<true, 8>),  grid, dim3(256), 0, s, t, (int)tr, (int)sp, sgn, 1, stamp, la_nw);
            else    hipLaunchKernelGGL((k_sweep16<false, 8>), grid, dim3(256), 0, s, t, (int)tr, (int)sp, sgn, 1, stamp, la_nw);
        } else {
            if (nt) hipLaunchKernelGGL((k_sweep16<true, 4>),  grid, dim3(256), 0, s, t, (int)tr, (int)sp, sgn, 1, stamp, la_nw);
            else    hipLaunchKernelGGL((k_sweep16<false, 4>), grid, dim3(256), 0, s, t, (int)tr, (int)sp, sgn, 1, stamp, la_nw);
        }
        return strips * (block / 64);
    }
    if (kmax <= 2)      launch_sweep_t<2>(t, grid, (int)tr, (int)sp, sgn, nt, stamp, la_nw, s);
    else if (kmax <= 4) launch_sweep_t<4>(t, grid, (int)tr, (int)sp, sgn, nt, stamp, la_nw, s);
    else if (kmax <= 8) launch_sweep_t<8>(t, grid, (int)tr, (int)sp, sgn, nt, stamp, la_nw, s);
    else                launch_sweep_t<16>(t, grid, (int)tr, (int)sp, sgn, nt, stamp, la_nw, s);
    return strips * (block / 64);
}

}  // namespace mi355x
