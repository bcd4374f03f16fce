// simplex_capi.hip -- the C ABI of libmi355x_simplex.so (include/mi355x_simplex.h).
//
// Host side of the drop-in boundary: handle management, uploads/downloads between the
// caller's tightly packed row-major host arrays and the padded HBM layout, the blind
// enqueue loop that keeps the whole price -> ratio -> pivot iteration on the device, the
// two-phase driver of src/simplex.lisp:402-452, and HIP-event timing of the update kernel.
// No CPU compute path exists here: without a device every entry point fails.
#include "../../include/mi355x_simplex.h"
#include "../../include/mi355x_simplex_tune.h"
#include "simplex_kernels.h"

#include <rccl/rccl.h>                    // types and prototypes only: the library is bound at run time
#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace mi355x;

namespace {

thread_local std::string g_err;
int g_select_mode = 0;                    // 0 auto, 1 single workgroup, 2 split (tuning/test hook)
int g_compact_enabled = 1;                // solve loops run on the compact representation
int g_handover_mode = 0;                  // 0 auto, 1 always the sequential re-elimination
int g_batch_mode = 0;                     // 0 auto, 1 lockstep launch pairs, 2 one workgroup per LP,
                                          // 3 blocked: look-ahead per LP + one sweep launch over all LPs
int g_tail_policy = 1;                    // a request that is not a whole number of blocks: 0 spread evenly, 1 full blocks + remainder
int g_la_mode = 0;                        // look-ahead: 0 auto, 1 two launches per step, 2 one persistent launch per block
int g_block_k = 0;                        // pivots selected ahead and applied per sweep: 0 = by size (16, or a wide
                                          // block of 28 where the sweep dominates), 1 = off, 2 .. 16, 24, 28
int g_resident_mode = 0;                  // resident solve (tableau in registers): 0 auto -- whenever the shape fits and
                                          // every other implementation knob is at its default --, 1 never, 2 whenever it fits
int g_batch_block_k = 0;                  // mirror of the blocked per-LP kernel's knob (0 = default)
int g_cp_exchange = 0;                    // column partition over RCCL, exchange B (the entering column):
                                          // 0 int64 SUM all-reduce of (owner's bits + zeros), 1 rooted
                                          // ncclBroadcast (root = the rank whose pricing winner won, read
                                          // back from the all-gather: one host synchronisation per pivot)

// The knobs above as a handle sees them: a handle takes a SNAPSHOT of them when it is created
// and never looks at the process-wide values again, so a host thread that turns a knob cannot
// change the path of a solve another thread has in flight on its own handle (handles are
// independent across threads: include/mi355x_simplex.h).  The tiling / placement knobs of
// simplex_kernels.hip (update variants, sweep shape, one-XCD placement, poll bounds, test faults)
// stay process-wide measurement and test hooks.
struct TuneSnapshot {
    int select_mode, compact_enabled, handover_mode, batch_mode, tail_policy, la_mode, block_k, resident_mode,
        batch_block_k;
};
TuneSnapshot tune_now()
{
    TuneSnapshot t;
    t.select_mode = g_select_mode; t.compact_enabled = g_compact_enabled; t.handover_mode = g_handover_mode;
    t.batch_mode = g_batch_mode; t.tail_policy = g_tail_policy; t.la_mode = g_la_mode; t.block_k = g_block_k;
    t.resident_mode = g_resident_mode; t.batch_block_k = g_batch_block_k;
    return t;
}

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? MI_NO_MEMORY : MI_HIP_ERROR,           \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                        __LINE__);                                                         \
    } while (0)

constexpr int64_t kTraceCap   = 1 << 20;
constexpr int     kTimingCap  = 4096;
constexpr int64_t kLdAlign    = 16;       // doubles: rows start on 128-byte boundaries
constexpr int     kPartCap    = 16384;    // single tableau: 8192 pricing + 8192 ratio partials
constexpr int     kBatchPartCap = 512;    // per LP of a batch

// debugging aid (MI355X_POISON_ALLOC=1): buffers that are supposed to be written before they are
// read are filled with 0xFF bytes (NaN doubles, -1 indices) instead of being left as allocated,
// so that a read of never-written memory shows up deterministically
bool poison_allocations() { static const bool on = getenv("MI355X_POISON_ALLOC") != nullptr; return on; }
void poison(void *p, size_t bytes, hipStream_t s) { if (p && poison_allocations()) (void)hipMemsetAsync(p, 0xff, bytes, s); }

int g_ld_extra = 0;                       // tuning hook: extra padding (doubles) per row
int64_t padded_ld(int64_t cols) { return (cols + kLdAlign - 1) / kLdAlign * kLdAlign + g_ld_extra; }

int device_count_checked()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

}  // namespace

struct mi355x_tab {
    TuneSnapshot tn = tune_now();         // the implementation knobs as they were when the handle was created
    int         device = 0;
    TabView     v{};
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    Ctl        *h_ctl = nullptr;          // pinned host mirror of the control block
    TabView     c{};                      // compact view [non-basic columns | RHS]; shares
                                          // basis / col / prow / ctl / trace / partials with v
    bool        compact = false;          // which representation currently holds the tableau
    bool        compact_failed = false;   // basis is not a set of unit columns: stay dense
    bool        unit_basis = false;       // basis columns verified to be exact unit vectors and
                                          // only pivoted by the solve loops since
    int64_t    *brow = nullptr;           // scratch for k_expand (var_count entries)
    int        *flag = nullptr;           // verification flag
    int         n_part = 0;               // pricing partials left by the last update (0 = none)
    int         part_is_max = -1;         // ... and the problem sense they were computed for
    int         shard_is_max = 1;         // sense last given to mi355x_shard_price
    int         shard_steps = 0;          // look-ahead steps of a shard enqueued since its last sweep
    int         timing_stride = 0;        // 0 = off, k = bracket every k-th update launch
    int64_t     update_launches = 0;
    int64_t     sweeps = 0;               // update launches so far: odd ones sweep bottom-up
    unsigned    la_epoch = 1;             // next epoch base of the persistent look-ahead kernel
    int         la_last_nw = 0;           // workgroups of the persistent launches (what a recovery rolls back against)
    // resident solve: exchange buffer, next epoch base, and whether its workgroups
    // once failed to become co-resident (the handle then stays on the established paths)
    unsigned long long *res_x = nullptr;
    unsigned    res_epoch = 1;
    bool        res_lost = false;
    bool        last_was_resident = false;  // kind of the launch enqueued last (what a kSyncLost status refers to)
    bool        la_lost = false;          // an exchange of the persistent look-ahead was lost once
                                          // (its workgroups were not co-resident): this handle
                                          // stays on the two-launch look-ahead
    int         n_timed = 0;
    std::vector<hipEvent_t> ev0, ev1;     // around the update / sweep launches
    int         n_timed_la = 0;
    std::vector<hipEvent_t> la0, la1;     // around the look-ahead of the same blocks
    // mi355x_tab_cancel (any thread): consumed by the blocking solve loop that next looks at it --
    // between two chunks of launches, when everything enqueued has completed
    std::atomic<int> cancel{0};
};

struct mi355x_batch {
    mi355x_tab *t = nullptr;              // same machinery, TabView::n_lps > 1
    // mi355x_batch_solve_async: the (host-driven) solve loop runs on a worker thread of the library
    std::thread worker;
    bool        running = false;
    int         worker_rc = MI_OK;
    std::string worker_err;
    std::vector<int32_t> w_status;
    std::vector<int64_t> w_pivots;
};

// Several sub-batches, one per device (or logical sub-batches on one device), behind one handle
struct mi355x_multibatch {
    int64_t n_lps = 0, rows = 0, cols = 0;
    std::vector<mi355x_batch *> sub;
    std::vector<int64_t> first;           // global index of each sub-batch's first LP (+ n_lps at the end)
};

namespace {

int use_device(const mi355x_tab *t)
{
    HIP_TRY(hipSetDevice(t->device));
    return MI_OK;
}

void free_tab(mi355x_tab *t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->own_stream) (void)hipStreamSynchronize(t->own_stream);
    for (auto e : t->ev0) (void)hipEventDestroy(e);
    for (auto e : t->ev1) (void)hipEventDestroy(e);
    for (auto e : t->la0) (void)hipEventDestroy(e);
    for (auto e : t->la1) (void)hipEventDestroy(e);
    (void)hipFree(t->v.M);
    (void)hipFree(t->v.basis);
    (void)hipFree(t->v.col);
    (void)hipFree(t->v.prow);
    (void)hipFree(t->v.rhs);
    (void)hipFree(t->v.ctl);
    (void)hipFree(t->v.trace_ec);
    (void)hipFree(t->v.trace_cr);
    (void)hipFree(t->v.part_v);
    (void)hipFree(t->v.part_i);
    (void)hipFree(t->v.part_s);
    (void)hipFree(t->v.bk_col);
    (void)hipFree(t->v.bk_prow);
    (void)hipFree(t->v.blk);
    (void)hipFree(t->v.bk_rmask);
    (void)hipFree(t->v.bk_smask);
    (void)hipFree(t->v.bk_smask2);
    (void)hipFree(t->v.la_px);
    (void)hipFree(t->v.la_rx);
    (void)hipFree(t->c.M);
    (void)hipFree(t->c.p2l);
    (void)hipFree(t->c.l2p);
    (void)hipFree(t->v.p2l);
    (void)hipFree(t->v.l2p);
    (void)hipFree(t->brow);
    (void)hipFree(t->flag);
    (void)hipFree(t->res_x);
    if (t->h_ctl) (void)hipHostFree(t->h_ctl);
    if (t->own_stream) (void)hipStreamDestroy(t->own_stream);
    delete t;
}

// allocate an empty handle of the given shape on `device`
int alloc_tab(mi355x_tab **out, int64_t rows, int64_t cols, int device, int64_t n_lps = 1,
              bool defer_dense = false)
{
    if (!out) return fail(MI_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (rows < 1 || cols < 1) return fail(MI_BAD_ARG, "rows=%lld cols=%lld must be >= 1",
                                          (long long)rows, (long long)cols);
    if (rows > 65535LL * 16) return fail(MI_BAD_ARG, "rows=%lld exceeds the supported 1048560",
                                         (long long)rows);
    if (n_lps < 1 || n_lps > 65535) return fail(MI_BAD_ARG, "n_lps=%lld outside [1,65535]", (long long)n_lps);
    const int ndev = device_count_checked();
    if (ndev <= 0) return fail(MI_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(MI_BAD_ARG, "device %d out of range [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    mi355x_tab *t = new (std::nothrow) mi355x_tab;
    if (!t) return fail(MI_NO_MEMORY, "host allocation failed");
    t->device = device;
    t->v.rows = rows;
    t->v.cols = cols;
    t->v.ld = padded_ld(cols);
    t->v.n_lps = n_lps;
    const int part_cap = n_lps == 1 ? kPartCap : kBatchPartCap;
    const int64_t nb = std::max<int64_t>(rows - 1, 1);
    if (n_lps > 1) {
        t->v.zs_M = rows * t->v.ld;
        t->v.zs_basis = nb;
        t->v.zs_col = rows;
        t->v.zs_prow = t->v.ld;
        t->v.zs_part = part_cap;
    }
    t->v.trace_cap = n_lps == 1 ? kTraceCap : 0;
    const size_t mbytes = (size_t)n_lps * (size_t)rows * (size_t)t->v.ld * sizeof(double);
    hipError_t e;
#define ALLOC(ptr, bytes)                                                                  \
    if ((e = hipMalloc((void **)&(ptr), (bytes))) != hipSuccess) {                         \
        free_tab(t);                                                                       \
        return fail(e == hipErrorOutOfMemory ? MI_NO_MEMORY : MI_HIP_ERROR,                \
                    "hipMalloc(%zu bytes) failed: %s", (size_t)(bytes), hipGetErrorString(e)); \
    }
    if (!defer_dense) ALLOC(t->v.M, mbytes);         // else: allocated by ensure_dense on demand
    ALLOC(t->v.basis, n_lps * nb * sizeof(int64_t));
    ALLOC(t->v.col, n_lps * rows * sizeof(double));
    ALLOC(t->v.prow, n_lps * t->v.ld * sizeof(double));
    if (n_lps == 1) ALLOC(t->v.rhs, rows * sizeof(double));
    ALLOC(t->v.ctl, n_lps * sizeof(Ctl));
    if (n_lps == 1) {
        ALLOC(t->v.trace_ec, kTraceCap * sizeof(int64_t));
        ALLOC(t->v.trace_cr, kTraceCap * sizeof(int64_t));
    }
    ALLOC(t->v.part_v, n_lps * part_cap * sizeof(double));
    ALLOC(t->v.part_i, n_lps * part_cap * sizeof(int64_t));
    ALLOC(t->v.part_s, n_lps * part_cap * sizeof(int64_t));
    t->v.part_cap = part_cap;
    if (n_lps > 1) {                                  // per-LP block state (look-ahead launch -> sweep launch)
        t->v.bk_stride = (rows + kLdAlign - 1) / kLdAlign * kLdAlign;
        t->v.zs_bk = (int64_t)kMaxBlock * t->v.bk_stride;
        t->v.zs_bkp = (int64_t)kMaxBlock * t->v.ld;
        t->v.zs_rm = t->v.bk_stride;
        t->v.zs_sm = t->v.ld;
        ALLOC(t->v.bk_col, (size_t)n_lps * t->v.zs_bk * sizeof(double));
        ALLOC(t->v.bk_prow, (size_t)n_lps * t->v.zs_bkp * sizeof(double));
        ALLOC(t->v.bk_rmask, (size_t)n_lps * t->v.zs_rm * sizeof(uint32_t));
        ALLOC(t->v.bk_smask, (size_t)n_lps * t->v.zs_sm * sizeof(uint32_t));
        ALLOC(t->v.blk, (size_t)n_lps * sizeof(BlockCtl));
    }
    if (n_lps == 1) {                                 // blocked pivoting
        t->v.bk_stride = (rows + kLdAlign - 1) / kLdAlign * kLdAlign;
        ALLOC(t->v.bk_col, (size_t)kWideBlock * t->v.bk_stride * sizeof(double));     // (room for wide blocks)
        ALLOC(t->v.bk_prow, (size_t)kWideBlock * t->v.ld * sizeof(double));
        ALLOC(t->v.blk, sizeof(BlockCtl));
        ALLOC(t->v.bk_rmask, (size_t)t->v.bk_stride * sizeof(uint32_t));
        ALLOC(t->v.bk_smask, (size_t)t->v.ld * sizeof(uint32_t));
        ALLOC(t->v.bk_smask2, (size_t)t->v.ld * sizeof(uint32_t));
        ALLOC(t->v.la_px, kMaxLaRecords * sizeof(ExchRec));
        ALLOC(t->v.la_rx, kMaxLaRecords * sizeof(ExchRec));
    }
#undef ALLOC
    if ((e = hipHostMalloc((void **)&t->h_ctl, n_lps * sizeof(Ctl))) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&t->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        free_tab(t);
        return fail(MI_HIP_ERROR, "stream/pinned allocation failed: %s", hipGetErrorString(e));
    }
    t->stream = t->own_stream;
    memset(t->h_ctl, 0, n_lps * sizeof(Ctl));
    poison(t->v.M, t->v.M ? mbytes : 0, t->stream);
    poison(t->v.col, n_lps * rows * sizeof(double), t->stream);
    poison(t->v.prow, n_lps * t->v.ld * sizeof(double), t->stream);
    poison(t->v.rhs, rows * sizeof(double), t->stream);
    poison(t->v.trace_ec, kTraceCap * sizeof(int64_t), t->stream);
    poison(t->v.trace_cr, kTraceCap * sizeof(int64_t), t->stream);
    // pricing / ratio partials are read by launches that may follow a no-op launch: defined contents
    if ((e = hipMemsetAsync(t->v.part_v, 0, n_lps * part_cap * sizeof(double), t->stream)) != hipSuccess ||
        (e = hipMemsetAsync(t->v.part_i, 0xff, n_lps * part_cap * sizeof(int64_t), t->stream)) != hipSuccess ||
        (e = hipMemsetAsync(t->v.part_s, 0, n_lps * part_cap * sizeof(int64_t), t->stream)) != hipSuccess) {
        free_tab(t);
        return fail(MI_HIP_ERROR, "memset failed: %s", hipGetErrorString(e));
    }
    if (n_lps > 1 && ((e = hipMemsetAsync(t->v.bk_col, 0, (size_t)n_lps * t->v.zs_bk * sizeof(double), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.bk_prow, 0, (size_t)n_lps * t->v.zs_bkp * sizeof(double), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.bk_rmask, 0, (size_t)n_lps * t->v.zs_rm * sizeof(uint32_t), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.bk_smask, 0, (size_t)n_lps * t->v.zs_sm * sizeof(uint32_t), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.blk, 0, (size_t)n_lps * sizeof(BlockCtl), t->stream)) != hipSuccess)) {
        free_tab(t);
        return fail(MI_HIP_ERROR, "memset failed: %s", hipGetErrorString(e));
    }
    if ((n_lps == 1 && ((e = hipMemsetAsync(t->v.blk, 0, sizeof(BlockCtl), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.bk_rmask, 0, t->v.bk_stride * sizeof(uint32_t), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.bk_smask, 0, t->v.ld * sizeof(uint32_t), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.bk_smask2, 0, t->v.ld * sizeof(uint32_t), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.la_px, 0, kMaxLaRecords * sizeof(ExchRec), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.la_rx, 0, kMaxLaRecords * sizeof(ExchRec), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.bk_col, 0, (size_t)kWideBlock * t->v.bk_stride * sizeof(double), t->stream)) != hipSuccess ||
                      (e = hipMemsetAsync(t->v.bk_prow, 0, (size_t)kWideBlock * t->v.ld * sizeof(double), t->stream)) != hipSuccess)) ||
        (e = hipMemsetAsync(t->v.ctl, 0, n_lps * sizeof(Ctl), t->stream)) != hipSuccess ||
        (e = hipMemsetAsync(t->v.basis, 0, n_lps * nb * sizeof(int64_t), t->stream)) != hipSuccess) {
        free_tab(t);
        return fail(MI_HIP_ERROR, "memset failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return MI_OK;
}

int upload(mi355x_tab *t, const double *hm, const int64_t *hb)
{
    const TabView &v = t->v;
    t->n_part = 0;
    t->compact = false;                   // the dense logical tableau is (re)defined by the caller
    t->compact_failed = false;
    t->unit_basis = false;
    if (!t->v.M)
        HIP_TRY(hipMalloc((void **)&t->v.M, (size_t)v.n_lps * v.rows * v.ld * sizeof(double)));
    if (hm) {
        const size_t all_rows = (size_t)v.rows * v.n_lps;     // LPs of a batch are stacked
        if (v.ld != v.cols)    // zero the padding columns once per upload
            HIP_TRY(hipMemsetAsync(v.M, 0, all_rows * v.ld * sizeof(double), t->stream));
        HIP_TRY(hipMemcpy2DAsync(v.M, v.ld * sizeof(double), hm, v.cols * sizeof(double),
                                 v.cols * sizeof(double), all_rows, hipMemcpyHostToDevice, t->stream));
    }
    if (hb && v.rows > 1)
        HIP_TRY(hipMemcpyAsync(v.basis, hb, v.n_lps * (v.rows - 1) * sizeof(int64_t),
                               hipMemcpyHostToDevice, t->stream));
    launch_ctl_reset(v, 0, /*reset_trace=*/1, t->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));   // host buffers may be released by the caller
    return MI_OK;
}

int read_ctl(mi355x_tab *t)
{
    HIP_TRY(hipMemcpyAsync(t->h_ctl, t->v.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return MI_OK;
}

// ---- representation changes -------------------------------------------
TabView &cur(mi355x_tab *t) { return t->compact ? t->c : t->v; }

// Rebuild the dense logical tableau from the compact representation (no-op when dense).
int ensure_dense(mi355x_tab *t)
{
    if (!t->compact) return MI_OK;
    if (!t->v.M) {                                    // handle born compact (mi355x_tab_create_compact)
        HIP_TRY(hipMalloc((void **)&t->v.M, (size_t)t->v.n_lps * t->v.rows * t->v.ld * sizeof(double)));
    }
    launch_expand(t->v, t->c, t->brow, t->stream);
    HIP_TRY(hipGetLastError());
    t->compact = false;
    t->n_part = 0;
    return MI_OK;
}

// Switch to [non-basic columns | RHS] if the basis columns are exactly unit vectors (they are
// for everything build-tableau produces and stay so under pivoting); otherwise stay dense.
int ensure_compact(mi355x_tab *t)
{
    TabView &v = t->v;
    const int64_t m = v.rows - 1, vc = v.cols - 1, n_nb = vc - m, nl = v.n_lps;
    if (t->compact || !t->tn.compact_enabled || t->compact_failed || m < 1 || n_nb < 1) return MI_OK;
    if (v.p2l) return MI_OK;                          // a compact column shard stays as it is
    t->compact_failed = true;                         // until proven otherwise
    std::vector<int64_t> basis((size_t)(nl * m));
    HIP_TRY(hipMemcpyAsync(basis.data(), v.basis, nl * m * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    std::vector<int64_t> l2p((size_t)(nl * vc), 0), p2l((size_t)(nl * n_nb), 0);
    for (int64_t k = 0; k < nl; ++k) {                // every LP of a batch must qualify
        int64_t *l = l2p.data() + k * vc, *p = p2l.data() + k * n_nb;
        for (int64_t i = 0; i < m; ++i) {
            const int64_t b = basis[(size_t)(k * m + i)];
            if (b < 0 || b >= vc || l[b] == -1) return MI_OK;           // out of range / repeated
            l[b] = -1;
        }
        int64_t slot = 0;
        for (int64_t cidx = 0; cidx < vc; ++cidx)
            if (l[cidx] != -1) { l[cidx] = slot; p[slot++] = cidx; }
    }
    if (!t->flag) HIP_TRY(hipMalloc((void **)&t->flag, sizeof(int)));
    HIP_TRY(hipMemsetAsync(t->flag, 0, sizeof(int), t->stream));
    launch_verify_basis(v, t->flag, t->stream);
    int bad = 1;
    HIP_TRY(hipMemcpyAsync(&bad, t->flag, sizeof(int), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    if (bad) return MI_OK;
    if (t->c.cols != n_nb + 1) {                      // first time: describe the compact view
        t->c = v;                                     // shares every auxiliary buffer
        t->c.M = nullptr; t->c.p2l = nullptr; t->c.l2p = nullptr;
        t->c.cols = n_nb + 1;
        t->c.ld = padded_ld(n_nb + 1);
        if (nl > 1) { t->c.zs_M = v.rows * t->c.ld; t->c.zs_p2l = n_nb; t->c.zs_l2p = vc; }
    }
    // allocate whatever is still missing (a failed attempt must not leave a half-built view)
    if (!t->c.M) {
        HIP_TRY(hipMalloc((void **)&t->c.M, (size_t)nl * v.rows * t->c.ld * sizeof(double)));
        poison(t->c.M, (size_t)nl * v.rows * t->c.ld * sizeof(double), t->stream);
    }
    if (!t->c.p2l) HIP_TRY(hipMalloc((void **)&t->c.p2l, nl * n_nb * sizeof(int64_t)));
    if (!t->c.l2p) HIP_TRY(hipMalloc((void **)&t->c.l2p, nl * vc * sizeof(int64_t)));
    if (!t->brow)  HIP_TRY(hipMalloc((void **)&t->brow, nl * vc * sizeof(int64_t)));
    HIP_TRY(hipMemcpyAsync(t->c.p2l, p2l.data(), nl * n_nb * sizeof(int64_t), hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->c.l2p, l2p.data(), nl * vc * sizeof(int64_t), hipMemcpyHostToDevice, t->stream));
    launch_compact(v, t->c, t->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));         // p2l / l2p host vectors go out of scope
    t->compact = true;
    t->compact_failed = false;
    t->unit_basis = true;
    t->n_part = 0;
    // the exchange buffer of the resident solve belongs to the representation, not to the first solve
    // (an allocation and a memset of 0.1 .. 4 MB would otherwise sit inside that solve: 20 us of config
    // 2's 1.2 ms)
    if (!t->res_x && t->tn.resident_mode != 1 && resident_plan(t->c, nullptr)) {
        const size_t bytes = resident_xbuf_bytes(t->c);
        HIP_TRY(hipMalloc((void **)&t->res_x, bytes));
        HIP_TRY(hipMemsetAsync(t->res_x, 0, bytes, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        t->res_epoch = 1;
    }
    return MI_OK;
}

// A select met an inf / NaN in the entering column while on the compact representation
// (status kNeedDense, the pivot was not applied): the reference would turn basic columns into
// NaNs, so go back to the dense logical tableau for good and let the pivot be redone there.
int fall_back_to_dense(mi355x_tab *t)
{
    int rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    t->compact_failed = true;
    t->unit_basis = false;
    launch_ctl_resume(t->v, t->stream, kNeedDense);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// select of one iteration; prices from the partials of the preceding update when they exist
void enqueue_select(mi355x_tab *t, int is_max, double f)
{
    const int np = (t->n_part > 0 && t->part_is_max == (is_max ? 1 : 0)) ? t->n_part : 0;
    // single-workgroup select for small tableaux (fewest launches), split select for large
    // ones (the strided column gather needs many workgroups' memory pipelines)
    const TabView &v = cur(t);
    bool split = (v.rows > 1024 || v.ld > 4096);
    if (t->tn.select_mode == 1) split = false;
    if (t->tn.select_mode == 2) split = true;
    if (split && select_split_supported(v)) launch_select_split(v, is_max, f, np, t->stream);
    else                                    launch_select(v, is_max, f, np, t->stream);
}

// update of one iteration (+ optional event pair around it); prices the new objective row
int enqueue_update(mi355x_tab *t, int is_max)
{
    const bool timed = t->timing_stride > 0 && t->n_timed < kTimingCap &&
                       (t->update_launches++ % t->timing_stride) == 0;
    if (timed) {
        if ((int)t->ev0.size() <= t->n_timed) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            t->ev0.push_back(a);
            t->ev1.push_back(b);
        }
        HIP_TRY(hipEventRecord(t->ev0[t->n_timed], t->stream));
    }
    t->n_part = launch_update(cur(t), is_max ? 1.0 : -1.0, 1, t->stream, t->sweeps++);
    t->part_is_max = is_max ? 1 : 0;
    if (timed) {
        HIP_TRY(hipEventRecord(t->ev1[t->n_timed], t->stream));
        t->n_timed++;
    }
    return MI_OK;
}

int enqueue_iteration(mi355x_tab *t, int is_max, double f)
{
    enqueue_select(t, is_max, f);
    return enqueue_update(t, is_max);
}

// ---- blocked pivoting: k look-ahead selects, then one sweep applies them all
// pivots per sweep of this handle: the knob as it was when the handle was created, or by size --
// 16 wherever the persistent look-ahead runs (its state lives in LDS), a wide block where the
// sweep dominates (wide_block_default), 16 otherwise
int block_size(const mi355x_tab *t)
{
    if (t->tn.block_k != 0) return t->tn.block_k;
    if (t->tn.la_mode != 1 && !t->la_lost && la_block_supported(t->c)) return kMaxBlock;
    const int w = wide_block_default(t->c);
    return w ? w : kMaxBlock;
}

bool block_mode(const mi355x_tab *t)
{
    if (t->tn.block_k == 1 || !t->compact || !block_supported(t->c)) return false;
    if (t->tn.select_mode == 1) return false;             // forced: single-workgroup select, per pivot
    if (t->tn.select_mode == 2) return true;
    // the persistent look-ahead pays at every size (a step costs less than the select + update
    // launches of one pivot); the two-launches-per-step form only where the split select is used
    if (t->tn.la_mode != 1 && la_block_supported(t->c)) return true;
    return t->c.rows > 1024 || t->c.ld > 4096;
}

int enqueue_block(mi355x_tab *t, int is_max, double f, int k)
{
    const TabView &v = t->c;
    int np = (t->n_part > 0 && t->part_is_max == (is_max ? 1 : 0)) ? t->n_part : 0;
    // (a handle that lost an exchange once stays on the two-launch form, forced mode 2 or not:
    // re-launching the persistent kernel for ever on a GPU that cannot co-schedule its workgroups
    // would never return)
    const bool persistent = t->tn.la_mode != 1 && !t->la_lost && k <= kMaxBlock && block_size(t) <= kMaxBlock &&
                            la_block_supported(v);
    // event pairs around FULL blocks only: the statistics are per (look-ahead of g_block_k
    // pivots, sweep of g_block_k pivots), the partial last block of a run is left out
    const bool timed = t->timing_stride > 0 && t->n_timed < kTimingCap && k == block_size(t) &&
                       (t->update_launches++ % t->timing_stride) == 0;
    auto ensure_events = [](std::vector<hipEvent_t> &a, std::vector<hipEvent_t> &b, int n) -> hipError_t {
        while ((int)a.size() <= n) {
            hipEvent_t x, y;
            hipError_t e = hipEventCreate(&x);
            if (e != hipSuccess) return e;
            if ((e = hipEventCreate(&y)) != hipSuccess) { (void)hipEventDestroy(x); return e; }
            a.push_back(x);
            b.push_back(y);
        }
        return hipSuccess;
    };
    if (timed) {
        HIP_TRY(ensure_events(t->la0, t->la1, t->n_timed_la));
        HIP_TRY(ensure_events(t->ev0, t->ev1, t->n_timed));
        HIP_TRY(hipEventRecord(t->la0[t->n_timed_la], t->stream));
    }
    unsigned stamp = 0;
    int la_nw = 0;
    t->last_was_resident = false;
    if (persistent) {
        if (t->la_epoch > 0x7fff0000u) {              // 32-bit tags: start over on clean records
            HIP_TRY(hipMemsetAsync(v.la_px, 0, kMaxLaRecords * sizeof(ExchRec), t->stream));
            HIP_TRY(hipMemsetAsync(v.la_rx, 0, kMaxLaRecords * sizeof(ExchRec), t->stream));
            t->la_epoch = 1;
        }
        stamp = t->la_epoch;
        la_nw = la_block_workgroups(v);
        launch_la_block(v, k, is_max, f, t->la_epoch, t->stream);
        t->la_last_nw = la_nw;
        t->la_epoch += 2 * kMaxBlock + 2;
    } else {
        for (int j = 0; j < k; ++j) np = launch_lookahead(v, j, is_max, f, np, t->stream);
    }
    if (timed) {
        HIP_TRY(hipEventRecord(t->la1[t->n_timed_la], t->stream));
        t->n_timed_la++;
        HIP_TRY(hipEventRecord(t->ev0[t->n_timed], t->stream));
    }
    t->n_part = launch_sweep(v, k, is_max ? 1.0 : -1.0, t->stream, stamp, la_nw);
    t->part_is_max = is_max ? 1 : 0;
    if (timed) {
        HIP_TRY(hipEventRecord(t->ev1[t->n_timed], t->stream));
        t->n_timed++;
    }
    return MI_OK;
}

// The persistent look-ahead gave up waiting for a workgroup's record (status kSyncLost): its
// workgroups were not all resident at the same time -- a GPU shared with other work.  Every
// workgroup times out on its own, so the leader's workgroup may have committed one pivot more than
// some other workgroup completed (that one gave up on the ratio exchange whose records the leader
// still saw arrive).  The sweep that followed applied only the pivots EVERY workgroup completed
// (BlockCtl::done); the bookkeeping of the one beyond is taken back here (k_la_rollback), and
// everything enqueued behind the failed launch was a no-op.  Nothing is lost: continue on the
// two-launch look-ahead, which needs no co-residency, for the rest of this handle's life.
int recover_lost_exchange(mi355x_tab *t)
{
    t->n_part = 0;
    if (t->last_was_resident) {
        // the resident solve's workgroups were not co-resident at their first exchange: nothing was
        // modified (it writes the tableau back only at the end of a launch that got through)
        t->res_lost = true;
    } else {
        t->la_lost = true;
        if (t->la_last_nw) launch_la_rollback(t->c, t->la_last_nw, t->stream);
    }
    launch_ctl_resume(t->v, t->stream, kSyncLost);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

int status_to_rc(int32_t st) { return st == kRunning ? MI_RUNNING : (int)st; }

// ---- the resident solve (tableaux that fit the register files; simplex_kernels.hip, k_resident)
bool knobs_at_default(const TuneSnapshot &k)
{
    return k.select_mode == 0 && k.la_mode == 0 && k.block_k == 0 && k.batch_mode == 0 && k.batch_block_k == 0;
}

bool resident_mode(const mi355x_tab *t)
{
    if (!t->compact || t->res_lost || t->tn.resident_mode == 1) return false;
    if (t->tn.resident_mode == 0 && !knobs_at_default(t->tn)) return false;   // an explicit knob asks for another path
    return resident_plan(t->c, nullptr);
}

// one launch of up to `cap` pivots per LP (cap <= 65536)
int enqueue_resident(mi355x_tab *t, int is_max, double f, int cap)
{
    if (!t->res_x) {
        const size_t bytes = resident_xbuf_bytes(t->c);
        HIP_TRY(hipMalloc((void **)&t->res_x, bytes));
        HIP_TRY(hipMemsetAsync(t->res_x, 0, bytes, t->stream));
        t->res_epoch = 1;
    }
    if (t->res_epoch > 0x7ffe0000u) {                 // 32-bit tags: start over on clean buffers
        HIP_TRY(hipMemsetAsync(t->res_x, 0, resident_xbuf_bytes(t->c), t->stream));
        t->res_epoch = 1;
    }
    // (mi355x_tab_timing_*: a resident launch is bracketed like an update launch)
    const bool timed = t->timing_stride > 0 && t->n_timed < kTimingCap &&
                       (t->update_launches++ % t->timing_stride) == 0;
    if (timed) {
        if ((int)t->ev0.size() <= t->n_timed) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            t->ev0.push_back(a);
            t->ev1.push_back(b);
        }
        HIP_TRY(hipEventRecord(t->ev0[t->n_timed], t->stream));
    }
    if (!launch_resident(t->c, t->res_x, is_max, f, cap, t->res_epoch, t->stream))
        return fail(MI_BAD_ARG, "resident launch refused");
    if (timed) {
        HIP_TRY(hipEventRecord(t->ev1[t->n_timed], t->stream));
        t->n_timed++;
    }
    t->res_epoch += (unsigned)cap + 2u;
    t->last_was_resident = true;
    t->n_part = 0;                                    // no pricing partials are left behind
    return MI_OK;
}

// A way out of a solve (the reference has no pivot cap and no anti-cycling rule, simplex.lisp:453-461;
// in Lisp a cycling LP is interruptible, a blocking foreign call is not).  Every blocking solve loop
// enqueues BOUNDED chunks of launches -- at most 64 blocks, 512 per-pivot iterations, one resident
// launch of 65536 pivots, 4096 pivots of a per-LP batch kernel -- and looks at this flag whenever it
// has read the status back, i.e. when everything enqueued has completed and the tableau is whole.
bool take_cancel(mi355x_tab *t) { return t->cancel.exchange(0, std::memory_order_acq_rel) != 0; }
// A request is aimed at the solve in flight (or, with none in flight, at the next one): whichever
// way that solve ends, the request ends with it.
struct CancelScope {
    std::atomic<int> &f;
    explicit CancelScope(std::atomic<int> &flag) : f(flag) {}
    ~CancelScope() { f.store(0, std::memory_order_release); }
};

// per-LP outcome of a batch as the host mirror of the control blocks holds it (an LP a cancelled
// solve left unfinished reports MI_RUNNING; its tableau is whole, a later solve call carries on)
int batch_report(mi355x_tab *t, int32_t *status, int64_t *n_pivots, int rc)
{
    for (int64_t i = 0; i < t->v.n_lps; ++i) {
        if (status) status[i] = status_to_rc(t->h_ctl[i].status);
        if (n_pivots) n_pivots[i] = t->h_ctl[i].n_pivots;
    }
    return rc;
}

}  // namespace

extern "C" {

int mi355x_abi_version(void) { return MI355X_SIMPLEX_ABI_VERSION; }
int mi355x_device_count(void) { return device_count_checked(); }
const char *mi355x_last_error(void) { return g_err.c_str(); }
// for host_problem.cpp / mps_reader.cpp (same library, not exported)
__attribute__((visibility("hidden"))) void mi355x_set_last_error_(const char *msg) { g_err = msg ? msg : ""; }
double mi355x_epsilon(void) { return kClEpsilon; }
const char *mi355x_update_kernel_name(void) { return update_kernel_symbol(); }

int mi355x_tab_create(mi355x_tab **out, int64_t rows, int64_t cols, const double *host_matrix,
                      const int64_t *host_basis, int device)
{
    if (!host_matrix) return fail(MI_BAD_ARG, "host_matrix is NULL");
    mi355x_tab *t = nullptr;
    int rc = alloc_tab(&t, rows, cols, device);
    if (rc != MI_OK) return rc;
    rc = upload(t, host_matrix, host_basis);
    if (rc != MI_OK) { free_tab(t); return rc; }
    *out = t;
    return MI_OK;
}

// everything of a compact handle but the stored matrix itself: allocations, column maps, basis,
// zeroed padding, control block -- enqueued on the handle's stream, not waited for
static int compact_prepare(mi355x_tab **out, int64_t rows, int64_t var_count, int64_t n_stored,
                           const int64_t *stored_cols, const int64_t *host_basis, int device)
{
    if (!out) return fail(MI_BAD_ARG, "out is NULL");
    *out = nullptr;
    const int64_t m = rows - 1;
    if (!stored_cols || !host_basis || m < 1 || n_stored < 1 || n_stored + m != var_count)
        return fail(MI_BAD_ARG, "compact upload needs rows >= 2 and n_stored + (rows-1) == var_count");
    std::vector<int64_t> l2p((size_t)var_count, -2);                 // -2: not yet accounted for
    for (int64_t i = 0; i < m; ++i) {
        const int64_t b = host_basis[i];
        if (b < 0 || b >= var_count || l2p[(size_t)b] != -2) return fail(MI_BAD_ARG, "bad basis entry %lld", (long long)b);
        l2p[(size_t)b] = -1;
    }
    for (int64_t j = 0; j < n_stored; ++j) {
        const int64_t g = stored_cols[j];
        if (g < 0 || g >= var_count || l2p[(size_t)g] != -2) return fail(MI_BAD_ARG, "bad stored column %lld", (long long)g);
        l2p[(size_t)g] = j;
    }
    mi355x_tab *t = nullptr;
    int rc = alloc_tab(&t, rows, var_count + 1, device, 1, /*defer_dense=*/true);
    if (rc != MI_OK) return rc;
    TabView &v = t->v;
    t->c = v;
    t->c.M = nullptr; t->c.p2l = nullptr; t->c.l2p = nullptr;
    t->c.cols = n_stored + 1;
    t->c.ld = padded_ld(n_stored + 1);
    hipError_t e = hipMalloc((void **)&t->c.M, (size_t)rows * t->c.ld * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&t->c.p2l, n_stored * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&t->c.l2p, var_count * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&t->brow, var_count * sizeof(int64_t));
    if (e == hipSuccess && t->c.ld != t->c.cols)                     // (the padding columns only)
        e = hipMemset2DAsync(t->c.M + t->c.cols, t->c.ld * sizeof(double), 0, (t->c.ld - t->c.cols) * sizeof(double),
                             rows, t->stream);
    // (pageable sources: these copies have left the host buffers when the calls return)
    if (e == hipSuccess) e = hipMemcpyAsync(t->c.p2l, stored_cols, n_stored * sizeof(int64_t), hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t->c.l2p, l2p.data(), var_count * sizeof(int64_t), hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(v.basis, host_basis, m * sizeof(int64_t), hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) { launch_ctl_reset(v, 0, 1, t->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);        // l2p is a local
    if (e != hipSuccess) {
        free_tab(t);
        return fail(e == hipErrorOutOfMemory ? MI_NO_MEMORY : MI_HIP_ERROR, "compact upload failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return MI_OK;
}

int mi355x_tab_create_compact(mi355x_tab **out, int64_t rows, int64_t var_count, int64_t n_stored,
                              const double *host_stored, const int64_t *stored_cols,
                              const int64_t *host_basis, int device)
{
    if (out) *out = nullptr;
    if (!host_stored) return fail(MI_BAD_ARG, "host_stored is NULL");
    mi355x_tab *t = nullptr;
    int rc = compact_prepare(&t, rows, var_count, n_stored, stored_cols, host_basis, device);
    if (rc != MI_OK) return rc;
    hipError_t e = hipMemcpy2DAsync(t->c.M, t->c.ld * sizeof(double), host_stored, t->c.cols * sizeof(double),
                                    t->c.cols * sizeof(double), rows, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e != hipSuccess) {
        free_tab(t);
        return fail(MI_HIP_ERROR, "compact upload failed: %s", hipGetErrorString(e));
    }
    t->compact = true;
    t->unit_basis = true;
    *out = t;
    return MI_OK;
}

// ---- the same with the stored matrix PRODUCED chunk by chunk (host_problem.cpp: build-tableau's rows
// are independent) into pinned staging buffers and copied while later rows are still being assembled:
// assembly, PCIe and nothing else overlap; no pageable 268 MB intermediate.  The staging pool is
// process-wide and kept (page-locking memory costs more than the copies it serves).
namespace {
struct StagePool {
    std::mutex mu;
    std::vector<double *> bufs;
    size_t bytes_each = 0;
    ~StagePool() { for (double *b : bufs) (void)hipHostFree(b); }
};
StagePool g_stage;
constexpr size_t kStageBytes = 4u << 20;
constexpr int    kStageBuffersPerWorker = 2, kStageMaxWorkers = 16;

int stage_pool_reserve(int n_buffers)
{
    std::lock_guard<std::mutex> lk(g_stage.mu);
    g_stage.bytes_each = kStageBytes;
    while ((int)g_stage.bufs.size() < n_buffers) {
        double *b = nullptr;
        HIP_TRY(hipHostMalloc((void **)&b, kStageBytes));
        g_stage.bufs.push_back(b);
    }
    return MI_OK;
}
std::mutex g_stream_build_mu;            // one streamed build at a time uses the pool
}  // namespace

extern "C" __attribute__((visibility("hidden")))
int mi355x_tab_create_compact_streamed_(mi355x_tab **out, int64_t rows, int64_t var_count, int64_t n_stored,
                                        const int64_t *stored_cols, const int64_t *host_basis, int device,
                                        void (*produce)(void *ctx, int64_t r0, int64_t r1, double *dst), void *ctx,
                                        int n_workers)
{
    if (out) *out = nullptr;
    if (!produce) return fail(MI_BAD_ARG, "no row producer");
    mi355x_tab *t = nullptr;
    int rc = compact_prepare(&t, rows, var_count, n_stored, stored_cols, host_basis, device);
    if (rc != MI_OK) return rc;
    const int64_t w = n_stored + 1;
    const int64_t rows_per_chunk = std::max<int64_t>(1, (int64_t)(kStageBytes / (w * sizeof(double))));
    const int64_t n_chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
    n_workers = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(n_workers, kStageMaxWorkers), n_chunks));
    std::lock_guard<std::mutex> build_lock(g_stream_build_mu);
    if ((size_t)w * sizeof(double) > kStageBytes ||                 // a row does not fit a staging buffer
        (rc = stage_pool_reserve(n_workers * kStageBuffersPerWorker)) != MI_OK) {
        // plain path: assemble everything, one copy
        std::unique_ptr<double[]> P(new (std::nothrow) double[(size_t)rows * w]);
        if (!P) { free_tab(t); return fail(MI_NO_MEMORY, "host allocation failed"); }
        produce(ctx, 0, rows, P.get());
        hipError_t e = hipMemcpy2DAsync(t->c.M, t->c.ld * sizeof(double), P.get(), w * sizeof(double), w * sizeof(double),
                                        rows, hipMemcpyHostToDevice, t->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
        if (e != hipSuccess) { free_tab(t); return fail(MI_HIP_ERROR, "compact upload failed: %s", hipGetErrorString(e)); }
    } else {
        std::atomic<int64_t> next{0};
        std::atomic<int> failed{0};
        auto worker = [&](int wi) {
            if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
            hipStream_t cs = nullptr;
            hipEvent_t ev[kStageBuffersPerWorker] = {};
            bool used[kStageBuffersPerWorker] = {};
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { failed = 1; return; }
            for (auto &e : ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) failed = 1;
            for (int turn = 0; !failed; ++turn) {
                const int64_t c = next.fetch_add(1);
                if (c >= n_chunks) break;
                const int b = turn % kStageBuffersPerWorker;
                double *buf = g_stage.bufs[(size_t)(wi * kStageBuffersPerWorker + b)];
                if (used[b] && hipEventSynchronize(ev[b]) != hipSuccess) { failed = 1; break; }   // its previous copy has left it
                const int64_t r0 = c * rows_per_chunk, r1 = std::min(rows, r0 + rows_per_chunk);
                produce(ctx, r0, r1, buf);
                if (hipMemcpy2DAsync(t->c.M + r0 * t->c.ld, t->c.ld * sizeof(double), buf, w * sizeof(double),
                                     w * sizeof(double), r1 - r0, hipMemcpyHostToDevice, cs) != hipSuccess ||
                    hipEventRecord(ev[b], cs) != hipSuccess) { failed = 1; break; }
                used[b] = true;
            }
            if (hipStreamSynchronize(cs) != hipSuccess) failed = 1;
            for (auto &e : ev) if (e) (void)hipEventDestroy(e);
            (void)hipStreamDestroy(cs);
        };
        std::vector<std::thread> pool;
        for (int k = 1; k < n_workers; ++k) pool.emplace_back(worker, k);
        worker(0);
        for (auto &th : pool) th.join();
        (void)hipSetDevice(device);
        if (failed) {
            (void)hipGetLastError();
            free_tab(t);
            return fail(MI_HIP_ERROR, "streamed compact upload failed");
        }
    }
    t->compact = true;
    t->unit_basis = true;
    *out = t;
    return MI_OK;
}

// Optional: pay the one-off costs now instead of inside the first solve -- the HIP context of
// `device`, the library's code object (loaded with the first launch), the pinned staging pool of
// the streamed uploads.  Idempotent.
int mi355x_init(int device)
{
    const int ndev = device_count_checked();
    if (ndev <= 0) return fail(MI_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(MI_BAD_ARG, "device %d out of range [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    mi355x_tab *t = nullptr;
    const double M[6] = {1.0, 1.0, 1.0, -1.0, 0.0, 0.0};            // max x, x <= 1: [A | I | b ; -c | 0 | 0]
    const int64_t b[1] = {1};
    int64_t k = 0;
    int rc = mi355x_tab_create(&t, 2, 3, M, b, device);
    if (rc != MI_OK) return rc;
    rc = mi355x_tab_solve(t, 1, 1024.0, 0, &k);
    mi355x_tab_destroy(t);
    if (rc != MI_OPTIMAL || k != 1) return rc < 0 ? rc : fail(MI_HIP_ERROR, "warm-up solve ended with status %d after %lld pivots", rc, (long long)k);
    {
        std::lock_guard<std::mutex> build_lock(g_stream_build_mu);
        rc = stage_pool_reserve(kStageMaxWorkers * kStageBuffersPerWorker);
    }
    return rc;
}

int mi355x_tab_upload(mi355x_tab *t, const double *host_matrix, const int64_t *host_basis)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    return upload(t, host_matrix, host_basis);
}

int mi355x_tab_copy(mi355x_tab **out, const mi355x_tab *src)
{
    if (!src) return fail(MI_BAD_ARG, "src is NULL");
    if (src->v.p2l)                         // a compact column shard has no dense logical form
        return fail(MI_UNSUPPORTED, "copy-tableau of a compact column shard is not supported");
    mi355x_tab *t = nullptr;
    int rc = use_device(src);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(const_cast<mi355x_tab *>(src));
    if (rc != MI_OK) return rc;
    rc = alloc_tab(&t, src->v.rows, src->v.cols, src->device);
    if (rc != MI_OK) return rc;
    hipError_t e = hipStreamSynchronize(src->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(t->v.M, src->v.M, (size_t)src->v.rows * src->v.ld * sizeof(double),
                           hipMemcpyDeviceToDevice, t->stream);
    if (e == hipSuccess && src->v.rows > 1)
        e = hipMemcpyAsync(t->v.basis, src->v.basis, (src->v.rows - 1) * sizeof(int64_t),
                           hipMemcpyDeviceToDevice, t->stream);
    if (e == hipSuccess) { launch_ctl_reset(t->v, 0, 1, t->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e != hipSuccess) { free_tab(t); return fail(MI_HIP_ERROR, "copy failed: %s", hipGetErrorString(e)); }
    *out = t;
    return MI_OK;
}

int mi355x_tab_create_synthetic(mi355x_tab **out, int64_t n_vars, int64_t n_cons, uint64_t seed,
                                int64_t col_begin, int64_t col_end, int device)
{
    if (n_vars < 1 || n_cons < 1) return fail(MI_BAD_ARG, "n_vars and n_cons must be >= 1");
    const int64_t vc = n_vars + n_cons;
    if (col_end < 0) col_end = vc;
    if (col_begin < 0 || col_begin >= col_end || col_end > vc)
        return fail(MI_BAD_ARG, "bad column slice [%lld,%lld) of %lld", (long long)col_begin,
                    (long long)col_end, (long long)vc);
    mi355x_tab *t = nullptr;
    int rc = alloc_tab(&t, n_cons + 1, (col_end - col_begin) + 1, device);
    if (rc != MI_OK) return rc;
    launch_synth_fill(t->v, n_vars, n_cons, seed, nullptr, col_begin, col_end, t->stream);
    launch_ctl_reset(t->v, 0, 1, t->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e != hipSuccess) { free_tab(t); return fail(MI_HIP_ERROR, "synthetic fill failed: %s", hipGetErrorString(e)); }
    *out = t;
    return MI_OK;
}

void mi355x_tab_destroy(mi355x_tab *t) { free_tab(t); }

int mi355x_tab_shape(const mi355x_tab *t, int64_t *rows, int64_t *cols, int64_t *ld)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    if (rows) *rows = t->v.rows;
    if (cols) *cols = t->v.cols;
    if (ld) *ld = t->v.ld;
    return MI_OK;
}

int mi355x_tab_layout(const mi355x_tab *t, int *compact, int64_t *stored_cols, int64_t *stored_ld)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    const TabView &v = t->compact ? t->c : t->v;
    if (compact) *compact = t->compact ? 1 : 0;
    if (stored_cols) *stored_cols = v.cols;
    if (stored_ld) *stored_ld = v.ld;
    return MI_OK;
}

int mi355x_tab_pivot(mi355x_tab *t, int64_t ec, int64_t cr)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    if (ec < 0 || ec >= t->v.cols || cr < 0 || cr >= t->v.rows)
        return fail(MI_BAD_ARG, "pivot (col %lld, row %lld) outside %lldx%lld", (long long)ec,
                    (long long)cr, (long long)t->v.rows, (long long)t->v.cols);
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    launch_prepare_pivot(t->v, ec, cr, t->stream);
    launch_update(t->v, 1.0, 0, t->stream);
    t->n_part = 0;
    t->unit_basis = false;                // a caller-chosen pivot may be anything
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    return MI_OK;
}

int mi355x_tab_price(mi355x_tab *t, int is_max, double f, int64_t *col)
{
    if (!t || !col) return fail(MI_BAD_ARG, "NULL argument");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    launch_price_only(t->v, is_max, f, t->stream);
    HIP_TRY(hipGetLastError());
    rc = read_ctl(t);
    if (rc != MI_OK) return rc;
    *col = t->h_ctl->ec;
    return MI_OK;
}

int mi355x_tab_ratio(mi355x_tab *t, int64_t ec, double f, int64_t *row)
{
    if (!t || !row) return fail(MI_BAD_ARG, "NULL argument");
    if (ec < 0 || ec >= t->v.cols) return fail(MI_BAD_ARG, "entering column %lld out of range", (long long)ec);
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    launch_ratio_only(t->v, ec, f, t->stream);
    HIP_TRY(hipGetLastError());
    rc = read_ctl(t);
    if (rc != MI_OK) return rc;
    *row = t->h_ctl->cr;
    return MI_OK;
}

int mi355x_tab_solve_async(mi355x_tab *t, int is_max, double f, int64_t n_pivots, int reset)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    if (n_pivots < 0) return fail(MI_BAD_ARG, "n_pivots < 0");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_compact(t);
    if (rc != MI_OK) return rc;
    if (reset) launch_ctl_reset(t->v, 0, 0, t->stream);
    if (resident_mode(t)) {
        // the tableau stays on chip for the whole request (one launch per 65536 pivots)
        for (int64_t left = n_pivots; left > 0; left -= 65536) {
            rc = enqueue_resident(t, is_max, f, (int)std::min<int64_t>(left, 65536));
            if (rc != MI_OK) return rc;
        }
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    if (block_mode(t)) {
        // whole blocks, then the remainder as one shorter block (k_sweep16 for the full ones, a
        // k_sweep with as few links as the remainder needs: 20 pivots 371 us, 40 pivots 627 us), or
        // (g_tail_policy 0) the remainder spread evenly over the blocks (378 / 656 us)
        const int bk = block_size(t);
        if (bk > kMaxBlock) {
            // wide blocks: full blocks, then the remainder -- as a block of up to 16 (the sweeps of
            // short blocks) preceded by one of 16 if it is longer than that
            int64_t left = n_pivots;
            while (left > 0) {
                const int64_t k = left >= bk ? bk : (left > kMaxBlock ? kMaxBlock : left);
                rc = enqueue_block(t, is_max, f, (int)k);
                if (rc != MI_OK) return rc;
                left -= k;
            }
            HIP_TRY(hipGetLastError());
            return MI_OK;
        }
        const int64_t nblk = (n_pivots + bk - 1) / bk;
        for (int64_t b = 0; b < nblk; ++b) {
            int64_t k = n_pivots / nblk + (b < n_pivots % nblk ? 1 : 0);
            if (t->tn.tail_policy == 1) k = (b + 1 < nblk || n_pivots % bk == 0) ? bk : n_pivots % bk;
            rc = enqueue_block(t, is_max, f, (int)k);
            if (rc != MI_OK) return rc;
        }
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    for (int64_t i = 0; i < n_pivots; ++i) {
        rc = enqueue_iteration(t, is_max, f);
        if (rc != MI_OK) return rc;
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

int mi355x_tab_reset(mi355x_tab *t, int64_t max_pivots)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    if (max_pivots < 0) return fail(MI_BAD_ARG, "max_pivots < 0");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    launch_ctl_reset(t->v, max_pivots, 0, t->stream);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

int mi355x_tab_sync(mi355x_tab *t, int64_t *n_pivots)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = read_ctl(t);
    if (rc != MI_OK) return rc;
    if (t->h_ctl->status == kSyncLost) {              // see recover_lost_exchange: the request may
        rc = recover_lost_exchange(t);                // have been cut short (fewer pivots than asked
        if (rc != MI_OK) return rc;                   // for); the count reported is what was applied
        rc = read_ctl(t);
        if (rc != MI_OK) return rc;
        if (n_pivots) *n_pivots = t->h_ctl->n_pivots;
        return MI_RUNNING;
    }
    if (n_pivots) *n_pivots = t->h_ctl->n_pivots;
    if (t->h_ctl->status == kNeedDense) {             // see fall_back_to_dense: further
        rc = fall_back_to_dense(t);                   // iterations continue on the dense tableau
        if (rc != MI_OK) return rc;
        return MI_RUNNING;
    }
    if (t->h_ctl->status == kResidentStuck)
        return fail(MI_HIP_ERROR, "the resident solve lost an exchange after its first one (hung GPU?): the tableau in HBM "
                                  "is as it was before that launch, but the request was not carried out");
    return status_to_rc(t->h_ctl->status);
}

int mi355x_tab_solve(mi355x_tab *t, int is_max, double f, int64_t max_pivots, int64_t *n_pivots)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    if (max_pivots < 0) return fail(MI_BAD_ARG, "max_pivots < 0");
    CancelScope cancel_scope(t->cancel);
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_compact(t);
    if (rc != MI_OK) return rc;
    launch_ctl_reset(t->v, max_pivots, 0, t->stream);
    if (resident_mode(t)) {
        // the whole solve with the tableau on chip: one launch per 65536 pivots, one read-back each
        for (;;) {
            rc = enqueue_resident(t, is_max, f, 65536);
            if (rc != MI_OK) return rc;
            HIP_TRY(hipGetLastError());
            rc = read_ctl(t);
            if (rc != MI_OK) return rc;
            const int32_t st = t->h_ctl->status;
            if (st == kRunning) {                     // (a launch ends with the tableau written back)
                if (take_cancel(t)) { if (n_pivots) *n_pivots = t->h_ctl->n_pivots; return MI_CANCELLED; }
                continue;
            }
            if (st == kSyncLost) {                    // its workgroups were not co-resident: nothing happened
                rc = recover_lost_exchange(t);
                if (rc != MI_OK) return rc;
                break;                                // the established paths below
            }
            if (st == kResidentStuck)
                return fail(MI_HIP_ERROR, "the resident solve lost an exchange after its first one (hung GPU?)");
            if (st == kNeedDense) {
                rc = fall_back_to_dense(t);           // redo that pivot, and the rest, densely (below)
                if (rc != MI_OK) return rc;
                break;
            }
            if (n_pivots) *n_pivots = t->h_ctl->n_pivots;
            return (int)st;
        }
    }
    if (block_mode(t)) {
        // blocks of g_block_k pivots, blind enqueue in growing chunks of blocks; a block whose
        // look-ahead terminates the solve still sweeps (applies what was selected before)
        int64_t blocks = 2;
        for (;;) {
            for (int64_t i = 0; i < blocks; ++i) {
                rc = enqueue_block(t, is_max, f, block_size(t));
                if (rc != MI_OK) return rc;
            }
            HIP_TRY(hipGetLastError());
            rc = read_ctl(t);
            if (rc != MI_OK) return rc;
            if (t->h_ctl->status == kSyncLost) {      // see recover_lost_exchange
                rc = recover_lost_exchange(t);
                if (rc != MI_OK) return rc;
                continue;
            }
            if (t->h_ctl->status != kRunning) break;
            // (every block ends with its sweep: the tableau is whole whenever the host looks)
            if (take_cancel(t)) { if (n_pivots) *n_pivots = t->h_ctl->n_pivots; return MI_CANCELLED; }
            if (blocks < 64) blocks *= 2;
        }
        if (t->h_ctl->status != kNeedDense) {
            if (n_pivots) *n_pivots = t->h_ctl->n_pivots;
            return (int)t->h_ctl->status;
        }
        rc = fall_back_to_dense(t);                   // redo that pivot, and the rest, densely
        if (rc != MI_OK) return rc;
    }
    // Blind enqueue in growing chunks, one status read-back per chunk.  The stream always
    // ends on a select (it is the select that detects optimality / unboundedness / the cap);
    // an update is a no-op unless the preceding select chose a pivot, so iterations enqueued
    // past termination cost a few empty launches and nothing else.
    enqueue_select(t, is_max, f);
    int64_t chunk = 16;
    for (;;) {
        for (int64_t i = 0; i < chunk; ++i) {
            rc = enqueue_update(t, is_max);
            if (rc != MI_OK) return rc;
            enqueue_select(t, is_max, f);
        }
        HIP_TRY(hipGetLastError());
        rc = read_ctl(t);
        if (rc != MI_OK) return rc;
        if (t->h_ctl->status == kNeedDense) {         // redo that pivot, and the rest, densely
            rc = fall_back_to_dense(t);
            if (rc != MI_OK) return rc;
            enqueue_select(t, is_max, f);
            continue;
        }
        if (t->h_ctl->status != kRunning) break;
        if (take_cancel(t)) {
            // the chunk ended on a select that chose a pivot (basis, column maps and pivot count
            // already say so): apply it, then the tableau is whole
            rc = enqueue_update(t, is_max);
            if (rc != MI_OK) return rc;
            HIP_TRY(hipGetLastError());
            rc = read_ctl(t);
            if (rc != MI_OK) return rc;
            if (n_pivots) *n_pivots = t->h_ctl->n_pivots;
            return MI_CANCELLED;
        }
        if (chunk < 512) chunk *= 2;
    }
    if (n_pivots) *n_pivots = t->h_ctl->n_pivots;
    return (int)t->h_ctl->status;
}

int mi355x_tab_cancel(mi355x_tab *t)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    t->cancel.store(1, std::memory_order_release);
    return MI_OK;
}

static int solve_two_phase_impl(mi355x_tab *art, mi355x_tab *mt, int main_is_max, double f, int64_t *n_pivots);
int mi355x_solve_two_phase(mi355x_tab *art, mi355x_tab *mt, int main_is_max, double f,
                           int64_t *n_pivots)
{
    if (!art || !mt) return fail(MI_BAD_ARG, "NULL handle");
    // mi355x_tab_cancel on EITHER handle stops the call (the caller cannot know which phase is
    // running, so it cancels both); whatever flag is left over does not outlive the call
    const int rc = solve_two_phase_impl(art, mt, main_is_max, f, n_pivots);
    (void)take_cancel(art);
    (void)take_cancel(mt);
    return rc;
}
static int solve_two_phase_impl(mi355x_tab *art, mi355x_tab *mt, int main_is_max, double f, int64_t *n_pivots)
{
    if (art->v.rows != mt->v.rows || art->v.cols < mt->v.cols || art->device != mt->device)
        return fail(MI_BAD_ARG, "artificial and main tableau do not match");
    const int64_t m = mt->v.rows - 1, num_vars = mt->v.cols - 1, num_art_vars = art->v.cols - 1;
    int64_t n1 = 0, n2 = 0;
    if (n_pivots) { n_pivots[0] = 0; n_pivots[1] = 0; }
    int rc = mi355x_tab_solve(art, /*is_max=*/0, f, 0, &n1);             // simplex.lisp:403
    if (n_pivots) n_pivots[0] = n1;
    if (rc != MI_OPTIMAL) return rc;
    rc = ensure_dense(art);                // the hand-over works on the logical tableaux
    if (rc != MI_OK) return rc;
    rc = ensure_dense(mt);
    if (rc != MI_OK) return rc;
    // (fp= 0 objective factor)                                             simplex.lisp:405-407
    double art_obj = 0.0;
    HIP_TRY(hipMemcpyAsync(&art_obj, art->v.M + m * art->v.ld + num_art_vars, sizeof(double),
                           hipMemcpyDeviceToHost, art->stream));
    HIP_TRY(hipStreamSynchronize(art->stream));
    const double diff = 0.0 - art_obj;
    if (!((diff < 0.0 ? -diff : diff) <= f * kClEpsilon)) return MI_INFEASIBLE;
    // degenerate artificials still basic: pivot them out               simplex.lisp:419-434
    std::vector<int64_t> basis((size_t)std::max<int64_t>(m, 1));
    if (m > 0) {
        HIP_TRY(hipMemcpyAsync(basis.data(), art->v.basis, m * sizeof(int64_t), hipMemcpyDeviceToHost,
                               art->stream));
        HIP_TRY(hipStreamSynchronize(art->stream));
    }
    std::vector<double> row;
    for (int64_t i = 0; i < m; ++i) {
        if (basis[i] < num_vars) continue;
        row.resize((size_t)art->v.cols);
        HIP_TRY(hipMemcpyAsync(row.data(), art->v.M + i * art->v.ld, art->v.cols * sizeof(double),
                               hipMemcpyDeviceToHost, art->stream));
        HIP_TRY(hipStreamSynchronize(art->stream));
        if (row[num_art_vars] != 0.0) return MI_ART_NONZERO;
        int64_t new_col = -1;
        for (int64_t j = 0; j < num_vars; ++j) {
            if (row[j] != 0.0 && std::find(basis.begin(), basis.begin() + m, j) == basis.begin() + m) {
                new_col = j;
                break;
            }
        }
        if (new_col < 0) return MI_ART_STUCK;
        rc = mi355x_tab_pivot(art, new_col, i);
        if (rc != MI_OK) return rc;
        basis[i] = new_col;
        ++n1;
    }
    if (n_pivots) n_pivots[0] = n1;
    // copy rows + basis, re-eliminate the objective row                simplex.lisp:437-451
    HIP_TRY(hipStreamSynchronize(mt->stream));
    launch_handover(art->v, mt->v, art->unit_basis && art->tn.handover_mode != 1, art->stream);
    mt->n_part = 0;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(art->stream));
    // (a request that arrived on the main handle during phase 1, or on either between the phases)
    if (art->cancel.load(std::memory_order_acquire) || mt->cancel.load(std::memory_order_acquire)) return MI_CANCELLED;
    rc = mi355x_tab_solve(mt, main_is_max, f, 0, &n2);                  // simplex.lisp:452
    if (n_pivots) n_pivots[1] = n2;
    return rc;
}

int mi355x_tab_download(mi355x_tab *t, double *hm, int64_t *hb, double *last_row, double *last_col)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    if (t->compact && !hm) {
        // what tableau-variable & co. need (objective row, RHS column, basis) straight from the
        // compact representation: no dense tableau is rebuilt (or even allocated)
        const TabView &c = t->c;
        const int64_t n_nb = c.cols - 1;
        if (hb && c.rows > 1)
            HIP_TRY(hipMemcpyAsync(hb, c.basis, (c.rows - 1) * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
        if (last_col)
            HIP_TRY(hipMemcpy2DAsync(last_col, sizeof(double), c.M + n_nb, c.ld * sizeof(double),
                                     sizeof(double), c.rows, hipMemcpyDeviceToHost, t->stream));
        if (last_row) {
            std::vector<double>  obj((size_t)c.cols);
            std::vector<int64_t> p2l((size_t)std::max<int64_t>(n_nb, 1));
            HIP_TRY(hipMemcpyAsync(obj.data(), c.M + (c.rows - 1) * c.ld, c.cols * sizeof(double),
                                   hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipMemcpyAsync(p2l.data(), c.p2l, n_nb * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            for (int64_t j = 0; j < t->v.cols; ++j) last_row[j] = 0.0;          // basic columns: +0
            for (int64_t j = 0; j < n_nb; ++j) last_row[p2l[(size_t)j]] = obj[(size_t)j];
            last_row[t->v.cols - 1] = obj[(size_t)n_nb];
        }
        HIP_TRY(hipStreamSynchronize(t->stream));
        return MI_OK;
    }
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    const TabView &v = t->v;
    if (hm)
        HIP_TRY(hipMemcpy2DAsync(hm, v.cols * sizeof(double), v.M, v.ld * sizeof(double),
                                 v.cols * sizeof(double), v.rows, hipMemcpyDeviceToHost, t->stream));
    if (hb && v.rows > 1)
        HIP_TRY(hipMemcpyAsync(hb, v.basis, (v.rows - 1) * sizeof(int64_t), hipMemcpyDeviceToHost,
                               t->stream));
    if (last_row)
        HIP_TRY(hipMemcpyAsync(last_row, v.M + (v.rows - 1) * v.ld, v.cols * sizeof(double),
                               hipMemcpyDeviceToHost, t->stream));
    if (last_col)
        HIP_TRY(hipMemcpy2DAsync(last_col, sizeof(double), v.M + (v.cols - 1), v.ld * sizeof(double),
                                 sizeof(double), v.rows, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return MI_OK;
}

int mi355x_tab_download_block(mi355x_tab *t, int64_t row0, int64_t n_rows, int64_t col0, int64_t n_cols,
                              double *host_block)
{
    if (!t || !host_block) return fail(MI_BAD_ARG, "NULL argument");
    if (row0 < 0 || n_rows < 1 || col0 < 0 || n_cols < 1 || row0 + n_rows > t->v.rows ||
        col0 + n_cols > t->v.cols)
        return fail(MI_BAD_ARG, "block [%lld,+%lld) x [%lld,+%lld) outside %lldx%lld", (long long)row0,
                    (long long)n_rows, (long long)col0, (long long)n_cols, (long long)t->v.rows,
                    (long long)t->v.cols);
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    const TabView &v = t->v;
    HIP_TRY(hipMemcpy2DAsync(host_block, n_cols * sizeof(double), v.M + row0 * v.ld + col0,
                             v.ld * sizeof(double), n_cols * sizeof(double), n_rows,
                             hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return MI_OK;
}

int mi355x_tab_trace(mi355x_tab *t, int64_t *ecs, int64_t *crs, int64_t cap, int64_t *n)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = read_ctl(t);
    if (rc != MI_OK) return rc;
    const int64_t total = t->h_ctl->trace_n;
    if (n) *n = total;
    const int64_t k = std::min<int64_t>(std::min<int64_t>(total, cap), kTraceCap);
    if (k > 0 && ecs) HIP_TRY(hipMemcpy(ecs, t->v.trace_ec, k * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (k > 0 && crs) HIP_TRY(hipMemcpy(crs, t->v.trace_cr, k * sizeof(int64_t), hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi355x_tab_set_stream(mi355x_tab *t, void *hip_stream, int use_own)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->stream = use_own ? t->own_stream : (hipStream_t)hip_stream;   // NULL = HIP's null stream
    return MI_OK;
}

int mi355x_tab_timing_enable(mi355x_tab *t, int enable)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    t->timing_stride = enable > 0 ? enable : 0;
    t->update_launches = 0;
    t->n_timed_la = 0;
    return MI_OK;
}

int mi355x_tab_timing_read_kind(mi355x_tab *t, int which, int64_t *n_launches, double *sum_ms, double *min_ms)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    if (which != 0 && which != 1) return fail(MI_BAD_ARG, "which must be 0 (update / sweep) or 1 (look-ahead)");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    HIP_TRY(hipStreamSynchronize(t->stream));
    std::vector<hipEvent_t> &a = which ? t->la0 : t->ev0, &b = which ? t->la1 : t->ev1;
    int &n = which ? t->n_timed_la : t->n_timed;
    double sum = 0.0, mn = 0.0;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, a[i], b[i]));
        sum += ms;
        if (i == 0 || ms < mn) mn = ms;
    }
    if (n_launches) *n_launches = n;
    if (sum_ms) *sum_ms = sum;
    if (min_ms) *min_ms = mn;
    n = 0;
    return MI_OK;
}

int mi355x_tab_timing_read(mi355x_tab *t, int64_t *n_launches, double *sum_ms, double *min_ms)
{
    return mi355x_tab_timing_read_kind(t, 0, n_launches, sum_ms, min_ms);
}

// ---- batches of independent LPs ---------------------------------------------------------
// The LPs of a batch share one shape and live stacked in one allocation; every kernel of the
// single-tableau path runs unchanged with grid.z = LP index, so the whole batch advances one
// simplex iteration per (select, update) launch pair and finished LPs simply stop.
int mi355x_batch_create(mi355x_batch **out, int64_t n_lps, int64_t rows, int64_t cols,
                        const double *host_matrices, const int64_t *host_bases, int device)
{
    if (!out) return fail(MI_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (!host_matrices) return fail(MI_BAD_ARG, "host_matrices is NULL");
    mi355x_tab *t = nullptr;
    int rc = alloc_tab(&t, rows, cols, device, n_lps);
    if (rc != MI_OK) return rc;
    rc = upload(t, host_matrices, host_bases);
    if (rc != MI_OK) { free_tab(t); return rc; }
    mi355x_batch *b = new (std::nothrow) mi355x_batch;
    if (!b) { free_tab(t); return fail(MI_NO_MEMORY, "host allocation failed"); }
    b->t = t;
    *out = b;
    return MI_OK;
}

int mi355x_batch_create_synthetic(mi355x_batch **out, int64_t n_lps, int64_t n_vars, int64_t n_cons,
                                  const uint64_t *seeds, int device)
{
    if (!out) return fail(MI_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (!seeds || n_vars < 1 || n_cons < 1) return fail(MI_BAD_ARG, "bad arguments");
    mi355x_tab *t = nullptr;
    int rc = alloc_tab(&t, n_cons + 1, n_vars + n_cons + 1, device, n_lps);
    if (rc != MI_OK) return rc;
    uint64_t *dseeds = nullptr;
    hipError_t e = hipMalloc((void **)&dseeds, n_lps * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpyAsync(dseeds, seeds, n_lps * sizeof(uint64_t), hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) {
        launch_synth_fill(t->v, n_vars, n_cons, 0, dseeds, 0, n_vars + n_cons, t->stream);
        launch_ctl_reset(t->v, 0, 1, t->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    (void)hipFree(dseeds);
    if (e != hipSuccess) { free_tab(t); return fail(MI_HIP_ERROR, "synthetic batch fill failed: %s", hipGetErrorString(e)); }
    mi355x_batch *b = new (std::nothrow) mi355x_batch;
    if (!b) { free_tab(t); return fail(MI_NO_MEMORY, "host allocation failed"); }
    b->t = t;
    *out = b;
    return MI_OK;
}

int mi355x_batch_prepare(mi355x_batch *b)
{
    if (!b || !b->t) return fail(MI_BAD_ARG, "batch is NULL");
    int rc = use_device(b->t);
    if (rc != MI_OK) return rc;
    rc = ensure_compact(b->t);
    if (rc != MI_OK) return rc;
    HIP_TRY(hipStreamSynchronize(b->t->stream));
    return MI_OK;
}

int mi355x_batch_solve(mi355x_batch *b, int is_max, double f, int64_t max_pivots, int32_t *status,
                       int64_t *n_pivots)
{
    if (!b || !b->t) return fail(MI_BAD_ARG, "batch is NULL");
    if (max_pivots < 0) return fail(MI_BAD_ARG, "max_pivots < 0");
    mi355x_tab *t = b->t;
    CancelScope cancel_scope(t->cancel);
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    const int64_t n = t->v.n_lps;
    rc = ensure_compact(t);
    if (rc != MI_OK) return rc;
    launch_ctl_reset(t->v, max_pivots, 0, t->stream);
    if (resident_mode(t)) {
        // every LP on chip, a few workgroups each, all LPs in one launch (they progress and finish
        // independently); whatever that launch could not finish continues on the paths below
        for (;;) {
            rc = enqueue_resident(t, is_max, f, 65536);
            if (rc != MI_OK) return rc;
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(t->h_ctl, t->v.ctl, n * sizeof(Ctl), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            bool running = false, other = false, lost = false;
            for (int64_t i = 0; i < n; ++i) {
                const int32_t st = t->h_ctl[i].status;
                if (st == kResidentStuck) return fail(MI_HIP_ERROR, "the resident solve lost an exchange after its first one (hung GPU?)");
                running |= st == kRunning;
                lost |= st == kSyncLost;
                other |= st == kNeedDense;
            }
            if (lost) {                               // some LP's workgroups were not co-resident
                rc = recover_lost_exchange(t);
                if (rc != MI_OK) return rc;
                break;
            }
            if (other) break;                         // an LP met an inf / NaN: the established path takes it
            if (running) {
                if (take_cancel(t)) return batch_report(t, status, n_pivots, MI_CANCELLED);
                continue;
            }
            for (int64_t i = 0; i < n; ++i) {
                if (status) status[i] = t->h_ctl[i].status;
                if (n_pivots) n_pivots[i] = t->h_ctl[i].n_pivots;
            }
            return MI_OK;
        }
    }
    // preferred: one launch, one workgroup per LP (k_batch_solve); lockstep launch pairs when
    // an LP is too large for the LDS budget (or when forced by the tuning hook)
    // measured (257x769 LPs, compact representation): one workgroup per LP 1.76 M pivots/s at
    // 128 LPs and 2.13 M at 1024 LPs; lockstep launch pairs 1.62 M and 1.25 M
    // default: blocked, with the sweep of a block as ONE launch over all LPs (every CU busy in the
    // part that moves the tableaux; the look-ahead of a block is one workgroup per LP): 128 LPs of
    // 512 x 256 3.5 -> 7.2 M pivots/s, 1024 LPs 8.2 -> 11.4 M against the all-in-one-workgroup
    // kernel.  Needs the compact representation and an LP whose block state fits the LDS.
    if ((t->tn.batch_mode == 0 || t->tn.batch_mode == 3) && t->compact) {
        bool split_ok = true;
        int64_t blocks = 2;
        for (;;) {
            for (int64_t i = 0; i < blocks && split_ok; ++i)
                split_ok = launch_batch_block_split(t->c, is_max, f, t->stream);
            if (!split_ok) break;
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(t->h_ctl, t->v.ctl, n * sizeof(Ctl), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            bool running = false, need_dense = false;
            for (int64_t i = 0; i < n; ++i) {
                running |= t->h_ctl[i].status == kRunning;
                need_dense |= t->h_ctl[i].status == kNeedDense;
            }
            if (need_dense) { split_ok = false; break; }      // finish below on the dense tableaux
            if (!running) {
                t->n_part = 0;
                for (int64_t i = 0; i < n; ++i) {
                    if (status) status[i] = t->h_ctl[i].status;
                    if (n_pivots) n_pivots[i] = t->h_ctl[i].n_pivots;
                }
                return MI_OK;
            }
            if (take_cancel(t)) { t->n_part = 0; return batch_report(t, status, n_pivots, MI_CANCELLED); }
            if (blocks < 8) blocks *= 2;
        }
        if (t->compact) {                                      // an LP met an inf / NaN: the established path
            bool need_dense = false;
            for (int64_t i = 0; i < n; ++i) need_dense |= t->h_ctl[i].status == kNeedDense;
            if (need_dense) {
                rc = fall_back_to_dense(t);
                if (rc != MI_OK) return rc;
            }
        }
    }
    const bool want_persistent = t->tn.batch_mode != 1;
    bool persistent = want_persistent && launch_batch_solve(cur(t), is_max, f, t->stream);
    if (persistent) t->n_part = 0;
    if (!persistent) enqueue_select(t, is_max, f);
    int64_t chunk = 16;
    for (;;) {
        if (persistent) {
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(t->h_ctl, t->v.ctl, n * sizeof(Ctl), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            bool need_dense = false, running = false;
            for (int64_t i = 0; i < n; ++i) {
                need_dense |= t->h_ctl[i].status == kNeedDense;
                running |= t->h_ctl[i].status == kRunning;
            }
            if (need_dense) {
                rc = fall_back_to_dense(t);           // some LP met an inf / NaN: finish densely
                if (rc != MI_OK) return rc;
            } else if (!running) {
                break;
            } else if (take_cancel(t)) {              // (a launch ends after kBatchLaunchCap pivots per LP at the latest)
                return batch_report(t, status, n_pivots, MI_CANCELLED);
            }
            if (!launch_batch_solve(cur(t), is_max, f, t->stream)) return fail(MI_HIP_ERROR, "batch relaunch failed");
            continue;
        }
        for (int64_t i = 0; i < chunk; ++i) {
            rc = enqueue_update(t, is_max);
            if (rc != MI_OK) return rc;
            enqueue_select(t, is_max, f);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(t->h_ctl, t->v.ctl, n * sizeof(Ctl), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        bool running = false, need_dense = false;
        for (int64_t i = 0; i < n; ++i) {
            running |= t->h_ctl[i].status == kRunning;
            need_dense |= t->h_ctl[i].status == kNeedDense;
        }
        if (need_dense) {
            // The chunk ended on a select: every LP that is still running has a pivot selected and
            // HALF done (on the compact representation the select already moved the leaving column
            // into the entering column's slot).  Finish those pivots before the representation
            // changes -- the update is a no-op for the LP(s) that asked for the dense tableau.
            rc = enqueue_update(t, is_max);
            if (rc != MI_OK) return rc;
            rc = fall_back_to_dense(t);
            if (rc != MI_OK) return rc;
            enqueue_select(t, is_max, f);
            continue;
        }
        if (!running) break;
        if (take_cancel(t)) {
            // the chunk ended on a select: finish the pivots it chose (see above), then report
            rc = enqueue_update(t, is_max);
            if (rc != MI_OK) return rc;
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(t->h_ctl, t->v.ctl, n * sizeof(Ctl), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            return batch_report(t, status, n_pivots, MI_CANCELLED);
        }
        if (chunk < 256) chunk *= 2;
    }
    return batch_report(t, status, n_pivots, MI_OK);
}

int mi355x_batch_cancel(mi355x_batch *b)
{
    if (!b || !b->t) return fail(MI_BAD_ARG, "batch is NULL");
    b->t->cancel.store(1, std::memory_order_release);
    return MI_OK;
}

int mi355x_multibatch_cancel(mi355x_multibatch *mb)
{
    if (!mb) return fail(MI_BAD_ARG, "handle is NULL");
    for (mi355x_batch *b : mb->sub) b->t->cancel.store(1, std::memory_order_release);
    return MI_OK;
}

int mi355x_batch_download(mi355x_batch *b, int64_t k, double *hm, int64_t *hb, double *last_row,
                          double *last_col)
{
    if (!b || !b->t) return fail(MI_BAD_ARG, "batch is NULL");
    mi355x_tab *t = b->t;
    const TabView &v = t->v;
    if (k < 0 || k >= v.n_lps) return fail(MI_BAD_ARG, "lp_index %lld out of range", (long long)k);
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    const double *M = v.M + k * v.rows * v.ld;
    if (hm)
        HIP_TRY(hipMemcpy2DAsync(hm, v.cols * sizeof(double), M, v.ld * sizeof(double),
                                 v.cols * sizeof(double), v.rows, hipMemcpyDeviceToHost, t->stream));
    if (hb && v.rows > 1)
        HIP_TRY(hipMemcpyAsync(hb, v.basis + k * std::max<int64_t>(v.rows - 1, 1),
                               (v.rows - 1) * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
    if (last_row)
        HIP_TRY(hipMemcpyAsync(last_row, M + (v.rows - 1) * v.ld, v.cols * sizeof(double),
                               hipMemcpyDeviceToHost, t->stream));
    if (last_col)
        HIP_TRY(hipMemcpy2DAsync(last_col, sizeof(double), M + (v.cols - 1), v.ld * sizeof(double),
                                 sizeof(double), v.rows, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return MI_OK;
}

int mi355x_batch_timing_enable(mi355x_batch *b, int enable)
{
    if (!b || !b->t) return fail(MI_BAD_ARG, "batch is NULL");
    return mi355x_tab_timing_enable(b->t, enable);
}

int mi355x_batch_timing_read(mi355x_batch *b, int64_t *n_launches, double *sum_ms, double *min_ms)
{
    if (!b || !b->t) return fail(MI_BAD_ARG, "batch is NULL");
    return mi355x_tab_timing_read(b->t, n_launches, sum_ms, min_ms);
}

void mi355x_batch_destroy(mi355x_batch *b)
{
    if (!b) return;
    if (b->worker.joinable()) b->worker.join();
    free_tab(b->t);
    delete b;
}

// ---- asynchronous batch solve: the loop of mi355x_batch_solve on a worker thread ---------------
// The batch solve is host-driven (blind chunks of launches, one status read-back per chunk), so
// "asynchronous" means: the library drives it from a thread of its own while the caller's thread
// goes on -- e.g. starts the sub-batches of the other GPUs.  One solve at a time per batch.
int mi355x_batch_solve_async(mi355x_batch *b, int is_max, double f, int64_t max_pivots)
{
    if (!b || !b->t) return fail(MI_BAD_ARG, "batch is NULL");
    if (max_pivots < 0) return fail(MI_BAD_ARG, "max_pivots < 0");
    if (b->running) return fail(MI_BAD_ARG, "a solve of this batch is already running (mi355x_batch_sync first)");
    if (b->worker.joinable()) b->worker.join();
    const int64_t n = b->t->v.n_lps;
    b->w_status.assign((size_t)n, 0);
    b->w_pivots.assign((size_t)n, 0);
    b->worker_rc = MI_OK;
    b->worker_err.clear();
    b->running = true;
    try {
        b->worker = std::thread([b, is_max, f, max_pivots]() {
            b->worker_rc = mi355x_batch_solve(b, is_max, f, max_pivots, b->w_status.data(), b->w_pivots.data());
            if (b->worker_rc != MI_OK) b->worker_err = g_err;        // g_err is thread-local
        });
    } catch (...) {
        b->running = false;
        return fail(MI_NO_MEMORY, "could not start the worker thread");
    }
    return MI_OK;
}

int mi355x_batch_sync(mi355x_batch *b, int32_t *status, int64_t *n_pivots)
{
    if (!b || !b->t) return fail(MI_BAD_ARG, "batch is NULL");
    if (!b->running) return fail(MI_BAD_ARG, "no asynchronous solve of this batch is in flight");
    if (b->worker.joinable()) b->worker.join();
    b->running = false;
    if (b->worker_rc < 0) { g_err = b->worker_err; return b->worker_rc; }
    const int64_t n = b->t->v.n_lps;
    if (status) std::copy(b->w_status.begin(), b->w_status.begin() + n, status);
    if (n_pivots) std::copy(b->w_pivots.begin(), b->w_pivots.begin() + n, n_pivots);
    return b->worker_rc;                              // MI_OK, or MI_CANCELLED (mi355x_batch_cancel)
}

// ---- one batch over several devices (BASELINE config 4: 1024 LPs over 8 GPUs) -------------------
// LP k lives in sub-batch k / ceil(n_lps / n_devices) (contiguous blocks); independent units, no
// communication.  One call solves all of them: every sub-batch's loop runs on its own worker
// thread (mi355x_batch_solve_async), the caller's thread only waits -- so a single-threaded host
// keeps all GPUs busy.  Fewer visible devices than sub-batches (device_ids == NULL): the
// sub-batches become logical sub-batches on device 0, each with its own stream.
static void mb_free(mi355x_multibatch *mb)
{
    if (!mb) return;
    for (mi355x_batch *b : mb->sub) mi355x_batch_destroy(b);
    delete mb;
}

static int mb_layout(mi355x_multibatch *mb, int64_t n_lps, int64_t rows, int64_t cols, int *n_devices,
                     const int *device_ids, std::vector<int> &devs)
{
    if (n_lps < 1 || *n_devices < 1) return fail(MI_BAD_ARG, "need n_lps >= 1 and n_devices >= 1");
    const int ndev = device_count_checked();
    if (ndev <= 0) return fail(MI_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if ((int64_t)*n_devices > n_lps) *n_devices = (int)n_lps;
    devs.resize((size_t)*n_devices);
    for (int d = 0; d < *n_devices; ++d) {
        if (device_ids) {
            if (device_ids[d] < 0 || device_ids[d] >= ndev) return fail(MI_BAD_ARG, "device %d out of range [0,%d)", device_ids[d], ndev);
            devs[(size_t)d] = device_ids[d];
        } else {
            devs[(size_t)d] = ndev >= *n_devices ? d : 0;
        }
    }
    mb->n_lps = n_lps; mb->rows = rows; mb->cols = cols;
    const int64_t per = (n_lps + *n_devices - 1) / *n_devices;
    mb->first.clear();
    for (int d = 0; d < *n_devices; ++d) mb->first.push_back(std::min<int64_t>((int64_t)d * per, n_lps));
    mb->first.push_back(n_lps);
    return MI_OK;
}

int mi355x_multibatch_create(mi355x_multibatch **out, int64_t n_lps, int64_t rows, int64_t cols,
                             const double *host_matrices, const int64_t *host_bases, int n_devices,
                             const int *device_ids)
{
    if (!out) return fail(MI_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (!host_matrices) return fail(MI_BAD_ARG, "host_matrices is NULL");
    mi355x_multibatch *mb = new (std::nothrow) mi355x_multibatch;
    if (!mb) return fail(MI_NO_MEMORY, "host allocation failed");
    std::vector<int> devs;
    int rc = mb_layout(mb, n_lps, rows, cols, &n_devices, device_ids, devs);
    for (int d = 0; rc == MI_OK && d < n_devices; ++d) {
        const int64_t k0 = mb->first[(size_t)d], k1 = mb->first[(size_t)d + 1];
        if (k1 <= k0) continue;
        mi355x_batch *b = nullptr;
        rc = mi355x_batch_create(&b, k1 - k0, rows, cols, host_matrices + k0 * rows * cols,
                                 host_bases ? host_bases + k0 * std::max<int64_t>(rows - 1, 0) : nullptr, devs[(size_t)d]);
        if (rc == MI_OK) mb->sub.push_back(b);
    }
    if (rc != MI_OK) { mb_free(mb); return rc; }
    // (sub-batches that would be empty were skipped: first[] keeps only the boundaries in use)
    std::vector<int64_t> f2;
    for (size_t d = 0; d + 1 < mb->first.size(); ++d)
        if (mb->first[d + 1] > mb->first[d]) f2.push_back(mb->first[d]);
    f2.push_back(n_lps);
    mb->first = f2;
    *out = mb;
    return MI_OK;
}

int mi355x_multibatch_create_synthetic(mi355x_multibatch **out, int64_t n_lps, int64_t n_vars, int64_t n_cons,
                                       const uint64_t *seeds, int n_devices, const int *device_ids)
{
    if (!out) return fail(MI_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (!seeds) return fail(MI_BAD_ARG, "seeds is NULL");
    mi355x_multibatch *mb = new (std::nothrow) mi355x_multibatch;
    if (!mb) return fail(MI_NO_MEMORY, "host allocation failed");
    std::vector<int> devs;
    int rc = mb_layout(mb, n_lps, n_cons + 1, n_vars + n_cons + 1, &n_devices, device_ids, devs);
    std::vector<int64_t> f2;
    for (int d = 0; rc == MI_OK && d < n_devices; ++d) {
        const int64_t k0 = mb->first[(size_t)d], k1 = mb->first[(size_t)d + 1];
        if (k1 <= k0) continue;
        mi355x_batch *b = nullptr;
        rc = mi355x_batch_create_synthetic(&b, k1 - k0, n_vars, n_cons, seeds + k0, devs[(size_t)d]);
        if (rc == MI_OK) { mb->sub.push_back(b); f2.push_back(k0); }
    }
    if (rc != MI_OK) { mb_free(mb); return rc; }
    f2.push_back(n_lps);
    mb->first = f2;
    *out = mb;
    return MI_OK;
}

int mi355x_multibatch_info(const mi355x_multibatch *mb, int *n_sub_batches, int *n_devices_used)
{
    if (!mb) return fail(MI_BAD_ARG, "handle is NULL");
    if (n_sub_batches) *n_sub_batches = (int)mb->sub.size();
    if (n_devices_used) {
        std::vector<int> seen;
        for (mi355x_batch *b : mb->sub)
            if (std::find(seen.begin(), seen.end(), b->t->device) == seen.end()) seen.push_back(b->t->device);
        *n_devices_used = (int)seen.size();
    }
    return MI_OK;
}

int mi355x_multibatch_solve(mi355x_multibatch *mb, int is_max, double f, int64_t max_pivots, int32_t *status,
                            int64_t *n_pivots)
{
    if (!mb) return fail(MI_BAD_ARG, "handle is NULL");
    int rc = MI_OK;
    size_t started = 0;
    for (; started < mb->sub.size(); ++started) {
        rc = mi355x_batch_prepare(mb->sub[started]);                 // representation change: before the worker starts
        if (rc == MI_OK) rc = mi355x_batch_solve_async(mb->sub[started], is_max, f, max_pivots);
        if (rc != MI_OK) break;
    }
    std::string err = rc != MI_OK ? g_err : std::string();
    for (size_t d = 0; d < started; ++d) {                           // wait for every worker that did start
        const int64_t k0 = mb->first[d];
        const int r2 = mi355x_batch_sync(mb->sub[d], status ? status + k0 : nullptr, n_pivots ? n_pivots + k0 : nullptr);
        if (r2 != MI_OK && rc == MI_OK) { rc = r2; err = g_err; }
    }
    if (rc != MI_OK) g_err = err;
    return rc;
}

int mi355x_multibatch_download(mi355x_multibatch *mb, int64_t k, double *hm, int64_t *hb, double *last_row,
                               double *last_col)
{
    if (!mb) return fail(MI_BAD_ARG, "handle is NULL");
    if (k < 0 || k >= mb->n_lps) return fail(MI_BAD_ARG, "lp_index %lld out of range", (long long)k);
    size_t d = 0;
    while (d + 1 < mb->sub.size() && k >= mb->first[d + 1]) ++d;
    return mi355x_batch_download(mb->sub[d], k - mb->first[d], hm, hb, last_row, last_col);
}

void mi355x_multibatch_destroy(mi355x_multibatch *mb) { mb_free(mb); }

// ---- n-solve-tableau, two-phase branch (src/simplex.lisp:402-452), for a batch -------------------
// Member k of `art` is the artificial tableau of the problem whose main tableau is member k of
// `main_mb` (build-tableau's two results, :326-328): phase 1 = the batch loop on `art` (min
// problems); per member the feasibility test (fp= 0 objective) :405-407 and the hand-over :437-451
// (mi355x_solve_two_phase's kernels on that member's slices of the two batches); phase 2 = the batch
// loop on `main_mb`.  A member whose degenerate artificials would have to be driven out of the basis
// first (:419-434 -- row fetches and single pivots, per member) is DECLINED: status MI_UNSUPPORTED,
// the caller runs that problem through mi355x_solve_two_phase (its tableaux in the caller's memory
// are untouched; what the batches hold of it is to be ignored).
static TabView mb_member_view(const TabView &v, int64_t k)
{
    TabView s = v;
    s.M += k * v.zs_M; s.basis += k * v.zs_basis; s.col += k * v.zs_col; s.prow += k * v.zs_prow;
    s.ctl += k;
    s.n_lps = 1;
    return s;
}

int mi355x_multibatch_solve_two_phase(mi355x_multibatch *art, mi355x_multibatch *main_mb, int main_is_max, double f,
                                      int32_t *status, int64_t *n_pivots)
{
    if (!art || !main_mb || !status) return fail(MI_BAD_ARG, "NULL argument");
    if (art->n_lps != main_mb->n_lps || art->rows != main_mb->rows || art->cols < main_mb->cols ||
        art->sub.size() != main_mb->sub.size() || art->first != main_mb->first)
        return fail(MI_BAD_ARG, "the artificial and the main batch do not match (members, rows, sub-batches)");
    for (size_t d = 0; d < art->sub.size(); ++d)
        if (art->sub[d]->t->device != main_mb->sub[d]->t->device)
            return fail(MI_BAD_ARG, "sub-batch %zu of the two batches lives on different devices", d);
    const int64_t n = art->n_lps, rows = art->rows, m = rows - 1, num_vars = main_mb->cols - 1, num_art_vars = art->cols - 1;
    std::vector<int32_t> st1((size_t)n, 0), st2((size_t)n, 0);
    std::vector<int64_t> np1((size_t)n, 0), np2((size_t)n, 0);
    int rc = mi355x_multibatch_solve(art, /*is_max=*/0, f, 0, st1.data(), np1.data());          // :403
    if (rc != MI_OK) return rc;
    std::vector<char> go((size_t)n, 0);
    std::vector<double> obj;
    std::vector<int64_t> basis;
    for (size_t d = 0; d < art->sub.size(); ++d) {
        mi355x_tab *at = art->sub[d]->t, *mt = main_mb->sub[d]->t;
        const int64_t k0 = art->first[d], nd = art->first[d + 1] - k0;
        if ((rc = use_device(at)) != MI_OK || (rc = ensure_dense(at)) != MI_OK || (rc = ensure_dense(mt)) != MI_OK) return rc;
        HIP_TRY(hipStreamSynchronize(mt->stream));
        obj.resize((size_t)nd);
        basis.resize((size_t)(nd * std::max<int64_t>(m, 1)));
        // objective value (last row, last column) and basis of every member of this sub-batch
        HIP_TRY(hipMemcpy2DAsync(obj.data(), sizeof(double), at->v.M + m * at->v.ld + num_art_vars,
                                 (size_t)rows * at->v.ld * sizeof(double), sizeof(double), (size_t)nd,
                                 hipMemcpyDeviceToHost, at->stream));
        if (m > 0)
            HIP_TRY(hipMemcpyAsync(basis.data(), at->v.basis, (size_t)(nd * m) * sizeof(int64_t), hipMemcpyDeviceToHost, at->stream));
        HIP_TRY(hipStreamSynchronize(at->stream));
        for (int64_t q = 0; q < nd; ++q) {
            const int64_t k = k0 + q;
            int32_t out = st1[(size_t)k];
            if (out == MI_OPTIMAL) {
                const double diff = 0.0 - obj[(size_t)q];                                        // (fp= 0 objective factor) :405-407
                if (!((diff < 0.0 ? -diff : diff) <= f * kClEpsilon)) out = MI_INFEASIBLE;
                else {
                    bool art_basic = false;
                    for (int64_t i = 0; i < m && !art_basic; ++i) art_basic = basis[(size_t)(q * m + i)] >= num_vars;
                    if (art_basic) out = MI_UNSUPPORTED;                                         // drive-out pivots: the one-problem path
                    else go[(size_t)k] = 1;
                }
            }
            status[k] = out;
            if (go[(size_t)k]) {
                launch_handover(mb_member_view(at->v, q), mb_member_view(mt->v, q),
                                at->unit_basis && at->tn.handover_mode != 1, at->stream);        // :437-451
            } else {
                // nothing is handed over: this member's main tableau must not run (the reference's loop
                // has no cap, and what it holds is no consistent tableau) -- an all-zero objective row
                // prices as optimal at once
                HIP_TRY(hipMemsetAsync(mt->v.M + q * mt->v.zs_M + m * mt->v.ld, 0, (size_t)mt->v.ld * sizeof(double), at->stream));
            }
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(at->stream));
        mt->n_part = 0;
        mt->unit_basis = false;
        mt->compact_failed = false;
    }
    rc = mi355x_multibatch_solve(main_mb, main_is_max, f, 0, st2.data(), np2.data());            // :452
    if (rc != MI_OK) return rc;
    for (int64_t k = 0; k < n; ++k) {
        if (go[(size_t)k]) status[k] = st2[(size_t)k];
        if (n_pivots) { n_pivots[2 * k] = np1[(size_t)k]; n_pivots[2 * k + 1] = go[(size_t)k] ? np2[(size_t)k] : 0; }
    }
    return MI_OK;
}

// ---- column-partitioned shards ------------------------------------------------------
static int shard_price_x(mi355x_tab *t, int is_max, int64_t col_offset, double *dev_out2, const P2pArgs &x);
int mi355x_shard_price(mi355x_tab *t, int is_max, int64_t col_offset, double *dev_out2)
{
    return shard_price_x(t, is_max, col_offset, dev_out2, P2pArgs());
}
static int shard_price_x(mi355x_tab *t, int is_max, int64_t col_offset, double *dev_out2, const P2pArgs &x)
{
    if (!t || !dev_out2) return fail(MI_BAD_ARG, "NULL argument");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    t->shard_is_max = is_max ? 1 : 0;
    t->v.col_bias = t->v.p2l ? 0 : col_offset;        // a dense shard's column 0 is global column col_offset
    const int np = (t->n_part > 0 && t->part_is_max == (is_max ? 1 : 0)) ? t->n_part : 0;
    launch_shard_price(t->v, is_max, col_offset, dev_out2, np, t->stream, x);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

int mi355x_shard_contribute(mi355x_tab *t, const double *dev_gathered, int n_shards,
                            int64_t col_offset, double f, int64_t *dev_col_bits, int64_t *dev_ec)
{
    if (!t || !dev_gathered || !dev_col_bits || !dev_ec) return fail(MI_BAD_ARG, "NULL argument");
    if (n_shards < 1) return fail(MI_BAD_ARG, "n_shards < 1");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    launch_shard_contribute(t->v, dev_gathered, n_shards, col_offset, f, dev_col_bits, dev_ec, t->stream);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

int mi355x_shard_pivot(mi355x_tab *t, const int64_t *dev_col_bits, const int64_t *dev_ec, double f)
{
    if (!t || !dev_col_bits || !dev_ec) return fail(MI_BAD_ARG, "NULL argument");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    launch_shard_prepare(t->v, reinterpret_cast<const double *>(dev_col_bits), dev_ec, f, t->stream);
    // the update prices the new local objective-row slice for the next mi355x_shard_price
    t->n_part = launch_update(t->v, t->shard_is_max ? 1.0 : -1.0, 1, t->stream);
    t->part_is_max = t->shard_is_max;
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// ---- blocked column shards: step j of a block, then the sweep --------------
static int shard_la_contribute_x(mi355x_tab *t, int j, const double *dev_gathered, int n_shards, int64_t col_offset,
                                 double f, int64_t *dev_col_bits, int64_t *dev_ec, const P2pArgs &x);
int mi355x_shard_la_contribute(mi355x_tab *t, int j, const double *dev_gathered, int n_shards,
                               int64_t col_offset, double f, int64_t *dev_col_bits, int64_t *dev_ec)
{
    return shard_la_contribute_x(t, j, dev_gathered, n_shards, col_offset, f, dev_col_bits, dev_ec, P2pArgs());
}
static int shard_la_contribute_x(mi355x_tab *t, int j, const double *dev_gathered, int n_shards, int64_t col_offset,
                                 double f, int64_t *dev_col_bits, int64_t *dev_ec, const P2pArgs &x)
{
    if (!t || !dev_gathered || !dev_col_bits || !dev_ec) return fail(MI_BAD_ARG, "NULL argument");
    if (n_shards < 1 || j < 0 || j >= kWideBlock) return fail(MI_BAD_ARG, "n_shards < 1 or step outside [0,%d)", kWideBlock);
    if (!t->v.blk) return fail(MI_UNSUPPORTED, "no block state on this handle");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    launch_shard_la_contribute(t->v, j, dev_gathered, n_shards, col_offset, f, dev_col_bits, dev_ec, t->stream, x);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

static int shard_la_pivot_x(mi355x_tab *t, int j, const int64_t *dev_col_bits, const int64_t *dev_ec, double f,
                            const P2pArgs &x);
int mi355x_shard_la_pivot(mi355x_tab *t, int j, const int64_t *dev_col_bits, const int64_t *dev_ec, double f)
{
    return shard_la_pivot_x(t, j, dev_col_bits, dev_ec, f, P2pArgs());
}
static int shard_la_pivot_x(mi355x_tab *t, int j, const int64_t *dev_col_bits, const int64_t *dev_ec, double f,
                            const P2pArgs &x)
{
    if (!t || !dev_col_bits || !dev_ec) return fail(MI_BAD_ARG, "NULL argument");
    if (j < 0 || j >= kWideBlock) return fail(MI_BAD_ARG, "step outside [0,%d)", kWideBlock);
    if (!t->v.blk) return fail(MI_UNSUPPORTED, "no block state on this handle");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    if (j + 1 > t->shard_steps) t->shard_steps = j + 1;
    // the step prices the local objective-row slice as it will be, for the next mi355x_shard_price
    t->n_part = launch_shard_la_prepare(t->v, j, reinterpret_cast<const double *>(dev_col_bits), dev_ec, f,
                                        t->shard_is_max, t->stream, x);
    t->part_is_max = t->shard_is_max;
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

int mi355x_shard_sweep(mi355x_tab *t)
{
    if (!t) return fail(MI_BAD_ARG, "handle is NULL");
    if (!t->v.blk) return fail(MI_UNSUPPORTED, "no block state on this handle");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    const bool timed = t->timing_stride > 0 && t->n_timed < kTimingCap &&
                       (t->update_launches++ % t->timing_stride) == 0;
    if (timed) {
        if ((int)t->ev0.size() <= t->n_timed) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            t->ev0.push_back(a);
            t->ev1.push_back(b);
        }
        HIP_TRY(hipEventRecord(t->ev0[t->n_timed], t->stream));
    }
    // (as many links as steps were enqueued since the last sweep: a wide block takes k_sweepw)
    t->n_part = launch_sweep(t->v, t->shard_steps > kMaxBlock ? t->shard_steps : kMaxBlock, t->shard_is_max ? 1.0 : -1.0, t->stream);
    t->shard_steps = 0;
    t->part_is_max = t->shard_is_max;
    if (timed) {
        HIP_TRY(hipEventRecord(t->ev1[t->n_timed], t->stream));
        t->n_timed++;
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

int mi355x_shard_set_compact(mi355x_tab *t, int64_t global_var_count, const int64_t *global_cols)
{
    if (!t || !global_cols) return fail(MI_BAD_ARG, "NULL argument");
    const int64_t n_local = t->v.cols - 1;
    if (t->v.n_lps != 1 || n_local < 1 || global_var_count < n_local)
        return fail(MI_BAD_ARG, "bad shard shape");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    std::vector<int64_t> l2p((size_t)global_var_count, -1);
    for (int64_t j = 0; j < n_local; ++j) {
        const int64_t g = global_cols[j];
        if (g == -1) continue;                      // a dead slot: stored and updated, never priced (see the header)
        if (g < 0 || g >= global_var_count || l2p[(size_t)g] != -1)
            return fail(MI_BAD_ARG, "global column %lld out of range or repeated", (long long)g);
        l2p[(size_t)g] = j;
    }
    if (!t->v.p2l) HIP_TRY(hipMalloc((void **)&t->v.p2l, n_local * sizeof(int64_t)));
    (void)hipFree(t->v.l2p);
    t->v.l2p = nullptr;
    HIP_TRY(hipMalloc((void **)&t->v.l2p, global_var_count * sizeof(int64_t)));
    HIP_TRY(hipMemcpyAsync(t->v.p2l, global_cols, n_local * sizeof(int64_t), hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->v.l2p, l2p.data(), global_var_count * sizeof(int64_t), hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->n_part = 0;
    return MI_OK;
}

int mi355x_shard_columns(mi355x_tab *t, int64_t *global_cols)
{
    if (!t || !global_cols) return fail(MI_BAD_ARG, "NULL argument");
    if (!t->v.p2l) return fail(MI_BAD_ARG, "not a compact shard");
    int rc = use_device(t);
    if (rc != MI_OK) return rc;
    HIP_TRY(hipMemcpyAsync(global_cols, t->v.p2l, (t->v.cols - 1) * sizeof(int64_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return MI_OK;
}

// ---- column-partitioned tableau: the whole solve behind one handle ----------------------------
// The driver of the per-shard steps above, in the library (include/mi355x_simplex.h,
// mi355x_colpart_*): per pivot  price -> exchange A -> la_contribute -> exchange B -> la_pivot,
// after 16 pivots (or before the host looks) sweep.  Exchanges: RCCL on the shards' streams when
// every shard has its own device (one host thread per shard in the one-process form, so the
// enqueue work of N shards runs in parallel; in the one-process-per-GPU form the single local
// shard is driven inline), device-local kernels with the same semantics when the shards are
// logical shards of one device.
}  // extern "C"

namespace {

// RCCL is bound at run time, to exactly ONE copy per process: a process that already holds a
// librccl (PyTorch wheels bundle their own, built against their own HIP runtime; two copies in one
// process interpose each other's symbols and corrupt the heap) uses that one, any other process
// (the Lisp host) gets librccl.so.1 of the ROCm installation.  No RCCL at all = MI_RCCL_ERROR from
// the entry points that need it; everything else in the library is unaffected.
struct RcclApi {
    decltype(&ncclGetUniqueId)    GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank = nullptr;
    decltype(&ncclCommInitAll)    CommInitAll = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy = nullptr;
    decltype(&ncclAllGather)      AllGather = nullptr;
    decltype(&ncclAllReduce)      AllReduce = nullptr;
    decltype(&ncclBroadcast)      Broadcast = nullptr;
    decltype(&ncclCommAbort)      CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string path, error;
    bool ok = false;
};

int rccl_find_loaded(struct dl_phdr_info *info, size_t, void *out)
{
    const char *name = info->dlpi_name ? info->dlpi_name : "";
    const char *base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "librccl.so", 10) == 0) { *static_cast<std::string *>(out) = name; return 1; }
    return 0;
}

const RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        std::string loaded;
        dl_iterate_phdr(rccl_find_loaded, &loaded);
        if (!loaded.empty()) { h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD); api.path = loaded; }
        for (const char *cand : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            if (h) break;
            h = dlopen(cand, RTLD_NOW | RTLD_LOCAL);
            if (h) api.path = cand;
        }
        if (!h) { api.error = "no librccl found (dlopen: " + std::string(dlerror() ? dlerror() : "?") + ")"; return; }
#define MI_RCCL_SYM(name)                                                              \
        api.name = reinterpret_cast<decltype(api.name)>(dlsym(h, "nccl" #name));        \
        if (!api.name) { api.error = "nccl" #name " missing in " + api.path; return; }
        MI_RCCL_SYM(GetUniqueId) MI_RCCL_SYM(CommInitRank) MI_RCCL_SYM(CommInitAll) MI_RCCL_SYM(CommDestroy)
        MI_RCCL_SYM(AllGather) MI_RCCL_SYM(AllReduce) MI_RCCL_SYM(Broadcast) MI_RCCL_SYM(CommAbort)
        MI_RCCL_SYM(GetErrorString)
#undef MI_RCCL_SYM
        api.ok = true;
    });
    return api;
}

#define RCCL_NEED()                                                                            \
    do {                                                                                       \
        if (!rccl().ok) return fail(MI_RCCL_ERROR, "RCCL unavailable: %s", rccl().error.c_str()); \
    } while (0)

#define RCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess)                                                                 \
            return fail(MI_RCCL_ERROR, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// exchange B on one device: out[r] = sum over the shards of their contributions (owner's bit
// patterns + zeros), exactly what the int64 SUM all-reduce delivers to every rank
__global__ __launch_bounds__(256) void k_local_sum(const long long *all, long long *out, int64_t rows, int n)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < rows;
         r += (int64_t)gridDim.x * blockDim.x) {
        long long acc = 0;
        for (int s = 0; s < n; ++s) acc += all[(int64_t)s * rows + r];
        out[r] = acc;
    }
}

// ---- exchange mode 2: the shards write straight into each other's memory (P2P over xGMI) -----------
// No collective and no host in the loop: every shard owns ONE fine-grained exchange buffer that its
// peers can write (peer access within a process, IPC handles between processes), laid out as
//     pairs  [2 parities][world][4 granules]     the local pricing winners (exchange A)
//     column [2 parities][2 x rows granules]     the entering column (exchange B)
// with every granule {tag = epoch of the pivot, 32 bits of payload} written by one 8-byte
// system-scope store and polled with system-scope loads until the tag matches -- self-validating,
// so nothing depends on the order in which stores from another GPU become visible, and there is no
// flag, fence or counter.  Producers never wait for anybody (a shard's pricing precedes its own
// waits in its stream), so the scheme cannot deadlock; a shard can run at most one pivot ahead of
// the slowest one (it needs that one's pair to go on), hence two parities.
// (layout and tag conventions: P2pLayout / P2pArgs in simplex_kernels.h.  The blocked shard steps run
// FUSED -- the pricing kernel pushes the pair, the contribution kernel waits for the pairs and pushes
// the column, the split look-ahead step waits for the column: four launches per pivot and shard; the
// kernels below are the unfused forms, used by the per-pivot path, the one-workgroup look-ahead step
// of small shards and the drive-out pivots of the two-phase hand-over)
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// exchange A, producer: my (key, global column) pair into slot `rank` of EVERY shard's buffer
// (only while the solve is running: iterations enqueued blind behind the terminating pivot must
// not overwrite, with later tags, a pair some peer has not polled yet)
__global__ __launch_bounds__(64) void k_p2p_push_pair(const Ctl *ctl, const double *send2, unsigned long long *const *peers,
                                                      P2pLayout lay, int rank, unsigned epoch)
{
    const int r = threadIdx.x;
    if (ctl->status != kRunning || r >= lay.world) return;
    const unsigned long long kb = (unsigned long long)__double_as_longlong(send2[0]);
    const unsigned long long cb = (unsigned long long)__double_as_longlong(send2[1]);
    unsigned long long *dst = peers[r] + lay.pair_off(epoch & 1u, rank);
    const unsigned long long tg = (unsigned long long)epoch << 32;
    st_sys(dst + 0, tg | (kb & 0xffffffffull));
    st_sys(dst + 1, tg | (kb >> 32));
    st_sys(dst + 2, tg | (cb & 0xffffffffull));
    st_sys(dst + 3, tg | (cb >> 32));
}
// exchange A, consumer: wait for every shard's pair of this pivot -> the plain `gathered` array
__global__ __launch_bounds__(64) void k_p2p_wait_pairs(Ctl *ctl, const unsigned long long *mine, P2pLayout lay,
                                                       unsigned epoch, double *gathered, unsigned max_spins)
{
    const int r = threadIdx.x;
    if (ctl->status != kRunning || r >= lay.world) return;      // (iterations enqueued past termination: no-ops)
    const unsigned long long *src = mine + lay.pair_off(epoch & 1u, r);
    unsigned long long g[4];
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) { g[k] = ld_sys(src + k); ok &= (unsigned)(g[k] >> 32) == epoch; }
        if (ok) break;
        if (spins > max_spins) { ctl->status = kExchangeLost; return; }
    }
    gathered[2 * r]     = __longlong_as_double((long long)(((g[1] & 0xffffffffull) << 32) | (g[0] & 0xffffffffull)));
    gathered[2 * r + 1] = __longlong_as_double((long long)(((g[3] & 0xffffffffull) << 32) | (g[2] & 0xffffffffull)));
}
// exchange B, producer: the shard that owns the entering column (the contribution kernel left its
// bit patterns in `bits`; everybody else holds zeros there and stays silent) writes it to every shard
__global__ __launch_bounds__(256) void k_p2p_push_column(TabView t, const long long *bits, const int64_t *ec_dev,
                                                         int64_t col_offset, unsigned long long *const *peers,
                                                         P2pLayout lay, unsigned epoch)
{
    const int64_t ec = *ec_dev;
    if (ec < 0) return;
    const int64_t lc = t.l2p ? t.l2p[ec] : ec - col_offset;
    if (!(lc >= 0 && lc < t.cols - 1)) return;                  // not mine
    const unsigned long long tg = (unsigned long long)epoch << 32;
    const int64_t off = lay.col_off(epoch & 1u);
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < t.rows; r += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long vb = (unsigned long long)bits[r];
        for (int q = 0; q < lay.world; ++q) {
            unsigned long long *dst = peers[q] + off + 2 * r;
            st_sys(dst, tg | (vb & 0xffffffffull));
            st_sys(dst + 1, tg | (vb >> 32));
        }
    }
}
// exchange B, consumer: wait for the column of this pivot -> the plain `bits_in` array
__global__ __launch_bounds__(256) void k_p2p_wait_column(Ctl *ctl, const int64_t *ec_dev, const unsigned long long *mine,
                                                         P2pLayout lay, unsigned epoch, int64_t rows, long long *bits_in,
                                                         unsigned max_spins)
{
    if (ctl->status != kRunning || *ec_dev < 0) return;         // nothing enters (every shard decides the same)
    const unsigned long long *src = mine + lay.col_off(epoch & 1u);
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long lo, hi;
        for (unsigned spins = 0;; ++spins) {
            lo = ld_sys(src + 2 * r);
            hi = ld_sys(src + 2 * r + 1);
            if ((unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch) break;
            if (spins > max_spins) { ctl->status = kExchangeLost; return; }
        }
        bits_in[r] = (long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull));
    }
}

struct CpShard {
    mi355x_tab *t = nullptr;
    int         device = 0, index = 0;        // physical device, global shard index
    int64_t     col_begin = 0, col_end = 0;   // global logical columns (dense) / initial columns (compact)
    double     *send = nullptr, *gathered = nullptr;
    long long  *bits = nullptr, *bits_in = nullptr;   // contribution / exchanged column
    int64_t    *ec = nullptr;
    ncclComm_t  comm = nullptr;
    double     *h_gathered = nullptr;         // pinned: the all-gathered winners (exchange B as a rooted broadcast)
    // exchange mode 2 (P2P): my fine-grained exchange buffer, the device array of every shard's
    // buffer address as THIS process maps it, and what had to be opened through IPC
    unsigned long long  *xch = nullptr;
    unsigned long long **d_peers = nullptr;
    std::vector<void *>  ipc_opened;
    bool        aborted = false;              // its communicator was aborted after a failure: no stream syncs
    // exchange timing (mi355x_colpart_exchange_timing): event quads around the two collectives of
    // sampled pivots -- [before all-gather, after, before all-reduce, after]
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
};

}  // namespace

struct mi355x_colpart {
    int     world = 1;                       // shards in total
    bool    rccl = false, multi_process = false, compact = false;
    int64_t rows = 0, var_count = 0;
    int     block = kMaxBlock, j = 0;        // pivots per sweep, steps of the current block enqueued
    int     is_max = 1;
    int     timing_stride = 0;               // 0 = no exchange timing, k = every k-th pivot
    int     exchange = 0;                    // g_cp_exchange when the handle was created (3 -> 2 with p2p_merged off)
    bool    p2p_merged = true;               // mode 2: a shard that has its device to itself steps in two launches
    unsigned xepoch = 0;                     // P2P exchange: pivots exchanged so far (the granules' tags)
    unsigned p2p_spins = 1u << 24;           // polls before a shard gives a peer up (kExchangeLost)
    bool     p2p_connected = false;          // mode 2: every peer's buffer is mapped
    P2pLayout lay{};
    std::vector<CpShard> sh;                 // the shards of THIS process
    // logical shards: one allocation each, shared by all of them
    double    *l_gathered = nullptr;
    long long *l_bits_all = nullptr, *l_bits_sum = nullptr;
    std::vector<int> thread_rc;
    bool    dead = false;                    // communicators aborted after a failure: only destroy is left
    std::atomic<int> cancel{0};              // mi355x_colpart_cancel (any thread; one-process forms)
};

namespace {

void cp_free(mi355x_colpart *p)
{
    if (!p) return;
    for (CpShard &s : p->sh) {
        // (a shard whose communicator had to be aborted may have collectives on its stream that
        // can never complete: do not wait for them)
        if (s.t && !s.aborted) { (void)hipSetDevice(s.device); (void)hipStreamSynchronize(s.t->stream); }
        if (s.comm && rccl().ok) (void)rccl().CommDestroy(s.comm);
        for (hipEvent_t e : s.ev) (void)hipEventDestroy(e);
        if (s.h_gathered) (void)hipHostFree(s.h_gathered);
        for (void *q : s.ipc_opened) (void)hipIpcCloseMemHandle(q);
        if (s.xch) { (void)hipSetDevice(s.device); (void)hipFree(s.xch); }
        if (s.d_peers) { (void)hipSetDevice(s.device); (void)hipFree(s.d_peers); }
        if (s.aborted && s.t) s.t->own_stream = nullptr;   // free_tab must not synchronise / destroy it either
    }
    for (CpShard &s : p->sh) {
        (void)hipSetDevice(s.device);
        if (p->rccl) { (void)hipFree(s.send); (void)hipFree(s.gathered); (void)hipFree(s.bits); }
        (void)hipFree(s.ec);
        if (s.t) free_tab(s.t);
    }
    if (!p->sh.empty()) (void)hipSetDevice(p->sh[0].device);
    (void)hipFree(p->l_gathered);
    (void)hipFree(p->l_bits_all);
    (void)hipFree(p->l_bits_sum);
    delete p;
}

// [begin, end) of shard r of n over `count` columns, sizes differing by at most one
void cp_partition(int64_t count, int n, int r, int64_t *b, int64_t *e)
{
    const int64_t base = count / n, extra = count % n;
    *b = r * base + std::min<int64_t>(r, extra);
    *e = *b + base + (r < extra ? 1 : 0);
}

// One process per GPU, exchange mode 2: map the other ranks' exchange buffers (`handles`: world x
// 64 bytes, the hipIpcMemHandle_t of every rank's buffer in rank order, this rank's own included)
int cp_p2p_connect(mi355x_colpart *p, const char *handles)
{
    CpShard &s = p->sh[0];
    HIP_TRY(hipSetDevice(s.device));
    std::vector<unsigned long long *> ptrs((size_t)p->world, nullptr);
    ptrs[(size_t)s.index] = s.xch;
    for (int r = 0; r < p->world; ++r) {
        if (r == s.index) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)64 * r, 64);
        void *q = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess));
        s.ipc_opened.push_back(q);
        ptrs[(size_t)r] = (unsigned long long *)q;
    }
    HIP_TRY(hipMemcpy(s.d_peers, ptrs.data(), p->world * sizeof(unsigned long long *), hipMemcpyHostToDevice));
    p->p2p_connected = true;
    return MI_OK;
}

// Exchange mode 2: one fine-grained buffer per shard, every shard's address of every buffer.
// One process: the shards' devices get peer access to each other.  One process per GPU: the IPC
// handle of this rank's buffer is all-gathered over the (already initialised) communicator and the
// other ranks' buffers are opened -- set-up only, the pivots themselves use no collective.
int cp_setup_p2p(mi355x_colpart *p)
{
    // (the pair kernels index by threadIdx.x < world on one wave and stage 2 x 64 doubles in LDS)
    if (p->world > 64)
        return fail(MI_BAD_ARG, "exchange mode 2 (P2P push) supports at most 64 shards, not %d", p->world);
    p->lay.world = p->world;
    p->lay.rows_p = (p->rows + 7) / 8 * 8;
    const size_t bytes = (size_t)p->lay.granules() * sizeof(unsigned long long);
    for (CpShard &s : p->sh) {
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipExtMallocWithFlags((void **)&s.xch, bytes, hipDeviceMallocFinegrained));
        HIP_TRY(hipMemset(s.xch, 0, bytes));
        HIP_TRY(hipMalloc((void **)&s.d_peers, p->world * sizeof(unsigned long long *)));
    }
    std::vector<unsigned long long *> ptrs((size_t)p->world, nullptr);
    if (!p->multi_process) {
        for (CpShard &s : p->sh) ptrs[(size_t)s.index] = s.xch;
        if (p->rccl)                                          // distinct devices: let them write to each other
            for (CpShard &a : p->sh) {
                HIP_TRY(hipSetDevice(a.device));
                for (CpShard &b : p->sh) {
                    if (a.device == b.device) continue;
                    const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                        return fail(MI_HIP_ERROR, "no peer access from device %d to device %d: %s", a.device, b.device,
                                    hipGetErrorString(e));
                    (void)hipGetLastError();
                }
            }
        for (CpShard &s : p->sh) {
            HIP_TRY(hipSetDevice(s.device));
            HIP_TRY(hipMemcpy(s.d_peers, ptrs.data(), p->world * sizeof(unsigned long long *), hipMemcpyHostToDevice));
        }
        p->p2p_connected = true;
        return MI_OK;
    }
    CpShard &s = p->sh[0];
    HIP_TRY(hipSetDevice(s.device));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (p->world == 1) {
        ptrs[0] = s.xch;
        HIP_TRY(hipMemcpy(s.d_peers, ptrs.data(), sizeof(unsigned long long *), hipMemcpyHostToDevice));
        p->p2p_connected = true;
        return MI_OK;
    }
    if (!s.comm) return MI_OK;             // no communicator: the host passes the handles (mi355x_colpart_p2p_connect)
    RCCL_NEED();
    hipIpcMemHandle_t mine;
    HIP_TRY(hipIpcGetMemHandle(&mine, s.xch));
    char *d_one = nullptr, *d_all = nullptr;
    HIP_TRY(hipMalloc((void **)&d_one, 64));
    HIP_TRY(hipMalloc((void **)&d_all, (size_t)64 * p->world));
    HIP_TRY(hipMemcpy(d_one, &mine, 64, hipMemcpyHostToDevice));
    RCCL_TRY(rccl().AllGather(d_one, d_all, 64, ncclChar, s.comm, s.t->stream));
    HIP_TRY(hipStreamSynchronize(s.t->stream));
    std::vector<char> all((size_t)64 * p->world);
    HIP_TRY(hipMemcpy(all.data(), d_all, (size_t)64 * p->world, hipMemcpyDeviceToHost));
    (void)hipFree(d_one); (void)hipFree(d_all);
    return cp_p2p_connect(p, all.data());
}

// the two exchanges of one pivot in mode 2, on shard s's stream (epoch = the pivot's tag)
int cp_p2p_push_pair(mi355x_colpart *p, CpShard &s, unsigned epoch)
{
    hipLaunchKernelGGL(k_p2p_push_pair, dim3(1), dim3(64), 0, s.t->stream, s.t->v.ctl, s.send, s.d_peers, p->lay, s.index, epoch);
    hipLaunchKernelGGL(k_p2p_wait_pairs, dim3(1), dim3(64), 0, s.t->stream, s.t->v.ctl, s.xch, p->lay, epoch, s.gathered,
                       p->p2p_spins);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}
int cp_p2p_push_wait_pairs_split(mi355x_colpart *p, CpShard &s, unsigned epoch, bool push)
{
    if (push) hipLaunchKernelGGL(k_p2p_push_pair, dim3(1), dim3(64), 0, s.t->stream, s.t->v.ctl, s.send, s.d_peers, p->lay, s.index, epoch);
    else      hipLaunchKernelGGL(k_p2p_wait_pairs, dim3(1), dim3(64), 0, s.t->stream, s.t->v.ctl, s.xch, p->lay, epoch,
                                 s.gathered, p->p2p_spins);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}
int cp_p2p_column(mi355x_colpart *p, CpShard &s, unsigned epoch, bool push, bool wait)
{
    int blocks = (int)((p->rows + 255) / 256);
    if (blocks > 256) blocks = 256;
    if (push) hipLaunchKernelGGL(k_p2p_push_column, dim3(blocks), dim3(256), 0, s.t->stream, s.t->v, s.bits, s.ec,
                                 s.col_begin, s.d_peers, p->lay, epoch);
    if (wait) hipLaunchKernelGGL(k_p2p_wait_column, dim3(blocks), dim3(256), 0, s.t->stream, s.t->v.ctl, s.ec, s.xch,
                                 p->lay, epoch, p->rows, s.bits_in, p->p2p_spins);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// exchange buffers (+ communicators, unless they are handed over from another handle) once the
// shards' handles exist
int cp_finish_setup(mi355x_colpart *p, const void *id128, int rank, bool make_comms = true)
{
    const int nl = (int)p->sh.size();
    // pivots per sweep of a shard's slice: the knob, or by the size of a shard (the block structure is
    // local to a shard -- the exchanges are per pivot -- so ranks need not even agree on it)
    if (g_block_k > 1) p->block = g_block_k;
    else if (g_block_k == 1) p->block = 1;
    else {
        const int w = (nl > 0 && p->sh[0].t) ? wide_block_default(p->sh[0].t->v) : 0;
        p->block = w ? w : kMaxBlock;
    }
    if (!p->rccl) {
        HIP_TRY(hipSetDevice(p->sh[0].device));
        HIP_TRY(hipMalloc((void **)&p->l_gathered, 2 * p->world * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&p->l_bits_all, (size_t)p->world * p->rows * sizeof(long long)));
        HIP_TRY(hipMalloc((void **)&p->l_bits_sum, p->rows * sizeof(long long)));
        HIP_TRY(hipMemset(p->l_gathered, 0, 2 * p->world * sizeof(double)));
        for (CpShard &s : p->sh) {
            s.send = p->l_gathered + 2 * s.index;           // "all-gather" = everyone writes its slot
            s.gathered = p->l_gathered;
            s.bits = p->l_bits_all + (int64_t)s.index * p->rows;
            s.bits_in = p->l_bits_sum;
            HIP_TRY(hipMalloc((void **)&s.ec, sizeof(int64_t)));
        }
        // the logical shards run one after the other on ONE stream (that of the first)
        for (CpShard &s : p->sh) s.t->stream = p->sh[0].t->own_stream;
        return p->exchange == 2 ? cp_setup_p2p(p) : MI_OK;
    }
    if (!(p->multi_process && !id128 && p->exchange == 2)) RCCL_NEED();
    for (CpShard &s : p->sh) {
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipMalloc((void **)&s.send, 2 * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&s.gathered, 2 * p->world * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&s.bits, p->rows * sizeof(long long)));
        HIP_TRY(hipMalloc((void **)&s.ec, sizeof(int64_t)));
        HIP_TRY(hipHostMalloc((void **)&s.h_gathered, 2 * p->world * sizeof(double)));
        s.bits_in = s.bits;                                  // all-reduce / broadcast in place
    }
    if (!make_comms) return p->exchange == 2 ? cp_setup_p2p(p) : MI_OK;
    if (p->multi_process) {
        if (id128) {                                         // (NULL: exchange mode 2 without a communicator)
            ncclUniqueId id;
            memcpy(&id, id128, sizeof id);
            HIP_TRY(hipSetDevice(p->sh[0].device));
            RCCL_TRY(rccl().CommInitRank(&p->sh[0].comm, p->world, id, rank));
        }
    } else {
        std::vector<ncclComm_t> comms((size_t)nl);
        std::vector<int> devs((size_t)nl);
        for (int i = 0; i < nl; ++i) devs[(size_t)i] = p->sh[(size_t)i].device;
        RCCL_TRY(rccl().CommInitAll(comms.data(), nl, devs.data()));
        for (int i = 0; i < nl; ++i) p->sh[(size_t)i].comm = comms[(size_t)i];
    }
    return p->exchange == 2 ? cp_setup_p2p(p) : MI_OK;
}

// ---- one pivot, as the three local steps of shard s with the exchanges between them
int cp_price(mi355x_colpart *p, CpShard &s)
{
    return mi355x_shard_price(s.t, p->is_max, s.col_begin, s.send);
}
int cp_contribute(mi355x_colpart *p, CpShard &s, double f)
{
    if (p->block > 1)
        return mi355x_shard_la_contribute(s.t, p->j, s.gathered, p->world, s.col_begin, f,
                                          (int64_t *)s.bits, s.ec);
    return mi355x_shard_contribute(s.t, s.gathered, p->world, s.col_begin, f, (int64_t *)s.bits, s.ec);
}
int cp_pivot(mi355x_colpart *p, CpShard &s, double f)
{
    if (p->block > 1) return mi355x_shard_la_pivot(s.t, p->j, (const int64_t *)s.bits_in, s.ec, f);
    return mi355x_shard_pivot(s.t, (const int64_t *)s.bits_in, s.ec, f);
}

// The rank whose (key, global column) pair wins the pricing -- vi_min's rule on the host: an empty
// pair (column -1) loses against anything, a NaN-in-column-0 marker (-2) or no candidate at all
// means nobody enters a column, and any root will do (everybody contributes zeros).
int cp_winner_rank(const double *g, int world)
{
    int best = -1;
    for (int k = 0; k < world; ++k) {
        const double v = g[2 * k], c = g[2 * k + 1];
        if (c < 0.0) continue;
        if (best < 0 || v < g[2 * best] || (v == g[2 * best] && c < g[2 * best + 1])) best = k;
    }
    return best < 0 ? 0 : best;
}

// Exchange mode 2, blocked shards: one pivot of shard s as its three fused phases --
//   0  pricing kernel, which also pushes this shard's pair into every shard's buffer
//   1  contribution kernel: waits for all pairs, chains, and (the owner) pushes the column
//   2  the look-ahead step: its split form waits for the column itself; the one-workgroup form of
//      small shards takes it from the (unfused) wait kernel
// On ONE stream (logical shards) phase k of every shard is enqueued before phase k + 1 of any.
int cp_fused_p2p_step(mi355x_colpart *p, CpShard &s, double f, int j, unsigned epoch, int phase)
{
    P2pArgs x;
    x.peers = s.d_peers; x.mine = s.xch; x.lay = p->lay; x.rank = s.index; x.epoch = epoch; x.max_spins = p->p2p_spins;
    if (phase == 0) return shard_price_x(s.t, p->is_max, s.col_begin, s.send, x);
    if (phase == 1) return shard_la_contribute_x(s.t, j, s.gathered, p->world, s.col_begin, f, (int64_t *)s.bits, s.ec, x);
    if (shard_la_split(s.t->v)) return shard_la_pivot_x(s.t, j, (const int64_t *)s.bits_in, s.ec, f, x);
    int rc = cp_p2p_column(p, s, epoch, false, true);
    if (rc != MI_OK) return rc;
    return mi355x_shard_la_pivot(s.t, j, (const int64_t *)s.bits_in, s.ec, f);
}

// Exchange mode 2, blocked: the step of shard s as TWO launches (k_shard_p2p_step + k_shard_la_scale)
// where a consumer and the producer it waits for may be the same kernel on different shards -- a
// shard that has its device (or at least its stream) to itself, or the only shard there is.
// Returns > 0: done; 0: this shard / state needs the separate launches; < 0: an error.
int cp_try_merged_step(mi355x_colpart *p, CpShard &s, double f, int j, unsigned epoch)
{
    mi355x_tab *t = s.t;
    const int np = (p->p2p_merged && t->n_part > 0 && t->part_is_max == (p->is_max ? 1 : 0)) ? t->n_part : 0;
    if (!(np > 0 && t->v.blk && j < kWideBlock)) return 0;
    P2pArgs x;
    x.peers = s.d_peers; x.mine = s.xch; x.lay = p->lay; x.rank = s.index; x.epoch = epoch; x.max_spins = p->p2p_spins;
    int rc = use_device(t);
    if (rc == MI_OK) rc = ensure_dense(t);
    if (rc != MI_OK) return rc;
    t->shard_is_max = p->is_max ? 1 : 0;
    t->v.col_bias = t->v.p2l ? 0 : s.col_begin;
    const int left = launch_shard_p2p_step(t->v, j, np, p->world, s.col_begin, f, t->shard_is_max, s.ec, t->stream, x);
    if (left > 0) {
        if (j + 1 > t->shard_steps) t->shard_steps = j + 1;     // (what mi355x_shard_la_pivot records)
        t->n_part = left;
        t->part_is_max = t->shard_is_max;
        HIP_TRY(hipGetLastError());
    }
    return left;
}

// n iterations of shard s over RCCL (its own thread in the one-process form).  j0 = step of the
// block the first iteration is; every shard runs the same sequence, so they meet in the collectives.
int cp_run_rccl(mi355x_colpart *p, CpShard &s, double f, int64_t n, int j0)
{
    HIP_TRY(hipSetDevice(s.device));
    int j = j0;
    for (int64_t i = 0; i < n; ++i) {
        const bool timed = p->timing_stride > 0 && i % p->timing_stride == 0 &&
                           (size_t)s.ev_used + 4 <= s.ev.size();
        hipEvent_t *e = timed ? &s.ev[(size_t)s.ev_used] : nullptr;
        const unsigned epoch = p->xepoch + (unsigned)i + 1u;     // (mode 2: the tag of this pivot's granules)
        if (p->exchange == 2 && p->block > 1) {
            // mode 2, blocked: the exchanges are INSIDE the step kernels -- two launches per pivot where
            // the shards' kernels run concurrently (this loop: one shard per device or process), four
            // for the first pivot after an upload and for small shards
            int rc = MI_OK;
            const int left = cp_try_merged_step(p, s, f, j, epoch);
            if (left < 0) return left;
            if (left == 0) {
                rc = cp_fused_p2p_step(p, s, f, j, epoch, 0);
                if (rc == MI_OK) rc = cp_fused_p2p_step(p, s, f, j, epoch, 1);
                if (rc == MI_OK) rc = cp_fused_p2p_step(p, s, f, j, epoch, 2);
            }
            if (rc != MI_OK) return rc;
            if (++j == p->block) {
                rc = mi355x_shard_sweep(s.t);
                if (rc != MI_OK) return rc;
                j = 0;
            }
            continue;
        }
        int rc = mi355x_shard_price(s.t, p->is_max, s.col_begin, s.send);
        if (rc != MI_OK) return rc;
        if (timed) HIP_TRY(hipEventRecord(e[0], s.t->stream));
        if (p->exchange == 2) { if ((rc = cp_p2p_push_pair(p, s, epoch)) != MI_OK) return rc; }
        else RCCL_TRY(rccl().AllGather(s.send, s.gathered, 2, ncclDouble, s.comm, s.t->stream));
        if (timed) HIP_TRY(hipEventRecord(e[1], s.t->stream));
        if (p->block > 1) rc = mi355x_shard_la_contribute(s.t, j, s.gathered, p->world, s.col_begin, f, (int64_t *)s.bits, s.ec);
        else              rc = mi355x_shard_contribute(s.t, s.gathered, p->world, s.col_begin, f, (int64_t *)s.bits, s.ec);
        if (rc != MI_OK) return rc;
        int root = -1;
        if (p->exchange == 1) {
            // exchange B as a rooted broadcast: the root is the rank whose local pricing winner is
            // the global one (lexicographic (key, column) minimum over the all-gathered pairs -- the
            // decision every shard takes on the device).  The host has to look: one stream
            // synchronisation per pivot, the price of a root RCCL wants as a host argument.
            HIP_TRY(hipMemcpyAsync(s.h_gathered, s.gathered, 2 * p->world * sizeof(double),
                                   hipMemcpyDeviceToHost, s.t->stream));
            HIP_TRY(hipStreamSynchronize(s.t->stream));
            root = cp_winner_rank(s.h_gathered, p->world);
        }
        if (timed) HIP_TRY(hipEventRecord(e[2], s.t->stream));
        if (p->exchange == 2) { if ((rc = cp_p2p_column(p, s, epoch, true, true)) != MI_OK) return rc; }
        else if (root >= 0)
            RCCL_TRY(rccl().Broadcast(s.bits, s.bits, (size_t)p->rows, ncclInt64, root, s.comm, s.t->stream));
        else
            RCCL_TRY(rccl().AllReduce(s.bits, s.bits, (size_t)p->rows, ncclInt64, ncclSum, s.comm, s.t->stream));
        if (timed) { HIP_TRY(hipEventRecord(e[3], s.t->stream)); s.ev_used += 4; }
        if (p->block > 1) rc = mi355x_shard_la_pivot(s.t, j, (const int64_t *)s.bits, s.ec, f);
        else              rc = mi355x_shard_pivot(s.t, (const int64_t *)s.bits, s.ec, f);
        if (rc != MI_OK) return rc;
        if (p->block > 1 && ++j == p->block) {
            rc = mi355x_shard_sweep(s.t);
            if (rc != MI_OK) return rc;
            j = 0;
        }
    }
    return MI_OK;
}

// A rank failed between two collectives: the other ranks' streams hold collectives that will never
// complete.  Abort every local communicator (ncclCommAbort tears the kernels down) and mark the
// shards, so that neither this call's caller nor the destroy path waits on those streams.
void cp_abort(mi355x_colpart *p)
{
    if (!p->rccl) return;
    const std::string keep = g_err;
    // (exchange mode 2 without a communicator has nothing to abort, but granules of the failed run
    // were already pushed under tags a later run would reuse: the handle is dead either way)
    for (CpShard &s : p->sh) {
        if (s.comm && rccl().ok) { (void)hipSetDevice(s.device); (void)rccl().CommAbort(s.comm); s.comm = nullptr; }
        s.aborted = true;
    }
    p->dead = true;
    g_err = keep;
}

// enqueue n iterations on every local shard; no host synchronisation
int cp_run(mi355x_colpart *p, double f, int64_t n)
{
    if (p->dead) return fail(MI_RCCL_ERROR, "this handle's communicators were aborted after an earlier failure");
    if (p->exchange == 2 && !p->p2p_connected)
        return fail(MI_BAD_ARG, "exchange mode 2: the ranks' buffers are not connected yet (mi355x_colpart_p2p_connect)");
    if (n <= 0) return MI_OK;
    if (p->rccl) {
        const int j0 = p->j;
        if (p->sh.size() == 1) {
            int rc = cp_run_rccl(p, p->sh[0], f, n, j0);
            if (rc != MI_OK) { cp_abort(p); return rc; }
        } else {
            // one host thread per shard: the shards' launches are enqueued in parallel and every
            // thread issues its own rank's collectives (the standard one-process multi-GPU form)
            p->thread_rc.assign(p->sh.size(), MI_OK);
            std::vector<std::string> errs(p->sh.size());
            std::vector<std::thread> th;
            for (size_t i = 0; i < p->sh.size(); ++i)
                th.emplace_back([p, f, n, j0, i, &errs]() {
                    p->thread_rc[i] = cp_run_rccl(p, p->sh[i], f, n, j0);
                    if (p->thread_rc[i] != MI_OK) errs[i] = g_err;       // g_err is thread-local
                });
            for (auto &x : th) x.join();
            for (size_t i = 0; i < p->sh.size(); ++i)
                if (p->thread_rc[i] != MI_OK) {
                    // the other ranks have enqueued collectives that can never complete now
                    cp_abort(p);
                    g_err = errs[i];
                    return p->thread_rc[i];
                }
        }
        if (p->block > 1) p->j = (int)((j0 + n) % p->block);
        p->xepoch += (unsigned)n;
        return MI_OK;
    }
    // logical shards on one device, one stream: step by step over all of them
    HIP_TRY(hipSetDevice(p->sh[0].device));
    hipStream_t st = p->sh[0].t->stream;
    for (int64_t i = 0; i < n; ++i) {
        int rc;
        const unsigned epoch = ++p->xepoch;
        if (p->exchange == 2 && p->block > 1) {
            // ONE shard in all: nobody to wait for, so the two-launch step is safe on this stream too
            const int left = p->world == 1 ? cp_try_merged_step(p, p->sh[0], f, p->j, epoch) : 0;
            if (left < 0) return left;
            if (left == 0)
                for (int phase = 0; phase < 3; ++phase)
                    for (CpShard &s : p->sh) if ((rc = cp_fused_p2p_step(p, s, f, p->j, epoch, phase)) != MI_OK) return rc;
            if (++p->j == p->block) {
                for (CpShard &s : p->sh) if ((rc = mi355x_shard_sweep(s.t)) != MI_OK) return rc;
                p->j = 0;
            }
            continue;
        }
        for (CpShard &s : p->sh) if ((rc = cp_price(p, s)) != MI_OK) return rc;
        if (p->exchange == 2) {
            // the P2P protocol on one device and ONE stream: all producers of an exchange are
            // enqueued before its consumers, so no consumer ever waits (same kernels, same buffers)
            for (CpShard &s : p->sh) if ((rc = cp_p2p_push_wait_pairs_split(p, s, epoch, true)) != MI_OK) return rc;
            for (CpShard &s : p->sh) if ((rc = cp_p2p_push_wait_pairs_split(p, s, epoch, false)) != MI_OK) return rc;
        }
        for (CpShard &s : p->sh) if ((rc = cp_contribute(p, s, f)) != MI_OK) return rc;
        if (p->exchange == 2) {
            for (CpShard &s : p->sh) if ((rc = cp_p2p_column(p, s, epoch, true, false)) != MI_OK) return rc;
            for (CpShard &s : p->sh) if ((rc = cp_p2p_column(p, s, epoch, false, true)) != MI_OK) return rc;
        } else {
            int blocks = (int)((p->rows + 255) / 256);
            if (blocks > 256) blocks = 256;
            hipLaunchKernelGGL(k_local_sum, dim3(blocks), dim3(256), 0, st, p->l_bits_all, p->l_bits_sum, p->rows, p->world);
        }
        for (CpShard &s : p->sh) if ((rc = cp_pivot(p, s, f)) != MI_OK) return rc;
        if (p->block > 1 && ++p->j == p->block) {
            for (CpShard &s : p->sh) if ((rc = mi355x_shard_sweep(s.t)) != MI_OK) return rc;
            p->j = 0;
        }
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// apply the pending pivots of an unfinished block (the tableau is whole whenever the host looks)
int cp_flush(mi355x_colpart *p)
{
    if (p->block > 1 && p->j > 0) {
        for (CpShard &s : p->sh) {
            int rc = mi355x_shard_sweep(s.t);
            if (rc != MI_OK) return rc;
        }
        p->j = 0;
    }
    return MI_OK;
}

int cp_status(mi355x_colpart *p, int64_t *n_pivots)
{
    int rc = cp_flush(p);
    if (rc != MI_OK) return rc;
    int st0 = 0;
    int64_t n0 = 0;
    for (size_t i = 0; i < p->sh.size(); ++i) {
        int64_t n = 0;
        const int st = mi355x_tab_sync(p->sh[i].t, &n);
        if (st < 0) return st;
        if (st == kExchangeLost)
            return fail(MI_RCCL_ERROR, "P2P exchange: a peer's data never arrived at shard %d (after %lld pivots)",
                        p->sh[i].index, (long long)n);
        if (i == 0) { st0 = st; n0 = n; }
        else if (st != st0 || n != n0)
            return fail(MI_HIP_ERROR, "shards disagree: (%d, %lld) vs (%d, %lld)", st0, (long long)n0, st, (long long)n);
    }
    if (n_pivots) *n_pivots = n0;
    return st0;
}

// devices for `world` shards of one process: one each when enough are visible, else all logical on 0.
// MI355X_COLPART_FORCE_RCCL=1 (test hook): a single shard also goes through its RCCL communicator
// (all-gather / all-reduce over one rank), so the collective code path runs on a one-GPU box.
bool cp_one_device_each(int world)
{
    if (world == 1) { const char *e = getenv("MI355X_COLPART_FORCE_RCCL"); return e && e[0] == '1'; }
    return device_count_checked() >= world;
}

}  // namespace

extern "C" {

int mi355x_rccl_unique_id(void *id128)
{
    if (!id128) return fail(MI_BAD_ARG, "id128 is NULL");
    RCCL_NEED();
    ncclUniqueId id;
    RCCL_TRY(rccl().GetUniqueId(&id));
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
    return MI_OK;
}

static int cp_create_synthetic(mi355x_colpart **out, int64_t n_vars, int64_t n_cons, uint64_t seed,
                               int world, int rank, int device, const void *id128)
{
    if (!out) return fail(MI_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (world < 1 || n_vars < world || n_cons < 1) return fail(MI_BAD_ARG, "need 1 <= shards <= n_vars and n_cons >= 1");
    const bool mp = rank >= 0;
    if (mp && (rank >= world || (!id128 && g_cp_exchange != 2 && g_cp_exchange != 3)))
        return fail(MI_BAD_ARG, "bad rank / id (a NULL id is accepted in exchange mode 2 only: no communicator, "
                                "the host connects the ranks with mi355x_colpart_p2p_handle / _p2p_connect)");
    if (device_count_checked() <= 0) return fail(MI_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    mi355x_colpart *p = new (std::nothrow) mi355x_colpart;
    if (!p) return fail(MI_NO_MEMORY, "host allocation failed");
    p->world = world;
    p->multi_process = mp;
    // (world == 1 in the one-process-per-GPU form: device-local exchanges, unless the test hook
    // MI355X_COLPART_FORCE_RCCL=1 asks for the one-rank ncclCommInitRank communicator)
    p->rccl = mp ? (world > 1 || cp_one_device_each(1)) : cp_one_device_each(world);
    p->exchange = g_cp_exchange == 3 ? 2 : g_cp_exchange;
    p->p2p_merged = g_cp_exchange != 3;
    // (a lone shard without a communicator exchanges with nobody: it takes the self-push form, whose
    // step is two launches instead of five)
    if (g_cp_exchange == 0 && p->world == 1 && !p->rccl) p->exchange = 2;
    p->compact = true;
    p->rows = n_cons + 1;
    p->var_count = n_vars + n_cons;
    const int first = mp ? rank : 0, last = mp ? rank + 1 : world;
    for (int r = first; r < last; ++r) {
        CpShard s;
        s.index = r;
        s.device = mp ? device : (p->rccl ? r : 0);
        cp_partition(n_vars, world, r, &s.col_begin, &s.col_end);      // compact: the structural columns
        int rc = mi355x_tab_create_synthetic(&s.t, n_vars, n_cons, seed, s.col_begin, s.col_end, s.device);
        if (rc == MI_OK) {
            std::vector<int64_t> cols((size_t)(s.col_end - s.col_begin));
            for (size_t k = 0; k < cols.size(); ++k) cols[k] = s.col_begin + (int64_t)k;
            rc = mi355x_shard_set_compact(s.t, p->var_count, cols.data());
        }
        p->sh.push_back(s);
        if (rc != MI_OK) { cp_free(p); return rc; }
    }
    int rc = cp_finish_setup(p, id128, rank);
    if (rc != MI_OK) { cp_free(p); return rc; }
    *out = p;
    return MI_OK;
}

int mi355x_colpart_create_synthetic(mi355x_colpart **out, int64_t n_vars, int64_t n_cons, uint64_t seed,
                                    int n_devices)
{
    return cp_create_synthetic(out, n_vars, n_cons, seed, n_devices, -1, 0, nullptr);
}

int mi355x_colpart_create_synthetic_rank(mi355x_colpart **out, int64_t n_vars, int64_t n_cons, uint64_t seed,
                                         int world, int rank, int device, const void *id128)
{
    if (rank < 0) return fail(MI_BAD_ARG, "rank < 0");
    return cp_create_synthetic(out, n_vars, n_cons, seed, world, rank, device, id128);
}

int mi355x_colpart_create(mi355x_colpart **out, int64_t rows, int64_t cols, const double *hm,
                          const int64_t *hb, int n_devices)
{
    return mi355x_colpart_create_on(out, rows, cols, hm, hb, n_devices, nullptr);
}

int mi355x_colpart_create_on(mi355x_colpart **out, int64_t rows, int64_t cols, const double *hm,
                             const int64_t *hb, int n_devices, const int *device_ids)
{
    if (!out) return fail(MI_BAD_ARG, "out is NULL");
    *out = nullptr;
    const int64_t m = rows - 1, vc = cols - 1;
    if (!hm || !hb || m < 1 || vc < 1 || n_devices < 1) return fail(MI_BAD_ARG, "bad arguments");
    const int ndev = device_count_checked();
    if (ndev <= 0) return fail(MI_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device_ids) {
        for (int i = 0; i < n_devices; ++i) {
            if (device_ids[i] < 0 || device_ids[i] >= ndev)
                return fail(MI_BAD_ARG, "device %d out of range [0,%d)", device_ids[i], ndev);
            for (int k = 0; k < i; ++k)
                if (device_ids[k] == device_ids[i]) return fail(MI_BAD_ARG, "device %d listed twice", device_ids[i]);
        }
    }
    // compact shards when the basis is a set of exact unit columns (+0.0 in the objective row)
    std::vector<char> basic((size_t)vc, 0);
    bool compact = true;
    for (int64_t i = 0; i < m && compact; ++i) {
        const int64_t b = hb[i];
        if (b < 0 || b >= vc || basic[(size_t)b]) { compact = false; break; }
        basic[(size_t)b] = 1;
        for (int64_t r = 0; r < rows && compact; ++r) {
            double v = hm[r * cols + b];
            uint64_t bits;
            memcpy(&bits, &v, 8);
            if (bits != (r == i ? 0x3FF0000000000000ull : 0ull)) compact = false;
        }
    }
    std::vector<int64_t> dist;                                   // the columns that are distributed
    for (int64_t c = 0; c < vc; ++c)
        if (!compact || !basic[(size_t)c]) dist.push_back(c);
    if (dist.empty()) return fail(MI_BAD_ARG, "no column to distribute");
    if ((int64_t)dist.size() < n_devices) n_devices = (int)dist.size();   // a shard needs a column: fewer shards (mi355x_colpart_info says how many)
    mi355x_colpart *p = new (std::nothrow) mi355x_colpart;
    if (!p) return fail(MI_NO_MEMORY, "host allocation failed");
    p->world = n_devices;
    p->rccl = cp_one_device_each(n_devices);
    p->exchange = g_cp_exchange == 3 ? 2 : g_cp_exchange;
    p->p2p_merged = g_cp_exchange != 3;
    // (a lone shard without a communicator exchanges with nobody: it takes the self-push form, whose
    // step is two launches instead of five)
    if (g_cp_exchange == 0 && p->world == 1 && !p->rccl) p->exchange = 2;
    p->compact = compact;
    p->rows = rows;
    p->var_count = vc;
    std::vector<double> loc;
    for (int r = 0; r < n_devices; ++r) {
        CpShard s;
        s.index = r;
        // one shard per listed device (default 0 .. n-1); logical shards all live on the first
        s.device = device_ids ? device_ids[p->rccl ? r : 0] : (p->rccl ? r : 0);
        int64_t b, e;
        cp_partition((int64_t)dist.size(), n_devices, r, &b, &e);
        const int64_t nloc = e - b;
        s.col_begin = dist[(size_t)b];                           // dense shards: a contiguous block of columns
        s.col_end = s.col_begin + nloc;
        loc.assign((size_t)(rows * (nloc + 1)), 0.0);
        for (int64_t rr = 0; rr < rows; ++rr) {
            for (int64_t k = 0; k < nloc; ++k) loc[(size_t)(rr * (nloc + 1) + k)] = hm[rr * cols + dist[(size_t)(b + k)]];
            loc[(size_t)(rr * (nloc + 1) + nloc)] = hm[rr * cols + vc];   // own copy of the RHS column
        }
        int rc = mi355x_tab_create(&s.t, rows, nloc + 1, loc.data(), hb, s.device);
        if (rc == MI_OK && compact) rc = mi355x_shard_set_compact(s.t, vc, dist.data() + b);
        p->sh.push_back(s);
        if (rc != MI_OK) { cp_free(p); return rc; }
    }
    int rc = cp_finish_setup(p, nullptr, -1);
    if (rc != MI_OK) { cp_free(p); return rc; }
    *out = p;
    return MI_OK;
}

int mi355x_colpart_p2p_handle(mi355x_colpart *p, void *handle64)
{
    if (!p || !handle64) return fail(MI_BAD_ARG, "NULL argument");
    if (p->exchange != 2 || !p->multi_process || p->sh.empty() || !p->sh[0].xch)
        return fail(MI_BAD_ARG, "not a one-process-per-GPU handle in exchange mode 2");
    HIP_TRY(hipSetDevice(p->sh[0].device));
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, p->sh[0].xch));
    memcpy(handle64, &h, 64);
    return MI_OK;
}

int mi355x_colpart_p2p_connect(mi355x_colpart *p, const void *handles)
{
    if (!p || !handles) return fail(MI_BAD_ARG, "NULL argument");
    if (p->exchange != 2 || !p->multi_process || p->sh.empty() || !p->sh[0].xch)
        return fail(MI_BAD_ARG, "not a one-process-per-GPU handle in exchange mode 2");
    if (p->p2p_connected) return MI_OK;
    return cp_p2p_connect(p, static_cast<const char *>(handles));
}

int mi355x_colpart_info(const mi355x_colpart *p, int *n_shards, int *n_devices_used, int *uses_rccl)
{
    if (!p) return fail(MI_BAD_ARG, "handle is NULL");
    if (n_shards) *n_shards = p->world;
    if (n_devices_used) *n_devices_used = p->rccl ? p->world : 1;
    if (uses_rccl) *uses_rccl = p->rccl ? 1 : 0;
    return MI_OK;
}

int mi355x_colpart_solve_async(mi355x_colpart *p, int is_max, double f, int64_t n_pivots, int reset)
{
    if (!p) return fail(MI_BAD_ARG, "handle is NULL");
    if (n_pivots < 0) return fail(MI_BAD_ARG, "n_pivots < 0");
    p->is_max = is_max ? 1 : 0;
    if (reset) {
        int rc = cp_flush(p);
        if (rc != MI_OK) return rc;
        for (CpShard &s : p->sh)
            if ((rc = mi355x_tab_reset(s.t, 0)) != MI_OK) return rc;
    }
    return cp_run(p, f, n_pivots);
}

// pivots per sweep of a shard's slice on this handle (1 = per-pivot updates)
int mi355x_colpart_block_size(const mi355x_colpart *p) { return p ? p->block : 0; }

int mi355x_colpart_exchange_timing_enable(mi355x_colpart *p, int stride, int max_samples)
{
    if (!p || stride < 0 || max_samples < 0 || max_samples > 4096) return fail(MI_BAD_ARG, "bad exchange-timing arguments");
    p->timing_stride = stride;
    for (CpShard &s : p->sh) {
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipStreamSynchronize(s.t->stream));
        s.ev_used = 0;
        while (s.ev.size() < (size_t)max_samples * 4) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            s.ev.push_back(e);
        }
    }
    return MI_OK;
}

int mi355x_colpart_exchange_timing_read(mi355x_colpart *p, int64_t *n_samples, double *allgather_us, double *allreduce_us)
{
    if (!p) return fail(MI_BAD_ARG, "handle is NULL");
    int64_t n = 0;
    double ag = 0.0, ar = 0.0;
    for (CpShard &s : p->sh) {                       // averages over the local shards' samples
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipStreamSynchronize(s.t->stream));
        for (int k = 0; k + 4 <= s.ev_used; k += 4) {
            float a = 0.f, b = 0.f;
            HIP_TRY(hipEventElapsedTime(&a, s.ev[(size_t)k], s.ev[(size_t)k + 1]));
            HIP_TRY(hipEventElapsedTime(&b, s.ev[(size_t)k + 2], s.ev[(size_t)k + 3]));
            ag += a * 1e3;
            ar += b * 1e3;
            ++n;
        }
        s.ev_used = 0;
    }
    if (n_samples) *n_samples = n;
    if (allgather_us) *allgather_us = n ? ag / n : 0.0;
    if (allreduce_us) *allreduce_us = n ? ar / n : 0.0;
    return MI_OK;
}

int mi355x_colpart_sync(mi355x_colpart *p, int64_t *n_pivots)
{
    if (!p) return fail(MI_BAD_ARG, "handle is NULL");
    return cp_status(p, n_pivots);
}

int mi355x_colpart_solve(mi355x_colpart *p, int is_max, double f, int64_t max_pivots, int64_t *n_pivots)
{
    if (!p) return fail(MI_BAD_ARG, "handle is NULL");
    if (max_pivots < 0) return fail(MI_BAD_ARG, "max_pivots < 0");
    CancelScope cancel_scope(p->cancel);
    p->is_max = is_max ? 1 : 0;
    int rc = cp_flush(p);
    if (rc != MI_OK) return rc;
    for (CpShard &s : p->sh)
        if ((rc = mi355x_tab_reset(s.t, max_pivots)) != MI_OK) return rc;
    // blind enqueue in chunks (an iteration after termination is a no-op on the device), one
    // status read-back per chunk -- the same decision on every shard / rank
    int64_t chunk = 64;
    for (;;) {
        rc = cp_run(p, f, chunk);
        if (rc != MI_OK) return rc;
        const int st = cp_status(p, n_pivots);
        if (st != MI_RUNNING) return st;
        // (cp_status has applied the pending pivots of an open block: the tableau is whole)
        if (p->cancel.exchange(0, std::memory_order_acq_rel)) return MI_CANCELLED;
        if (chunk < 256) chunk *= 2;
    }
}

int mi355x_colpart_cancel(mi355x_colpart *p)
{
    if (!p) return fail(MI_BAD_ARG, "handle is NULL");
    // one process per GPU: the ranks would have to take the decision at the same chunk, or the
    // collectives of the others never complete -- bound such solves with max_pivots instead
    if (p->multi_process)
        return fail(MI_UNSUPPORTED, "cancel is not available on a one-process-per-GPU handle: solve in max_pivots chunks");
    p->cancel.store(1, std::memory_order_release);
    return MI_OK;
}

}  // extern "C"

namespace {

// fn(shard) on every local shard: one after the other, except over RCCL with several shards in
// this process, where every shard's collectives must be issued from its own thread
template <class F> int cp_each_shard(mi355x_colpart *p, F fn)
{
    if (!p->rccl || p->sh.size() == 1) {
        for (CpShard &s : p->sh) {
            const int rc = fn(s);
            if (rc != MI_OK) return rc;
        }
        return MI_OK;
    }
    std::vector<int> rcs(p->sh.size(), MI_OK);
    std::vector<std::string> errs(p->sh.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < p->sh.size(); ++i)
        th.emplace_back([&, i]() { rcs[i] = fn(p->sh[i]); if (rcs[i] != MI_OK) errs[i] = g_err; });
    for (auto &x : th) x.join();
    for (size_t i = 0; i < p->sh.size(); ++i)
        if (rcs[i] != MI_OK) { cp_abort(p); g_err = errs[i]; return rcs[i]; }
    return MI_OK;
}

// n-pivot-row with a column AND row the caller chose (src/simplex.lisp:434), on every shard: the
// owner contributes the column, exchange B delivers it, every shard normalises its slice of the
// row and updates its slice (per-pivot kernels; no pending block may be open)
int cp_forced_pivot(mi355x_colpart *p, int64_t ec, int64_t cr)
{
    if (p->dead) return fail(MI_RCCL_ERROR, "this handle's communicators were aborted after an earlier failure");
    auto contribute = [&](CpShard &s) -> int {
        mi355x_tab *t = s.t;
        HIP_TRY(hipSetDevice(s.device));
        t->v.col_bias = t->v.p2l ? 0 : s.col_begin;
        launch_shard_forced_contribute(t->v, ec, s.col_begin, (int64_t *)s.bits, s.ec, t->stream);
        HIP_TRY(hipGetLastError());
        return MI_OK;
    };
    auto pivot = [&](CpShard &s) -> int {
        mi355x_tab *t = s.t;
        HIP_TRY(hipSetDevice(s.device));
        launch_shard_prepare(t->v, reinterpret_cast<const double *>(s.bits_in), s.ec, 1024.0, t->stream, cr);
        (void)launch_update(t->v, 1.0, 0, t->stream);
        t->n_part = 0;
        HIP_TRY(hipGetLastError());
        return MI_OK;
    };
    const unsigned epoch = ++p->xepoch;
    if (!p->rccl) {
        int rc;
        for (CpShard &s : p->sh) if ((rc = contribute(s)) != MI_OK) return rc;
        if (p->exchange == 2) {
            for (CpShard &s : p->sh) if ((rc = cp_p2p_column(p, s, epoch, true, false)) != MI_OK) return rc;
            for (CpShard &s : p->sh) if ((rc = cp_p2p_column(p, s, epoch, false, true)) != MI_OK) return rc;
        } else {
            int blocks = (int)((p->rows + 255) / 256);
            if (blocks > 256) blocks = 256;
            hipLaunchKernelGGL(k_local_sum, dim3(blocks), dim3(256), 0, p->sh[0].t->stream, p->l_bits_all, p->l_bits_sum,
                               p->rows, p->world);
        }
        for (CpShard &s : p->sh) if ((rc = pivot(s)) != MI_OK) return rc;
        return MI_OK;
    }
    return cp_each_shard(p, [&](CpShard &s) -> int {
        int rc = contribute(s);
        if (rc != MI_OK) return rc;
        if (p->exchange == 2) { if ((rc = cp_p2p_column(p, s, epoch, true, true)) != MI_OK) return rc; }
        else RCCL_TRY(rccl().AllReduce(s.bits, s.bits, (size_t)p->rows, ncclInt64, ncclSum, s.comm, s.t->stream));
        return pivot(s);
    });
}

}  // namespace

extern "C" {

int mi355x_colpart_solve_two_phase(mi355x_colpart *art, int64_t main_cols, const double *main_obj,
                                   int main_is_max, double f, int64_t *n_pivots, mi355x_colpart **main_out)
{
    if (main_out) *main_out = nullptr;
    if (n_pivots) { n_pivots[0] = 0; n_pivots[1] = 0; }
    if (!art || !main_obj || !main_out) return fail(MI_BAD_ARG, "NULL argument");
    const int64_t rows = art->rows, m = rows - 1, num_art_vars = art->var_count, num_vars = main_cols - 1;
    if (num_vars < 1 || num_vars > num_art_vars)
        return fail(MI_BAD_ARG, "main tableau has %lld columns, the artificial one %lld", (long long)main_cols,
                    (long long)(num_art_vars + 1));
    if (art->multi_process || (int)art->sh.size() != art->world)
        return fail(MI_UNSUPPORTED, "the two-phase hand-over needs every shard in this process");
    if (!art->compact)
        return fail(MI_UNSUPPORTED, "the artificial tableau's basis is not a set of unit columns (dense shards): "
                                    "the column-parallel hand-over does not apply");
    int64_t n1 = 0, n2 = 0;
    int rc = mi355x_colpart_solve(art, /*is_max=*/0, f, 0, &n1);                  // simplex.lisp:403
    if (n_pivots) n_pivots[0] = n1;
    if (rc != MI_OPTIMAL) return rc;
    std::vector<int64_t> basis((size_t)std::max<int64_t>(m, 1));
    std::vector<double> last_col((size_t)rows);
    rc = mi355x_colpart_download(art, nullptr, basis.data(), nullptr, last_col.data());
    if (rc != MI_OK) return rc;
    // (fp= 0 objective factor)                                                    simplex.lisp:405-407
    const double diff = 0.0 - last_col[(size_t)m];
    if (!((diff < 0.0 ? -diff : diff) <= f * kClEpsilon)) return MI_INFEASIBLE;
    // degenerate artificials still basic: pivot them out                        simplex.lisp:419-434
    bool reset_done = false;
    std::vector<double> rowbuf;
    std::vector<int64_t> gcols;
    for (int64_t i = 0; i < m; ++i) {
        if (basis[(size_t)i] < num_vars) continue;
        // the RHS entry of row i (every shard holds the RHS column)
        {
            CpShard &s0 = art->sh[0];
            HIP_TRY(hipSetDevice(s0.device));
            double rhs_i = 0.0;
            HIP_TRY(hipMemcpyAsync(&rhs_i, s0.t->v.M + i * s0.t->v.ld + (s0.t->v.cols - 1), sizeof(double),
                                   hipMemcpyDeviceToHost, s0.t->stream));
            HIP_TRY(hipStreamSynchronize(s0.t->stream));
            if (rhs_i != 0.0) return MI_ART_NONZERO;
        }
        // first non-basic column of the main problem with a non-zero entry in row i: only stored
        // columns can qualify (a basic column other than basis[i] holds +0 there)
        int64_t new_col = -1;
        double new_val = 0.0;
        for (CpShard &s : art->sh) {
            const int64_t nloc = s.t->v.cols - 1;
            gcols.resize((size_t)nloc);
            rowbuf.resize((size_t)nloc);
            if ((rc = mi355x_shard_columns(s.t, gcols.data())) != MI_OK) return rc;
            HIP_TRY(hipMemcpyAsync(rowbuf.data(), s.t->v.M + i * s.t->v.ld, nloc * sizeof(double),
                                   hipMemcpyDeviceToHost, s.t->stream));
            HIP_TRY(hipStreamSynchronize(s.t->stream));
            for (int64_t k = 0; k < nloc; ++k) {
                const int64_t g = gcols[(size_t)k];
                if (g >= 0 && g < num_vars && rowbuf[(size_t)k] != 0.0 && (new_col < 0 || g < new_col)) {
                    new_col = g;
                    new_val = rowbuf[(size_t)k];
                }
            }
        }
        if (new_col < 0) return MI_ART_STUCK;
        // A NEGATIVE pivot element turns the +0 entries the basic columns hold in the pivot row into
        // -0.0 (+0 / negative), and compact shards store no basic column: the bits of the reference's
        // tableau cannot be kept.  (The ratio test only ever picks positive pivot elements; a
        // drive-out pivot takes whatever is there.)  The caller's tableaux are untouched: it solves
        // them on one device, where the drive-out pivots run on the dense tableau.
        if (new_val < 0.0)
            return fail(MI_UNSUPPORTED, "drive-out pivot on a negative element (row %lld, column %lld): not "
                                        "representable on compact column shards", (long long)i, (long long)new_col);
        if (!reset_done) {                                    // phase 1 left every shard's status at OPTIMAL
            for (CpShard &s : art->sh)
                if ((rc = mi355x_tab_reset(s.t, 0)) != MI_OK) return rc;
            reset_done = true;
        }
        rc = cp_forced_pivot(art, new_col, i);
        if (rc != MI_OK) return rc;
        basis[(size_t)i] = new_col;
        ++n1;
    }
    if (n_pivots) n_pivots[0] = n1;
    if (reset_done) {                                         // a non-finite column in a drive-out pivot
        int64_t np = 0;
        const int st = cp_status(art, &np);
        if (st != MI_RUNNING) return st < 0 ? st : (st == MI_NONFINITE ? MI_NONFINITE : fail(MI_HIP_ERROR, "drive-out pivot failed (status %d)", st));
    }
    // ---- hand-over (simplex.lisp:437-451): the main tableau's shards are the artificial shards'
    // main-problem columns, gathered slot by slot on each device; the objective row is the main
    // tableau's own, re-eliminated over the basic rows
    // The column-parallel re-elimination takes every scale from the ORIGINAL objective row; the
    // reference (simplex.lisp:447) reads it from the row as reduced so far.  The two agree while the
    // basic columns are exact unit columns AND every product scale * (+0) is +0 -- an inf / NaN
    // objective coefficient on a basic column would turn later scales into NaNs there.  Declined
    // like the negative drive-out pivot: the caller solves on one device (sequential form).
    for (int64_t i = 0; i < m; ++i) {
        const double sc = main_obj[basis[(size_t)i]];
        if (!((sc < 0.0 ? -sc : sc) <= 1.7976931348623157e308))
            return fail(MI_UNSUPPORTED, "non-finite objective coefficient on basic column %lld: the column-parallel "
                                        "hand-over does not apply", (long long)basis[(size_t)i]);
    }
    mi355x_colpart *mp = new (std::nothrow) mi355x_colpart;
    if (!mp) return fail(MI_NO_MEMORY, "host allocation failed");
    mp->world = art->world;
    mp->rccl = art->rccl;
    mp->compact = true;
    mp->rows = rows;
    mp->var_count = num_vars;
    mp->exchange = art->exchange;
    mp->xepoch = 0;                                           // (its own buffers, zeroed: tags start over)
    std::vector<double> scales((size_t)std::max<int64_t>(m, 1));
    for (int64_t i = 0; i < m; ++i) scales[(size_t)i] = main_obj[basis[(size_t)i]];
    std::vector<int64_t> keep, newcols;
    std::vector<double> obj0;
    for (CpShard &as : art->sh) {
        const int64_t nloc = as.t->v.cols - 1;
        gcols.resize((size_t)nloc);
        rc = mi355x_shard_columns(as.t, gcols.data());
        if (rc != MI_OK) { cp_free(mp); return rc; }
        keep.clear(); newcols.clear(); obj0.clear();
        for (int64_t k = 0; k < nloc; ++k)
            if (gcols[(size_t)k] >= 0 && gcols[(size_t)k] < num_vars) {
                keep.push_back(k); newcols.push_back(gcols[(size_t)k]); obj0.push_back(main_obj[gcols[(size_t)k]]);
            }
        if (keep.empty()) {                                   // nothing but artificial columns here: keep ONE slot as a
            keep.push_back(0); newcols.push_back(-1); obj0.push_back(0.0);   // dead slot (a shard stores a column)
        }
        obj0.push_back(main_obj[num_vars]);
        const int64_t nk = (int64_t)keep.size();
        CpShard ns;
        ns.index = as.index;
        ns.device = as.device;
        ns.col_begin = 0; ns.col_end = nk;
        rc = alloc_tab(&ns.t, rows, nk + 1, ns.device);
        mp->sh.push_back(ns);
        if (rc != MI_OK) { cp_free(mp); return rc; }
        mi355x_tab *nt = mp->sh.back().t;
        int64_t *d_keep = nullptr;
        double *d_obj0 = nullptr, *d_scales = nullptr;
        hipError_t e = hipMalloc((void **)&d_keep, nk * sizeof(int64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&d_obj0, (nk + 1) * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_scales, std::max<int64_t>(m, 1) * sizeof(double));
        hipStream_t st = as.t->stream;                        // behind everything phase 1 enqueued on this shard
        if (e == hipSuccess) e = hipMemcpyAsync(d_keep, keep.data(), nk * sizeof(int64_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_obj0, obj0.data(), (nk + 1) * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess && m > 0) e = hipMemcpyAsync(d_scales, scales.data(), m * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(nt->stream);          // alloc_tab's memsets
        if (e == hipSuccess) e = hipMemsetAsync(nt->v.M, 0, (size_t)rows * nt->v.ld * sizeof(double), st);   // padding columns
        if (e == hipSuccess) {
            launch_shard_handover(as.t->v, nt->v, d_keep, d_obj0, d_scales, st);
            launch_ctl_reset(nt->v, 0, 1, st);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_keep); (void)hipFree(d_obj0); (void)hipFree(d_scales);
        if (e != hipSuccess) { cp_free(mp); return fail(MI_HIP_ERROR, "hand-over failed: %s", hipGetErrorString(e)); }
        rc = mi355x_shard_set_compact(nt, num_vars, newcols.data());
        if (rc != MI_OK) { cp_free(mp); return rc; }
    }
    rc = cp_finish_setup(mp, nullptr, -1, /*make_comms=*/false);
    if (rc != MI_OK) { cp_free(mp); return rc; }
    // the communicators move to the main tableau: the artificial one can still be read, not solved
    for (size_t i = 0; i < art->sh.size(); ++i) { mp->sh[i].comm = art->sh[i].comm; art->sh[i].comm = nullptr; }
    art->dead = art->rccl;
    *main_out = mp;
    rc = mi355x_colpart_solve(mp, main_is_max, f, 0, &n2);                       // simplex.lisp:452
    if (n_pivots) n_pivots[1] = n2;
    return rc;
}

int mi355x_colpart_trace(mi355x_colpart *p, int64_t *ecs, int64_t *crs, int64_t cap, int64_t *n)
{
    if (!p) return fail(MI_BAD_ARG, "handle is NULL");
    int rc = cp_flush(p);
    if (rc != MI_OK) return rc;
    return mi355x_tab_trace(p->sh[0].t, ecs, crs, cap, n);
}

int mi355x_colpart_download(mi355x_colpart *p, double *hm, int64_t *hb, double *last_row, double *last_col)
{
    if (!p) return fail(MI_BAD_ARG, "handle is NULL");
    int rc = cp_flush(p);
    if (rc != MI_OK) return rc;
    const int64_t rows = p->rows, vc = p->var_count, cols = vc + 1, m = rows - 1;
    const bool whole = (int)p->sh.size() == p->world;
    if ((hm || last_row) && !whole)
        return fail(MI_UNSUPPORTED, "the logical tableau / objective row need every shard in this process");
    std::vector<int64_t> basis((size_t)std::max<int64_t>(m, 1));
    std::vector<double> loc;
    std::vector<int64_t> gcols;
    if (hm) std::fill(hm, hm + rows * cols, 0.0);
    if (last_row) std::fill(last_row, last_row + cols, 0.0);
    for (size_t si = 0; si < p->sh.size(); ++si) {
        CpShard &s = p->sh[si];
        const int64_t nloc = s.t->v.cols - 1;
        const bool first = si == 0;
        if (hm || last_row) {
            gcols.resize((size_t)nloc);
            if (p->compact) { if ((rc = mi355x_shard_columns(s.t, gcols.data())) != MI_OK) return rc; }
            else for (int64_t k = 0; k < nloc; ++k) gcols[(size_t)k] = s.col_begin + k;
        }
        if (hm) {
            loc.resize((size_t)(rows * (nloc + 1)));
            rc = mi355x_tab_download(s.t, loc.data(), first ? basis.data() : nullptr, nullptr, nullptr);
            if (rc != MI_OK) return rc;
            for (int64_t r = 0; r < rows; ++r) {
                for (int64_t k = 0; k < nloc; ++k)
                    if (gcols[(size_t)k] >= 0) hm[r * cols + gcols[(size_t)k]] = loc[(size_t)(r * (nloc + 1) + k)];   // (< 0: a dead slot)
                if (first) hm[r * cols + vc] = loc[(size_t)(r * (nloc + 1) + nloc)];
            }
            if (last_row) for (int64_t k = 0; k < nloc; ++k)
                if (gcols[(size_t)k] >= 0) last_row[gcols[(size_t)k]] = loc[(size_t)(m * (nloc + 1) + k)];
            if (last_row && first) last_row[vc] = loc[(size_t)(m * (nloc + 1) + nloc)];
        } else if (last_row) {
            loc.resize((size_t)(nloc + 1));
            rc = mi355x_tab_download(s.t, nullptr, first ? basis.data() : nullptr, loc.data(), nullptr);
            if (rc != MI_OK) return rc;
            for (int64_t k = 0; k < nloc; ++k)
                if (gcols[(size_t)k] >= 0) last_row[gcols[(size_t)k]] = loc[(size_t)k];
            if (first) last_row[vc] = loc[(size_t)nloc];
        } else if (first && (hb || last_col)) {
            rc = mi355x_tab_download(s.t, nullptr, basis.data(), nullptr, nullptr);
            if (rc != MI_OK) return rc;
        }
    }
    if (hm && p->compact)                                        // basic columns are stored nowhere: unit vectors
        for (int64_t i = 0; i < m; ++i) hm[i * cols + basis[(size_t)i]] = 1.0;
    if (hb) std::copy(basis.begin(), basis.begin() + m, hb);
    if (last_col) {
        rc = mi355x_tab_download(p->sh[0].t, nullptr, nullptr, nullptr, last_col);   // every shard holds the RHS column
        if (rc != MI_OK) return rc;
    }
    return MI_OK;
}

void mi355x_colpart_destroy(mi355x_colpart *p) { cp_free(p); }

// tuning hooks, used by bench.py / the microbenchmark only (not part of the reference boundary)
int         mi355x_tune_variant_count(void) { return update_variant_count(); }
const char *mi355x_tune_variant_name(int v) { return (v >= 0 && v < update_variant_count()) ? update_variant_name(v) : ""; }
int         mi355x_tune_set_variant(int v) { set_update_variant(v); return get_update_variant(); }
int         mi355x_tune_set_select_mode(int mode) { g_select_mode = mode; return g_select_mode; }
int         mi355x_tune_set_ld_extra(int doubles) { g_ld_extra = (doubles > 0 ? doubles : 0) / 16 * 16; return g_ld_extra; }
int         mi355x_tune_set_handover_mode(int mode) { g_handover_mode = mode; return mode; }
int         mi355x_tune_set_batch_mode(int mode) { g_batch_mode = mode; return g_batch_mode; }
int         mi355x_tune_set_alternate_sweep(int on) { set_alternate_sweep(on); return on; }
// pivots one tableau-update launch of this handle applies in its current representation
int         mi355x_tab_block_size(mi355x_tab *t) { return (t && block_mode(t)) ? block_size(t) : 1; }
int         mi355x_debug_rhs(mi355x_tab *t, double *out, int64_t n, int clear)
{
    double *buf = t->v.rhs ? t->v.rhs : t->v.col;          // batches have no rhs buffer: their col buffer
    if (hipMemcpy(out, buf, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (clear) (void)hipMemset(buf, 0, n * sizeof(double));
    return 0;
}
int         mi355x_tune_set_batch_block(int k) { set_batch_block(k); g_batch_block_k = k; return k; }
int         mi355x_tune_set_resident(int mode) { g_resident_mode = (mode == 1 || mode == 2) ? mode : 0; return g_resident_mode; }
#ifdef MI355X_TEST_HOOKS
int         mi355x_tune_set_resident_fault(int on) { set_resident_fault(on); return on; }
#endif
int         mi355x_tune_set_resident_poll(int mode) { set_resident_poll(mode); return mode; }
int         mi355x_tune_set_resident_lds(int mode) { mode = mode > 0 ? 1 : 0; set_resident_lds(mode); return mode; }
int         mi355x_tab_resident(mi355x_tab *t) { return (t && resident_mode(t)) ? 1 : 0; }
int         mi355x_tune_set_lookahead_mode(int mode) { g_la_mode = mode; return g_la_mode; }
int         mi355x_tune_set_block(int k)
{
    // 0: by size; 1: off; 2 .. 16; wide blocks: 24 or 28 (anything else above 16 -> the next smaller size)
    if (k <= 0) g_block_k = 0;
    else if (k <= kMaxBlock) g_block_k = k;
    else g_block_k = k >= 28 ? 28 : (k >= 24 ? 24 : kMaxBlock);
    return g_block_k;
}
int         mi355x_tune_set_sweep_shape(int tr, int nt) { set_sweep_shape(tr, nt); return tr; }
int         mi355x_tune_set_compact(int on) { g_compact_enabled = on ? 1 : 0; return g_compact_enabled; }
int         mi355x_tune_set_sweep_impl(int impl) { set_sweep_impl(impl); return impl; }
int         mi355x_tune_set_shard_la_split(int mode) { set_shard_la_split(mode); return mode; }
int         mi355x_tune_set_tail_policy(int p) { g_tail_policy = p == 1 ? 1 : 0; return g_tail_policy; }
int         mi355x_tune_set_colpart_exchange(int mode) { g_cp_exchange = (mode >= 1 && mode <= 3) ? mode : 0; return g_cp_exchange; }
/* measurement aid: the sweep of the CURRENT pending list launched n more times (the list is not
 * consumed by a sweep); average launch duration by HIP events.  Leaves the tableau meaningless. */
int         mi355x_debug_repeat_sweep(mi355x_tab *t, int n, double *avg_us)
{
    if (!t || n < 1 || !avg_us || !t->compact) return MI_BAD_ARG;
    if (hipSetDevice(t->device) != hipSuccess) return MI_HIP_ERROR;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return MI_HIP_ERROR;
    for (int i = 0; i < 3; ++i) (void)launch_sweep(t->c, block_size(t), 1.0, t->stream, 0);
    (void)hipEventRecord(a, t->stream);
    for (int i = 0; i < n; ++i) (void)launch_sweep(t->c, block_size(t), 1.0, t->stream, 0);
    (void)hipEventRecord(b, t->stream);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *avg_us = ms * 1e3 / n;
    return MI_OK;
}
// persistent look-ahead: all workgroups on one XCD (1, default) or spread (0); polls before a
// workgroup gives up on a record (0 = default 2^21); test hook: the last workgroup stops
// publishing from step `step_plus_1 - 1` of every block on (0 = off)
int         mi355x_tune_set_la_one_xcd(int on) { set_la_one_xcd(on); return on; }
int         mi355x_tune_set_la_max_spins(unsigned n) { set_la_max_spins(n); return (int)n; }
#ifdef MI355X_TEST_HOOKS
int         mi355x_tune_set_la_fault(int step_plus_1) { set_la_fault(step_plus_1); return step_plus_1; }
#endif
// 1 once an exchange of the persistent look-ahead was lost on this handle (it then stays on the
// two-launch look-ahead)
int         mi355x_tab_la_lost(const mi355x_tab *t) { return (t && t->la_lost) ? 1 : 0; }

}  // extern "C"
