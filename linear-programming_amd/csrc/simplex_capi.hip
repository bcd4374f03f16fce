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
#include <chrono>
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

int g_cp_la_block = 0;                    // exchange mode 2, blocked: the shard's look-ahead of a block as ONE persistent
                                          // launch (k_shard_la_block): 0 wherever it fits, 1 never, 2 = 0 (reserved)
int g_cp_self_hop = 0;                    // measurement: a lone shard runs exchange A against its own buffer
unsigned g_cp_p2p_spins = 0;              // polls before a shard gives a peer up (0: 2^24)

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
    unsigned long long *h_seq = nullptr;  // pinned: sequence number of the last control block k_ctl_publish handed over (read_ctls)
    unsigned long long  seq_next = 0;
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
    bool        la_lost = false;          // an exchange of the persistent look-ahead was lost (its
                                          // workgroups were not co-resident): the handle runs the
                                          // two-launch look-ahead -- until la_rearm_in clean blocks have
                                          // gone by, then the persistent form gets another try
    int         la_losses = 0;            // how often that happened on this handle
    int64_t     la_rearm_in = 0;          // two-launch blocks left before the next try
    // which implementation the dispatcher actually enqueued, per launch class (mi355x_tab_path_counts)
    int64_t     path_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool        primed = false;           // every (look-ahead, sweep) form this handle's requests can pick was launched once (prime_block_kernels)
    int         n_timed = 0;
    std::vector<hipEvent_t> ev0, ev1;     // around the update / sweep launches
    int         n_timed_la = 0;
    std::vector<hipEvent_t> la0, la1;     // around the look-ahead of the same blocks
    // mi355x_tab_cancel (any thread): consumed by the blocking solve loop that next looks at it --
    // between two chunks of launches, when everything enqueued has completed
    std::atomic<int> cancel{0};
    // mi355x_solve_two_phase: the call's two handles look at each other's flag too (a caller cannot
    // know which phase is running), and the flags are cleared when the CALL ends, not a phase
    std::atomic<int> *cancel_peer = nullptr;
    bool        cancel_scoped_outside = false;
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

// The entry points, by path (this file stays ONE translation unit):
#include "capi_tab_impl.inc"   // one tableau behind a handle: internal helpers
#include "capi_tab.inc"        // mi355x_tab_*, mi355x_solve_two_phase, mi355x_init
#include "capi_batch.inc"      // mi355x_batch_*, mi355x_multibatch_*
#include "capi_shard.inc"      // mi355x_shard_*
#include "capi_colpart.inc"    // mi355x_colpart_*, mi355x_rccl_unique_id
#include "capi_tune.inc"       // mi355x_tune_*, mi355x_debug_*
