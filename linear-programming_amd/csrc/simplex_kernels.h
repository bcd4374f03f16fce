// simplex_kernels.h -- device-side data structures and launcher prototypes of the
// MI355X dense-simplex backend (host side of the launchers lives in
// simplex_kernels.hip, the C ABI in simplex_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355x {

// Common Lisp DOUBLE-FLOAT-EPSILON = 2^-53 (1 + 2^-52)  (src/utils.lisp:84-124)
constexpr double kClEpsilon = 1.1102230246251568e-16;

// device-side status word; values >= 0 other than kRunning are the MI_* outcomes
constexpr int32_t kRunning = 100;
// compact representation only: the entering column holds an inf / NaN.  The reference would
// turn every basic column into NaNs (x - inf*0), which a representation that does not store
// basic columns cannot reproduce: the pivot is NOT applied and the host re-runs it on the dense
// tableau (single tableaux, batches) or reports MI_NONFINITE (compact column shards).
constexpr int32_t kNeedDense = 101;
// persistent look-ahead kernel only: a workgroup gave up waiting for the others' records (cannot
// happen while all its workgroups are resident; bounded so that a bug reports instead of hanging)
constexpr int32_t kSyncLost = 102;
// resident solve only: an exchange was lost AFTER the first one (the first is the co-residency
// check and ends as kSyncLost): cannot happen short of a hung GPU; nothing was written back, but
// the launch may have been partly observed -- reported as an error, never as a result
constexpr int32_t kResidentStuck = 103;

// Control block living in device memory: the whole price -> ratio -> pivot loop
// runs without host round trips, kernels communicate through this struct.
struct Ctl {
    int32_t status;       // kRunning, or MI_OPTIMAL / MI_UNBOUNDED / MI_MAX_PIVOTS
    int32_t poison;       // set by k_select_gather when it meets a non-finite column entry
    int64_t ec;           // entering column chosen by the last select
    int64_t cr;           // pivot row chosen by the last select
    int64_t n_pivots;     // pivots since the last reset
    int64_t max_pivots;   // 0 = no cap
    int64_t trace_n;      // pivots recorded in the trace buffers
    int64_t slot;         // physical column of `ec` (== ec unless the representation is compact)
};

// Blocked pivoting (single compact tableaux): up to kMaxBlock pivots are SELECTED ahead of the
// tableau update -- each look-ahead step evaluates just the objective row, one column, the RHS
// column and one row as they would be after the pending pivots -- and then applied to every
// stored element in ONE sweep (k rank-1 updates per element, in pivot order, operands and
// roundings unchanged).  This block lives next to the control block.
constexpr int kMaxBlock = 16;
constexpr int kLaBlockMax = 24;            // pending pivots one launch of the persistent look-ahead can hold (6 KB of LDS each)
// WIDE blocks (round 4): where the sweep dominates an iteration -- tableaux / column shards of a GB
// and more, which the persistent look-ahead does not fit anyway -- up to kWideBlock pivots are
// pending per pass (24 or 28 in practice: what three waves per SIMD can hold of prow operands,
// kernels_sweep.inc).  The list, the col_i / prow_i buffers and the row masks have room for
// kWideBlock pivots on every single-tableau handle; pivots 16 .. 31 record their slot hand-overs in
// a second mask array (bk_smask2, layout of bk_smask).  Everything that lives in LDS (persistent
// look-ahead, batches) stays at kMaxBlock.
constexpr int kWideBlock = 32;
struct BlockCtl {
    int64_t n_pending;            // pivots selected but not yet applied by a sweep
    int64_t stamp;                // persistent look-ahead: epoch base of the launch that wrote the list
    int64_t cr[kWideBlock];       // their pivot rows ...
    int64_t slot[kWideBlock];     // ... and the physical slots their entering columns gave up
    // persistent look-ahead: (stamp << 8 | steps) -- how many steps of the launch with that stamp
    // workgroup w completed (everything it owns of col_i / prow_i stored).  The leader raises
    // n_pending on its own; a workgroup that gave up on an exchange (kSyncLost) may be one step
    // behind it, so the sweep applies min(n_pending, min over w of steps) pivots and the host's
    // recovery takes the bookkeeping of the pivot beyond that back (k_la_rollback).
    int64_t done[256];            // kMaxShardLaWorkgroups entries (single tableaux use the first kMaxLaWorkgroups)
    int64_t ec[kWideBlock];       // persistent look-ahead: the logical columns that entered (k_la_rollback)
    int64_t prev[kWideBlock];     // column shards (k_shard_la_block): basis[cr] as it was before the pivot (k_shard_la_rollback)
};

// Record one workgroup of the persistent look-ahead kernel publishes per exchange: eight
// self-validating granules {tag = 32-bit epoch, 32 bits of payload}, each written by one 8-byte
// store (a reduction candidate (value, index, payload), a flag and two doubles; layout in
// simplex_kernels.hip, la_exchange).  Since round 5 the buffer of kMaxLaRecords records is laid out
// TRANSPOSED -- granule j of record R at 8-byte word j * kMaxLaRecords + R -- so that the collecting
// wave reads consecutive granules with consecutive lanes; ExchRec only sizes the buffer.
struct ExchRec {
    unsigned long long g[8];
};
// Up to kLaWaveRecordsMaxNw workgroups every WAVE publishes a record and every wave polls them all (the
// launch can sit on one XCD: 32 CUs); from there up to kMaxLaWorkgroups every WORKGROUP publishes one
// record (its four waves' winners reduced through LDS) and its first wave polls -- the poll traffic of
// the first form grows with the square of the workgroups, the LDS hop of the second does not
// (DESIGN_experiments.md R5.16 / R5.18).  Both forms use the same transposed buffers.
constexpr int kLaWaveRecordsMaxNw = 32;
constexpr int kMaxLaWorkgroups = 64;
// A column shard's persistent look-ahead (k_shard_la_block, round 6) runs one workgroup per 256 rows of the
// WHOLE column -- 129 for config 5 however many GPUs share it -- with a record (a 64-byte line) per workgroup
// and one polling wave each; the sweeps read BlockCtl::done with up to four entries per lane.
constexpr int kMaxShardLaWorkgroups = 256;
static_assert(sizeof(BlockCtl::done) / sizeof(int64_t) == kMaxShardLaWorkgroups, "BlockCtl::done");
constexpr int kMaxLaRecords = 4 * kLaWaveRecordsMaxNw;  // one record per wave of a 256-thread workgroup / one per workgroup
static_assert(kMaxLaWorkgroups <= 64 && kMaxLaWorkgroups <= kMaxLaRecords,
              "a sweep's wave reads BlockCtl::done with one lane per workgroup; per-workgroup records are collected one per lane");

// Column partition, exchange mode 2 (the shards write into each other's fine-grained buffers):
// layout of one shard's exchange buffer in 8-byte granules {tag = pivot number, 32 bits of payload}
//     pairs  [2 parities][world][4]      the local pricing winners (key, global column)
//     column [2 parities][2 x rows_p]    the entering column
// and what a kernel needs to take part: every shard's buffer as THIS process maps it, the own one,
// the own rank, the tag of this pivot and the poll bound.  peers == nullptr: not in this mode.
constexpr int32_t kExchangeLost = 104;      // device status: a peer's data never arrived (-> MI_RCCL_ERROR)
struct P2pLayout {
    int world; int64_t rows_p;
    __host__ __device__ int64_t pair_off(unsigned par, int r) const { return ((int64_t)par * world + r) * 4; }
    __host__ __device__ int64_t col_off(unsigned par) const { return (int64_t)2 * world * 4 + (int64_t)par * 2 * rows_p; }
    __host__ __device__ int64_t granules() const { return (int64_t)2 * world * 4 + (int64_t)4 * rows_p; }
};
struct P2pArgs {
    unsigned long long *const *peers = nullptr;
    unsigned long long *mine = nullptr;
    P2pLayout lay{};
    int rank = 0;
    unsigned epoch = 0, max_spins = 0;
};

// One tableau in HBM.  Row-major, leading dimension ld (a multiple of 16 doubles so
// that every row starts on a 128-byte boundary and 16-byte vector accesses never
// straddle rows); columns [cols, ld) are padding and hold zeros.
struct TabView {
    double  *M;
    int64_t  ld, rows, cols;
    int64_t *basis;       // rows-1 entries
    double  *col;         // snapshot of the entering column before the pivot, rows entries
    double  *prow;        // normalised pivot row, ld entries (padding zero)
    double  *rhs;         // contiguous snapshot of the RHS column (column shards only), rows entries
    Ctl     *ctl;
    int64_t *trace_ec;    // may be null
    int64_t *trace_cr;
    int64_t  trace_cap;
    double  *part_v;      // per-wave pricing partials left by k_update (key space); the
    int64_t *part_i;      // upper half holds the ratio-test partials of the split select
    int64_t *part_s;      // payload of each partial (physical slot / pivot-element bits)
    int      part_cap;
    // compact representation (non-basic columns + RHS only): physical slot <-> logical column
    // maps; both null for the dense logical layout
    int64_t *p2l;         // cols-1 entries: logical column stored in physical slot j
    int64_t *l2p;         // logical var_count entries: slot of a logical column, -1 if basic
    int64_t  col_bias;    // dense column shard: global index of its column 0 (0 everywhere else)
    // blocked pivoting: entering-column snapshots (kMaxBlock x bk_stride), normalised pivot rows
    // (kMaxBlock x ld of the view in use) and the pending list; null when not available
    double   *bk_col, *bk_prow;
    BlockCtl *blk;
    int64_t   bk_stride;
    // bit i of bk_rmask[r]: row r is the pivot row of pending pivot i; bits i / 16+i of
    // bk_smask[pair]: the even / odd column of that pair is the slot pending pivot i gave up
    uint32_t *bk_rmask, *bk_smask;
    uint32_t *bk_smask2;          // slot hand-overs of pending pivots 16 .. 31 (wide blocks; null for batches)
    ExchRec  *la_px, *la_rx;      // kMaxLaRecords pricing / ratio records (persistent look-ahead)
    // batch of n_lps same-shape LPs: per-LP element strides (all zero for a single tableau)
    int64_t  n_lps;
    int64_t  zs_M, zs_basis, zs_col, zs_prow, zs_part, zs_p2l, zs_l2p;
    int64_t  zs_bk, zs_bkp, zs_rm, zs_sm;       // per-LP block state: bk_col, bk_prow, bk_rmask, bk_smask (blk: one BlockCtl each)
};

// One shard of a k_shard_la_block launch (an array of these in device memory, indexed by blockIdx.y:
// one entry per device-local shard -- one where a shard has its GPU to itself, the logical shards of a
// one-GPU test box all in ONE launch, whose workgroups are then co-resident by construction).
struct ShardLaunch {
    TabView t;
    unsigned long long *const *peers;   // every shard's exchange buffer as this process maps it
    unsigned long long *mine;           // this shard's own
    int64_t col_offset;                 // dense shards: global index of local column 0
    int rank, pad;
};

struct UpdateShape {
    int strips, strip_pairs, tr, row_chunks, waves_per_block, n_partials;
};

// select: price -> gather column -> ratio test -> normalise pivot row (one workgroup)
// n_part > 0: price from the partials the preceding launch_update(..., price=1) left behind
void launch_select(const TabView &t, int is_max, double fp_factor, int n_part, hipStream_t s);
// the same select as two multi-workgroup launches (large tableaux)
void launch_select_split(const TabView &t, int is_max, double fp_factor, int n_part, hipStream_t s);
bool select_split_supported(const TabView &t);
// pieces of it, for the step-wise ABI entry points
void launch_price_only(const TabView &t, int is_max, double fp_factor, hipStream_t s);
void launch_ratio_only(const TabView &t, int64_t ec, double fp_factor, hipStream_t s);
void launch_prepare_pivot(const TabView &t, int64_t ec, int64_t cr, hipStream_t s);
// the bandwidth kernel: M[r][c] -= col[r] * prow[c] (r != cr), M[cr][c] = prow[c]
// returns the number of pricing partials written (0 if price == 0)
int  launch_update(const TabView &t, double sgn, int price, hipStream_t s, int64_t launch_index = 0);
void set_alternate_sweep(int on);
// blocked pivoting: look-ahead step j of a block (select pivot j as if pivots 0..j-1 of the block
// had been applied), and the sweep that applies the whole block.  n_part as for launch_select;
// launch_lookahead returns the number of pricing partials it leaves for step j+1.
bool block_supported(const TabView &t);
int  launch_lookahead(const TabView &t, int j, int is_max, double fp_factor, int n_part, hipStream_t s);
// stamp != 0: apply the pending list only if the look-ahead launch with that epoch base wrote it,
// and of it only the pivots all la_nw workgroups of that launch completed (BlockCtl::done)
int  launch_sweep(const TabView &t, int kmax, double sgn, hipStream_t s, unsigned stamp = 0, int la_nw = 0);
// wide blocks: the block sizes k_sweepw is compiled for, and the one a view of this size should run
// with (0: stay at kMaxBlock)
bool wide_block_size_ok(int k);
int  wide_block_default(const TabView &t);
void set_sweep_skew(int rows);             // k_sweepw_ring, one round of workgroups: rows by which the first / last third of the tiles are taller / shorter (-1: by the tile height)
void set_sweep_xmap(int on);               // k_sweepw_ring: workgroup -> (strip, tile) by XCD (see the kernel)
void set_sweepw_ring(int on);              // wide sweeps through the LDS ring (default) or the register form
int  sweep_kind(int kmax);                 // 2 wide pair, 1 k_sweep16, 0 k_sweep<..>: what launch_sweep picks
// after a lost exchange (kSyncLost): undo the bookkeeping (column maps, basis, pivot count, trace)
// of the pivots the leader committed but the sweep did not apply
void launch_la_rollback(const TabView &t, int la_nw, hipStream_t s);
int  la_block_workgroups(const TabView &t);
// the whole look-ahead of a block (steps 0 .. ksteps-1) as ONE launch of a few persistent
// workgroups that exchange their reduction candidates through la_px / la_rx; epoch_base (> 0)
// must grow by at least 2*kMaxBlock+2 from launch to launch on the same tableau (the records
// are zeroed whenever the host wraps it around)
bool la_block_supported(const TabView &t);
void launch_la_block(const TabView &t, int ksteps, int is_max, double fp_factor,
                     unsigned epoch_base, hipStream_t s);
// the same with the kernel picked by `form` (a block size: <= kMaxBlock the 16-step look-ahead, above
// it the 24-step one) instead of by ksteps -- prime_block_kernels launches each with 0 steps
void launch_la_block_form(const TabView &t, int form, int ksteps, int is_max, double fp_factor,
                          unsigned epoch_base, hipStream_t s);
// tuning / test hooks of the persistent look-ahead: all its workgroups on one XCD (default on),
// polls before a workgroup gives up waiting (kSyncLost), a workgroup that stops publishing
void set_la_one_xcd(int on);
void set_la_max_spins(unsigned n);
void set_la_fault(int step_plus_1);
void set_sweep_shape(int tr, int nt);      // tuning hook
void set_sweep_impl(int impl);             // tuning / test hook: 0 k_sweep16 for full blocks (default), 1 k_sweep always
UpdateShape update_shape(const TabView &t);
// column-partitioned shards (one shard = one handle)
// x.peers != nullptr (exchange mode 2, FUSED form): the kernel also is the producer / consumer of
// the exchange next to it -- the pricing kernel pushes the pair to every shard, the contribution
// kernel waits for all pairs and (the owner) pushes the column, the split look-ahead step waits for
// the column -- so that a pivot is four launches and no collective
void launch_shard_price(const TabView &t, int is_max, int64_t col_offset, double *out2, int n_part,
                        hipStream_t s, const P2pArgs &x = P2pArgs());
void launch_shard_contribute(const TabView &t, const double *gathered, int n_shards,
                             int64_t col_offset, double fp_factor, int64_t *bits_out,
                             int64_t *ec_out, hipStream_t s);
// forced_cr >= 0: no ratio test, the pivot row is the caller's
void launch_shard_prepare(const TabView &t, const double *col, const int64_t *ec_dev,
                          double fp_factor, hipStream_t s, int64_t forced_cr = -1);
// the owner of logical column ec contributes it (a pivot the caller chose)
void launch_shard_forced_contribute(const TabView &t, int64_t ec, int64_t col_offset, int64_t *bits_out,
                                    int64_t *ec_out, hipStream_t s);
// two-phase hand-over of one compact column shard: slots keep[] of `art` + its RHS copy -> `mt`,
// objective row obj0 re-eliminated with the given scales (src/simplex.lisp:437-451)
void launch_shard_handover(const TabView &art, const TabView &mt, const int64_t *keep, const double *obj0,
                           const double *scales, hipStream_t s);
// ... and their blocked forms: step j of a block (no update), the sweep is launch_sweep
void launch_shard_la_contribute(const TabView &t, int j, const double *gathered, int n_shards,
                                int64_t col_offset, double fp_factor, int64_t *bits_out,
                                int64_t *ec_out, hipStream_t s, const P2pArgs &x = P2pArgs());
int  launch_shard_la_prepare(const TabView &t, int j, const double *col, const int64_t *ec_dev,
                             double fp_factor, int is_max, hipStream_t s, const P2pArgs &x = P2pArgs());
int  launch_shard_p2p_step(const TabView &t, int j, int n_part, int n_shards, int64_t col_offset, double fp_factor,
                           int is_max, int64_t *ec_dev, hipStream_t s, const P2pArgs &x);
// the look-ahead of a whole block of every device-local shard as ONE persistent launch (exchange mode 2;
// kernels_shard_block.inc): workgroups a shard needs (0: not available for this view), the launch, and
// the recovery's bookkeeping roll-back
int  shard_la_block_workgroups(const TabView &t);
void launch_shard_la_block(const ShardLaunch *dev_shards, int n_local, int nw_max, const P2pLayout &lay, int ksteps,
                           int is_max, double fp_factor, unsigned epoch_base, unsigned xepoch_base,
                           unsigned p2p_spins, int hop, hipStream_t s);
void launch_shard_la_rollback(const TabView &t, int la_nw, hipStream_t s);
void set_shard_la_fault(int step_plus_1);  // test hook (test build): as set_la_fault, the last workgroup of the last local shard
bool shard_la_split(const TabView &t);     // the look-ahead step of this shard is the multi-workgroup pair
void set_shard_la_split(int mode);         // tuning / test hook: 0 by size, 1 one workgroup, 2 split over many
// two-phase hand-over (src/simplex.lisp:437-451)
// unit_basis: the basic columns of `art` are known to be exact unit vectors (column-parallel
// re-elimination); otherwise the sequential form
void launch_handover(const TabView &art, const TabView &main_tab, bool unit_basis, hipStream_t s);
void launch_handover_objective_columns(const TabView &mt, hipStream_t s);   // scales in mt.col[0 .. m)
// whole-batch solve, one workgroup per LP (false: an LP does not fit the LDS budget)
bool launch_batch_solve(const TabView &t, int is_max, double fp_factor, hipStream_t s);
// one block of every LP of a batch: look-ahead (one workgroup per LP, 16 pivots selected ahead)
// then ONE sweep launch over all LPs (grid.z = LP); false: not available for this shape
bool launch_batch_block_split(const TabView &t, int is_max, double fp_factor, hipStream_t s);
void set_batch_block(int k);       // tuning hook: pivots per pass of the blocked per-LP kernel (1 = off)
// dense logical tableau <-> compact representation
void launch_verify_basis(const TabView &t, int *flag, hipStream_t s);
void launch_compact(const TabView &dense, const TabView &compact, hipStream_t s);
void launch_expand(const TabView &dense, const TabView &compact, int64_t *brow, hipStream_t s);
void launch_ctl_reset(const TabView &t, int64_t max_pivots, int reset_trace, hipStream_t s);
void launch_ctl_finish(const TabView &t, hipStream_t s);
// the first n control blocks -> pinned coherent host memory, then `seq` -> *host_seq (system-scope release)
void launch_ctl_publish(const TabView &t, int64_t n, void *host_dst, unsigned long long *host_seq,
                        unsigned long long seq, hipStream_t s);
// `from` -> kRunning (kNeedDense: after the host has rebuilt the dense tableau; kSyncLost: after
// it has switched the handle to the two-launch look-ahead)
void launch_ctl_resume(const TabView &t, hipStream_t s, int32_t from);
// synthetic LP straight into HBM
void launch_synth_fill(const TabView &t, int64_t n_vars, int64_t n_cons, uint64_t seed,
                       const uint64_t *dev_seeds, int64_t col_begin, int64_t col_end, hipStream_t s);

// The resident solve (tableaux whose stored part fits the register files: DESIGN.md): the strip
// layout of a compact view, the exchange buffer it needs (per handle; zero it once), and ONE launch
// that runs up to `cap` pivots of n-solve-tableau per LP with the tableau on chip.  epoch_base must
// grow by at least cap + 2 from launch to launch on the same buffer.
struct ResidentPlan { int TR, CW, G; int64_t slot_granules, lp_granules; };
struct ResidentArgs {
    double sgn, price_tol, ratio_thr;
    unsigned long long *xbuf;
    int64_t xs_lp, xs_slot;
    int G, cap;
    unsigned epoch_base, spins_first, spins;
    int fault;
};
bool   resident_plan(const TabView &compact, ResidentPlan *p);
size_t resident_xbuf_bytes(const TabView &compact);
bool   launch_resident(const TabView &compact, unsigned long long *xbuf, int is_max, double fp_factor,
                       int cap, unsigned epoch_base, hipStream_t s);
void   set_resident_fault(int on);      // test hook: the last workgroup of every LP never publishes
void   set_resident_lds(int mode);      // tuning hook: 1 = part of the strip in LDS (three workgroups per CU), 0 = never
void   set_resident_poll(int mode);     // tuning hook: who polls the exchange records (0 by size, 1 wave 0, 2 every wave)

int         update_variant_count();
const char *update_variant_name(int v);
void        set_update_variant(int v);      // tuning hook (bench / microbench only)
int         get_update_variant();
const char *update_kernel_symbol();

}  // namespace mi355x
