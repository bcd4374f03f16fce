/* mi355x_simplex_tune.h -- tuning, measurement and test hooks of libmi355x_simplex.so.
 *
 * NOT part of the drop-in boundary (that is mi355x_simplex.h, every entry of which restates a
 * function of the reference): nothing here corresponds to anything in
 * neil-lindquist/linear-programming.  bench.py, tools/ and tests/ use these to pick between
 * implementations of the same bit-identical path and to read back measurement data.
 *
 * State model.  The knobs that choose WHICH implementation of the solve loop runs (select mode,
 * compact, block, look-ahead mode, tail policy, hand-over mode, batch mode, batch block, resident,
 * column-partition exchange) are SNAPSHOTTED INTO A HANDLE WHEN IT IS CREATED: set them, then create
 * the handle that should run that way; a handle never changes path because another thread turned
 * a knob, so the per-handle thread-safety promise of mi355x_simplex.h holds whatever other threads
 * do with these hooks.  The remaining hooks (update-kernel tiling, sweep shape / implementation,
 * shard look-ahead split, one-XCD placement, poll bounds, resident poll mode) are
 * process-wide and read at every launch: measurement and test use only, change them while no other
 * thread is inside the library.  Every hook returns the value now in effect.
 */
#ifndef MI355X_SIMPLEX_TUNE_H
#define MI355X_SIMPLEX_TUNE_H

#include "mi355x_simplex.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the per-pivot update kernel (k_update): compiled tilings ------------------------- */
int         mi355x_tune_variant_count(void);
const char *mi355x_tune_variant_name(int variant);
int         mi355x_tune_set_variant(int variant);            /* 0 = default (by size)           */
int         mi355x_tune_set_alternate_sweep(int on);         /* odd launches sweep bottom-up     */
int         mi355x_tune_set_ld_extra(int doubles);           /* extra row padding, multiple of 16 */

/* ---- which implementation of the solve loop runs -------------------------------------- */
int         mi355x_tune_set_select_mode(int mode);           /* 0 auto, 1 one workgroup, 2 split */
int         mi355x_tune_set_compact(int on);                 /* 1: [non-basic | RHS] (default)   */
int         mi355x_tune_set_block(int k);                    /* pivots per sweep: 0 by size (default: 16,
                                                                or a wide block of 28 where the sweep
                                                                dominates), 1 off, 2 .. 16, 24, 28    */
int         mi355x_tune_set_lookahead_mode(int mode);        /* 0 auto, 1 two launches per step,
                                                                2 one persistent launch per block
                                                                (tableaux of up to 64 look-ahead
                                                                workgroups = 16384 rows and 32768
                                                                stored columns; auto: of those with more
                                                                than 32, the ones below 2e9 bytes stored) */
int         mi355x_tune_set_sweep_shape(int rows_per_workgroup, int nontemporal /* -1 by size */);
int         mi355x_tune_set_sweep_impl(int impl);            /* 0 k_sweep16 for full blocks, 1 k_sweep
                                                                always, 3 the two-launch form of the wide
                                                                sweeps (k_sweepw<16> + k_sweepw_rest);
                                                                4 / 8: rows per step of k_sweep16      */
int         mi355x_tune_set_sweepw_ring(int on);             /* wide sweeps (k > 16 pivots per pass): 1
                                                                (default) the tile's rows travel through a
                                                                per-wave LDS ring (k_sweepw_ring), 0 the
                                                                register form of round 4 (k_sweepw); A/B of
                                                                the ring's cache policy: 2 no non-temporal
                                                                access, 3 non-temporal stores only        */
int         mi355x_tune_set_prime(int on);                   /* 1 (default): a handle's first block is
                                                                preceded by one EMPTY block of every kernel
                                                                form its requests can pick, so that no later
                                                                request pays a kernel's first launch; 0 off */
int         mi355x_tune_set_sweep_xcd_map(int on);           /* k_sweepw_ring: 1 the (strip, tile) pairs in
                                                                strip-major order cut into one run per XCD
                                                                (traffic 1.06 x instead of 1.14 x, 1.5 %
                                                                slower), 0 (default) the grid's own order */
int         mi355x_tune_set_sweep_skew(int rows);            /* k_sweepw_ring where ONE round of workgroups (three
                                                                per CU) covers the tableau: the thirds of the
                                                                tiles in dispatch order are tr + rows / tr /
                                                                tr - rows tall -- a SIMD serves its oldest wave
                                                                first.  -1 (default): 5/16 of the tile height
                                                                (config 3: 28 of 92, 108.5 -> 102.5 us per
                                                                pass); 0: equal tiles */
int         mi355x_tune_set_ctl_wait(int mode);              /* how a status read-back waits: 2 (default) a
                                                                kernel publishes the control block to pinned
                                                                memory and the host polls its sequence number,
                                                                1 copy + polled hipStreamQuery, 0 copy +
                                                                hipStreamSynchronize                      */
int         mi355x_tune_set_shard_la_split(int mode);        /* column shards, local look-ahead step:
                                                                0 by size, 1 one workgroup, 2 many */
int         mi355x_tune_set_colpart_exchange(int mode);      /* column partition over RCCL, how the
                                                                entering column travels (read when a
                                                                handle is created): 0 int64 SUM
                                                                all-reduce of owner's bits + zeros
                                                                (no host sync), 1 ncclBroadcast from
                                                                the owner (root read back from the
                                                                all-gather: one host sync per pivot),
                                                                2 no collective at all: the shards
                                                                write pricing pairs and the entering
                                                                column straight into each other's
                                                                fine-grained buffers (P2P / IPC over
                                                                xGMI), self-validating granules; a
                                                                shard that has its device to itself
                                                                steps in two launches, 3 the same
                                                                with the four launches per step of
                                                                shards that share a stream         */
/* one process per GPU in exchange mode 2: the ranks' exchange buffers are mapped into each other's
 * address space through IPC handles.  With a communicator (an id from mi355x_rccl_unique_id) the
 * library all-gathers the handles itself when the handle is created; with id128 == NULL no
 * communicator exists at all (RCCL is not touched) and the host connects the ranks: every rank
 * takes its 64-byte handle, the host gathers them in rank order (world x 64 bytes) and hands the
 * table to every rank before the first solve. */
int         mi355x_colpart_p2p_handle(mi355x_colpart *p, void *handle64);
int         mi355x_colpart_p2p_connect(mi355x_colpart *p, const void *handles);
int         mi355x_tune_set_tail_policy(int p);              /* n pivots, n not a multiple of the block:
                                                                0 spread evenly, 1 full blocks + remainder */
int         mi355x_tune_set_handover_mode(int mode);         /* 0 auto, 1 sequential re-elimination */
int         mi355x_tune_set_batch_mode(int mode);            /* 0 auto, 1 lockstep, 2 all in one workgroup
                                                                per LP, 3 look-ahead per LP + sweeps over all LPs */
int         mi355x_tune_set_batch_block(int k);              /* blocked per-LP kernel, 1 = off   */
int         mi355x_tune_set_resident(int mode);              /* resident solve (the stored tableau in
                                                                registers, one exchange per pivot):
                                                                0 auto = whenever the shape fits AND
                                                                every knob above is at its default,
                                                                1 never, 2 whenever the shape fits  */
int         mi355x_tune_set_resident_poll(int mode);         /* who polls the exchange records:
                                                                0 by size (every wave when an LP has
                                                                <= 8 workgroups), 1 wave 0, 2 every */
int         mi355x_tune_set_resident_lds(int mode);          /* 1: 24 of a strip's 64 columns in LDS (three
                                                                workgroups per CU; shapes of <= 256
                                                                constraints) -- measured NOT faster, see
                                                                kernels_launch.inc; 0 (default): never */
/* 1 when the solve entry points would run this handle (in its current representation) resident */
int         mi355x_tab_resident(mi355x_tab *t);

/* ---- persistent look-ahead (k_la_block) ------------------------------------------------ */
int         mi355x_tune_set_la_one_xcd(int on);              /* 1 (default): all its workgroups on
                                                                one XCD, verified inside the launch
                                                                (up to 32 workgroups; more are spread
                                                                over the chip whatever this says)   */
int         mi355x_tune_set_la_max_spins(unsigned polls);    /* polls before a workgroup gives up
                                                                on a record; 0 = default (2^21)  */
/* How often an exchange of the persistent look-ahead was lost on this handle (0 = never).  After a
 * loss the solve carries on on the two-launch look-ahead; the persistent form is re-armed after 1024
 * clean blocks (four times as many after each further loss).  mi355x_debug_set_la_rearm overrides the
 * number of clean blocks left (tests). */
int         mi355x_tab_la_lost(const mi355x_tab *t);
int         mi355x_debug_set_la_rearm(mi355x_tab *t, int64_t blocks);
/* Which implementation the dispatcher actually enqueued on this handle so far, launches by class:
 * out8[0] per-pivot updates (k_update), [1] persistent look-ahead blocks (k_la_block), [2] two-launch
 * look-ahead blocks (k_la_gather / k_la_scale per step), [3] k_sweep16 sweeps, [4] wide sweeps
 * (k_sweepw + k_sweepw_rest), [5] short sweeps (k_sweep), [6] resident launches (k_resident),
 * [7] split selects (k_select_gather + k_select_scale). */
int         mi355x_tab_path_counts(const mi355x_tab *t, int64_t *out8);

/* ---- measurement ------------------------------------------------------------------------ */
/* HIP-event brackets on the handle's stream (what `mi355x_tab_timing_*` of the main header
 * brackets is the tableau update / sweep; `which` selects another kernel class):
 * 0 update / sweep, 1 look-ahead (select) */
int         mi355x_tab_timing_read_kind(mi355x_tab *t, int which, int64_t *n_launches,
                                        double *sum_ms, double *min_ms);
/* column partition over RCCL: HIP-event brackets around the two per-pivot collectives of every
 * `stride`-th pivot (at most max_samples per shard and per solve call; 0 = off); _read waits for
 * the shards' streams and returns the averages over this process's shards since the last read */
int         mi355x_colpart_block_size(const mi355x_colpart *p);   /* pivots per sweep of a shard's slice */
/* exchange mode 2, blocked shards: the look-ahead of a whole block as ONE persistent launch per device
 * (k_shard_la_block).  out4: [0] blocks enqueued that way, [1] exchanges it lost, [2] the handle is demoted
 * to the two-launch step right now, [3] the next block would take the persistent form */
int         mi355x_colpart_la_stats(mi355x_colpart *p, int64_t *out4);
int         mi355x_colpart_is_compact(const mi355x_colpart *p);   /* 1 compact shards, 0 dense shards */
int         mi355x_colpart_debug_set_la_rearm(mi355x_colpart *p, int64_t blocks);   /* as mi355x_debug_set_la_rearm */
int         mi355x_colpart_debug_rhs(mi355x_colpart *p, int shard, double *out, int64_t n, int clear);   /* mi355x_debug_rhs of a local shard */
int         mi355x_tune_set_shard_la_block(int mode);        /* read when a handle is created: 0 (default) the
                                                                persistent block launch wherever it fits, 1 never */
int         mi355x_tune_set_shard_self_hop(int on);          /* measurement: a LONE shard runs exchange A (its
                                                                pricing pair) against its own buffer -- what a
                                                                shard of several pays minus the wire        */
int         mi355x_tune_set_p2p_spins(unsigned n);           /* polls before a shard gives a peer's granules
                                                                up (read at creation; 0 = default 2^24)      */
int         mi355x_colpart_exchange_timing_enable(mi355x_colpart *p, int stride, int max_samples);
int         mi355x_colpart_exchange_timing_read(mi355x_colpart *p, int64_t *n_samples,
                                                double *allgather_us, double *allreduce_us);
/* measurement aid: launches the sweep of the handle's CURRENT pending list n more times (a sweep
 * does not consume the list) and returns the average launch duration (HIP events).  The tableau
 * is meaningless afterwards. */
int         mi355x_debug_repeat_sweep(mi355x_tab *t, int n, double *avg_us);
/* debugging aid: copies n doubles of the handle's scratch `rhs` buffer (the per-phase clocks of
 * a -DMI355X_LA_TIMING build); clear != 0 zeroes it afterwards */
int         mi355x_debug_rhs(mi355x_tab *t, double *out, int64_t n, int clear);
int         mi355x_debug_last_wait(double *out2);            /* host microseconds of the last status read-back:
                                                                [0] launching k_ctl_publish, [1] polling its
                                                                sequence number (bench.py's steady-state leg)   */

/* ---- fault injection: TEST BUILD ONLY ---------------------------------------------------- */
/* Compiled in with -DMI355X_TEST_HOOKS (libmi355x_simplex_test.so, built next to the product
 * library by linear-programming_amd/build.py and loaded by the tests that need it); the product
 * library neither exports these nor contains the code paths behind them. */
#ifdef MI355X_TEST_HOOKS
int         mi355x_tune_set_resident_fault(int on);          /* the last workgroup of every LP never
                                                                publishes (co-residency lost)    */
int         mi355x_tune_set_shard_la_fault(int step_plus_1); /* the same two faults (below) in the last
                                                                workgroup of the last local shard of a
                                                                k_shard_la_block launch                  */
int         mi355x_tune_set_la_fault(int step_plus_1);       /* > 0: the last workgroup stops
                                                                publishing from that step on; < 0: it
                                                                publishes its ratio record of step
                                                                -step_plus_1 - 1 and then gives up
                                                                alone (the leader commits a pivot
                                                                that workgroup never stored)     */
#endif

#ifdef __cplusplus
}
#endif
#endif
