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
//
// ONE translation unit, split by path (each part is included below, in dependency order):
//     kernels_common.inc          reductions, pricing candidates + NaN rules, gathers, column maps
//     kernels_select_update.inc   per-pivot path: k_select*, step-wise pieces, per-pivot shard steps, k_update
//     kernels_lookahead.inc       pending-pivot chain, k_la_gather / k_la_scale
//     kernels_shard.inc           look-ahead step of a column shard (k_shard_la_*, k_shard_p2p_step)
//     kernels_la_block.inc        persistent look-ahead k_la_block + hand-off protocol, k_la_rollback
//     kernels_shard_block.inc     a column shard's look-ahead of a whole block as one persistent launch (k_shard_la_block)
//     kernels_sweep.inc           k_sweep, k_sweep16
//     kernels_batch.inc           k_batch_solve, k_batch_block
//     kernels_resident.inc        k_resident
//     kernels_layout.inc          dense <-> compact, control block, two-phase hand-over, synthetic LPs
//     kernels_launch.inc          host-side launchers, tuning state
#include "simplex_kernels.h"
#include <type_traits>

namespace mi355x {

#include "kernels_common.inc"
#include "kernels_select_update.inc"
#include "kernels_lookahead.inc"
#include "kernels_shard.inc"
#include "kernels_la_block.inc"
#include "kernels_shard_block.inc"
#include "kernels_sweep.inc"
#include "kernels_batch.inc"
#include "kernels_resident.inc"
#include "kernels_layout.inc"
#include "kernels_launch.inc"

}  // namespace mi355x
