// shard_step_skel.hip -- what a column shard's look-ahead step would cost as ONE persistent launch
// per block (the "k_shard_la_block" the round-4 review asks for), measured on a skeleton BEFORE
// building it: the memory accesses, the two record exchanges and the dependent chain of a step at the
// shape of one 8-GPU shard of config 5 (32769 x 8193 stored, ld 8194: 2.15 GB), with the arithmetic
// of a step reduced to its dependences.  Not product code and not a solver: the "pivots" are hashes
// of the reductions' winners, so that every workgroup takes the same decisions from the same records
// (as k_la_block does) and the column / row addresses of a step are unknown until its exchange ends.
//
// One step, G workgroups of 256 threads, thread g owning rows g, g + 256 G, ... and column pair g:
//   X1  every wave polls the G pricing records (64 B each: eight self-validating granules, sc1
//       loads, no fence -- the hand-off protocol of kernels_la_block.inc) -> entering column;
//   C   strided read of that column (one entry per owned row), chained through the pending pivots
//       (their col entries / prow entries requested BEFORE the poll: they do not depend on it),
//       ratio test, workgroup reduction, ratio record out;
//   X2  every wave polls the G ratio records -> pivot row;
//   R   contiguous read of that row (pair g < ld / 2), chained through the pending pivots, prow_j
//       written, objective-row pair updated and priced, workgroup reduction, pricing record out.
// The pending pivots' col entry of my row and prow pair of my column pair stay in LDS (as in k_la_block);
// what ANOTHER workgroup needs of them (prow_i[entering column], col_i[pivot row]) goes write-through
// to global memory and comes back through sc1 loads issued together with the column / row load.
// Reductions: DPP inside a row of 16 lanes, one barrier per workgroup reduction; a wave requests all
// the records it polls (up to three per lane) before it waits.  (First form, `profiles/
// r05_shard_step_skeleton_v1.txt`: ds_bpermute reductions, two barriers, records polled 64 at a
// time, pending entries re-read from global memory: 15.2 us per step at 17 workgroups.)
// Modes: 3 = everything, 1 = exchanges only, 2 = memory + chains only (workgroups free-running).
// Spins are bounded: a lost exchange sets err and every workgroup leaves.
//   hipcc -O3 --offload-arch=gfx950 -o shard_step_skel shard_step_skel.hip && ./shard_step_skel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned long long u64;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Rec { u64 g[8]; };
constexpr int K = 24;                                     // pivots per block
constexpr unsigned kMaxSpins = 1u << 18;
__device__ int g_nap = 0;                                 // s_sleep between failed polls (0: none)
__device__ __forceinline__ void nap() { if (g_nap) __builtin_amdgcn_s_sleep(2); }

struct VI { double v; int i; double s; };
__device__ __forceinline__ VI vi_min(const VI &a, const VI &b)
{
    const bool better = (b.v < a.v) | ((b.v == a.v) & (b.i < a.i));
    const bool take_b = (a.i < 0) | ((b.i >= 0) & better);
    VI r; r.v = take_b ? b.v : a.v; r.i = take_b ? b.i : a.i; r.s = take_b ? b.s : a.s;
    return r;
}
template <int CTRL> __device__ __forceinline__ int dpp32(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dppd(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = dpp32<CTRL>((int)(b & 0xffffffffll)), hi = dpp32<CTRL>((int)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
template <int CTRL> __device__ __forceinline__ VI dpp_vi(const VI &x) { VI y; y.v = dppd<CTRL>(x.v); y.i = dpp32<CTRL>(x.i); y.s = dppd<CTRL>(x.s); return y; }
__device__ __forceinline__ VI shfl_vi(const VI &x, int off) { VI y; y.v = __shfl_down(x.v, off, 64); y.i = __shfl_down(x.i, off, 64); y.s = __shfl_down(x.s, off, 64); return y; }
// lane 0 holds the minimum; everybody gets it through readfirstlane
__device__ __forceinline__ VI wave_min(VI x)
{
    x = vi_min(x, shfl_vi(x, 32));
    x = vi_min(x, shfl_vi(x, 16));
    x = vi_min(x, dpp_vi<0x108>(x));
    x = vi_min(x, dpp_vi<0x104>(x));
    x = vi_min(x, dpp_vi<0x102>(x));
    x = vi_min(x, dpp_vi<0x101>(x));
    const long long vb = __double_as_longlong(x.v), sb = __double_as_longlong(x.s);
    VI r;
    r.v = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)vb >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)vb)));
    r.s = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)sb >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)sb)));
    r.i = __builtin_amdgcn_readfirstlane(x.i);
    return r;
}
__device__ __forceinline__ void publish(Rec *r, unsigned tag, const VI &c)
{
    // lanes 0..7 of the calling wave: one granule each, ONE untorn write-through store per granule
    const int k = threadIdx.x & 63;
    if (k < 8) {
        const u64 vb = (u64)__double_as_longlong(c.v), sb = (u64)__double_as_longlong(c.s);
        unsigned pay = 0;
        if (k == 0) pay = (unsigned)vb;            else if (k == 1) pay = (unsigned)(vb >> 32);
        else if (k == 2) pay = (unsigned)c.i;      else if (k == 3) pay = (unsigned)sb;
        else if (k == 4) pay = (unsigned)(sb >> 32);
        __hip_atomic_store(&r->g[k], ((u64)tag << 32) | pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool decode(const v4u &q0, const v4u &q1, const v4u &q2, const v4u &q3, unsigned tag, VI &c)
{
    const bool ok = q0.y == tag && q0.w == tag && q1.y == tag && q1.w == tag &&
                    q2.y == tag && q2.w == tag && q3.y == tag && q3.w == tag;
    c.v = __longlong_as_double((long long)(((u64)q0.z << 32) | q0.x));
    c.i = (int)q1.x;
    c.s = __longlong_as_double((long long)(((u64)q2.x << 32) | q1.z));
    return ok;
}
// every wave for itself: all G <= 192 records, lane l takes records l, l + 64, l + 128 -- all twelve
// 16-byte loads in flight before the one wait; false = lost
template <int NR>
__device__ __forceinline__ bool poll_reduce(const Rec *recs, int G, unsigned tag, VI &out)
{
    const int lane = threadIdx.x & 63;
    const Rec *r0 = recs + (lane < G ? lane : 0), *r1 = recs + (lane + 64 < G ? lane + 64 : 0), *r2 = recs + (lane + 128 < G ? lane + 128 : 0);
    VI c0, c1, c2;
    for (unsigned spins = 0;; ++spins) {
        v4u a0, a1, a2, a3, b0, b1, b2, b3, d0, d1, d2, d3;
        bool ok;
        if (NR == 1) {
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                         "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(r0) : "memory");
            ok = decode(a0, a1, a2, a3, tag, c0);
        } else {
            asm volatile("global_load_dwordx4 %0, %12, off sc1\n\tglobal_load_dwordx4 %1, %12, off offset:16 sc1\n\t"
                         "global_load_dwordx4 %2, %12, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %12, off offset:48 sc1\n\t"
                         "global_load_dwordx4 %4, %13, off sc1\n\tglobal_load_dwordx4 %5, %13, off offset:16 sc1\n\t"
                         "global_load_dwordx4 %6, %13, off offset:32 sc1\n\tglobal_load_dwordx4 %7, %13, off offset:48 sc1\n\t"
                         "global_load_dwordx4 %8, %14, off sc1\n\tglobal_load_dwordx4 %9, %14, off offset:16 sc1\n\t"
                         "global_load_dwordx4 %10, %14, off offset:32 sc1\n\tglobal_load_dwordx4 %11, %14, off offset:48 sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3),
                           "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                         : "v"(r0), "v"(r1), "v"(r2) : "memory");
            ok = decode(a0, a1, a2, a3, tag, c0);
            ok &= decode(b0, b1, b2, b3, tag, c1);
            ok &= decode(d0, d1, d2, d3, tag, c2);
        }
        if (__all(ok)) break;
        if (spins > kMaxSpins) return false;
        nap();
    }
    if (lane >= G) c0.i = -1;
    if (NR > 1) {
        if (lane + 64 >= G) c1.i = -1;
        if (lane + 128 >= G) c2.i = -1;
        c0 = vi_min(c0, vi_min(c1, c2));
    }
    out = wave_min(c0);
    return true;
}
// transposed records (kernels_la_block.inc, R5.3 / R5.5): granule k of workgroup w at tr[k * kTrStride + w], so a
// wave's poll of one granule is ONE coalesced request per 64 workgroups instead of 64 lines 64 bytes apart
constexpr int kTrStride = 256;
__device__ __forceinline__ void publish_tr(u64 *tr, int w, unsigned tag, const VI &c)
{
    const int k = threadIdx.x & 63;
    if (k < 5) {
        const u64 vb = (u64)__double_as_longlong(c.v), sb = (u64)__double_as_longlong(c.s);
        const unsigned pay = k == 0 ? (unsigned)vb : k == 1 ? (unsigned)(vb >> 32) : k == 2 ? (unsigned)c.i : k == 3 ? (unsigned)sb : (unsigned)(sb >> 32);
        __hip_atomic_store(&tr[k * kTrStride + w], ((u64)tag << 32) | pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int NR>
__device__ __forceinline__ bool poll_reduce_tr(const u64 *tr, int G, unsigned tag, VI &out)
{
    const int lane = threadIdx.x & 63;
    u64 q[NR][5];
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int w = lane + 64 * n;
                q[n][k] = __hip_atomic_load(&tr[k * kTrStride + (w < G ? w : 0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int k = 0; k < 5; ++k) ok &= (unsigned)(q[n][k] >> 32) == tag;
        if (__all(ok)) break;
        if (spins > kMaxSpins) return false;
        nap();
    }
    VI acc; acc.v = 0.0; acc.i = -1; acc.s = 0.0;
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        VI c;
        c.v = __longlong_as_double((long long)(((q[n][1] & 0xffffffffull) << 32) | (q[n][0] & 0xffffffffull)));
        c.i = lane + 64 * n < G ? (int)(unsigned)q[n][2] : -1;
        c.s = __longlong_as_double((long long)(((q[n][4] & 0xffffffffull) << 32) | (q[n][3] & 0xffffffffull)));
        acc = vi_min(acc, c);
    }
    out = wave_min(acc);
    return true;
}
// the exchange in one of four forms: XM bit 0 = transposed records, bit 1 = ONE polling wave per workgroup
// (the others take its result out of LDS behind a barrier)
template <int NR, int XM>
__device__ __forceinline__ bool exchange(Rec *recs, u64 *tr, int G, unsigned tag, const VI &mine, VI &out, VI *s_b, int *s_ok)
{
    if (threadIdx.x < 64) {
        if (XM & 1) publish_tr(tr, blockIdx.x, tag, mine);
        else        publish(recs + blockIdx.x, tag, mine);
    }
    if (XM == 4) {
        // two hops: workgroup 0 alone collects the G records and publishes the winner; everybody else polls that
        // ONE record (slot 255 of the same array)
        if (threadIdx.x < 64) {
            VI r; r.v = 0.0; r.i = -1; r.s = 0.0;
            bool ok;
            if (blockIdx.x == 0) {
                ok = poll_reduce<NR>(recs, G, tag, r);
                publish(recs + 255, tag, r);
            } else {
                ok = poll_reduce<1>(recs + 255, 1, tag, r);
            }
            if (threadIdx.x == 0) { *s_b = r; *s_ok = ok; }
        }
        __syncthreads();
        out = *s_b;
        return *s_ok != 0;
    }
    if (XM & 2) {
        if (threadIdx.x < 64) {
            VI r; r.v = 0.0; r.i = -1; r.s = 0.0;
            const bool ok = (XM & 1) ? poll_reduce_tr<NR>(tr, G, tag, r) : poll_reduce<NR>(recs, G, tag, r);
            if (threadIdx.x == 0) { *s_b = r; *s_ok = ok; }
        }
        __syncthreads();
        out = *s_b;
        return *s_ok != 0;
    }
    return (XM & 1) ? poll_reduce_tr<NR>(tr, G, tag, out) : poll_reduce<NR>(recs, G, tag, out);
}
// one barrier per reduction: the partials alternate between two sets of slots
__device__ __forceinline__ VI block_min(VI x, VI (*s_red)[4], int &par)
{
    x = wave_min(x);
    if ((threadIdx.x & 63) == 0) s_red[par][threadIdx.x >> 6] = x;
    __syncthreads();
    VI r = s_red[par][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) r = vi_min(r, s_red[par][w]);
    par ^= 1;
    return r;
}
__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) { a ^= b * 0x9E3779B1u; a ^= a >> 15; a *= 0x85EBCA77u; a ^= a >> 13; return a; }

// NR: records per lane in a poll (1: G <= 64, 3: G <= 192)
template <int NR, int XM>
__global__ __launch_bounds__(256) void k_skel(double *M, double *bk_col, double *bk_prow, Rec *recA, Rec *recB, u64 *trA, u64 *trB,
                                              int R, int ld, int ksteps, int mode, unsigned epoch0, int *err)
{
    __shared__ VI s_red[2][4];
    __shared__ VI s_b;
    __shared__ int s_ok;
    __shared__ double  s_col[K][256];                   // pending pivots: col_i[my row]
    __shared__ double2 s_prow[K][256];                  //                 prow_i[my column pair]
    int par = 0;
    const int G = gridDim.x, tid = threadIdx.x;
    const int g = blockIdx.x * 256 + tid;
    const int ldv = ld >> 1, vcl = ld - 1, m = R - 1;
    const bool xch = mode & 1, mem = mode & 2;
    const double2 *M2 = reinterpret_cast<const double2 *>(M);
    const bool in = g < ldv, row = g < R;
    double rhs = row ? 1.0 + (g & 1023) * 1e-3 : 0.0;
    double2 z = in ? M2[(int64_t)m * ldv + g] : make_double2(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < K; ++i) { s_col[i][tid] = 0.0; s_prow[i][tid] = make_double2(0.0, 0.0); }
    // the block's first pricing record
    VI pbv;
    {
        VI c; c.v = in ? z.x : 0.0; c.i = in ? 2 * g : -1; c.s = 0.0;
        pbv = block_min(c, s_red, par);
    }
    VI e; e.v = 0.0; e.i = 0; e.s = 0.0;
    for (int j = 0; j < ksteps; ++j) {
        const unsigned tag = epoch0 + (unsigned)j;
        // ---- X1: the entering column
        if (xch) {
            if (!exchange<NR, XM>(recA, trA, G, tag, pbv, e, &s_b, &s_ok)) { if (tid == 0) *err = 1; return; }
        }
        const int lc = (int)(mix((unsigned)e.i, (unsigned)j * 2u + 1u) % (unsigned)vcl);
        // ---- C: the column (strided), the pending prow entries at that column (other workgroups' stores:
        // sc1), all requested together; chained, ratio-tested
        VI best; best.v = 0.0; best.i = -1; best.s = 0.0;
        if (mem) {
            double a = row ? M[(int64_t)g * ld + lc] : 0.0;
            double pl[K];
#pragma unroll
            for (int i = 0; i < K; ++i)
                pl[i] = i < j ? __hip_atomic_load(&bk_prow[(int64_t)i * ld + lc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i)
                a = a - s_col[i][tid] * pl[i];                       // (entries past the last pending pivot are +0.0)
            if (j > 0) rhs = rhs - s_col[j - 1][tid] * 1e-6;
            s_col[j][tid] = a;
            if (row) __hip_atomic_store(&bk_col[(int64_t)j * R + g], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g < m && a > 1e-3) { best.v = rhs / a; best.i = g; best.s = a; }
        } else {
            best.v = (double)(mix((unsigned)g, (unsigned)j) & 0xffff); best.i = g < m ? g : -1;
        }
        best = block_min(best, s_red, par);
        // ---- X2: the pivot row
        VI q2 = best;
        if (xch) {
            if (!exchange<NR, XM>(recB, trB, G, tag, best, q2, &s_b, &s_ok)) { if (tid == 0) *err = 2; return; }
        }
        const int cr = (int)(mix((unsigned)q2.i, (unsigned)j * 2u + 2u) % (unsigned)m);
        // ---- R: the pivot row (contiguous), the pending col entries at that row (sc1), together;
        // chained, scaled, prow_j out, objective row priced
        VI pb; pb.v = 0.0; pb.i = -1; pb.s = 0.0;
        if (mem) {
            double2 y = in ? M2[(int64_t)cr * ldv + g] : make_double2(0.0, 0.0);
            double cc[K];
#pragma unroll
            for (int i = 0; i < K; ++i)
                cc[i] = i < j ? __hip_atomic_load(&bk_col[(int64_t)i * R + cr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const double2 pi = s_prow[i][tid];
                y.x = y.x - cc[i] * pi.x; y.y = y.y - cc[i] * pi.y;
                z.x = z.x - 1e-4 * pi.x;  z.y = z.y - 1e-4 * pi.y;
            }
            const double piv = q2.s != 0.0 ? q2.s : 1.0;
            y.x = y.x / piv; y.y = y.y / piv;
            s_prow[j][tid] = y;
            if (in) {
                __hip_atomic_store(&bk_prow[(int64_t)j * ld + 2 * g], y.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&bk_prow[(int64_t)j * ld + 2 * g + 1], y.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                z.x = z.x - 1e-4 * y.x; z.y = z.y - 1e-4 * y.y;
                VI c; c.v = z.x; c.i = 2 * g; c.s = 0.0; pb = c;
                c.v = z.y; c.i = 2 * g + 1; pb = vi_min(pb, c);
            }
            // the stores above are what the NEXT exchange's records vouch for: drained before publishing
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            pb.v = (double)(mix((unsigned)g, (unsigned)j + 77u) & 0xffff); pb.i = in ? 2 * g : -1;
        }
        pbv = block_min(pb, s_red, par);
        if (!xch) e = pbv;
    }
    if (g == 0) M[(int64_t)m * ld + ld - 1] = z.x + rhs;             // keep everything live
}

__global__ void k_fill(double *M, int64_t n, int ld)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned h = mix((unsigned)(i / ld), (unsigned)(i % ld));
        M[i] = 0.25 + (h & 0xffff) * (1.0 / 65536.0);
    }
}

struct Bufs { double *M, *bk_col, *bk_prow; Rec *recA, *recB; u64 *trA, *trB; int *err; };
template <int NR, int XM>
static void run(const char *what, int R, int ld, int G, const Bufs &B)
{
    static const char *xm[5] = {"record per lane, every wave polls", "transposed, every wave polls", "record per lane, one wave polls", "transposed, one wave polls", "workgroup 0 collects + broadcasts"};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode : {3, 1, 2}) {
        if (mode == 2 && XM != 0) continue;                               // (no exchange in it)
        const int blocks = 40;
        unsigned epoch = 1;
        hipMemset(B.recA, 0, sizeof(Rec) * 512); hipMemset(B.recB, 0, sizeof(Rec) * 512);
        hipMemset(B.trA, 0, 8 * 8 * kTrStride); hipMemset(B.trB, 0, 8 * 8 * kTrStride); hipMemset(B.err, 0, 4);
        for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL((k_skel<NR, XM>), dim3(G), dim3(256), 0, 0, B.M, B.bk_col, B.bk_prow, B.recA, B.recB, B.trA, B.trB, R, ld, K, mode, epoch, B.err); epoch += K + 1; }
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int w = 0; w < blocks; ++w) { hipLaunchKernelGGL((k_skel<NR, XM>), dim3(G), dim3(256), 0, 0, B.M, B.bk_col, B.bk_prow, B.recA, B.recB, B.trA, B.trB, R, ld, K, mode, epoch, B.err); epoch += K + 1; }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        int herr = 0; hipMemcpy(&herr, B.err, 4, hipMemcpyDeviceToHost);
        printf("%-24s %3d workgroups, %-34s %-22s %6.2f us per step (%.1f us per block of %d, err=%d)\n", what, G, mode == 2 ? "-" : xm[XM],
               mode == 3 ? "exchanges + memory:" : mode == 1 ? "exchanges only:" : "memory + chains only:", ms * 1e3 / (blocks * K), ms * 1e3 / blocks, K, herr);
        fflush(stdout);
    }
}
template <int NR>
static void run_all(const char *what, int R, int ld, int G, const Bufs &B)
{
    run<NR, 0>(what, R, ld, G, B); run<NR, 1>(what, R, ld, G, B); run<NR, 2>(what, R, ld, G, B); run<NR, 3>(what, R, ld, G, B); run<NR, 4>(what, R, ld, G, B);
    int one = 1, zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_nap), &one, 4);
    printf("with s_sleep 2 between failed polls:\n");
    run<NR, 0>(what, R, ld, G, B); run<NR, 2>(what, R, ld, G, B); run<NR, 4>(what, R, ld, G, B);
    hipMemcpyToSymbol(HIP_SYMBOL(g_nap), &zero, 4);
}

int main()
{
    const int R = 32769, ld = 8194;
    Bufs B;
    if (hipMalloc(&B.M, (size_t)R * ld * 8) != hipSuccess) { printf("no memory\n"); return 1; }
    hipMalloc(&B.bk_col, (size_t)K * R * 8); hipMalloc(&B.bk_prow, (size_t)K * ld * 8);
    hipMalloc(&B.recA, sizeof(Rec) * 512); hipMalloc(&B.recB, sizeof(Rec) * 512); hipMalloc(&B.err, 4);
    hipMalloc(&B.trA, 8 * 8 * kTrStride); hipMalloc(&B.trB, 8 * 8 * kTrStride);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, B.M, (int64_t)R * ld, ld);
    hipMemset(B.bk_col, 0, (size_t)K * R * 8); hipMemset(B.bk_prow, 0, (size_t)K * ld * 8);
    hipDeviceSynchronize();
    // calibration: the shape k_la_block serves today (config 3: 4097 rows, 17 workgroups; 161 us per block
    // of 24 = 6.7 us per step in the product) inside the big buffer
    run_all<1>("config-3 shape, 4097 rows", 4097, ld, 17, B);
    run_all<1>("8192 rows", 8192, ld, 32, B);
    run_all<3>("16385 rows", 16385, ld, 65, B);
    run_all<3>("8-GPU shard, 32769 rows", R, ld, 129, B);
    return 0;
}
