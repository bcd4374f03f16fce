// sweep32.hip -- which shape should a sweep that applies 32 pending pivots per pass have?
//
// k_sweep16 (product) = a thread owns a column PAIR (16-byte accesses), keeps its 16 prow pairs in
// 64 VGPRs, streams 4 + 4 rows, col operands through hand-issued SGPR chunk loads.  Doubling the
// links with the same shape doubles the prow registers (128 VGPRs: 2 waves per SIMD, what the
// round-2 experiment k_sweep32 measured: 178-215 us at config 3).  Here: the same skeleton with the
// knobs that keep the register budget at the K = 16 level --
//     CPT   columns per thread: 2 (pair, 16-byte accesses) or 1 (8-byte accesses, 32 prow values in
//           64 VGPRs)
//     U     rows per step (two register sets: U rows computed while the next U travel)
//     K     links (pending pivots) per pass
//     tr    rows per workgroup (the prow prologue is paid once per tile)
// col operands arrive exactly as in the product kernels: 32 SGPRs per chunk (s_load_dwordx8 x 4 for
// U = 4: 4 pivots x 4 rows; s_load_dwordx16 x 2 for U = 8: 2 pivots x 8 rows), two chunk sets
// alternating, issue and wait in separate asm statements.
//
// usage: sweep32 [rows ld]      default 32769 x 8208 (one 8-GPU-sized shard of config 5, 2.15 GB)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o sweep32 sweep32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef double vec2d __attribute__((ext_vector_type(2)));
typedef int    v8i  __attribute__((ext_vector_type(8)));
typedef int    v16i __attribute__((ext_vector_type(16)));
typedef double v4d  __attribute__((ext_vector_type(4)));
typedef double v8d  __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U> struct ColChunk;
template <> struct ColChunk<4> {
    static constexpr int CP = 4;
    v8i c[4];
    __device__ __forceinline__ void issue(const double *base, unsigned o1)
    {
        const unsigned o2 = 2u * o1, o3 = 3u * o1;
        asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, %5\n\t"
                     "s_load_dwordx8 %2, %4, %6\n\ts_load_dwordx8 %3, %4, %7"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3])
                     : "s"(base), "s"(o1), "s"(o2), "s"(o3));
    }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c[0]), "+s"(c[1]), "+s"(c[2]), "+s"(c[3])); }
    __device__ __forceinline__ double col(int i, int u) const { return __builtin_bit_cast(v4d, c[i])[u]; }
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

template <int CPT> struct Elem;
template <> struct Elem<2> { typedef vec2d T; };
template <> struct Elem<1> { typedef double T; };

template <int K, int U, int CPT, bool NT>
__global__ __launch_bounds__(256) void k_sweep(double *M, const double *prow, const double *col, int64_t ld,
                                               int64_t rows, int64_t col_stride, int tr, int strips)
{
    typedef typename Elem<CPT>::T E;
    constexpr int CP = ColChunk<U>::CP, NCH = K / CP;
    const int bx = blockIdx.x % strips, by = blockIdx.x / strips;
    const int64_t lde = ld / CPT;                                      // row length in elements of E
    const int64_t e = (int64_t)bx * 256 + threadIdx.x;
    if (e >= lde) return;
    const int64_t r0 = (int64_t)by * tr, r1 = r0 + tr < rows ? r0 + tr : rows;
    E *Mp = reinterpret_cast<E *>(M) + e;
    auto ld2 = [&](int64_t r) -> E {
        if constexpr (NT) return __builtin_nontemporal_load(Mp + r * lde);
        else              return Mp[r * lde];
    };
    auto st2 = [&](int64_t r, E v) {
        if constexpr (NT) __builtin_nontemporal_store(v, Mp + r * lde);
        else              Mp[r * lde] = v;
    };
    E xa[U], xb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { xa[u] = E(0); xb[u] = E(0); if (r0 + u < r1) xa[u] = ld2(r0 + u); }
    E p[K];
#pragma unroll
    for (int i = 0; i < K; ++i) p[i] = reinterpret_cast<const E *>(prow)[(int64_t)i * lde + e];
    const unsigned o1 = (unsigned)(col_stride * 8);
    const int64_t chunk_stride = (int64_t)CP * col_stride;
    auto step = [&](E (&cur)[U], E (&nxt)[U], const int64_t r) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (r + U + u < r1) nxt[u] = ld2(r + U + u);
        const double *cb = col + r;
        ColChunk<U> A, B;
        auto apply = [&](const ColChunk<U> &c, const int i0) {
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const E pi = p[i0 + i];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double cv = c.col(i, u);
                    if constexpr (CPT == 2) {
                        const double m0 = cv * pi.x, m1 = cv * pi.y;
                        cur[u].x = cur[u].x - m0;
                        cur[u].y = cur[u].y - m1;
                    } else {
                        const double m0 = cv * pi;
                        cur[u] = cur[u] - m0;
                    }
                }
            }
        };
        A.issue(cb, o1);
        A.wait();
#pragma unroll
        for (int c = 0; c < NCH; c += 2) {
            B.issue(cb + (int64_t)(c + 1) * chunk_stride, o1);
            apply(A, c * CP);
            B.wait();
            if (c + 2 < NCH) A.issue(cb + (int64_t)(c + 2) * chunk_stride, o1);
            apply(B, (c + 1) * CP);
            if (c + 2 < NCH) A.wait();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (r + u < r1) st2(r + u, cur[u]);
    };
    for (int64_t r = r0; r < r1; r += 2 * U) {
        step(xa, xb, r);
        if (r + U < r1) step(xb, xa, r + U);
    }
}

// pair layout, U = 4: the prow pairs of the first KR links in registers, those of the last K - KR in
// LDS (re-read once per step of four rows: 16 B per lane and link against 16 f64 instructions)
template <int K, int KR, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_sweep_mix(double *M, const double *prow, const double *col, int64_t ld,
                                                   int64_t rows, int64_t col_stride, int tr, int strips)
{
    constexpr int U = 4, CP = 4, NCH = K / CP, KL = K - KR;
    __shared__ vec2d s_p[(KL > 0 ? KL : 1) * 256];
    const int bx = blockIdx.x % strips, by = blockIdx.x / strips;
    const int64_t lde = ld / 2;
    const int64_t e = (int64_t)bx * 256 + threadIdx.x;
    if (e >= lde) return;
    const int64_t r0 = (int64_t)by * tr, r1 = r0 + tr < rows ? r0 + tr : rows;
    vec2d *Mp = reinterpret_cast<vec2d *>(M) + e;
    auto ld2 = [&](int64_t r) -> vec2d {
        if constexpr (NT) return __builtin_nontemporal_load(Mp + r * lde);
        else              return Mp[r * lde];
    };
    auto st2 = [&](int64_t r, vec2d v) {
        if constexpr (NT) __builtin_nontemporal_store(v, Mp + r * lde);
        else              Mp[r * lde] = v;
    };
    vec2d xa[U], xb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { xa[u] = vec2d(0); xb[u] = vec2d(0); if (r0 + u < r1) xa[u] = ld2(r0 + u); }
    vec2d p[KR > 0 ? KR : 1];
#pragma unroll
    for (int i = 0; i < KR; ++i) p[i] = reinterpret_cast<const vec2d *>(prow)[(int64_t)i * lde + e];
#pragma unroll
    for (int i = 0; i < KL; ++i) s_p[i * 256 + threadIdx.x] = reinterpret_cast<const vec2d *>(prow)[(int64_t)(KR + i) * lde + e];
    // (a thread reads back only what it wrote: no barrier)
    const unsigned o1 = (unsigned)(col_stride * 8);
    const int64_t chunk_stride = (int64_t)CP * col_stride;
    auto step = [&](vec2d (&cur)[U], vec2d (&nxt)[U], const int64_t r) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (r + U + u < r1) nxt[u] = ld2(r + U + u);
        const double *cb = col + r;
        ColChunk<U> A, B;
        auto apply = [&](const ColChunk<U> &c, const int i0) {
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const vec2d pi = (i0 + i < KR) ? p[(i0 + i < KR) ? i0 + i : 0] : s_p[((i0 + i >= KR) ? i0 + i - KR : 0) * 256 + threadIdx.x];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double cv = c.col(i, u);
                    const double m0 = cv * pi.x, m1 = cv * pi.y;
                    cur[u].x = cur[u].x - m0;
                    cur[u].y = cur[u].y - m1;
                }
            }
        };
        A.issue(cb, o1);
        A.wait();
#pragma unroll
        for (int c = 0; c < NCH; c += 2) {
            if (c + 1 < NCH) B.issue(cb + (int64_t)(c + 1) * chunk_stride, o1);
            apply(A, c * CP);
            if (c + 1 < NCH) {
                B.wait();
                if (c + 2 < NCH) A.issue(cb + (int64_t)(c + 2) * chunk_stride, o1);
                apply(B, (c + 1) * CP);
                if (c + 2 < NCH) A.wait();
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (r + u < r1) st2(r + u, cur[u]);
    };
    for (int64_t r = r0; r < r1; r += 2 * U) {
        step(xa, xb, r);
        if (r + U < r1) step(xb, xa, r + U);
    }
}

template <int K, int KR, bool NT>
static void run_mix(double *M, const double *prow, const double *col, int64_t ld, int64_t rows, int64_t col_stride, int tr)
{
    const int strips = (int)((ld / 2 + 255) / 256);
    const int nb = (int)((rows + tr - 1) / tr);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = rows * ld > (1ll << 28) ? 6 : 20;
    auto launch = [&]() {
        hipLaunchKernelGGL((k_sweep_mix<K, KR, NT>), dim3(strips * nb), dim3(256), 0, 0, M, prow, col, ld, rows, col_stride, tr, strips);
    };
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, gb = 2.0 * rows * ld * 8 / 1e9;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_sweep_mix<K, KR, NT>)));
    printf("pairs, U=4     K=%2d (%2d in registers, %2d in LDS) nt=%d tr=%4d : %9.1f us  %5.2f TB/s  %7.2f us per pivot   (%d VGPRs, %d B LDS)\n",
           K, KR, K - KR, (int)NT, tr, us, gb / us * 1e-3, us / K, fa.numRegs, (int)fa.sharedSizeBytes);
    fflush(stdout);
}

template <int K, int U, int CPT, bool NT>
static void run(const char *what, double *M, const double *prow, const double *col, int64_t ld, int64_t rows,
                int64_t col_stride, int tr)
{
    const int strips = (int)((ld / CPT + 255) / 256);
    const int nb = (int)((rows + tr - 1) / tr);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = rows * ld > (1ll << 28) ? 6 : 20;
    auto launch = [&]() {
        hipLaunchKernelGGL((k_sweep<K, U, CPT, NT>), dim3(strips * nb), dim3(256), 0, 0, M, prow, col, ld, rows, col_stride, tr, strips);
    };
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, gb = 2.0 * rows * ld * 8 / 1e9;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_sweep<K, U, CPT, NT>)));
    printf("%-14s K=%2d U=%d cols/thread=%d nt=%d tr=%4d : %9.1f us  %5.2f TB/s  %7.2f us per pivot   (%d VGPRs, %d spilled)\n",
           what, K, U, CPT, (int)NT, tr, us, gb / us * 1e-3, us / K, fa.numRegs, (int)(fa.localSizeBytes / 4));
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 32769, ld = argc > 2 ? atoll(argv[2]) : 8208;
    const int64_t cs = (rows + 64 + 15) / 16 * 16;
    double *M, *prow, *col;
    CK(hipMalloc(&M, rows * ld * 8)); CK(hipMalloc(&prow, 32 * ld * 8)); CK(hipMalloc(&col, 32 * cs * 8));
    CK(hipMemset(M, 0, rows * ld * 8)); CK(hipMemset(prow, 0, 32 * ld * 8)); CK(hipMemset(col, 0, 32 * cs * 8));
    printf("tableau %lld x %lld doubles = %.2f GB stored\n", (long long)rows, (long long)ld, rows * ld * 8 / 1e9);
    const bool big = rows * ld * 8 > (300ll << 20);
#define RUN(K, U, CPT, TR) do { if (big) run<K, U, CPT, true>("", M, prow, col, ld, rows, cs, TR); else run<K, U, CPT, false>("", M, prow, col, ld, rows, cs, TR); } while (0)
#define MIX(K, KR, TR) do { if (big) run_mix<K, KR, true>(M, prow, col, ld, rows, cs, TR); else run_mix<K, KR, false>(M, prow, col, ld, rows, cs, TR); } while (0)
    if (argc > 3) {   // the first survey (round 4): columns per thread, rows per step
        RUN(16, 4, 2, 32); RUN(32, 4, 2, 32); RUN(32, 8, 1, 64); RUN(32, 4, 1, 64); RUN(16, 8, 1, 64); RUN(24, 4, 2, 64);
        return 0;
    }
    for (int tr : {32, 64}) {
        if (tr == 32) { MIX(16, 16, 32); MIX(20, 20, 32); MIX(24, 24, 32); MIX(28, 28, 32); MIX(32, 32, 32); }
        else          { MIX(16, 16, 64); MIX(20, 20, 64); MIX(24, 24, 64); MIX(28, 28, 64); MIX(32, 32, 64); }
    }
    // the last links' prow pairs from LDS
    MIX(32, 24, 32); MIX(32, 24, 64); MIX(32, 24, 128);
    MIX(32, 16, 32); MIX(32, 16, 64);
    MIX(32, 20, 64);
    MIX(28, 20, 64);
    MIX(24, 16, 64);
    MIX(16, 8, 32);
    MIX(32, 0, 64);
    return 0;
}
