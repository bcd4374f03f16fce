// sweep_stream.hip -- what separates the blocked sweep (k_sweep16: ~5.3 TB/s) from the per-pivot
// update (k_update: 6.8 TB/s) on the SAME bytes?  One kernel skeleton -- a [tr rows x 512 columns]
// tile per workgroup, rows streamed through registers U at a time with the next step's loads in
// flight, K dependent (mul, sub) links applied to every element from register-resident `prow`
// pairs and SGPR `col` values -- timed over the config-3 tableau (4097 x 8208 f64) while ONE knob
// moves at a time:
//     K      links per element (0 = pure read-modify-write stream, 16 = the sweep)
//     tr     rows per workgroup          occ    waves per SIMD (capped with dynamic LDS)
//     pro    bytes of per-tile prologue loads (the sweep's prow: 16 x 16 B per thread)
//     order  0 = x-fastest tiles (strips of one row band are neighbours), 1 = y-fastest
//
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o sweep_stream sweep_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double vec2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int I> __device__ __forceinline__ double bcast16(double v)   // lane I of every row of 16 lanes
{
    double c;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(c) : "v"(v), "n"(I));
    return c;
}
template <int I, int K, int U> struct Links {
    static __device__ __forceinline__ void run(vec2d (&cur)[U], const double (&cv)[U], const vec2d (&p)[K > 0 ? K : 1])
    {
        if constexpr (I < K) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double c = bcast16<I>(cv[u]);
                cur[u].x = cur[u].x - c * p[I].x;
                cur[u].y = cur[u].y - c * p[I].y;
            }
            Links<I + 1, K, U>::run(cur, cv, p);
        }
    }
};

template <int K, int U, bool PRO, int MODE = 0>
__global__ __launch_bounds__(256) void k_stream(double *M, const double *prow, const double *col, int64_t ld,
                                                int64_t rows, int tr, int strips, int order)
{
    extern __shared__ double occupancy_pad[];
    int bx, by;
    if (order == 0) { bx = blockIdx.x % strips; by = blockIdx.x / strips; }
    else            { const int nb = gridDim.x / strips; by = blockIdx.x % nb; bx = blockIdx.x / nb; }
    const int64_t ldv = ld >> 1;
    const int64_t pair = (int64_t)bx * 256 + threadIdx.x;
    if (pair >= ldv) return;
    const int64_t r0 = (int64_t)by * tr, r1 = r0 + tr < rows ? r0 + tr : rows;
    vec2d *Mp = reinterpret_cast<vec2d *>(M) + pair;
    constexpr int KP = K > 0 ? K : 1;
    vec2d p[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        if (PRO) p[i] = reinterpret_cast<const vec2d *>(prow)[(int64_t)i * ldv + pair];
        else   { p[i].x = 1e-9 * (i + 1); p[i].y = 2e-9 * (i + 1); }
    }
    vec2d xa[U], xb[U];
    double ca[U], cb[U];                         // MODE 5: lane l holds col[l % 16][row] of the step's rows
    const double *colp = col + (int64_t)(threadIdx.x & 15) * rows;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        xa[u] = vec2d{0, 0}; xb[u] = xa[u]; ca[u] = 0; cb[u] = 0;
        if (r0 + u < r1) { xa[u] = Mp[(r0 + u) * ldv]; if (MODE == 5) ca[u] = colp[r0 + u]; }
    }
    auto step = [&](vec2d (&cur)[U], vec2d (&nxt)[U], double (&cc)[U], double (&cn)[U], int64_t r) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (r + U + u < r1) { nxt[u] = Mp[(r + U + u) * ldv]; if (MODE == 5) cn[u] = colp[r + U + u]; }
        if (MODE == 5) {
            Links<0, K, U>::run(cur, cc, p);
        } else if (MODE >= 10) {                // sleep MODE - 10 units of 64 cycles
            __builtin_amdgcn_s_sleep(MODE - 10);
        } else if (MODE == 1) {                        // no arithmetic: the wave sleeps for about as long
            __builtin_amdgcn_s_sleep(16);
        } else if (MODE == 2) {                 // same instruction count in f32 (2 x 32-bit halves)
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float cv = (float)col[(int64_t)i * rows + r + u];
                    float a = __builtin_bit_cast(float, (int)__double2loint(cur[u].x)), b = __builtin_bit_cast(float, (int)__double2loint(cur[u].y));
                    a = a - cv * (float)p[i].x; b = b - cv * (float)p[i].y;
                    cur[u].x = __hiloint2double(__double2hiint(cur[u].x), __builtin_bit_cast(int, a));
                    cur[u].y = __hiloint2double(__double2hiint(cur[u].y), __builtin_bit_cast(int, b));
                }
            }
        } else if (MODE == 3) {                 // f64 links, col values made up in registers (no scalar loads)
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double cv = __builtin_bit_cast(double, (unsigned long long)(0x3e45798ee2308c3aull + (unsigned long long)(r + u + i)));
                    cur[u].x = cur[u].x - cv * p[i].x;
                    cur[u].y = cur[u].y - cv * p[i].y;
                }
            }
        } else if (MODE == 4) {                 // scalar loads as in the sweep, but no arithmetic with them
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int u = 0; u < U; ++u) acc += col[(int64_t)i * rows + r + u];
            }
            cur[0].x += acc;
        } else {
#pragma unroll
        for (int i = 0; i < K; ++i) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double cv = col[(int64_t)i * rows + r + u];          // uniform: scalar load
                cur[u].x = cur[u].x - cv * p[i].x;
                cur[u].y = cur[u].y - cv * p[i].y;
            }
        }
        }
        if (K == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) { cur[u].x += p[0].x; cur[u].y += p[0].y; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (r + u < r1) Mp[(r + u) * ldv] = cur[u];
    };
    for (int64_t r = r0; r < r1; r += 2 * U) {
        step(xa, xb, ca, cb, r);
        if (r + U < r1) step(xb, xa, cb, ca, r + U);
    }
}

template <int K, int U, bool PRO, int MODE = 0>
static double run(double *M, const double *prow, const double *col, int64_t ld, int64_t rows, int tr, int occ, int order)
{
    const int strips = (int)(((ld >> 1) + 255) / 256);
    const int nb = (int)((rows + tr - 1) / tr);
    // occupancy cap: occ workgroups of 256 threads per CU <=> occ waves per SIMD; LDS 160 KB per CU
    const size_t lds = occ >= 8 ? 0 : (size_t)(160 * 1024 / occ) - 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<K, U, PRO, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<K, U, PRO, MODE>), dim3(strips * nb), dim3(256), lds, 0, M, prow, col, ld, rows, tr, strips, order);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<K, U, PRO, MODE>), dim3(strips * nb), dim3(256), lds, 0, M, prow, col, ld, rows, tr, strips, order);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main(int argc, char **argv)
{
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 4097, ld = argc > 2 ? atoll(argv[2]) : 8208;
    double *M, *prow, *col;
    CK(hipMalloc(&M, rows * ld * 8)); CK(hipMalloc(&prow, 32 * ld * 8)); CK(hipMalloc(&col, 32 * (rows + 64) * 8));
    CK(hipMemset(M, 0, rows * ld * 8)); CK(hipMemset(prow, 0, 32 * ld * 8)); CK(hipMemset(col, 0, 32 * (rows + 64) * 8));
    const double gb = 2.0 * rows * ld * 8 / 1e9;
    auto show = [&](const char *what, int K, int U, int pro, int tr, int occ, int order, double us) {
        printf("%-22s K=%2d U=%d prologue=%d tr=%3d occ=%d order=%d : %7.1f us  %5.2f TB/s\n", what, K, U, pro, tr, occ, order, us, gb / us * 1e3 / 1e3 * 1e-3 * 1e3);
        fflush(stdout);
    };
    show("K=8 f64", 8, 4, 1, 32, 4, 0, run<8, 4, true>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=8 sleep instead", 8, 4, 1, 32, 4, 0, run<8, 4, true, 1>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=8 f32 instead", 8, 4, 1, 32, 4, 0, run<8, 4, true, 2>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=16 sleep instead", 16, 4, 1, 32, 3, 0, run<16, 4, true, 1>(M, prow, col, ld, rows, 32, 3, 0));
    show("K=8 f64, no s_load", 8, 4, 1, 32, 4, 0, run<8, 4, true, 3>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=8 s_load only", 8, 4, 1, 32, 4, 0, run<8, 4, true, 4>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=4 f64, no s_load", 4, 4, 1, 32, 4, 0, run<4, 4, true, 3>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=8 vector col + dpp", 8, 4, 1, 32, 4, 0, run<8, 4, true, 5>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=16 vector col + dpp", 16, 4, 1, 32, 3, 0, run<16, 4, true, 5>(M, prow, col, ld, rows, 32, 3, 0));
    show("K=16 vector col + dpp", 16, 4, 1, 32, 4, 0, run<16, 4, true, 5>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=16 vector col + dpp", 16, 4, 1, 16, 4, 0, run<16, 4, true, 5>(M, prow, col, ld, rows, 16, 4, 0));
    show("K=16 f64, no s_load", 16, 4, 1, 32, 4, 0, run<16, 4, true, 3>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=16 sgpr col", 16, 4, 1, 32, 4, 0, run<16, 4, true, 0>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=0", 0, 4, 0, 32, 4, 0, run<0, 4, false>(M, prow, col, ld, rows, 32, 4, 0));
    show("K=0 + sleep", 0, 4, 0, 32, 4, 0, run<0, 4, false, 1>(M, prow, col, ld, rows, 32, 4, 0));
    if (argc > 3 && argv[3][0] == 'k') {      // links per pass: 16 vs 32 (two look-ahead blocks per sweep)
        for (int occ : {2, 3}) {
            show("K=16 f64, no s_load", 16, 4, 1, 32, occ, 0, run<16, 4, true, 3>(M, prow, col, ld, rows, 32, occ, 0));
            show("K=24 f64, no s_load", 24, 4, 1, 32, occ, 0, run<24, 4, true, 3>(M, prow, col, ld, rows, 32, occ, 0));
            show("K=32 f64, no s_load", 32, 4, 1, 32, occ, 0, run<32, 4, true, 3>(M, prow, col, ld, rows, 32, occ, 0));
            show("K=32 f64, no s_load U=2", 32, 2, 1, 32, occ, 0, run<32, 2, true, 3>(M, prow, col, ld, rows, 32, occ, 0));
        }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'p') {      // pacing study
        for (int occ : {1, 2, 3, 4, 6, 8}) show("K=16 f64, no s_load", 16, 4, 1, 32, occ, 0, run<16, 4, true, 3>(M, prow, col, ld, rows, 32, occ, 0));
        for (int occ : {1, 2, 4, 8}) {
            show("K=0 sleep 0", 0, 4, 0, 32, occ, 0, run<0, 4, false, 10>(M, prow, col, ld, rows, 32, occ, 0));
            show("K=0 sleep 4", 0, 4, 0, 32, occ, 0, run<0, 4, false, 14>(M, prow, col, ld, rows, 32, occ, 0));
            show("K=0 sleep 8", 0, 4, 0, 32, occ, 0, run<0, 4, false, 18>(M, prow, col, ld, rows, 32, occ, 0));
            show("K=0 sleep 16", 0, 4, 0, 32, occ, 0, run<0, 4, false, 26>(M, prow, col, ld, rows, 32, occ, 0));
            show("K=0 sleep 32", 0, 4, 0, 32, occ, 0, run<0, 4, false, 42>(M, prow, col, ld, rows, 32, occ, 0));
            show("K=0 sleep 64", 0, 4, 0, 32, occ, 0, run<0, 4, false, 74>(M, prow, col, ld, rows, 32, occ, 0));
        }
        return 0;
    }
    if (argc > 3) return 0;
    // 1. the update's shape and the sweep's shape, links 0 / 16
    show("update-like", 0, 4, 0, 4, 8, 0, run<0, 4, false>(M, prow, col, ld, rows, 4, 8, 0));
    show("update-like occ 3", 0, 4, 0, 4, 3, 0, run<0, 4, false>(M, prow, col, ld, rows, 4, 3, 0));
    show("sweep tile, no links", 0, 4, 0, 32, 8, 0, run<0, 4, false>(M, prow, col, ld, rows, 32, 8, 0));
    show("sweep tile, no links", 0, 4, 0, 32, 3, 0, run<0, 4, false>(M, prow, col, ld, rows, 32, 3, 0));
    show("  + prologue", 0, 4, 1, 32, 3, 0, run<0, 4, true>(M, prow, col, ld, rows, 32, 3, 0));
    for (int occ : {2, 3, 4}) {
        show("sweep, 4 links", 4, 4, 1, 32, occ, 0, run<4, 4, true>(M, prow, col, ld, rows, 32, occ, 0));
        show("sweep, 8 links", 8, 4, 1, 32, occ, 0, run<8, 4, true>(M, prow, col, ld, rows, 32, occ, 0));
        show("sweep, 16 links", 16, 4, 1, 32, occ, 0, run<16, 4, true>(M, prow, col, ld, rows, 32, occ, 0));
    }
    // 2. tile height and dispatch order at 16 links
    for (int tr : {8, 16, 32, 64, 128})
        for (int order : {0, 1})
            show("sweep, 16 links", 16, 4, 1, tr, 3, order, run<16, 4, true>(M, prow, col, ld, rows, tr, 3, order));
    // 3. two rows per step (half the registers in flight per wave, more waves)
    for (int occ : {3, 4, 5, 6})
        show("sweep, 16 links, U=2", 16, 2, 1, 32, occ, 0, run<16, 2, true>(M, prow, col, ld, rows, 32, occ, 0));
    return 0;
}
