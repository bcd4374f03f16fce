// xchg_variants.hip -- what one all-to-all exchange between the persistent look-ahead workgroups
// costs on gfx950, by protocol, with the payload hand-off checked word for word (also next to a
// streaming kernel that keeps HBM busy: hand-off bugs hide on an idle chip).
//
//   V0  payload + epoch with a RELEASE store, consumers poll with ACQUIRE loads (round-1 form)
//   V1  self-validating 8-byte granules {tag, value}: relaxed write-through (sc1) stores, relaxed
//       L1-bypassing (sc1) polls, no fence anywhere; data handed over with sc1 stores, every
//       storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the
//       publish, consumers read it with sc1 loads
//   V2  V1 with all participants on ONE XCD (grid = 8 x nw, blocks with b % 8 != 0 exit)
//   V3  one XCD, plain stores (kept in that XCD's L2) + sc1 loads -- only valid when every
//       participant really sits on the same XCD (checked with HW_REG_XCC_ID, reported)
//
// build: hipcc --offload-arch=gfx950 -O3 -o xchg_variants xchg_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Rec { unsigned long long g[8]; };                 // one 64-byte line per workgroup
constexpr int kMaxW = 32;

template <class T> __device__ __forceinline__ void st_wt(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ T ld_l2(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// V: protocol; data: doubles handed over per thread per exchange (0 or 1)
template <int V>
__global__ __launch_bounds__(256) void k_xchg(Rec *recs, double *data, int n, int nw_logical, int *err,
                                              unsigned *xcc_seen, unsigned long long *cycles)
{
    int w = blockIdx.x, nw = gridDim.x;
    if (V >= 2) {                                        // one XCD: only every 8th block takes part
        if (w % 8 != 0) return;
        w /= 8; nw = nw_logical;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    __shared__ double s_res;
    if (tid == 0) atomicOr(xcc_seen, 1u << xcc_id());
    double acc = 0.0;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= n; ++it) {
        Rec *mine = recs + (size_t)(it & 1) * kMaxW + w;
        double *dbuf = data + (size_t)(it & 1) * kMaxW * 256;
        // payload: every thread hands one double to the thread of the same index in workgroup w+1
        const double payload = (double)it * 1024.0 + w * 256 + tid;
        if (V == 0 || V == 3) dbuf[w * 256 + tid] = payload;
        else                  st_wt(&dbuf[w * 256 + tid], payload);
        if (V != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave drains
        __syncthreads();
        if (tid < 64) {
            const double v = acc + w;
            const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
            if (V == 0) {
                if (lane == 0) {
                    st_wt(&mine->g[0], vb);
                    st_wt(&mine->g[1], (unsigned long long)w);
                    __hip_atomic_store(&mine->g[2], (unsigned long long)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                const unsigned val = lane == 0 ? (unsigned)vb : lane == 1 ? (unsigned)(vb >> 32) : (unsigned)w;
                const unsigned long long gr = ((unsigned long long)(unsigned)it << 32) | val;
                if (lane < 5) {
                    if (V == 3) *(volatile unsigned long long *)&mine->g[lane] = gr;
                    else        st_wt(&mine->g[lane], gr);
                }
            }
            const Rec *r = recs + (size_t)(it & 1) * kMaxW + (lane < nw ? lane : 0);
            unsigned spins = 0;
            double got = 1e300;
            if (V == 0) {
                bool ok = lane >= nw;
                while (!__all(ok)) {
                    if (!ok) ok = __hip_atomic_load(&r->g[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)it;
                    if (++spins > (1u << 22)) { *err = 1; break; }
                }
                if (lane < nw) got = __longlong_as_double((long long)ld_l2(&r->g[0]));
            } else {
                unsigned long long g0, g1, g2, g3, g4;
                for (;;) {
                    g0 = ld_l2(&r->g[0]); g1 = ld_l2(&r->g[1]); g2 = ld_l2(&r->g[2]);
                    g3 = ld_l2(&r->g[3]); g4 = ld_l2(&r->g[4]);
                    const bool ok = lane >= nw || ((unsigned)(g0 >> 32) == (unsigned)it && (unsigned)(g1 >> 32) == (unsigned)it &&
                                                   (unsigned)(g2 >> 32) == (unsigned)it && (unsigned)(g3 >> 32) == (unsigned)it &&
                                                   (unsigned)(g4 >> 32) == (unsigned)it);
                    if (__all(ok)) break;
                    if (++spins > (1u << 22)) { *err = 1; break; }
                }
                if (lane < nw) got = __longlong_as_double((long long)(((g1 & 0xffffffffull) << 32) | (g0 & 0xffffffffull)));
            }
            for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(got, off, 64); got = o < got ? o : got; }
            if (lane == 0) s_res = got;
        }
        __syncthreads();
        if (s_res != acc) *err = 2;
        // the hand-off: read what workgroup w+1 (mod nw) stored this exchange
        const int src = (w + 1) % nw;
        const double want = (double)it * 1024.0 + src * 256 + tid;
        double seen;
        if (V == 0) seen = dbuf[src * 256 + tid];       // covered by wave 0's acquire + the barrier
        else        seen = ld_l2(&dbuf[src * 256 + tid]);
        if (seen != want) atomicAdd(err + 1, 1);
        acc += 1.0;
    }
    if (tid == 0 && w == 0) cycles[0] = wall_clock64() - t0;
}

__global__ __launch_bounds__(256) void k_stream(double *a, double *b, size_t n, int reps)
{
    for (int rp = 0; rp < reps; ++rp)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            b[i] = a[i] * 1.0000001 + (double)rp;
}

template <int V>
static void run(const char *name, int nw, int n, Rec *recs, double *data, int *err, unsigned *xcc,
                unsigned long long *cyc, bool loaded, double *sa, double *sb, size_t sn, hipStream_t s2)
{
    hipMemset(recs, 0, 2 * kMaxW * sizeof(Rec)); hipMemset(err, 0, 8); hipMemset(xcc, 0, 4);
    hipMemset(data, 0, 2 * kMaxW * 256 * sizeof(double));
    hipDeviceSynchronize();
    if (loaded) hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, s2, sa, sb, sn, 40);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_xchg<V>, dim3(V >= 2 ? nw * 8 : nw), dim3(256), 0, 0, recs, data, n, nw, err, xcc, cyc);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipDeviceSynchronize();
    int herr[2] = {0, 0}; unsigned hx = 0;
    hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost); hipMemcpy(&hx, xcc, 4, hipMemcpyDeviceToHost);
    printf("%-28s nw=%2d %s: %.3f us per exchange  err=%d stale_words=%d xcc_mask=0x%02x\n", name, nw,
           loaded ? "LOADED" : "idle  ", ms * 1e3 / n, herr[0], herr[1], hx);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main()
{
    Rec *recs; double *data; int *err; unsigned *xcc; unsigned long long *cyc;
    hipMalloc(&recs, 2 * kMaxW * sizeof(Rec)); hipMalloc(&data, 2 * kMaxW * 256 * sizeof(double));
    hipMalloc(&err, 8); hipMalloc(&xcc, 4); hipMalloc(&cyc, 8);
    const size_t sn = (size_t)1 << 27;                   // 2 x 1 GiB streaming buffers
    double *sa, *sb; hipMalloc(&sa, sn * 8); hipMalloc(&sb, sn * 8); hipMemset(sa, 0, sn * 8);
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int n = 4000;
    for (int loaded = 0; loaded < 2; ++loaded)
        for (int nw : {5, 9, 17, 32}) {
            run<0>("V0 release/acquire", nw, n, recs, data, err, xcc, cyc, loaded, sa, sb, sn, s2);
            run<1>("V1 granules sc1", nw, n, recs, data, err, xcc, cyc, loaded, sa, sb, sn, s2);
            run<2>("V2 granules sc1, one XCD", nw, n, recs, data, err, xcc, cyc, loaded, sa, sb, sn, s2);
            run<3>("V3 plain stores, one XCD", nw, n, recs, data, err, xcc, cyc, loaded, sa, sb, sn, s2);
        }
    return 0;
}
