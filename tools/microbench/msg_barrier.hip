// msg_barrier.hip -- exchange + synchronisation between NW persistent workgroups by message
// passing: every workgroup publishes a small record (payload words, then an epoch flag with
// release semantics at agent scope) and every workgroup's first wave polls all NW flags in
// parallel (lane l polls workgroup l's flag) before reading the payloads.  No read-modify-write
// atomics, so nothing serialises.  Spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>

struct Rec { double v; long long i; long long s; unsigned long long epoch; };   // 32 bytes

template <int MODE>
__global__ __launch_bounds__(256) void k_msg(Rec *slots, int n, int *err, double *out)
{
    const int nw = gridDim.x, w = blockIdx.x, lane = threadIdx.x & 63;
    __shared__ double s_best;
    double acc = 0.0;
    for (int it = 1; it <= n; ++it) {
        Rec *mine = slots + (size_t)(it & 1) * 64 + w;
        if (threadIdx.x == 0) {
            __hip_atomic_store(&mine->v, acc + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine->i, (long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine->s, (long long)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 0) __hip_atomic_store(&mine->epoch, (unsigned long long)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else { __builtin_amdgcn_s_waitcnt(0); __hip_atomic_store(&mine->epoch, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (threadIdx.x < 64) {
            const Rec *r = slots + (size_t)(it & 1) * 64 + lane;
            unsigned spins = 0;
            bool ok = lane >= nw;
            while (!__all(ok)) {
                if (!ok) ok = __hip_atomic_load(&r->epoch, MODE == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)it;
                if (++spins > (1u << 22)) { *err = 1; break; }
            }
            double v = 1e300;
            if (lane < nw) v = __hip_atomic_load(&r->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
            if (lane == 0) s_best = v;
        }
        __syncthreads();
        if (s_best != acc) { *err = 2; }
        acc += 1.0;
        __syncthreads();
    }
    if (threadIdx.x == 0 && w == 0) out[0] = acc;
}

int main()
{
    Rec *slots; int *err; double *out;
    hipMalloc(&slots, 2 * 64 * sizeof(Rec)); hipMalloc(&err, 4); hipMalloc(&out, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode) for (int nw : {8, 17, 33, 50}) {
        const int n = 4000;
        hipMemset(slots, 0, 2 * 64 * sizeof(Rec)); hipMemset(err, 0, 4);
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k_msg<0>, dim3(nw), dim3(256), 0, 0, slots, n, err, out);
        else           hipLaunchKernelGGL(k_msg<1>, dim3(nw), dim3(256), 0, 0, slots, n, err, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("mode %d (%s), %2d workgroups: %.3f us per exchange (err=%d)\n", mode,
               mode == 0 ? "release/acquire atomics" : "relaxed + s_waitcnt", nw, ms * 1e3 / n, herr);
    }
    return 0;
}
