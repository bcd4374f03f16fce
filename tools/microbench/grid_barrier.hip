// grid_barrier.hip -- cost of an in-kernel barrier across a few dozen workgroups on gfx950
// (arrive: agent-scope atomic add on a monotonically increasing counter; wait: spin on an
// agent-scope load), with and without a small data exchange through global memory, against the
// cost of a kernel boundary.  Spins are bounded: a broken barrier reports instead of hanging.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ bool grid_barrier(unsigned *bar, unsigned target)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_barriers(unsigned *bar, double *buf, int n, int exchange, int *err)
{
    const unsigned nwg = gridDim.x;
    double acc = 0.0;
    for (int it = 0; it < n; ++it) {
        if (exchange) {
            // every workgroup publishes 256 doubles, then reads its neighbour's after the barrier
            buf[(size_t)(it & 1) * nwg * 256 + blockIdx.x * 256 + threadIdx.x] = acc + it + threadIdx.x;
        }
        if (!grid_barrier(bar, (unsigned)(it + 1) * nwg)) { if (threadIdx.x == 0) *err = 1; return; }
        if (exchange) {
            const unsigned nb = (blockIdx.x + 1) % nwg;
            const double v = buf[(size_t)(it & 1) * nwg * 256 + nb * 256 + threadIdx.x];
            if (v != acc + it + threadIdx.x) { *err = 2; }
            acc = v * 0.0 + acc + 1.0;                       // every workgroup keeps the same acc
        }
    }
    if (exchange && threadIdx.x == 0 && blockIdx.x == 0) buf[0] = acc;
}

__global__ void k_empty(double *buf) { if (buf == nullptr) buf[0] = 1; }

int main()
{
    unsigned *bar; double *buf; int *err;
    hipMalloc(&bar, 4); hipMalloc(&buf, 2 * 64 * 256 * 8); hipMalloc(&err, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nwg : {8, 33, 50, 64}) for (int ex = 0; ex < 2; ++ex) {
        const int n = 2000;
        hipMemset(bar, 0, 4); hipMemset(err, 0, 4);
        hipLaunchKernelGGL(k_barriers, dim3(nwg), dim3(256), 0, 0, bar, buf, 10, ex, err);
        hipMemset(bar, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_barriers, dim3(nwg), dim3(256), 0, 0, bar, buf, n, ex, err);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("%2d workgroups, exchange=%d: %.3f us per barrier (err=%d)\n", nwg, ex, ms * 1e3 / n, herr);
    }
    const int n = 2000;
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(33), dim3(256), 0, 0, buf);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("kernel boundary (empty 33-workgroup launches back to back): %.3f us per launch\n", ms * 1e3 / n);
    return 0;
}
