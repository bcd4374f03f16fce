// f64_rate.hip -- what the vector ALU of gfx950 sustains for UNFUSED double-precision
// multiply + subtract chains (the arithmetic of the simplex sweep), with the multiplier held in a
// VGPR, an SGPR, or read from LDS as a broadcast.  Build: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off f64_rate.hip -o f64_rate ; prints wave-instructions per SIMD-cycle figures.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kChains = 8, kIters = 2048, kPiv = 16;

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(double *out, const double *in, int iters)
{
    __shared__ double lds[kPiv][16];
    if (threadIdx.x < kPiv * 16) lds[threadIdx.x / 16][threadIdx.x % 16] = in[threadIdx.x];
    __syncthreads();
    double x[kChains], p[kPiv];
    for (int c = 0; c < kChains; ++c) x[c] = in[c] + threadIdx.x;
    for (int i = 0; i < kPiv; ++i) p[i] = in[32 + i] + 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kPiv; ++i) {
#pragma unroll
            for (int c = 0; c < kChains; ++c) {
                double s;
                if (MODE == 0) s = p[(i + 1) % kPiv];                       // VGPR
                else if (MODE == 1) s = in[64 + ((it + i) & 15)];            // uniform -> SGPR
                else s = lds[i][(c >> 1) + (it & 3)];                        // LDS broadcast
                if (MODE == 3) x[c] = fma(-s, p[i], x[c]);
                else { const double m = s * p[i]; x[c] = x[c] - m; }
            }
        }
    }
    double acc = 0;
    for (int c = 0; c < kChains; ++c) acc += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char *name, double *out, const double *in)
{
    const int blocks = 256 * 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, 16);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, kIters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 256 * kIters * kPiv * kChains * (MODE == 3 ? 1 : 2);   // lane-ops
    const double wave_instr = ops / 64;
    printf("%-28s %8.3f ms  %7.2f T lane-ops/s  %.2f cycles per wave-instruction per SIMD (2.4 GHz, 1024 SIMDs)\n",
           name, ms, ops / (ms * 1e-3) / 1e12, (ms * 1e-3 * 2.4e9 * 1024) / wave_instr);
}

int main()
{
    double *in, *out;
    hipMalloc(&in, 4096 * sizeof(double));
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    double h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 1.0 + 1e-6 * i;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    run<0>("mul+sub f64, VGPR operand", out, in);
    run<1>("mul+sub f64, SGPR operand", out, in);
    run<2>("mul+sub f64, LDS broadcast", out, in);
    run<3>("fma f64, VGPR operand", out, in);
    return 0;
}
