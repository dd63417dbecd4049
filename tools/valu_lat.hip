// valu_lat.hip -- latency / issue cost (cycles) of the fp64 VALU instructions the diagonal block's pivot loop is
// made of, one wave per SIMD: dependent chains of v_fma_f64, v_rsq_f64, and the v_readlane -> SGPR -> VALU round trip.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_lat.hip -o tools/bin/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NCH, int KIND>
__global__ __launch_bounds__(256) void k(double* sink, unsigned long long* cyc, int iters)
{
    double x[NCH];
    const double a = 1.0 - threadIdx.x * 1e-12, b = threadIdx.x * 1e-13;
#pragma unroll
    for (int i = 0; i < NCH; i++) x[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                if (KIND == 0) x[i] = __builtin_fma(x[i], a, b);
                else if (KIND == 1) x[i] = __builtin_amdgcn_rsq(x[i]);
                else if (KIND == 2) {       // readlane round trip: lane (u) of x -> SGPR -> fma
                    unsigned long long v = __builtin_bit_cast(unsigned long long, x[i]);
                    unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, u), hi = __builtin_amdgcn_readlane((int)(v >> 32), u);
                    double s = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                    x[i] = __builtin_fma(s, a, b);
                } else if (KIND == 3) x[i] = x[i] * a;
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; i++) s += x[i];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    if (s == 12345.678) sink[threadIdx.x] = s;
}
template <int NCH, int KIND> void run(double* sink, unsigned long long* cyc, const char* what)
{
    const int iters = 500;
    hipLaunchKernelGGL((k<NCH, KIND>), dim3(1), dim3(256), 0, 0, sink, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s chains %d: %.1f cycles per instruction (%.1f per round of %d)\n", what, NCH, (double)h / (iters * 8 * NCH),
           (double)h / (iters * 8), NCH);
}
int main()
{
    double* sink; unsigned long long* cyc;
    hipMalloc(&sink, 4096 * 8); hipMalloc(&cyc, 4096 * 8);
    run<1, 0>(sink, cyc, "v_fma_f64"); run<2, 0>(sink, cyc, "v_fma_f64"); run<4, 0>(sink, cyc, "v_fma_f64"); run<8, 0>(sink, cyc, "v_fma_f64");
    run<1, 3>(sink, cyc, "v_mul_f64"); run<4, 3>(sink, cyc, "v_mul_f64");
    run<1, 1>(sink, cyc, "v_rsq_f64"); run<2, 1>(sink, cyc, "v_rsq_f64"); run<4, 1>(sink, cyc, "v_rsq_f64");
    run<1, 2>(sink, cyc, "readlane x2 + v_fma_f64"); run<2, 2>(sink, cyc, "readlane x2 + v_fma_f64"); run<4, 2>(sink, cyc, "readlane x2 + v_fma_f64");
    return 0;
}
