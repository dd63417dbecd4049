// mfma_lat.hip -- cycles per v_mfma_f64_16x16x4_f64 from ONE workgroup (one wave per SIMD): NCH independent
// accumulation chains, back to back, no memory traffic.  hipcc --offload-arch=gfx950 -O3 tools/mfma_lat.hip -o tools/bin/mfma_lat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ __launch_bounds__(256) void k(double* sink, unsigned long long* cyc, int iters)
{
    d4 acc[NCH];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < NCH; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NCH; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (s == 12345.678) sink[threadIdx.x] = s;
}
template <int NCH> void run(double* sink, unsigned long long* cyc, int blocks)
{
    const int iters = 2000;
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, sink, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("chains %d, %d workgroup(s): %.1f cycles per MFMA per wave\n", NCH, blocks, (double)h / (iters * NCH));
}
int main()
{
    double* sink; unsigned long long* cyc;
    hipMalloc(&sink, 4096 * 8); hipMalloc(&cyc, 4096 * 8);
    for (int blocks : {1, 512}) { run<1>(sink, cyc, blocks); run<2>(sink, cyc, blocks); run<4>(sink, cyc, blocks); run<8>(sink, cyc, blocks); run<16>(sink, cyc, blocks); }
    return 0;
}
