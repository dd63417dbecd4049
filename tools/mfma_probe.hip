// fp64 MFMA / VALU peak probe with in-kernel clock (s_memtime / s_memrealtime).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* sink, int iters, unsigned long long* clk)
{
    d4 acc[NACC];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k_fma(double* sink, int iters, unsigned long long* clk)
{
    double acc[16];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    if (s == 12345.678) sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename F>
void run(const char* name, F launch, int blocks, int iters, double flop_per_thread_iter_wave)
{
    double* sink; unsigned long long* clk;
    hipMalloc(&sink, blocks * 256 * 8); hipMalloc(&clk, blocks * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(sink, iters / 10, clk);
    hipDeviceSynchronize();
    hipEventRecord(a); launch(sink, iters, clk); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    double tf = (double)blocks * 4 * iters * flop_per_thread_iter_wave / (ms * 1e-3) / 1e12;
    printf("%-28s blocks=%4d  %.3f ms  %.1f TFLOP/s  clk=%.2f GHz  cyc/iter=%.1f\n", name, blocks, ms, tf, ghz,
           (double)h[0] / iters);
    hipFree(sink); hipFree(clk);
}

int main()
{
    const int it = 20000;
    for (int bpc : {1, 2, 4}) {
        int blocks = 256 * bpc;
        run("mfma_f64 4acc", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, it, 4 * 2048.0);
        run("mfma_f64 8acc", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, it, 8 * 2048.0);
        run("mfma_f64 16acc", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, it, 16 * 2048.0);
        run("v_fma_f64 16acc", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, it, 16 * 64 * 2.0);
    }
    // few blocks: clock when the chip is lightly loaded
    run("mfma_f64 8acc (16 blocks)", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL(k_mfma<8>, dim3(16), dim3(256), 0, 0, s, i, c); }, 16, it, 8 * 2048.0);
    return 0;
}
