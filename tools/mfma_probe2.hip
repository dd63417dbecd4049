// how fast can one SIMD issue v_mfma_f64_16x16x4_f64 with a realistic operand pattern?
// 4x4 accumulator tiles, 4 A + 4 B operand registers refreshed (cheaply) every k-step, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int LDS, int FILL>
__global__ __launch_bounds__(256, 2) void k(double* sink, int iters, unsigned long long* clk)
{
    __shared__ double sm[2048];
    d4 acc[4][4];
    double a[4], b[4];
    for (int i = 0; i < 2048; i += 256) sm[i + threadIdx.x] = 1.0 + 1e-9 * (i + threadIdx.x);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; m++) {
        a[m] = 1.0 + threadIdx.x * 1e-9 * (m + 1);
        b[m] = 1.0 - threadIdx.x * 1e-9 * (m + 1);
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (LDS) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                a[m] = sm[(threadIdx.x + 64 * m + it) & 2047];
                b[m] = sm[(threadIdx.x + 64 * m + 256 + it) & 2047];
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) { a[m] += 1e-12; b[m] -= 1e-12; }
        }
        int filler = it;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n < 4; n++) {
                acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < FILL; f++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(filler));
            }
        if (filler == -12345) sink[0] = filler;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (s == 12345.678) sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <typename F> void run(const char* name, F launch, int blocks, int iters)
{
    double* sink; unsigned long long* clk;
    hipMalloc(&sink, blocks * 256 * 8); hipMalloc(&clk, blocks * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(sink, iters / 10, clk); hipDeviceSynchronize();
    hipEventRecord(a); launch(sink, iters, clk); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    double tf = (double)blocks * 4 * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12;
    int wps = blocks / 256;
    printf("%-22s waves/SIMD=%d  %.1f TFLOP/s (wall)  in-kernel cycles per MFMA per wave = %.1f\n", name, wps, tf, (double)h / iters / 16);
    hipFree(sink); hipFree(clk);
}
int main()
{
    for (int wps : {1, 2}) {
        int blocks = 256 * wps;
        run("regs only", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL((k<0, 0>), dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, 4000);
        run("LDS operand reads", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL((k<1, 0>), dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, 4000);
        run("regs + 1 VALU/MFMA", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL((k<0, 1>), dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, 4000);
        run("regs + 4 VALU/MFMA", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL((k<0, 4>), dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, 4000);
        run("LDS + 2 VALU/MFMA", [&](double* s, int i, unsigned long long* c) { hipLaunchKernelGGL((k<1, 2>), dim3(blocks), dim3(256), 0, 0, s, i, c); }, blocks, 4000);
    }
    return 0;
}
