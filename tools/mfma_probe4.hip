// Private (per-wave) staging, no workgroup barrier: per 32 MFMAs (BK=8) each wave loads its own 64x8 A and B
// strips (8 x 16 B per lane), writes them to its own LDS region and reads fragments back.  Compare with the
// shared-staging stage of mfma_probe3 (~60 TF/s with random data and strided loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int AHEAD>
__global__ __launch_bounds__(256, 2) void k(double* sink, const double* __restrict__ src, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    char* my = sm + wave * 16640;                          // 2 buffers x (A 4160 + B 4160)
    d4 acc[4][4];
    d2 g[8], g2[8];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
    // my strip: rows (lane>>2) + 16 q, 16-byte chunk lane&3 of an 8-double k range
    const double* gp = src + ((size_t)((blockIdx.x * 4 + wave) % 120) * 64 + (lane >> 2)) * 8192 + (lane & 3) * 2;
    const int wofs = (lane & 3) * 1040 + (lane >> 2) * 16;
    const int rofs = (lane & 15) * 16 + (lane >> 5) * 1040 + ((lane >> 4) & 1) * 8;
#pragma unroll
    for (int q = 0; q < 8; q++) { g[q] = *(const d2*)(gp + (size_t)(q & 3) * 16 * 8192 + (q >> 2) * 64 * 8192); g2[q] = g[q]; }
#pragma unroll
    for (int q = 0; q < 8; q++) *(d2*)(my + (q >> 2) * 4160 + wofs + (q & 3) * 256) = g[q];
    for (int it = 0; it < iters; it++) {
        const int buf = (it & 1) * 8320, nb = 8320 - buf;
        if (AHEAD) {
#pragma unroll
            for (int q = 0; q < 8; q++) { g[q] = g2[q]; g2[q] = *(const d2*)(gp + (size_t)(q & 3) * 16 * 8192 + (q >> 2) * 64 * 8192 + ((it + 2) & 511) * 8); }
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) g[q] = *(const d2*)(gp + (size_t)(q & 3) * 16 * 8192 + (q >> 2) * 64 * 8192 + ((it + 1) & 511) * 8);
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                a[m] = *(const double*)(my + buf + rofs + kk * 2080 + m * 256);
                b[m] = *(const double*)(my + buf + 4160 + rofs + kk * 2080 + m * 256);
            }
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) *(d2*)(my + nb + (q >> 2) * 4160 + wofs + (q & 3) * 256) = g[q];
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (s == 12345.678) sink[blockIdx.x * 256 + t] = s;
}

template <int AHEAD> void run(const char* name)
{
    const int blocks = 512, iters = 4000;
    const size_t nsrc = (size_t)8192 * 8192;
    double *sink, *src;
    hipMalloc(&sink, blocks * 256 * 8); hipMalloc(&src, nsrc * 8);
    static double* h = nullptr;
    if (!h) { h = (double*)malloc(nsrc * 8); srand(7); for (size_t i = 0; i < nsrc; i++) h[i] = (rand() / (double)RAND_MAX - 0.5) * 2.0; }
    hipMemcpy(src, h, nsrc * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k<AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 66560);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<AHEAD>), dim3(blocks), dim3(256), 66560, 0, sink, src, 400);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<AHEAD>), dim3(blocks), dim3(256), 66560, 0, sink, src, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-60s %.1f TFLOP/s\n", name, (double)blocks * 4 * iters * 32 * 2048.0 / (ms * 1e-3) / 1e12);
    hipFree(sink); hipFree(src);
}
int main()
{
    run<0>("private staging, BK=8, no barrier, loads 1 stage ahead");
    run<1>("private staging, BK=8, no barrier, loads 2 stages ahead");
    run<0>("private staging, BK=8, no barrier, loads 1 stage ahead (again)");
    return 0;
}
