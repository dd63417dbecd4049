// Which part of the tile-product stage costs MFMA throughput?  Synthetic stage = 64 MFMAs (4x4 tiles x 4 k-steps)
// + optional 16 ds_read2_b64-equivalents, 8 ds_write_b128, 8 global_load_dwordx4, 1 barrier.  2 WGs of 4 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int RD, int WR, int GL, int BAR, int PAT, int OPT = 0>
__global__ __launch_bounds__(256, 2) void k(double* sink, const double* __restrict__ src, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char sm[];
    d4 acc[4][4];
    double a[4], b[4];
    d2 g[8];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 8192; i += 256) ((double*)sm)[i] = PAT ? src[i] : 1.0 + 1e-9 * i;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; m++) {
        a[m] = PAT ? src[t + 256 * m] : 1.0 + t * 1e-9 * (m + 1);
        b[m] = PAT ? src[t + 256 * m + 1024] : 1.0 - t * 1e-9 * (m + 1);
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int q = 0; q < 8; q++) g[q] = (d2){1.0, 1.0};
    const int rbase = (lane & 15) * 16 + (lane >> 5) * 2064 + ((lane >> 4) & 1) * 8;
    const int wbase = (t & 7) * 2064 + (t >> 3) * 16;
    const double* gp = PAT ? src + ((size_t)(blockIdx.x % 60) * 128 + (t >> 3)) * 8192 + (t & 7) * 2
                           : src + (size_t)blockIdx.x * 4096 + t * 2;
    if (OPT == 1 && (blockIdx.x & 1)) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    d2 g2[8];
#pragma unroll
    for (int q = 0; q < 8; q++) g2[q] = (d2){1.0, 1.0};
    for (int it = 0; it < iters; it++) {
        if (GL && PAT == 2) {
#pragma unroll
            for (int q = 0; q < 8; q++) { g[q] = g2[q]; g2[q] = *(const d2*)(gp + (size_t)(q >> 1) * 32 * 8192 + ((it + 1) & 255) * 16); }
        } else if (GL) {
#pragma unroll
            for (int q = 0; q < 8; q++)
                g[q] = PAT ? *(const d2*)(gp + (size_t)(q >> 1) * 32 * 8192 + (q & 1) * 64 * 8192 * 0 + (it & 255) * 16)
                           : *(const d2*)(gp + q * 512 + (it & 63) * 16);
        }
        if (OPT == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            if (OPT == 3 && kk == 2 && WR) {
#pragma unroll
                for (int q = 0; q < 8; q++) *(d2*)(sm + 33024 + (q & 1) * 16512 + wbase + (q >> 1) * 512) = g[q];
            }
            if (RD) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    a[m] = *(const double*)(sm + rbase + kk * 4128 + m * 256);
                    b[m] = *(const double*)(sm + 16512 + rbase + kk * 4128 + m * 256);
                }
            }
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (OPT == 2) __builtin_amdgcn_s_setprio(0);
        if (WR && OPT != 3) {
#pragma unroll
            for (int q = 0; q < 8; q++) *(d2*)(sm + 33024 + (q & 1) * 16512 + wbase + (q >> 1) * 512) = g[q];
        }
        if (BAR) __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
#pragma unroll
    for (int q = 0; q < 8; q++) s += g[q][0];
    if (s == 12345.678) sink[blockIdx.x * 256 + t] = s;
}

template <int RD, int WR, int GL, int BAR, int PAT, int OPT = 0> void run(const char* name)
{
    const int blocks = 512, iters = 2000;
    double *sink, *src;
    const size_t nsrc = (size_t)8192 * 8192;
    hipMalloc(&sink, blocks * 256 * 8); hipMalloc(&src, nsrc * 8);
    {
        static double* h = nullptr;
        if (!h) { h = (double*)malloc(nsrc * 8); srand(7); for (size_t i = 0; i < nsrc; i++) h[i] = (rand() / (double)RAND_MAX - 0.5) * 2.0; }
        hipMemcpy(src, h, nsrc * 8, hipMemcpyHostToDevice);
    }
    hipFuncSetAttribute((const void*)k<RD, WR, GL, BAR, PAT, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 66048);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<RD, WR, GL, BAR, PAT, OPT>), dim3(blocks), dim3(256), 66048, 0, sink, src, 200);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<RD, WR, GL, BAR, PAT, OPT>), dim3(blocks), dim3(256), 66048, 0, sink, src, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %.1f TFLOP/s\n", name, (double)blocks * 4 * iters * 64 * 2048.0 / (ms * 1e-3) / 1e12);
    hipFree(sink); hipFree(src);
}
int main()
{
    run<1, 1, 1, 1, 1, 0>("full stage (random, strided loads)");
    run<1, 1, 1, 1, 1, 1>("  + odd workgroups start half a stage late");
    run<1, 1, 1, 1, 1, 2>("  + s_setprio(1) around the MFMA block");
    run<1, 1, 1, 1, 1, 3>("  + LDS writes in the middle of the MFMA block");
    run<1, 1, 1, 1, 2, 3>("  + loads 2 ahead + writes in the middle");
    run<1, 1, 1, 1, 1, 0>("full stage (again)");
    return 0;
}
