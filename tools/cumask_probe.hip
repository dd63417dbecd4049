// does hipExtStreamCreateWithCUMask partition the chip?  two streams with disjoint masks, run concurrently.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(double* sink, int iters)
{
    d4 acc[8];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void k_where(unsigned* out)
{
    unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
    unsigned hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
static float timeit(hipStream_t s, int blocks, int iters, double* sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s); hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, s, sink, iters); hipEventRecord(b, s);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const bool try_mask = argc > 1;
    double* sink; hipMalloc(&sink, 4096 * 256 * 8);
    unsigned* where; hipMalloc(&where, 4096 * 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    hipStream_t s0; hipStreamCreate(&s0);
    printf("unmasked 512 blocks: %.3f ms\n", timeit(s0, 512, 2000, sink));
    printf("unmasked 512 blocks: %.3f ms\n", timeit(s0, 512, 2000, sink));
    // mask A: first 8 bits of each 32 (guess: bit index = CU id in some order); try 16 CUs total
    for (int variant = 0; try_mask && variant < 1; variant++) {
        std::vector<uint32_t> m(8, 0), inv(8, 0xffffffffu);
        if (variant == 0) { m[0] = 0xffff; }                          // bits 0..15
        if (variant == 1) { for (int w = 0; w < 8; w++) m[w] = 0x3; } // bits 32w, 32w+1
        if (variant == 2) { m[0] = 0xffffffffu; }                     // bits 0..31
        for (int w = 0; w < 8; w++) inv[w] = ~m[w];
        hipStream_t sa, sb;
        hipError_t e1 = hipExtStreamCreateWithCUMask(&sa, 8, m.data());
        hipError_t e2 = hipExtStreamCreateWithCUMask(&sb, 8, inv.data());
        printf("variant %d create: %s / %s\n", variant, hipGetErrorString(e1), hipGetErrorString(e2));
        if (e1 != hipSuccess || e2 != hipSuccess) continue;
        float ta = timeit(sa, 512, 2000, sink);
        float tb = timeit(sb, 512, 2000, sink);
        printf("  small-mask stream 512 blocks: %.3f ms ; complement stream: %.3f ms\n", ta, tb);
        // concurrency: long kernel on complement, short on small mask
        hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
        hipEventRecord(a, sb);
        hipLaunchKernelGGL(k_mfma, dim3(4096), dim3(256), 0, sb, sink, 2000);
        hipEventRecord(b, sb);
        float tsmall = timeit(sa, 16, 200, sink);
        hipEventSynchronize(b); float tbig; hipEventElapsedTime(&tbig, a, b);
        printf("  concurrent: big(4096 blocks) %.3f ms, small(16 blocks, 200 it) %.3f ms\n", tbig, tsmall);
        hipStreamDestroy(sa); hipStreamDestroy(sb);
    }
    // same concurrency test with plain + high-priority streams
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t sp, sq; hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi); hipStreamCreateWithPriority(&sq, hipStreamNonBlocking, lo);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, sq);
    hipLaunchKernelGGL(k_mfma, dim3(4096), dim3(256), 0, sq, sink, 2000);
    hipEventRecord(b, sq);
    float tsmall = timeit(sp, 16, 200, sink);
    hipEventSynchronize(b); float tbig; hipEventElapsedTime(&tbig, a, b);
    printf("priority streams (range %d..%d): big %.3f ms, small(16 blocks, 200 it) %.3f ms (alone: %.3f)\n", lo, hi, tbig, tsmall, timeit(sp, 16, 200, sink));
    return 0;
}
