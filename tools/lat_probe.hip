// dependent-chain latencies (cycles) of the instructions on the Cholesky pivot chain, one wave per CU
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int src)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffull), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double seed)
{
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) x = __builtin_fma(x, y, 1e-9);                       // dependent v_fma_f64
            if (MODE == 1) x = __builtin_amdgcn_rsq(x) + 1.0;                    // rsq + add
            if (MODE == 2) x = x * y;                                           // dependent v_mul_f64
            if (MODE == 3) x = rl(x, u) * y;                                    // readlane pair + mul
            if (MODE == 4) x = __builtin_amdgcn_rsq(x);                          // rsq only
            if (MODE == 5) { float f = (float)x; f = __builtin_amdgcn_rsqf(f); x = (double)f + 1.0; }  // cvt, rsq_f32, cvt, add
            if (MODE == 6) x = __builtin_amdgcn_rcp(x) + 1.0;                    // rcp + add
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double* o; unsigned long long* c; hipMalloc(&o, 64 * 8); hipMalloc(&c, 8);
    const char* names[] = {"v_fma_f64 dependent", "v_rsq_f64 + v_add_f64", "v_mul_f64 dependent", "readlane x2 + v_mul_f64",
                           "v_rsq_f64 dependent", "cvt + v_rsq_f32 + cvt + add", "v_rcp_f64 + v_add_f64"};
    for (int m = 0; m < 7; m++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, o, c, 1.5); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, o, c, 1.5); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, o, c, 1.5); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, o, c, 1.5); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, o, c, 1.5); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, o, c, 1.5); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, o, c, 1.5); break;
            }
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-32s %.1f cycles per step\n", names[m], (double)h / 4096.0);
    }
    return 0;
}
