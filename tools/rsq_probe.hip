// precision of v_rsq_f64 and of Newton steps on it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, double* y3, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    y0[i] = y;
    double e = __builtin_fma(-v * y, y, 1.0); y = __builtin_fma(0.5 * y, e, y); y1[i] = y;
    e = __builtin_fma(-v * y, y, 1.0); y = __builtin_fma(0.5 * y, e, y); y2[i] = y;
    e = __builtin_fma(-v * y, y, 1.0); y = __builtin_fma(0.5 * y, e, y); y3[i] = y;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), r[4];
    srand(3);
    for (int i = 0; i < n; i++) x[i] = exp((rand() / (double)RAND_MAX - 0.5) * 40.0);
    double *dx, *dy[4];
    hipMalloc(&dx, n * 8); hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (int j = 0; j < 4; j++) { hipMalloc(&dy[j], n * 8); r[j].resize(n); }
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy[0], dy[1], dy[2], dy[3], n);
    for (int j = 0; j < 4; j++) hipMemcpy(r[j].data(), dy[j], n * 8, hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; j++) {
        long double m = 0;
        for (int i = 0; i < n; i++) {
            long double ex = 1.0L / sqrtl((long double)x[i]);
            long double e = fabsl(((long double)r[j][i] - ex) / ex);
            if (e > m) m = e;
        }
        printf("rsq + %d Newton: max rel err %.3Le (2^%.1Lf)\n", j, m, log2l(m));
    }
    return 0;
}
