// gemm_k_bench.hip -- the tile product as a function of its depth: C[m x n] = A[m x k] B[n x k]^T on 4096
// uniform 128x128 tiles (8192 x 8192 output), k = 128 .. 8192.  t(k) = t0 + c*k separates the per-tile cost
// (prologue, epilogue, workgroup launch) from the steady-state rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cugp_amd/csrc tools/gemm_k_bench.hip -o tools/bin/gemm_k_bench
#include "../cugp_amd/csrc/kernels.hip"

#include <cstdio>
#include <vector>

using namespace cugp;

// the same product with persistent workgroups: `slots` workgroups walk the tile list
__global__ __launch_bounds__(256, 2) void k_gemm_persistent(const double* __restrict__ A, const double* __restrict__ B,
                                                            double* __restrict__ C, int n, int k, int mt, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ti = t % mt, tj = t / mt;
        d4 acc[4][4];
        acc_zero(acc);
        tile_nt<false>(A + (size_t)ti * TILE * k, k, B + (size_t)tj * TILE * k, k, 0, k, acc, smem);
        tile_store(C + (size_t)ti * TILE * n + tj * TILE, n, acc, 1.0);
        __syncthreads();
    }
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int m = 8192, n = 8192, kmax = 8192;
    double *A, *B, *C;
    hipMalloc(&A, (size_t)m * kmax * 8); hipMalloc(&B, (size_t)n * kmax * 8); hipMalloc(&C, (size_t)m * n * 8);
    std::vector<double> h((size_t)m * kmax);
    unsigned long long st = 88172645463325252ull;
    for (double& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (double)(st >> 11) / 9007199254740992.0 - 0.5; }
    hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k : {128, 256, 512, 1024, 2048, 8192}) {
        // operands are read as m x k with leading dimension k: the first k columns of a packed m x k matrix
        launch_test_gemm_nt(A, B, C, m, n, k, 0);
        hipDeviceSynchronize();
        const int reps = k <= 1024 ? 20 : 4;
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) launch_test_gemm_nt(A, B, C, m, n, k, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        const double tiles = (double)(m / 128) * (n / 128), rounds = tiles / 512.0;
        printf("k=%5d  %8.1f us  %5.1f TFLOP/s   per tile-slot %6.2f us  (%.2f us per 128 of k)", k, ms * 1e3,
               2.0 * m * n * k / ms / 1e9, ms * 1e3 / rounds, ms * 1e3 / rounds / (k / 128.0));
        if (k <= 1024) {
            (void)hipFuncSetAttribute((const void*)k_gemm_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            for (int slots : {512, 768}) {
                hipLaunchKernelGGL(k_gemm_persistent, dim3(slots), dim3(256), GEMM_LDS, 0, A, B, C, n, k, m / 128, (int)tiles);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < reps; r++)
                    hipLaunchKernelGGL(k_gemm_persistent, dim3(slots), dim3(256), GEMM_LDS, 0, A, B, C, n, k, m / 128, (int)tiles);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms2; hipEventElapsedTime(&ms2, e0, e1);
                ms2 /= reps;
                printf("   persistent x%d: %8.1f us %5.1f TF", slots, ms2 * 1e3, 2.0 * m * n * k / ms2 / 1e9);
            }
        }
        printf("\n");
    }
    return 0;
}
