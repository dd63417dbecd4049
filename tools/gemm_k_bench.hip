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
    int it = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, it++) {
        const int ti = t % mt, tj = t / mt;
        d4 acc[4][4];
        acc_zero(acc);
        if (it == 3) TILE_STAMP(0);
#ifdef CUGP_TILE_STAMPS
        if (threadIdx.x == 0 && blockIdx.x == 200) g_tile_stamp_on = (it == 3);
#endif
#ifdef CUGP_TILE_STAMPS
        if (threadIdx.x == 0 && (blockIdx.x == 200 || blockIdx.x == 7) && it < 10)
            g_tile_stamps[(blockIdx.x == 7 ? 16 : 48) + it] = __builtin_amdgcn_s_memrealtime();
#endif
        tile_nt<false>(A + (size_t)ti * TILE * k, k, B + (size_t)tj * TILE * k, k, 0, k, acc, smem);
        if (it == 3) TILE_STAMP(2);
        tile_store(C + (size_t)ti * TILE * n + tj * TILE, n, acc, 1.0);
        if (it == 3) TILE_STAMP(3);
#ifdef CUGP_TILE_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (it == 3) TILE_STAMP(4);
#endif
        __syncthreads();
        if (it == 3) TILE_STAMP(5);
        if (it == 4) TILE_STAMP(6);                       // (stamp 1 is overwritten by every tile: last one wins)
    }
#ifdef CUGP_TILE_STAMPS
    if (threadIdx.x == 0 && (blockIdx.x == 200 || blockIdx.x == 7) && it < 10)
        g_tile_stamps[(blockIdx.x == 7 ? 16 : 48) + it] = __builtin_amdgcn_s_memrealtime();
#endif
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
            for (int slots : {512, 256}) {
                hipLaunchKernelGGL(k_gemm_persistent, dim3(slots), dim3(256), GEMM_LDS, 0, A, B, C, n, k, m / 128, (int)tiles);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < reps; r++)
                    hipLaunchKernelGGL(k_gemm_persistent, dim3(slots), dim3(256), GEMM_LDS, 0, A, B, C, n, k, m / 128, (int)tiles);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms2; hipEventElapsedTime(&ms2, e0, e1);
                ms2 /= reps;
                printf("   persistent x%d: %8.1f us %5.1f TF", slots, ms2 * 1e3, 2.0 * m * n * k / ms2 / 1e9);
#ifdef CUGP_TILE_STAMPS
                {
                    unsigned long long st[64];
                    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tile_stamps), sizeof(st));
                    printf("\n      tile times (us) workgroup 7:");
                    for (int i = 0; i < 8; i++) printf(" %.1f", (st[17 + i] - st[16 + i]) / 100.0);
                    printf("   workgroup 200:");
                    for (int i = 0; i < 8; i++) printf(" %.1f", (st[49 + i] - st[48 + i]) / 100.0);
                    printf("\n      [%d workgroups] tile 3 (cycles): prologue %lld, k loop %lld, store issue %lld, store drain %lld, barrier %lld",
                           slots, (long long)(st[1] - st[0]), (long long)(st[2] - st[1]), (long long)(st[3] - st[2]),
                           (long long)(st[4] - st[3]), (long long)(st[5] - st[4]));
                    printf("\n      two tiles: %lld core cycles in %.2f us (100 MHz counter) = %.0f MHz",
                           (long long)(st[6] - st[0]), (st[38] - st[32]) / 100.0, (st[6] - st[0]) / ((st[38] - st[32]) / 100.0));
                    printf("\n      stamps (cycles from tile start): loop end %lld, stores issued %lld, stores done %lld, barrier %lld, next tile at %lld",
                           (long long)(st[2] - st[0]), (long long)(st[3] - st[0]), (long long)(st[4] - st[0]),
                           (long long)(st[5] - st[0]), (long long)(st[6] - st[0]));
                }
#endif
            }
        }
        printf("\n");
    }
    return 0;
}
