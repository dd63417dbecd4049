// wide_bench.hip -- k_syrk_wide alone: achieved TF/s by K (kw k tiles), region and grid bound.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cugp_amd/csrc tools/wide_bench.hip -o tools/bin/wide_bench
#include "../cugp_amd/csrc/kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace cugp;

// ---- round 5: variants of the accumulate-form tile pass (the review's item 2) ----
// C -= acc with the accumulators started from zero: the C tile is read in the epilogue, 16 doubles per lane at a
// time (one row group m of the wave), instead of in front of the K loop; one extra rounding per entry and pass.
template <bool STREAM, int WM>
__device__ __forceinline__ void tile_sub_store(double* __restrict__ C, int ldc, const d4 (&acc)[WM][WM])
{
    const int tid = opaque_tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < WM; m++) {
        d2 v[4][WM / 2];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++) {
                const d2* src = (const d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
                v[r][np] = STREAM ? __builtin_nontemporal_load(src) : *src;
            }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++) {
                d2* dst = (d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
                const d2 o = (d2){v[r][np][0] - acc[m][2 * np][r], v[r][np][1] - acc[m][2 * np + 1][r]};
                if (STREAM) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
    }
}

// the same with the loads of row group m + 1 issued BEFORE row group m is updated and stored (two groups in flight)
template <int WM>
__device__ __forceinline__ void tile_sub_store_db(double* __restrict__ C, int ldc, const d4 (&acc)[WM][WM])
{
    const int tid = opaque_tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    d2 v[2][4][WM / 2];
    auto load = [&](int m, d2 (&dst)[4][WM / 2]) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++)
                dst[r][np] = *(const d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
    };
    load(0, v[0]);
#pragma unroll
    for (int m = 0; m < WM; m++) {
        if (m + 1 < WM) load(m + 1, v[(m + 1) & 1]);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++) {
                d2* dst = (d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
                *dst = (d2){v[m & 1][r][np][0] - acc[m][2 * np][r], v[m & 1][r][np][1] - acc[m][2 * np + 1][r]};
            }
    }
}

// all four row groups requested at once, then updated and stored (needs 64 more registers: may not fit beside the accumulators)
template <int WM>
__device__ __forceinline__ void tile_sub_store_all(double* __restrict__ C, int ldc, const d4 (&acc)[WM][WM])
{
    const int tid = opaque_tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    d2 v[WM][4][WM / 2];
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++)
                v[m][r][np] = *(const d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++) {
                d2* dst = (d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
                *dst = (d2){v[m][r][np][0] - acc[m][2 * np][r], v[m][r][np][1] - acc[m][2 * np + 1][r]};
            }
}

// MODE 0: the product's form (accumulators from C, K loop, store); 1: C in the epilogue, non-temporal; 2: the same
// with plain accesses; 3: no C read at all (timing only: what the read costs); 4: neither read nor write (timing only)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_wide_var(double* __restrict__ A, int ld, int k0, int kw, int ca, int cb, int nfull, int rev)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __builtin_amdgcn_s_setprio(1);
    const int xg = blockIdx.x & 7;
    const int xq = nfull >> 3, xr = nfull & 7;
    const int tlin = (xg < xr ? xg * (xq + 1) : xr * (xq + 1) + (xg - xr) * xq) + (blockIdx.x >> 3);
    int ti, tj;
    trap_index(rev ? nfull - 1 - tlin : tlin, cb - ca, ti, tj);
    const int i0 = (ca + ti) * TILE, j0 = (ca + tj) * TILE;
    double* C = A + (size_t)i0 * ld + j0;
    d4 acc[4][4];
    if (MODE == 0) {
        tile_load<true>(C, ld, acc);
        tile_nt<true>(A + (size_t)i0 * ld, ld, A + (size_t)j0 * ld, ld, k0 * TILE, (k0 + kw) * TILE, acc, smem);
        tile_store<true>(C, ld, acc, 1.0);
    } else {
        acc_zero(acc);
        tile_nt<false>(A + (size_t)i0 * ld, ld, A + (size_t)j0 * ld, ld, k0 * TILE, (k0 + kw) * TILE, acc, smem);
        if (MODE == 1) tile_sub_store<true>(C, ld, acc);
        else if (MODE == 2) tile_sub_store<false>(C, ld, acc);
        else if (MODE == 5) tile_sub_store_db(C, ld, acc);
        else if (MODE == 6) tile_sub_store_all(C, ld, acc);
        else if (MODE == 3) tile_store<true>(C, ld, acc, -1e-300);
        else if (acc[0][0][0] == 12345.678) tile_store<true>(C, ld, acc, 1.0);
    }
}

template <int MODE>
static int launch_wide_var(double* A, int ld, int nt, int k0, int kw, int ca, int cb, int rev)
{
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)k_wide_var<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS); attr = true; }
    const int ntiles = trap_count(nt - ca, cb - ca);
    hipLaunchKernelGGL(k_wide_var<MODE>, dim3(ntiles), dim3(256), GEMM_LDS, 0, A, ld, k0, kw, ca, cb, ntiles, rev);
    return ntiles;
}

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int nt = 64, n = nt * TILE;
    double* A;
    hipMalloc(&A, (size_t)n * n * 8);
    {
        std::vector<double> h((size_t)n * 1024);
        srand(1);
        for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
        for (int r = 0; r < n; r += 1024) hipMemcpy(A + (size_t)r * n, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    struct Case { int k0, kw, ca, cb; };
    const Case cases[] = {{0, 4, 8, 64}, {0, 4, 14, 64}, {0, 4, 19, 64}, {0, 4, 32, 64}, {0, 1, 8, 64}, {0, 2, 8, 64},
                          {0, 8, 8, 64}, {0, 4, 8, 12}, {0, 4, 8, 16}, {0, 8, 19, 64}, {0, 4, 19, 42}};
    const int grids[] = {1 << 20, 512, 448, 1024};
    for (const Case& c : cases) {
        for (int G : grids) {
            (void)G;
            int tiles = launch_syrk_wide(A, n, nt, c.k0, c.kw, c.ca, c.cb, 0, 0);
            hipDeviceSynchronize();
            hipEventRecord(a);
            const int reps = 5;
            for (int r = 0; r < reps; r++) launch_syrk_wide(A, n, nt, c.k0, c.kw, c.ca, c.cb, r & 1, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            ms /= reps;
            const double flop = (double)tiles * TILE * TILE * (c.kw * TILE) * 2.0;
            printf("cols [%2d,%2d) kw %d  grid %7d: %5d tiles (%.2f rounds of 512)  %8.1f us  %5.1f TF/s (all tile flop)\n",
                   c.ca, c.cb, c.kw, G, tiles, tiles / 512.0, ms * 1e3, flop / (ms * 1e-3) / 1e12);
        }
    }
    // round 5: the C tile in front of the K loop (0) or in the epilogue (1 non-temporal, 2 plain), and what it costs (3, 4)
    {
        typedef int (*Fn)(double*, int, int, int, int, int, int, int);
        const Fn fns[] = {launch_wide_var<0>, launch_wide_var<1>, launch_wide_var<2>, launch_wide_var<3>, launch_wide_var<4>, launch_wide_var<5>, launch_wide_var<6>};
        const char* nm[] = {"acc from C (product form)", "C in the epilogue, nt", "C in the epilogue, plain", "no C read (timing only)", "no C read, no write (timing only)",
                            "epilogue, two row groups in flight", "epilogue, all four requested at once"};
        const Case vc[] = {{0, 1, 8, 64}, {0, 2, 8, 64}, {0, 4, 8, 64}, {0, 4, 19, 64}, {0, 8, 8, 64}, {0, 16, 19, 64}};
        for (const Case& c : vc)
            for (int rep = 0; rep < 2; rep++)
                for (int m = 0; m < 7; m++) {
                    int tiles = fns[m](A, n, nt, c.k0, c.kw, c.ca, c.cb, 0);
                    hipDeviceSynchronize();
                    hipEventRecord(a);
                    const int reps = 6;
                    for (int r = 0; r < reps; r++) fns[m](A, n, nt, c.k0, c.kw, c.ca, c.cb, r & 1);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms;
                    hipEventElapsedTime(&ms, a, b);
                    ms /= reps;
                    const double flop = (double)tiles * TILE * TILE * (c.kw * TILE) * 2.0;
                    printf("VAR cols [%2d,%2d) K %4d %5d tiles (%.2f rounds)  %-36s %8.1f us  %5.1f TF/s\n", c.ca, c.cb, c.kw * TILE,
                           tiles, tiles / 512.0, nm[m], ms * 1e3, flop / (ms * 1e-3) / 1e12);
                }
    }
    // the plain uniform product at the same K for reference (k_test_gemm, K = 512: C = A(:, :512) A(:, :512)^T on 4096 x 4096)
    {
        double* C;
        hipMalloc(&C, (size_t)4096 * 4096 * 8);
        for (int K : {128, 256, 512, 1024}) {
            launch_test_gemm_nt(A, A, C, 4096, 4096, K, 0);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 5; r++) launch_test_gemm_nt(A, A, C, 4096, 4096, K, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            ms /= 5;
            printf("k_test_gemm 4096x4096 (1024 tiles, lda = K) K=%4d: %8.1f us  %5.1f TF/s\n", K, ms * 1e3,
                   2.0 * 4096 * 4096 * K / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
