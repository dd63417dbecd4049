// wide_bench.hip -- k_syrk_wide alone: achieved TF/s by K (kw k tiles), region and grid bound.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cugp_amd/csrc tools/wide_bench.hip -o tools/bin/wide_bench
#include "../cugp_amd/csrc/kernels.hip"

#include <cmath>
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

// ---- round 5: the 8-wave tile product once more, now with the C tile in the epilogue ----
// 512 threads, waves 4 x 2, 32 x 64 outputs per wave (2 x 4 MFMA tiles, 32 accumulators), the LDS image of the 4-wave
// product; two such workgroups put FOUR waves on every SIMD, so while one workgroup is in its prologue or epilogue the
// other still has two waves per SIMD -- what the fp64 MFMA pipe needs to run at its full rate.
template <bool NEGA>
__device__ __forceinline__ void tile_nt8(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg, int ldb,
                                         int kbeg, int kend, d4 (&acc)[2][4], char* smem)
{
    typedef Geo<4> G;
    const int t = opaque_tid();
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int srow = t >> 3, skp = t & 7;                 // 64 rows per pass, 2 passes
    const double* ag = Ag + (size_t)srow * lda + 2 * skp;
    const double* bg = Bg + (size_t)srow * ldb + 2 * skp;
    d2 ra[2], rb[2];
    int wa[2], wb[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        wa[q] = skp * G::PLANE + (srow + 64 * q) * 16;
        wb[q] = G::OPER + skp * G::PLANE + bpos<4>(srow + 64 * q) * 16;
    }
    const int fr = lane & 15, fk = lane >> 4;
    const int abase = (fk >> 1) * G::PLANE + (wr * 32 + fr) * 16 + (fk & 1) * 8;
    const int bbase = G::OPER + (fk >> 1) * G::PLANE + (wc * 64 + fr) * 16 + (fk & 1) * 8;
    const int nk = (kend - kbeg) / BK;
    if (nk <= 0) return;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        ra[q] = *(const d2*)(ag + (size_t)(64 * q) * lda + kbeg);
        rb[q] = *(const d2*)(bg + (size_t)(64 * q) * ldb + kbeg);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        *(d2*)(smem + wa[q]) = NEGA ? -ra[q] : ra[q];
        *(d2*)(smem + wb[q]) = rb[q];
    }
    __syncthreads();
#define STAGE8(HALF, MORE, KNEXT)                                                                   \
    do {                                                                                            \
        const char* cur_ = smem + (HALF) * G::STAGE;                                                \
        char* nxt_ = smem + (1 - (HALF)) * G::STAGE;                                                \
        const bool more_ = (MORE);                                                                  \
        if (more_) {                                                                                \
            const int k_ = (KNEXT);                                                                 \
            _Pragma("unroll") for (int q = 0; q < 2; q++) {                                         \
                ra[q] = *(const d2*)(ag + (size_t)(64 * q) * lda + k_);                             \
                rb[q] = *(const d2*)(bg + (size_t)(64 * q) * ldb + k_);                             \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int kk = 0; kk < BK / 4; kk++) {                                     \
            double a[2], b[4];                                                                      \
            _Pragma("unroll") for (int m = 0; m < 2; m++) a[m] = *(const double*)(cur_ + abase + kk * 2 * G::PLANE + m * 256); \
            _Pragma("unroll") for (int n = 0; n < 4; n++) b[n] = *(const double*)(cur_ + bbase + kk * 2 * G::PLANE + n * 256); \
            _Pragma("unroll") for (int m = 0; m < 2; m++)                                           \
                _Pragma("unroll") for (int n = 0; n < 4; n++)                                       \
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0); \
        }                                                                                           \
        if (more_) {                                                                                \
            _Pragma("unroll") for (int q = 0; q < 2; q++) {                                         \
                *(d2*)(nxt_ + wa[q]) = NEGA ? -ra[q] : ra[q];                                       \
                *(d2*)(nxt_ + wb[q]) = rb[q];                                                       \
            }                                                                                       \
        }                                                                                           \
        __syncthreads();                                                                            \
    } while (0)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        STAGE8(0, true, kbeg + (kt + 1) * BK);
        STAGE8(1, kt + 2 < nk, kbeg + (kt + 2) * BK);
    }
    if (kt < nk) STAGE8(0, false, 0);
#undef STAGE8
}

// MODE: 0 C -= acc in the epilogue (plain), 1 plain product store (no C read: C = -acc), 2 neither
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_wide8(double* __restrict__ A, int ld, int k0, int kw, int ca, int cb, int nfull, int rev)
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
    d4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
    tile_nt8<false>(A + (size_t)i0 * ld, ld, A + (size_t)j0 * ld, ld, k0 * TILE, (k0 + kw) * TILE, acc, smem);
    const int tid = opaque_tid(), lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    if (MODE == 2 && acc[0][0][0] != 12345.678) return;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        d2 v[4][2];
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int np = 0; np < 2; np++)
                    v[r][np] = *(const d2*)(C + (size_t)(wr * 32 + m * 16 + (lane >> 4) + 4 * r) * ld + wc * 64 + np * 32 + 2 * (lane & 15));
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < 2; np++) {
                d2* dst = (d2*)(C + (size_t)(wr * 32 + m * 16 + (lane >> 4) + 4 * r) * ld + wc * 64 + np * 32 + 2 * (lane & 15));
                if (MODE == 0) *dst = (d2){v[r][np][0] - acc[m][2 * np][r], v[r][np][1] - acc[m][2 * np + 1][r]};
                else *dst = (d2){-acc[m][2 * np][r], -acc[m][2 * np + 1][r]};
            }
    }
}

template <int MODE>
static int launch_wide8(double* A, int ld, int nt, int k0, int kw, int ca, int cb, int rev)
{
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)k_wide8<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS); attr = true; }
    const int ntiles = trap_count(nt - ca, cb - ca);
    hipLaunchKernelGGL(k_wide8<MODE>, dim3(ntiles), dim3(512), GEMM_LDS, 0, A, ld, k0, kw, ca, cb, ntiles, rev);
    return ntiles;
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
    {   // the 8-wave product against the 4-wave one on identical inputs (one pass over cols [8,64), K = 512)
        double* A2;
        hipMalloc(&A2, (size_t)n * n * 8);
        hipMemcpy(A2, A, (size_t)n * n * 8, hipMemcpyDeviceToDevice);
        launch_wide_var<2>(A, n, nt, 0, 4, 8, 64, 0);
        launch_wide8<0>(A2, n, nt, 0, 4, 8, 64, 0);
        hipDeviceSynchronize();
        std::vector<double> r1((size_t)n), r2((size_t)n);
        double worst = 0, big = 0;
        for (int row : {1024, 1500, 4095, 5000, 8191}) {
            hipMemcpy(r1.data(), A + (size_t)row * n, (size_t)n * 8, hipMemcpyDeviceToHost);
            hipMemcpy(r2.data(), A2 + (size_t)row * n, (size_t)n * 8, hipMemcpyDeviceToHost);
            for (int c = 0; c < n; c++) { const double d = fabs(r1[c] - r2[c]); if (d > worst) worst = d; if (fabs(r1[c]) > big) big = fabs(r1[c]); }
        }
        printf("CHECK 8-wave vs 4-wave epilogue form: max |difference| %.3e (entries up to %.3e)\n", worst, big);
        hipFree(A2);
    }
    // round 5: the C tile in front of the K loop (0) or in the epilogue (1 non-temporal, 2 plain), and what it costs (3, 4)
    {
        typedef int (*Fn)(double*, int, int, int, int, int, int, int);
        const Fn fns[] = {launch_wide_var<0>, launch_wide8<0>, launch_wide_var<2>, launch_wide_var<3>, launch_wide_var<4>, launch_wide8<1>, launch_wide8<2>};
        const char* nm[] = {"acc from C (product form)", "8 WAVES, C in the epilogue", "C in the epilogue, plain", "no C read (timing only)", "no C read, no write (timing only)",
                            "8 WAVES, no C read", "8 WAVES, neither"};
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
