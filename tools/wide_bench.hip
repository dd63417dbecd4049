// wide_bench.hip -- k_syrk_wide alone: achieved TF/s by K (kw k tiles), region and grid bound.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cugp_amd/csrc tools/wide_bench.hip -o tools/bin/wide_bench
#include "../cugp_amd/csrc/kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace cugp;

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
