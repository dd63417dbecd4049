// corun_bench.hip -- do two large tile-product launches on two streams cost more side by side than one after the other?
// (k_lauum share a=60 w=4 on matrix B + k_syrk_wide K=512 on matrix A, 64 x 64 tiles each; and each with a copy of itself)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I cugp_amd/csrc tools/corun_bench.hip -o tools/bin/corun_bench
#include "../cugp_amd/csrc/kernels.hip"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace cugp;

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int nt = 64, n = nt * TILE;
    double *A, *B, *U, *A2;
    const size_t bytes = (size_t)n * n * 8;
    hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&U, bytes); hipMalloc(&A2, bytes);
    {
        std::vector<double> h((size_t)n * 1024);
        srand(1);
        for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
        for (double* P : {A, U, A2})
            for (int r = 0; r < n; r += 1024) hipMemcpy(P + (size_t)r * n, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    hipMemset(B, 0, bytes);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    auto wide = [&](double* M, hipStream_t s, int kw) { launch_syrk_wide(M, n, nt, 0, kw, 8, 64, 0, s); };
    auto lau = [&](hipStream_t s) { launch_lauum(U, B, n, 60, 4, s); };
    auto bord = [&](hipStream_t s) { launch_trtri_border1(A2, A2, U, n, 32, 32, 28, 32, s); };   // 32 x 32 tiles, K = 512
    auto timeit = [&](const char* what, auto&& f) {
        f();
        hipDeviceSynchronize();
        double best = 1e30;
        for (int r = 0; r < 5; r++) {
            auto t0 = std::chrono::steady_clock::now();
            f();
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        printf("%-58s %8.1f us\n", what, best);
        return best;
    };
    const double w = timeit("k_syrk_wide K=512 (1596 tiles)", [&] { wide(A, s1, 4); });
    const double l = timeit("k_lauum share a=60 w=4 (2080 tiles)", [&] { lau(s1); });
    const double b = timeit("k_trtri_border step 1 (1024 tiles, K=512)", [&] { bord(s1); });
    const double w2 = timeit("k_syrk_wide K=2048", [&] { wide(A, s1, 16); });
    timeit("wide K=512 then lauum, one stream", [&] { wide(A, s1, 4); lau(s1); });
    timeit("wide K=512 beside lauum, two streams", [&] { wide(A, s1, 4); lau(s2); });
    timeit("wide K=512 beside a second wide K=512 (other matrix)", [&] { wide(A, s1, 4); wide(A2, s2, 4); });
    timeit("lauum beside border", [&] { lau(s1); bord(s2); });
    timeit("wide K=2048 beside lauum + border (same stream)", [&] { wide(A, s1, 16); lau(s2); bord(s2); });
    printf("sums: wide+lauum %.1f  2 x wide %.1f  lauum+border %.1f  wide2048+lauum+border %.1f\n", w + l, 2 * w, l + b, w2 + l + b);
    return 0;
}
