// ld_bench.hip -- does the leading dimension (a power of two at N = 8192: rows 64 KiB apart) cost the tile products
// memory-channel conflicts?  k_syrk_wide (K = 128..2048 trailing updates) and the whole-matrix k_lauum on a
// 64 x 64-tile matrix with ld = 8192 + pad doubles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I cugp_amd/csrc tools/ld_bench.hip -o tools/bin/ld_bench
// Round 3: no effect (pads of 16..256 doubles: every figure within 1 %), so ld stays npad.  The same bench under
// `rocprofv3 --pmc FETCH_SIZE --kernel-trace` gave the fetch volumes of the XCD-patch experiment in DESIGN.md section 8.
#include "../cugp_amd/csrc/kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace cugp;

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int nt = 64, n = nt * TILE;
    const int maxpad = 512;
    double *A, *B;
    hipMalloc(&A, (size_t)n * (n + maxpad) * 8);
    hipMalloc(&B, (size_t)n * (n + maxpad) * 8);
    {
        std::vector<double> h((size_t)(n + maxpad) * 1024);
        srand(1);
        for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
        for (int r = 0; r < n; r += 1024) hipMemcpy(A + (size_t)r * (n + maxpad), h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    hipMemset(B, 0, (size_t)n * (n + maxpad) * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int pad : {0, 16, 32, 64, 128, 256, 0}) {
        const int ld = n + pad;
        for (int kw : {1, 4, 16}) {
            int tiles = launch_syrk_wide(A, ld, nt, 0, kw, 8, 64, 0, 0);
            hipDeviceSynchronize();
            hipEventRecord(a);
            const int reps = 5;
            for (int r = 0; r < reps; r++) launch_syrk_wide(A, ld, nt, 0, kw, 8, 64, r & 1, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            ms /= reps;
            const double flop = (double)tiles * TILE * TILE * (kw * TILE) * 2.0;
            printf("ld 8192+%3d  k_syrk_wide K=%4d %5d tiles  %8.1f us  %5.1f TF/s\n", pad, kw * TILE, tiles, ms * 1e3,
                   flop / (ms * 1e-3) / 1e12);
        }
        for (int blk : {60, 0}) {     // one share of K^-1 (rows [60,64): K = 512 for all but the last rows) / the whole product
            const int aa = blk, ww = nt - blk;
            launch_lauum(A, B, ld, aa, ww, 0);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 3; r++) launch_lauum(A, B, ld, aa, ww, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            ms /= 3;
            double kt = 0;                                  // k tiles summed over the output tiles
            for (int ti = 0; ti < nt; ti++) kt += (double)(ti + 1) * (nt - (ti < aa ? aa : ti));
            printf("ld 8192+%3d  k_lauum a=%2d w=%2d  %8.1f us  %5.1f TF/s\n", pad, aa, ww, ms * 1e3,
                   kt * 2.0 * TILE * TILE * TILE / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
