// slot_probe.hip -- where do the workgroups of a persistent tile-product launch run, and for how long?
// 512 workgroups x 8 tiles (K=128..1024); per workgroup: start, end (100 MHz counter), XCC / SE / CU ids.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cugp_amd/csrc tools/slot_probe.hip -o tools/bin/slot_probe
#include "../cugp_amd/csrc/kernels.hip"

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

using namespace cugp;

struct Rec { unsigned long long t0, t1; unsigned hwid, xcc; };

__global__ __launch_bounds__(256, 2) void k_probe(const double* __restrict__ A, const double* __restrict__ B,
                                                  double* __restrict__ C, int n, int k, int mt, int ntiles, Rec* rec)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ti = t % mt, tj = t / mt;
        d4 acc[4][4];
        acc_zero(acc);
        tile_nt<false>(A + (size_t)ti * TILE * k, k, B + (size_t)tj * TILE * k, k, 0, k, acc, smem);
        tile_store(C + (size_t)ti * TILE * n + tj * TILE, n, acc, 1.0);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[blockIdx.x] = Rec{t0, __builtin_amdgcn_s_memrealtime(), hw, xcc};
    }
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    printf("device: %s, %d CUs, clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    const int m = 8192, n = 8192, kmax = 1024;
    double *A, *B, *C;
    Rec* rec;
    (void)hipMalloc(&A, (size_t)m * kmax * 8); (void)hipMalloc(&B, (size_t)n * kmax * 8); (void)hipMalloc(&C, (size_t)m * n * 8);
    (void)hipMalloc(&rec, 4096 * sizeof(Rec));
    (void)hipMemset(A, 0, (size_t)m * kmax * 8); (void)hipMemset(B, 0, (size_t)n * kmax * 8);
    (void)hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    for (int k : {128, 1024})
        for (int slots : {512, 256, 4096}) {
            for (int rep = 0; rep < 2; rep++)
                hipLaunchKernelGGL(k_probe, dim3(slots), dim3(256), GEMM_LDS, 0, A, B, C, n, k, m / 128, 4096, rec);
            (void)hipDeviceSynchronize();
            std::vector<Rec> h(slots);
            (void)hipMemcpy(h.data(), rec, slots * sizeof(Rec), hipMemcpyDeviceToHost);
            unsigned long long tmin = ~0ull, tmax = 0;
            for (auto& r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
            std::vector<double> dur, start;
            std::map<unsigned, int> per_cu;
            for (auto& r : h) {
                dur.push_back((r.t1 - r.t0) / 100.0);
                start.push_back((r.t0 - tmin) / 100.0);
                // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13] on gfx9
                per_cu[(r.xcc & 15) << 16 | ((r.hwid >> 13) & 7) << 8 | ((r.hwid >> 8) & 15)]++;
            }
            std::sort(dur.begin(), dur.end()); std::sort(start.begin(), start.end());
            std::map<int, int> occ;
            for (auto& kv : per_cu) occ[kv.second]++;
            printf("k=%4d grid=%4d: span %.1f us; workgroup duration min %.1f med %.1f max %.1f; start med %.1f p90 %.1f max %.1f; "
                   "%zu distinct CUs, workgroups per CU:", k, slots, (tmax - tmin) / 100.0, dur.front(), dur[dur.size() / 2],
                   dur.back(), start[start.size() / 2], start[start.size() * 9 / 10], start.back(), per_cu.size());
            for (auto& kv : occ) printf(" %dx%d", kv.second, kv.first);
            printf("\n");
        }
    return 0;
}
