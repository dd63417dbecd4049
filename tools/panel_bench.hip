// panel_bench.hip -- cycles of ONE 16-pivot panel factor of the diagonal block (diagonal micro tile + the 112 rows
// below it + the identity rows that become the tile's inverse, three waves side by side), checked on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cugp_amd/csrc tools/panel_bench.hip -o tools/bin/panel_bench
#include "../cugp_amd/csrc/kernels.hip"

#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace cugp;

__global__ __launch_bounds__(256) void k_panel(const double* __restrict__ A, int ld, double* __restrict__ out,
                                               double* __restrict__ gd, unsigned long long* __restrict__ cyc)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[TILE + 4 * MT];
    double* rinv = sm + NLT * MTS;
    double* zz = red + TILE;
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    {
        const int r = t >> 4, c = t & 15;
        for (int bi = 0; bi < NMT; bi++)
            for (int bj = 0; bj <= bi; bj++)
                sm[mt_off(bi, bj) + r * (MT + 1) + c] = A[(size_t)(bi * MT + r) * ld + bj * MT + c];
        if (t < 2 * MT) zz[t] = t == MT - 1 ? 1.0 : 0.0;
    }
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    const int rows = (NMT - 1) * MT;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    PanelLanes pl;
    double pr[MT];
    if (wave <= 2) panel_load(sm, 0, rows, wave * 48, wave == 2, red + wave * 2 * MT, zz, pl, pr);
    lds_barrier();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (wave <= 2) panel_factor(pl, pr, wave == 0 ? gd : nullptr, TILE, rinv, red + wave * 2 * MT, (unsigned*)(zz + 2 * MT - 1), 3u);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    lds_barrier();
    if ((t & 63) == 0) { cyc[wave * 2] = t1 - t0; cyc[wave * 2 + 1] = t2 - t1; }
    {   // column block 0 back out: tile (bi,0), bi = 1..7 (the diagonal tile went to gd), its inverse and 1/L_ii
        const int r = t >> 4, c = t & 15;
        for (int bi = 1; bi < NMT; bi++) out[(size_t)(bi * MT + r) * MT + c] = sm[mt_off(bi, 0) + r * (MT + 1) + c];
        out[(size_t)TILE * MT + r * MT + c] = sm[mt_off(0, 0) + r * (MT + 1) + c];
        if (t < MT) out[t] = rinv[t];
    }
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int n = TILE;
    std::vector<double> h((size_t)n * n);
    srand(3);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double v = (rand() / (double)RAND_MAX - 0.5) * 0.5;
            if (i == j) v = 9.0 + v;
            h[(size_t)i * n + j] = h[(size_t)j * n + i] = v;
        }
    // host reference of the panel: L00 = chol(A00), X = A10 L00^-T
    std::vector<double> L(h);
    for (int c = 0; c < MT; c++) {
        double d = L[(size_t)c * n + c];
        for (int k = 0; k < c; k++) d -= L[(size_t)c * n + k] * L[(size_t)c * n + k];
        d = sqrt(d);
        L[(size_t)c * n + c] = d;
        for (int i = c + 1; i < n; i++) {
            double v = L[(size_t)i * n + c];
            for (int k = 0; k < c; k++) v -= L[(size_t)i * n + k] * L[(size_t)c * n + k];
            L[(size_t)i * n + c] = v / d;
        }
    }
    double *dA, *dout, *dgd; unsigned long long* dc;
    hipMalloc(&dA, h.size() * 8); hipMalloc(&dout, (n + MT) * MT * 8); hipMalloc(&dgd, n * n * 8); hipMalloc(&dc, 64 * 8);
    hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_panel, hipFuncAttributeMaxDynamicSharedMemorySize, POTF2_LDS);
    unsigned long long best[8];
    for (int i = 0; i < 8; i++) best[i] = ~0ull;
    std::vector<double> o((size_t)(n + MT) * MT), gd((size_t)n * n);
    for (int rep = 0; rep < 5; rep++) {
        hipMemset(dgd, 0, n * n * 8);
        hipLaunchKernelGGL(k_panel, dim3(1), dim3(256), POTF2_LDS, 0, dA, n, dout, dgd, dc);
        hipDeviceSynchronize();
        unsigned long long c[8];
        hipMemcpy(c, dc, sizeof c, hipMemcpyDeviceToHost);
        for (int i = 0; i < 8; i++) if (c[i] < best[i]) best[i] = c[i];
    }
    hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(gd.data(), dgd, gd.size() * 8, hipMemcpyDeviceToHost);
    double err = 0, errd = 0, errr = 0, erri = 0;
    for (int i = MT; i < n; i++)
        for (int c = 0; c < MT; c++) err = fmax(err, fabs(o[(size_t)i * MT + c] - L[(size_t)i * n + c]));
    for (int i = 0; i < MT; i++) {
        for (int c = 0; c <= i; c++) errd = fmax(errd, fabs(gd[(size_t)i * TILE + c] - L[(size_t)i * n + c]));
        errr = fmax(errr, fabs(o[i] - 1.0 / L[(size_t)i * n + i]));
    }
    for (int i = 0; i < MT; i++)                      // T = inverse tile from the identity rows: L00 T = I
        for (int j = 0; j < MT; j++) {
            double v = 0;
            for (int k = 0; k < MT; k++) v += (k <= i ? L[(size_t)i * n + k] : 0.0) * o[(size_t)TILE * MT + k * MT + j];
            erri = fmax(erri, fabs(v - (i == j ? 1.0 : 0.0)));
        }
    printf("load+barrier %llu / %llu / %llu cycles, 16 pivots %llu / %llu / %llu cycles (waves 0,1,2); max err rows %.2e "
           "diag %.2e 1/Lii %.2e |L00 T - I| %.2e\n", best[0], best[2], best[4], best[1], best[3], best[5], err, errd, errr, erri);
    return 0;
}
