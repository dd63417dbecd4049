// chain_bench.hip -- times the latency-critical kernels of the Cholesky chain alone (potf2, strip TRSM,
// next-diagonal update) and prints in-kernel cycle stamps of potf2.  Diagnostic build: -DCUGP_STAMPS.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCUGP_STAMPS -I cugp_amd/csrc tools/chain_bench.hip -o tools/bin/chain_bench
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"
#include "../cugp_amd/csrc/kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace cugp;

static float timeit(void (*fn)(void*), void* ctx, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    fn(ctx);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) fn(ctx);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

struct Ctx { double *A, *A0, *d16, *d64, *ld; int n, nt; };

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int nt = 16, n = nt * TILE;
    std::vector<double> h((size_t)n * n);
    srand(1);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double v = (rand() / (double)RAND_MAX - 0.5) * 0.01;
            if (i == j) v = 4.0 + v;
            h[(size_t)i * n + j] = h[(size_t)j * n + i] = v;
        }
    Ctx c; c.n = n; c.nt = nt;
    hipMalloc(&c.A, h.size() * 8); hipMalloc(&c.A0, h.size() * 8);
    hipMalloc(&c.d16, (size_t)nt * 8 * 256 * 8); hipMalloc(&c.d64, (size_t)nt * 8192 * 8); hipMalloc(&c.ld, nt * 8);
    hipMemcpy(c.A0, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(c.A, c.A0, h.size() * 8, hipMemcpyDeviceToDevice);

    auto potf2 = [](void* p) { Ctx* c = (Ctx*)p; hipMemcpyAsync(c->A, c->A0, (size_t)TILE * c->n * 8, hipMemcpyDeviceToDevice, 0); launch_potf2(c->A, c->n, 0, c->d16, c->d64, c->ld, 0); };
    auto copy_only = [](void* p) { Ctx* c = (Ctx*)p; hipMemcpyAsync(c->A, c->A0, (size_t)TILE * c->n * 8, hipMemcpyDeviceToDevice, 0); };
    float t_copy = timeit(copy_only, &c, 50);
    float t_potf2 = timeit(potf2, &c, 50);
    printf("potf2 (128x128 diagonal block): %.2f us (incl. %.2f us restore copy)\n", t_potf2, t_copy);
    unsigned long long st[128];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st);
    printf("  stamps (cycles): load %llu\n", st[1] - st[0]);
    for (int q = 0; q < 8; q++)
        printf("  q=%d: panel factor beside the older panels' update of column q+1 %llu   panel q into column q+1 %llu\n", q,
               st[2 + 2 * q] - (q ? st[1 + 2 * q] : st[1]), st[3 + 2 * q] - st[2 + 2 * q]);
    for (int q = 0; q < 8; q++)
        printf("  q=%d: waves reach the barrier after %lld %lld %lld %lld cycles\n", q, (long long)(st[64 + 4 * q] - (q ? st[1 + 2 * q] : st[1])),
               (long long)(st[65 + 4 * q] - (q ? st[1 + 2 * q] : st[1])), (long long)(st[66 + 4 * q] - (q ? st[1 + 2 * q] : st[1])),
               (long long)(st[67 + 4 * q] - (q ? st[1 + 2 * q] : st[1])));
    printf("  tail: second half of the pair (6,7) + logdet %llu | second half of 64-block 1 %llu | its store %llu | total %llu cycles\n",
           st[31] - st[30], st[32] - st[31], st[33] - st[32], st[33] - st[0]);

    auto trsm2 = [](void* p) { Ctx* c = (Ctx*)p; launch_trsm_inv64(c->A, c->d64, c->n, 0, c->nt, 0); };
    printf("trsm_inv64 (%d strips): %.2f us\n", (nt - 1) * 8, timeit(trsm2, &c, 50));
    unsigned* tk; hipMalloc(&tk, nt * 4);
    struct C2 { Ctx* c; unsigned* tk; int kb; } c2{&c, tk, 0}, c3{&c, tk, 12};
    auto step = [](void* p) { C2* q = (C2*)p; hipMemsetAsync(q->tk, 0, q->c->nt * 4, 0);
                              launch_syrk_step(q->c->A, q->c->n, q->kb, q->c->nt, q->c->d16, q->c->d64, q->c->ld, q->tk, 0); };
    printf("syrk_step kb=0 (119 tiles + next diagonal block): %.2f us\n", timeit(step, &c2, 20));
    printf("syrk_step kb=12 (5 tiles + next diagonal block): %.2f us\n", timeit(step, &c3, 20));
    return 0;
}
