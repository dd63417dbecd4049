// waitvalue_probe.hip -- what a hipStreamWaitValue32 gate costs on MI355X (ROCm 7.2): a resident kernel on stream C
// writes a flag, a launch on stream M is held back by hipStreamWaitValue32 on that flag, and answers through a second
// flag the resident kernel polls.  Printed per kind of flag memory: flag write -> first instruction of the gated
// kernel (s_memrealtime, 100 MHz, one clock for the whole chip) and the whole round trip; plus the cost of an already
// satisfied gate in front of an empty kernel.  (Round 5: can the factorisation's chain release the off-chain launches
// of a step without a kernel boundary of its own?  DESIGN section 8.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/waitvalue_probe.hip -o tools/bin/waitvalue_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); }         \
    } while (0)

__global__ void k_ping(unsigned* flagA, unsigned* flagB, unsigned long long* st, int n, unsigned spin_cap)
{
    if (threadIdx.x != 0) return;
    bool dead = false;
    for (int i = 1; i <= n && !dead; i++) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_store(flagA, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned s = 0;
        while (__hip_atomic_load(flagB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)i) {
            __builtin_amdgcn_s_sleep(4);
            if (++s > spin_cap) { dead = true; break; }
        }
        st[2 * i] = t0;
        st[2 * i + 1] = dead ? 0ull : __builtin_amdgcn_s_memrealtime();
    }
    // whatever happened: every gate still waiting on this flag opens
    __hip_atomic_store(flagA, 0x7fffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_pong(unsigned* flagB, unsigned long long* stB, int i)
{
    if (threadIdx.x != 0) return;
    stB[i] = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(flagB, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_empty() {}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (can != 1) return 0;
    hipStream_t C, M;
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
    const int n = 40;
    unsigned long long *st, *stB;
    CK(hipMalloc(&st, (2 * n + 4) * 8));
    CK(hipMalloc(&stB, (n + 2) * 8));
    const char* names[] = {"hipMalloc", "hipExtMallocWithFlags(SignalMemory)", "hipExtMallocWithFlags(Uncached)",
                           "hipExtMallocWithFlags(Finegrained)", "hipHostMalloc(coherent)"};
    for (int kind = 0; kind < 5; kind++) {
        unsigned *fa = nullptr, *fb = nullptr;
        hipError_t e = hipSuccess;
        if (kind == 0) { e = hipMalloc(&fa, 64); if (e == hipSuccess) e = hipMalloc(&fb, 64); }
        else if (kind < 4) {
            const unsigned fl = kind == 1 ? hipMallocSignalMemory : (kind == 2 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
            e = hipExtMallocWithFlags((void**)&fa, kind == 1 ? 8 : 64, fl);
            if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&fb, kind == 1 ? 8 : 64, fl);
        } else {
            e = hipHostMalloc(&fa, 64, hipHostMallocCoherent);
            if (e == hipSuccess) e = hipHostMalloc(&fb, 64, hipHostMallocCoherent);
        }
        if (e != hipSuccess) { printf("%-40s allocation: %s\n", names[kind], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipMemset(fa, 0, 8)); CK(hipMemset(fb, 0, 8));
        CK(hipMemset(st, 0, (2 * n + 4) * 8)); CK(hipMemset(stB, 0, (n + 2) * 8));
        CK(hipDeviceSynchronize());
        // the gated launches first (they only hold their own stream), then the resident kernel
        bool ok = true;
        int queued = 0;
        for (int i = 1; i <= n && ok; i++) {
            e = hipStreamWaitValue32(M, fa, (unsigned)i, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { printf("%-40s hipStreamWaitValue32: %s\n", names[kind], hipGetErrorString(e)); (void)hipGetLastError(); ok = false; break; }
            hipLaunchKernelGGL(k_pong, dim3(1), dim3(64), 0, M, fb, stB, i);
            queued = i;
        }
        // spin cap ~ 0.25 s per round trip at most
        hipLaunchKernelGGL(k_ping, dim3(1), dim3(64), 0, C, fa, fb, st, queued, 400000u);
        CK(hipStreamSynchronize(C));
        CK(hipMemsetD32Async((hipDeviceptr_t)fa, 0x7fffffff, 1, C));   // second way out for gates the kernel's store did not open
        CK(hipStreamSynchronize(C));
        CK(hipStreamSynchronize(M));
        if (!ok && queued == 0) continue;
        std::vector<unsigned long long> h(2 * n + 4), hb(n + 2);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), stB, hb.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> rel, rt;
        int dead = 0;
        for (int i = 2; i <= queued; i++) {                  // (the first round trip warms everything up)
            if (h[2 * i + 1] == 0) { dead++; continue; }
            rel.push_back((double)(hb[i] - h[2 * i]) * 0.01);
            rt.push_back((double)(h[2 * i + 1] - h[2 * i]) * 0.01);
        }
        if (rel.empty()) { printf("%-40s no round trip completed (%d timed out)\n", names[kind], dead); continue; }
        std::sort(rel.begin(), rel.end()); std::sort(rt.begin(), rt.end());
        printf("%-40s flag write -> gated kernel's first instruction: min %.2f  median %.2f  max %.2f us;  round trip: min %.2f  median %.2f  max %.2f us  (%zu samples, %d timed out)\n",
               names[kind], rel.front(), rel[rel.size() / 2], rel.back(), rt.front(), rt[rt.size() / 2], rt.back(), rel.size(), dead);
        // an already satisfied gate in front of an empty kernel, against the empty kernel alone
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int pass = 0; pass < 2; pass++) {
            const int reps = 200;
            for (int w = 0; w < 2; w++) {
                if (w == 1) CK(hipEventRecord(a, M));
                for (int r = 0; r < reps; r++) {
                    if (pass == 1) CK(hipStreamWaitValue32(M, fa, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, M);
                }
            }
            CK(hipEventRecord(b, M));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("    %s: %.2f us per launch\n", pass ? "satisfied gate + empty kernel" : "empty kernel alone            ", ms * 1e3 / reps);
        }
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
        if (kind < 4) { CK(hipFree(fa)); CK(hipFree(fb)); } else { CK(hipHostFree(fa)); CK(hipHostFree(fb)); }
    }
    return 0;
}
