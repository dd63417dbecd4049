// rccl_driver.cpp -- the reference's multi-process BCM driver (cuda_scalingdist/main.cpp:70-233 master/worker loop,
// cg_solver.cpp:72-213 gathers of log-likelihood and gradient, :245-279 hyper-parameter broadcast), as a C++ host
// would write it on libcugp + RCCL: one process per GPU, expert k on rank k mod W (cg_solver.cpp:93).  The objective
// goes through the library's own exchange (cugp_comm_*, cugp_bcm_loglik_grad_allgather: evaluation, ncclAllGather of
// every rank's {LL_k, g_k} rows and the copy to the host as ONE sequence on the evaluation's stream), summed in expert
// order on every rank -- no TCP, no host wait before the collective.  Prediction: per-rank product-of-experts partial
// sums (distributed_gp/BCM.cpp:45-62), one ncclAllReduce of 2 x nt doubles by the driver itself, cugp_poe_finish.
//
//   rccl_driver <id file> <rank> <world> <data file> <K> <rows per expert> <d> <nt> [device]
//     id file  : rank 0 writes the ncclUniqueId there, the others wait for it (any shared path; no MPI needed)
//     data file: K*rows*d doubles X, K*rows doubles y, nt*d doubles Xt, 3 doubles hp (raw, row-major)
//   prints one JSON object (rank 0): the all-reduced results and, for comparison, what the library's own
//   single-process sums give on the same experts (world == 1: must be bit-identical).
//
//   hipcc -O2 -std=c++17 tests/cpp/rccl_driver.cpp -Iinclude -Lcugp_amd/lib -lcugp -lrccl -Wl,-rpath,cugp_amd/lib
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cugp.h"

#define CK(call)                                                                      \
    do {                                                                              \
        const int rc_ = (call);                                                       \
        if (rc_ != 0) { fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, cugp_last_error()); return 2; } \
    } while (0)
#define HK(call)                                                                      \
    do {                                                                              \
        const hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } \
    } while (0)
#define NK(call)                                                                      \
    do {                                                                              \
        const ncclResult_t r_ = (call);                                               \
        if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); return 4; } \
    } while (0)

static void print_vec(const char* name, const double* v, int n, bool comma = true)
{
    printf("\"%s\": [", name);
    for (int i = 0; i < n; i++) printf("%s%.17g", i ? ", " : "", v[i]);
    printf("]%s\n", comma ? "," : "");
}

int main(int argc, char** argv)
{
    if (argc < 9) { fprintf(stderr, "usage: see the header of rccl_driver.cpp\n"); return 1; }
    const char* idfile = argv[1];
    const int rank = atoi(argv[2]), world = atoi(argv[3]);
    const char* datafile = argv[4];
    const int K = atoi(argv[5]), rows = atoi(argv[6]), d = atoi(argv[7]), nt = atoi(argv[8]);
    const int device = argc > 9 ? atoi(argv[9]) : rank;

    std::vector<double> X((size_t)K * rows * d), y((size_t)K * rows), Xt((size_t)nt * d);
    double hp[3];
    FILE* f = fopen(datafile, "rb");
    if (!f || fread(X.data(), 8, X.size(), f) != X.size() || fread(y.data(), 8, y.size(), f) != y.size() ||
        fread(Xt.data(), 8, Xt.size(), f) != Xt.size() || fread(hp, 8, 3, f) != 3) { fprintf(stderr, "bad data file\n"); return 1; }
    fclose(f);

    // ---- communicator: rank 0 publishes the id through a file (stands in for the reference's TCP hand-shake) ----
    HK(hipSetDevice(device));
    ncclUniqueId id;                                   // the driver's own communicator (prediction) ...
    unsigned char lid[128];                            // ... and the id of the library's (objective)
    if (rank == 0) {
        NK(ncclGetUniqueId(&id));
        CK(cugp_comm_unique_id(lid, (int)sizeof lid));
        std::string tmp = std::string(idfile) + ".tmp";
        FILE* g = fopen(tmp.c_str(), "wb");
        if (!g || fwrite(&id, sizeof id, 1, g) != 1 || fwrite(lid, sizeof lid, 1, g) != 1) return 1;
        fclose(g);
        rename(tmp.c_str(), idfile);
    } else {
        FILE* g = nullptr;
        for (int tries = 0; tries < 600 && !(g = fopen(idfile, "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!g || fread(&id, sizeof id, 1, g) != 1 || fread(lid, sizeof lid, 1, g) != 1) { fprintf(stderr, "no id file\n"); return 1; }
        fclose(g);
    }
    ncclComm_t comm;
    NK(ncclCommInitRank(&comm, world, id, rank));
    cugp_comm* lc = nullptr;
    CK(cugp_comm_create(lid, (int)sizeof lid, rank, world, device, &lc));
    hipStream_t cs;
    HK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));

    // ---- this rank's experts: k = rank, rank + W, ... (cg_solver.cpp:93) ----
    std::vector<int> mine;
    for (int k = rank; k < K; k += world) mine.push_back(k);
    const int nl = (int)mine.size();
    cugp_bcm* b = nullptr;
    if (nl > 0) {
        std::vector<int> r(nl, rows);
        CK(cugp_bcm_create(nl, r.data(), d, device, &b));
        for (int i = 0; i < nl; i++)
            CK(cugp_bcm_set_expert_data(b, i, X.data() + (size_t)mine[i] * rows * d, y.data() + (size_t)mine[i] * rows));
        CK(cugp_bcm_set_loghyper(b, hp));
    }

    // ---- objective: evaluation, all-gather and copy to the host are one stream sequence inside the library ----
    const int per = (K + world - 1) / world;                                // row slots per rank
    std::vector<double> gathered((size_t)world * per * 4), hrows((size_t)K * 4);
    CK(cugp_bcm_loglik_grad_allgather(b, lc, per, gathered.data()));
    for (int k = 0; k < K; k++)                                              // expert k = rank (k mod W)'s (k / W)-th
        memcpy(&hrows[4 * (size_t)k], &gathered[4 * ((size_t)(k % world) * per + k / world)], 4 * sizeof(double));
    double ll = 0.0, g[3] = {0, 0, 0};
    for (int k = 0; k < K; k++) {                                            // expert order, as BCM.cpp:161-197
        ll = ll + hrows[4 * k];
        for (int j = 0; j < 3; j++) g[j] = k == 0 ? hrows[4 * k + 1 + j] : g[j] + hrows[4 * k + 1 + j];
    }

    // ---- prediction: product of experts over all ranks ----
    std::vector<double> sp(nt, 0.0), spm(nt, 0.0), mean(nt), var(nt);
    if (nl > 0) CK(cugp_bcm_predict_partial(b, Xt.data(), nt, sp.data(), spm.data()));
    double* dpp = nullptr;
    HK(hipMalloc((void**)&dpp, (size_t)2 * nt * sizeof(double)));
    HK(hipMemcpyAsync(dpp, sp.data(), nt * sizeof(double), hipMemcpyHostToDevice, cs));
    HK(hipMemcpyAsync(dpp + nt, spm.data(), nt * sizeof(double), hipMemcpyHostToDevice, cs));
    NK(ncclAllReduce(dpp, dpp, (size_t)2 * nt, ncclDouble, ncclSum, comm, cs));
    HK(hipMemcpyAsync(sp.data(), dpp, nt * sizeof(double), hipMemcpyDeviceToHost, cs));
    HK(hipMemcpyAsync(spm.data(), dpp + nt, nt * sizeof(double), hipMemcpyDeviceToHost, cs));
    HK(hipStreamSynchronize(cs));
    CK(cugp_poe_finish(sp.data(), spm.data(), nt, mean.data(), var.data()));

    if (rank == 0) {
        printf("{\"world\": %d, \"K\": %d,\n\"ll\": %.17g,\n", world, K, ll);
        print_vec("grad", g, 3);
        std::vector<double> per(K);
        for (int k = 0; k < K; k++) per[k] = hrows[4 * k];
        print_vec("per_expert_ll", per.data(), K);
        print_vec("pred_mean", mean.data(), nt);
        print_vec("pred_var", var.data(), nt);
        if (world == 1) {                                                    // the library's own single-process sums
            double ll1, g1[3];
            std::vector<double> m1(nt), v1(nt);
            CK(cugp_bcm_loglik_grad(b, &ll1, g1, nullptr));
            CK(cugp_bcm_predict(b, Xt.data(), nt, m1.data(), v1.data()));
            printf("\"direct_ll\": %.17g,\n", ll1);
            print_vec("direct_grad", g1, 3);
            print_vec("direct_pred_mean", m1.data(), nt);
            print_vec("direct_pred_var", v1.data(), nt);
        }
        printf("\"ok\": 1}\n");
    }
    if (b) cugp_bcm_destroy(b);
    cugp_comm_destroy(lc);
    (void)hipFree(dpp);
    ncclCommDestroy(comm);
    return 0;
}
