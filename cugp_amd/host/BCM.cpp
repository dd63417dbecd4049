// BCM.cpp -- class BCM over the C-ABI (see BCM.h).
#include "BCM.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cugp.h"

namespace {
void must(int rc, const char *what)
{
    if (rc != CUGP_OK) throw std::runtime_error(std::string(what) + ": " + cugp_last_error());
}
}  // namespace

namespace {
std::vector<int> all_devices(int K)
{
    int cnt = 0;
    must(cugp_device_count(&cnt), "cugp_device_count");
    if (cnt > K) cnt = K;
    std::vector<int> d(cnt > 0 ? cnt : 1, 0);
    for (int i = 0; i < (int)d.size(); i++) d[i] = i;
    return d;
}
}  // namespace

BCM::BCM(double **inp, double *out, int N, int D, int K)
    : BCM(inp, out, N, D, K, all_devices(K).data(), (int)all_devices(K).size()) {}

BCM::BCM(double **inp, double *out, int N, int D, int K, int device) : BCM(inp, out, N, D, K, &device, 1) {}

BCM::BCM(double **inp, double *out, int N, int D, int K, const int *devices, int ndev)
    : handle(nullptr), num_experts(K), dim(D), log_hyper_bcm{0, 0, 0}
{
    std::vector<double> flat((size_t)N * D);
    for (int i = 0; i < N; i++) memcpy(&flat[(size_t)i * D], inp[i], D * sizeof(double));
    must(cugp_bcm_create_split_multi(flat.data(), out, N, D, K, ndev, devices, &handle), "BCM");
}

BCM::~BCM()
{
    if (handle) cugp_bcm_destroy(handle);
}

void BCM::set_BCM_log_hyperparam(double *hp)
{
    for (int i = 0; i < 3; i++) log_hyper_bcm[i] = hp[i];
    must(cugp_bcm_set_loghyper(handle, log_hyper_bcm), "cugp_bcm_set_loghyper");
}

// BCM.cpp:132-151: the reference returns the SUM over experts of their (identical) vectors, i.e. K * hp
void BCM::get_BCM_log_hyperparam(double *hp)
{
    for (int i = 0; i < 3; i++) {
        double s = log_hyper_bcm[i];
        for (int k = 1; k < num_experts; k++) s += log_hyper_bcm[i];
        hp[i] = s;
    }
}

void BCM::get_loghyperparam(double *hp)
{
    for (int i = 0; i < 3; i++) hp[i] = log_hyper_bcm[i];
}

void BCM::get_BCM_gradient_hyper(double *g) { must(cugp_bcm_loglik_grad(handle, nullptr, g, nullptr), "cugp_bcm_loglik_grad"); }

double BCM::get_BCM_loglikelihood()
{
    double ll = 0;
    must(cugp_bcm_loglik_grad(handle, &ll, nullptr, nullptr), "cugp_bcm_loglik_grad");
    return ll;
}

void BCM::compute_BCM_test_means_and_var(double **Xtest, double *tmeanvec, double *tvarvec, int size)
{
    std::vector<double> xt((size_t)size * dim);
    for (int i = 0; i < size; i++) memcpy(&xt[(size_t)i * dim], Xtest[i], dim * sizeof(double));
    must(cugp_bcm_predict(handle, xt.data(), size, tmeanvec, tvarvec), "cugp_bcm_predict");
}

double BCM::get_BCM_negative_log_predprob(double *actual, double *predmean, double *predvar, int TS)
{
    double out = 0;
    must(cugp_nlpp(actual, predmean, predvar, TS, &out), "cugp_nlpp");
    return out;
}

void cg_solve(BCM &pobj)
{
    int nev = 0;
    must(cugp_bcm_cg_solve(pobj.native(), 100, nullptr, 0, &nev), "cugp_bcm_cg_solve");
    double hp[3];
    must(cugp_bcm_get_loghyper(pobj.native(), hp), "cugp_bcm_get_loghyper");
    pobj.set_BCM_log_hyperparam(hp);
    printf("\n\n PLEASE-SEE 3 : %lf, %lf, %lf\n\n", hp[0], hp[1], hp[2]);
}
