// BCM.cpp -- class BCM over the C-ABI (see BCM.h).
#include "BCM.h"
#include "bcm_solve.h"

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

struct BCM::Shared {
    cugp_bcm *handle;
    double log_hyper_bcm[3];
    int refs;
    bool have_eval;               // ll / g below belong to log_hyper_bcm (one evaluation serves the pair of calls
    double ll, g[3];              // the reference's cg_solve makes at every probe, distributed_ver1.cpp:87-88)
    void eval()
    {
        if (have_eval) return;
        must(cugp_bcm_loglik_grad(handle, &ll, g, nullptr), "cugp_bcm_loglik_grad");
        have_eval = true;
    }
};

void BCM::init(double **inp, double *out, int N, int D, int K, const int *devices, int ndev)
{
    s = new Shared();
    s->handle = nullptr;
    s->log_hyper_bcm[0] = s->log_hyper_bcm[1] = s->log_hyper_bcm[2] = 0.0;
    s->refs = 1;
    s->have_eval = false;
    num_experts = K;
    dim = D;
    std::vector<double> flat((size_t)N * D);
    for (int i = 0; i < N; i++) memcpy(&flat[(size_t)i * D], inp[i], D * sizeof(double));
    const int rc = cugp_bcm_create_split_multi(flat.data(), out, N, D, K, ndev, devices, &s->handle);
    if (rc != CUGP_OK) {
        delete s;
        s = nullptr;
        must(rc, "BCM");
    }
}

BCM::BCM(double **inp, double *out, int N, int D, int K)
{
    std::vector<int> d = all_devices(K);
    init(inp, out, N, D, K, d.data(), (int)d.size());
}

BCM::BCM(double **inp, double *out, int N, int D, int K, int device) { init(inp, out, N, D, K, &device, 1); }

BCM::BCM(double **inp, double *out, int N, int D, int K, const int *devices, int ndev) { init(inp, out, N, D, K, devices, ndev); }

// copies share the device-side model (the reference passes a BCM by value, distributed_ver1.cpp:13,285)
BCM::BCM(const BCM &o) : s(o.s), num_experts(o.num_experts), dim(o.dim)
{
    if (s) s->refs++;
}

BCM &BCM::operator=(const BCM &o)
{
    if (o.s) o.s->refs++;
    if (s && --s->refs == 0) {
        if (s->handle) cugp_bcm_destroy(s->handle);
        delete s;
    }
    s = o.s;
    num_experts = o.num_experts;
    dim = o.dim;
    return *this;
}

BCM::~BCM()
{
    if (s && --s->refs == 0) {
        if (s->handle) cugp_bcm_destroy(s->handle);
        delete s;
    }
}

cugp_bcm *BCM::native() { return s ? s->handle : nullptr; }

void BCM::set_BCM_log_hyperparam(double *hp)
{
    for (int i = 0; i < 3; i++) s->log_hyper_bcm[i] = hp[i];
    s->have_eval = false;
    must(cugp_bcm_set_loghyper(s->handle, s->log_hyper_bcm), "cugp_bcm_set_loghyper");
}

// BCM.cpp:132-151: the reference returns the SUM over experts of their (identical) vectors, i.e. K * hp
void BCM::get_BCM_log_hyperparam(double *hp)
{
    for (int i = 0; i < 3; i++) {
        double sum = s->log_hyper_bcm[i];
        for (int k = 1; k < num_experts; k++) sum += s->log_hyper_bcm[i];
        hp[i] = sum;
    }
}

void BCM::get_loghyperparam(double *hp)
{
    for (int i = 0; i < 3; i++) hp[i] = s->log_hyper_bcm[i];
}

void BCM::get_BCM_gradient_hyper(double *g)
{
    s->eval();
    for (int i = 0; i < 3; i++) g[i] = s->g[i];
}

double BCM::get_BCM_loglikelihood()
{
    s->eval();
    return s->ll;
}

void BCM::compute_BCM_test_means_and_var(double **Xtest, double *tmeanvec, double *tvarvec, int size)
{
    std::vector<double> xt((size_t)size * dim);
    for (int i = 0; i < size; i++) memcpy(&xt[(size_t)i * dim], Xtest[i], dim * sizeof(double));
    must(cugp_bcm_predict(s->handle, xt.data(), size, tmeanvec, tvarvec), "cugp_bcm_predict");
}

double BCM::get_BCM_negative_log_predprob(double *actual, double *predmean, double *predvar, int TS)
{
    double out = 0;
    must(cugp_nlpp(actual, predmean, predvar, TS, &out), "cugp_nlpp");
    return out;
}

void cugp_cg_solve(BCM &pobj)
{
    int nev = 0;
    must(cugp_bcm_cg_solve(pobj.native(), 100, nullptr, 0, &nev), "cugp_bcm_cg_solve");
    double hp[3];
    must(cugp_bcm_get_loghyper(pobj.native(), hp), "cugp_bcm_get_loghyper");
    pobj.set_BCM_log_hyperparam(hp);
    printf("\n\n PLEASE-SEE 3 : %lf, %lf, %lf\n\n", hp[0], hp[1], hp[2]);
}
