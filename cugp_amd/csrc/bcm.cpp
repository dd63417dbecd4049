// bcm.cpp -- product-of-experts ("BCM") over the experts resident on one GPU.
//
// Mirrors class BCM (distributed_gp/BCM.h:2-27, BCM.cpp): K experts, each a full GP on its own rows,
// the objective is the plain sum of the experts' log-likelihoods / gradients (BCM.cpp:153-198) and
// the prediction is a product of experts with no prior-precision correction (BCM.cpp:45-62).
// The experts are padded to a common size and evaluated as a GROUP: one sequence of launches in which
// blockIdx.y selects the expert (group.h) -- 16 experts driven from 16 streams were bounded by the
// command processor's dispatch rate, not by the CUs.  When a group evaluation is not possible (experts
// with different hyper-parameters, profiling on) every expert is enqueued on its own stream before the
// first result is fetched.  Sums are taken in expert order on the host either way, so the result does
// not depend on completion order.
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/cugp.h"
#include "group.h"

struct cugp_bcm {
    cugp_group* group = nullptr;     // all experts in one set of launches (null: one expert, or shapes differ)
    std::vector<cugp_gp*> experts;
    std::vector<int> rows;
    int d = 0, device = 0;
    double hp[3] = {0, 0, 0};
};

extern "C" {

int cugp_bcm_create(int nexperts, const int* rows, int d, int device, cugp_bcm** out)
{
    if (!out || nexperts <= 0 || !rows || d <= 0) return CUGP_ERR_INVALID;
    cugp_bcm* b = new (std::nothrow) cugp_bcm;
    if (!b) return CUGP_ERR_NOMEM;
    b->d = d;
    b->device = device;
    // Common padded size (identity padding) so that the experts can share launches -- unless their row counts
    // differ by more than a tile or ~6 %: then padding the small ones would cost more than it gains.
    int nmax = 0, nmin = rows[0];
    for (int k = 0; k < nexperts; k++) {
        if (rows[k] <= 0) { delete b; return CUGP_ERR_INVALID; }
        nmax = rows[k] > nmax ? rows[k] : nmax;
        nmin = rows[k] < nmin ? rows[k] : nmin;
    }
    const int tmax = (nmax + 127) / 128, tmin = (nmin + 127) / 128;
    const int pad_to = (tmax - tmin <= (tmin / 16 > 1 ? tmin / 16 : 1)) ? nmax : 0;
    for (int k = 0; k < nexperts; k++) {
        cugp_gp* g = nullptr;
        int rc = cugp_create_padded(rows[k], d, device, pad_to, &g);
        if (rc) { cugp_bcm_destroy(b); return rc; }
        // several experts on one device already fill each other's idle time; the extra streams only cost launches
        if (nexperts > 1) cugp_set_overlap(g, 0);
        b->experts.push_back(g);
        b->rows.push_back(rows[k]);
    }
    if (nexperts > 1 && cugp_group_create(b->experts.data(), nexperts, &b->group) != CUGP_OK) b->group = nullptr;
    *out = b;
    return CUGP_OK;
}

// BCM.cpp:85-110 -- expert k gets rows [k*floor(N/K), ...), the last one also the remainder
int cugp_bcm_create_split(const double* X, const double* y, int N, int D, int K, int device, cugp_bcm** out)
{
    if (!X || !y || N <= 0 || D <= 0 || K <= 0 || K > N) return CUGP_ERR_INVALID;
    std::vector<int> rows(K), off(K);
    const int part = N / K;
    int start = 0;
    for (int k = 0; k < K; k++) {
        off[k] = start;
        rows[k] = (k == K - 1) ? (N - start) : part;
        start += part;
    }
    int rc = cugp_bcm_create(K, rows.data(), D, device, out);
    if (rc) return rc;
    for (int k = 0; k < K; k++) {
        rc = cugp_bcm_set_expert_data(*out, k, X + (size_t)off[k] * D, y + off[k]);
        if (rc) { cugp_bcm_destroy(*out); *out = nullptr; return rc; }
    }
    return CUGP_OK;
}

int cugp_bcm_destroy(cugp_bcm* b)
{
    if (!b) return CUGP_OK;
    cugp_group_destroy(b->group);
    for (cugp_gp* g : b->experts) cugp_destroy(g);
    delete b;
    return CUGP_OK;
}

int cugp_bcm_num_experts(const cugp_bcm* b, int* k)
{
    if (!b || !k) return CUGP_ERR_INVALID;
    *k = (int)b->experts.size();
    return CUGP_OK;
}

int cugp_bcm_expert(cugp_bcm* b, int k, cugp_gp** gp)
{
    if (!b || !gp || k < 0 || k >= (int)b->experts.size()) return CUGP_ERR_INVALID;
    *gp = b->experts[k];
    return CUGP_OK;
}

int cugp_bcm_set_expert_data(cugp_bcm* b, int k, const double* X, const double* y)
{
    if (!b || k < 0 || k >= (int)b->experts.size()) return CUGP_ERR_INVALID;
    return cugp_set_data(b->experts[k], X, y);
}

int cugp_bcm_set_loghyper(cugp_bcm* b, const double hp[3])
{
    if (!b || !hp) return CUGP_ERR_INVALID;
    for (int i = 0; i < 3; i++) b->hp[i] = hp[i];
    for (cugp_gp* g : b->experts) {
        int rc = cugp_set_loghyper(g, b->hp);
        if (rc) return rc;
    }
    return CUGP_OK;
}

int cugp_bcm_get_loghyper(const cugp_bcm* b, double hp[3])
{
    if (!b || !hp) return CUGP_ERR_INVALID;
    for (int i = 0; i < 3; i++) hp[i] = b->hp[i];
    return CUGP_OK;
}

// rows[k] = {LL_k, g_k[0..2]} for every expert of this device (what a multi-device BCM all-reduces)
int cugp_bcm_loglik_grad_rows(cugp_bcm* b, double* rows)
{
    if (!b || !rows) return CUGP_ERR_INVALID;
    const size_t K = b->experts.size();
    std::vector<double> lk(K), gk3(3 * K);
    bool grouped = false;
    if (b->group) {
        const int rc = cugp_group_eval(b->group, 1, lk.data(), gk3.data());
        if (rc == CUGP_OK) grouped = true;
        else if (rc != CUGP_ERR_INVALID) return rc;      // INVALID: not possible as a group right now
    }
    if (!grouped) {
        for (cugp_gp* e : b->experts) {                  // all experts in flight before the first fetch
            int rc = cugp_loglik_grad_enqueue(e, 1);
            if (rc) return rc;
        }
        for (size_t k = 0; k < K; k++) {
            int rc = cugp_loglik_grad_fetch(b->experts[k], &lk[k], &gk3[3 * k]);
            if (rc) return rc;
        }
    }
    for (size_t k = 0; k < K; k++) {
        rows[4 * k] = lk[k];
        for (int i = 0; i < 3; i++) rows[4 * k + 1 + i] = gk3[3 * k + i];
    }
    return CUGP_OK;
}

int cugp_bcm_loglik_grad(cugp_bcm* b, double* ll, double g[3], double* per_expert_ll)
{
    if (!b) return CUGP_ERR_INVALID;
    const size_t K = b->experts.size();
    std::vector<double> rows(4 * K);
    int rc = cugp_bcm_loglik_grad_rows(b, rows.data());
    if (rc) return rc;
    double sll = 0.0, sg[3] = {0, 0, 0};
    for (size_t k = 0; k < K; k++) {
        const double l = rows[4 * k];
        const double* gk = &rows[4 * k + 1];
        sll = sll + l;                                   // BCM.cpp:190-194
        for (int i = 0; i < 3; i++) sg[i] = (k == 0) ? gk[i] : sg[i] + gk[i];   // BCM.cpp:161-173
        if (per_expert_ll) per_expert_ll[k] = l;
    }
    if (ll) *ll = sll;
    if (g)
        for (int i = 0; i < 3; i++) g[i] = sg[i];
    return CUGP_OK;
}

int cugp_bcm_predict_partial(cugp_bcm* b, const double* Xt, int nt, double* sum_prec, double* sum_prec_mean)
{
    if (!b || !Xt || nt <= 0 || !sum_prec || !sum_prec_mean) return CUGP_ERR_INVALID;
    std::vector<double> m(nt), v(nt);
    for (int i = 0; i < nt; i++) sum_prec[i] = sum_prec_mean[i] = 0.0;
    for (cugp_gp* e : b->experts) {
        int rc = cugp_predict(e, Xt, nt, m.data(), v.data());
        if (rc) return rc;
        for (int i = 0; i < nt; i++) {                   // BCM.cpp:51-55
            const double inv = 1.0 / v[i];
            sum_prec[i] += inv;
            sum_prec_mean[i] += inv * m[i];
        }
    }
    return CUGP_OK;
}

int cugp_poe_finish(const double* sum_prec, const double* sum_prec_mean, int nt, double* mean, double* var)
{
    if (!sum_prec || !sum_prec_mean || !mean || !var || nt <= 0) return CUGP_ERR_INVALID;
    for (int i = 0; i < nt; i++) {                       // BCM.cpp:56-60
        const double tv = 1.0 / sum_prec[i];
        var[i] = tv;
        mean[i] = tv * sum_prec_mean[i];
    }
    return CUGP_OK;
}

int cugp_bcm_predict(cugp_bcm* b, const double* Xt, int nt, double* mean, double* var)
{
    if (!b || nt <= 0) return CUGP_ERR_INVALID;
    std::vector<double> sp(nt), spm(nt);
    int rc = cugp_bcm_predict_partial(b, Xt, nt, sp.data(), spm.data());
    if (rc) return rc;
    return cugp_poe_finish(sp.data(), spm.data(), nt, mean, var);
}

namespace {
void bcm_objective(void* ctx, const double th[3], double* f, double g[3])
{
    cugp_bcm* b = (cugp_bcm*)ctx;
    double ll = NAN;
    cugp_bcm_set_loghyper(b, th);
    if (cugp_bcm_loglik_grad(b, &ll, g, nullptr) != CUGP_OK) { ll = NAN; g[0] = g[1] = g[2] = NAN; }
    *f = -1.0 * ll;
}
}  // namespace

int cugp_bcm_cg_solve(cugp_bcm* b, int budget, double* trace, int trace_cap, int* nevals)
{
    if (!b) return CUGP_ERR_INVALID;
    double th[3] = {b->hp[0], b->hp[1], b->hp[2]};
    int rc = cugp_cg_minimize(bcm_objective, b, th, budget, trace, trace_cap, nevals);
    if (rc) return rc;
    return cugp_bcm_set_loghyper(b, th);
}

}  // extern "C"
