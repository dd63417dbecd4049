// ref_bcm_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" driver (our code) around the reference's OWN distributed_gp
// sources compiled where they lie (oracle/Makefile): distributed_gp/covkernel.cpp
// + BCM.cpp + distributed_ver1.cpp (for cg_solve(BCM)) + common/matrixops.cpp.
// Produces oracle/_ref/libref_bcm.so -- prediction and product-of-experts
// goldens come from here (the serial copy of covkernel.cpp dumps whole vectors
// to stdout in its prediction routine; the arithmetic is identical).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>

#include "covkernel.h"   // reference: distributed_gp/covkernel.h (via -I)
#include "BCM.h"         // reference: distributed_gp/BCM.h

void cg_solve(BCM pobj); // reference: distributed_gp/distributed_ver1.cpp:13

namespace {

struct StdoutTo {
    int saved;
    explicit StdoutTo(const char *path) {
        fflush(stdout);
        saved = dup(1);
        int fd = open(path ? path : "/dev/null", O_WRONLY | O_CREAT | O_TRUNC, 0644);
        dup2(fd, 1);
        close(fd);
    }
    ~StdoutTo() {
        fflush(stdout);
        dup2(saved, 1);
        close(saved);
    }
};

double **rows_copy(const double *flat, int n, int m) {
    double **r = new double *[n];
    for (int i = 0; i < n; i++) {
        r[i] = new double[m];
        memcpy(r[i], flat + (size_t)i * m, (size_t)m * sizeof(double));
    }
    return r;
}

struct BcmHandle {
    BCM *bcm;
    double **X;
    double *y;
    int N, D, K;
};

}  // namespace

extern "C" {

void *ref_bcm_create(const double *X, const double *y, int N, int D, int K) {
    BcmHandle *h = new BcmHandle;
    h->X = rows_copy(X, N, D);
    h->y = new double[N];
    memcpy(h->y, y, (size_t)N * sizeof(double));
    h->N = N; h->D = D; h->K = K;
    h->bcm = new BCM(h->X, h->y, N, D, K);
    return h;
}

void ref_bcm_set_loghyper(void *vh, const double *hp) {
    double t[3] = { hp[0], hp[1], hp[2] };
    ((BcmHandle *)vh)->bcm->set_BCM_log_hyperparam(t);
}

void ref_bcm_get_loghyper(void *vh, double *hp) {
    ((BcmHandle *)vh)->bcm->get_loghyperparam(hp);
}

// stdout (per-expert "LL of Expert k" lines) captured in logpath when given
double ref_bcm_loglik(void *vh, const char *logpath) {
    StdoutTo q(logpath);
    return ((BcmHandle *)vh)->bcm->get_BCM_loglikelihood();
}

void ref_bcm_grad(void *vh, double *g) {
    StdoutTo q(NULL);
    ((BcmHandle *)vh)->bcm->get_BCM_gradient_hyper(g);
}

void ref_bcm_predict(void *vh, const double *Xt, int nt, double *mean, double *var) {
    BcmHandle *h = (BcmHandle *)vh;
    StdoutTo q(NULL);
    double **Xtr = rows_copy(Xt, nt, h->D);
    h->bcm->compute_BCM_test_means_and_var(Xtr, mean, var, nt);
}

double ref_bcm_nlpp(void *vh, const double *actual, const double *mean, const double *var, int nt) {
    return ((BcmHandle *)vh)->bcm->get_BCM_negative_log_predprob(
        const_cast<double *>(actual), const_cast<double *>(mean), const_cast<double *>(var), nt);
}

void ref_bcm_cg_solve(void *vh, const char *logpath) {
    StdoutTo q(logpath);
    cg_solve(*((BcmHandle *)vh)->bcm);   // by value, as distributed_ver1.cpp:285 does
}

// single-expert prediction through the distributed_gp copy of Covsum
void ref_gp_predict(const double *X, const double *y, int n, int d, const double *hp,
                    const double *Xt, int nt, double *mean, double *var) {
    StdoutTo q(NULL);
    Covsum *gp = new Covsum(n, d);
    double t[3] = { hp[0], hp[1], hp[2] };
    gp->set_loghyperparam(t);
    double **Xr = rows_copy(X, n, d), **Xtr = rows_copy(Xt, nt, d);
    gp->compute_test_means_and_variances(Xr, const_cast<double *>(y), Xtr, mean, var, nt);
}

}  // extern "C"
