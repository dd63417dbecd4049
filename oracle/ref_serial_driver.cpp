// ref_serial_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver (our code) around the reference's OWN serial sources,
// compiled where they lie (see oracle/Makefile): cpp_serial_gp/covkernel.cpp +
// common/matrixops.cpp.  Product of the build is oracle/_ref/libref_serial.so,
// used only to pin the restatement in gp_oracle.c and to generate
// tests/golden/*.json (tests/golden/make_golden.py).  No reference source is
// copied: this file only declares and calls the reference's interface
// (cpp_serial_gp/covkernel.h:20-37, common/matrixops.h:5-25).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>

#include "covkernel.h"   // reference: cpp_serial_gp/covkernel.h (via -I)
#include "matrixops.h"   // reference: common/matrixops.h (via -I)

namespace {

// The reference prints progress to stdout from inside the arithmetic; route it
// to a log file (or /dev/null) for the duration of a call.
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

void rows_free(double **r, int n) {
    for (int i = 0; i < n; i++) delete[] r[i];
    delete[] r;
}

struct Handle {
    Covsum *gp;
    int n, d;
};

}  // namespace

extern "C" {

void *ref_gp_create(int n, int d) {
    Handle *h = new Handle;
    h->gp = new Covsum(n, d);
    h->n = n; h->d = d;
    return h;
}

void ref_gp_set_loghyper(void *vh, const double *hp) {
    double t[3] = { hp[0], hp[1], hp[2] };
    ((Handle *)vh)->gp->set_loghyperparam(t);
}

void ref_gp_get_loghyper(void *vh, double *hp) {
    double *p = ((Handle *)vh)->gp->get_loghyperparam();
    for (int i = 0; i < 3; i++) hp[i] = p[i];
}

double ref_gp_loglik(void *vh, const double *X, const double *y) {
    Handle *h = (Handle *)vh;
    StdoutTo q(NULL);
    double **Xr = rows_copy(X, h->n, h->d);
    double ll = h->gp->compute_loglikelihood(Xr, const_cast<double *>(y));
    rows_free(Xr, h->n);
    return ll;
}

void ref_gp_grad(void *vh, const double *X, const double *y, double *g) {
    Handle *h = (Handle *)vh;
    StdoutTo q(NULL);
    double **Xr = rows_copy(X, h->n, h->d);
    double *p = h->gp->compute_gradient_loghyperparam(Xr, const_cast<double *>(y));
    for (int i = 0; i < 3; i++) g[i] = p[i];
    rows_free(Xr, h->n);
}

void ref_gp_K_train(void *vh, const double *X, double *Kout) {
    Handle *h = (Handle *)vh;
    StdoutTo q(NULL);
    double **Xr = rows_copy(X, h->n, h->d);
    double **K = rows_copy(Kout, h->n, h->n);
    h->gp->compute_K_train(Xr, K);
    for (int i = 0; i < h->n; i++) memcpy(Kout + (size_t)i * h->n, K[i], h->n * sizeof(double));
    rows_free(Xr, h->n);
    rows_free(K, h->n);
}

void ref_gp_k_test(void *vh, const double *X, const double *xt, double *out) {
    Handle *h = (Handle *)vh;
    double **Xr = rows_copy(X, h->n, h->d);
    h->gp->compute_k_test(Xr, const_cast<double *>(xt), out);
    rows_free(Xr, h->n);
}

// runs Covsum::cg_solve with stdout (the "PLEASE-SEE" trace) captured in logpath
void ref_gp_cg_solve(void *vh, const double *X, const double *y, const char *logpath) {
    Handle *h = (Handle *)vh;
    StdoutTo q(logpath);
    double **Xr = rows_copy(X, h->n, h->d);
    h->gp->cg_solve(Xr, const_cast<double *>(y), true);
    rows_free(Xr, h->n);
}

void ref_gp_rprop_solve(void *vh, const double *X, const double *y, const char *logpath) {
    Handle *h = (Handle *)vh;
    StdoutTo q(logpath);
    double **Xr = rows_copy(X, h->n, h->d);
    h->gp->rprop_solve(Xr, const_cast<double *>(y), true);
    rows_free(Xr, h->n);
}

double ref_gp_nlpp(void *vh, const double *actual, const double *mean, const double *var, int nt) {
    StdoutTo q(NULL);
    return ((Handle *)vh)->gp->get_negative_log_predprob(const_cast<double *>(actual),
                                                         const_cast<double *>(mean),
                                                         const_cast<double *>(var), nt);
}

void ref_get_cholesky(const double *in, double *out, int n) {
    double **a = rows_copy(in, n, n), **l = rows_copy(in, n, n);
    get_cholesky(a, l, n);
    for (int i = 0; i < n; i++) memcpy(out + (size_t)i * n, l[i], n * sizeof(double));
    rows_free(a, n); rows_free(l, n);
}

void ref_K_inverse(const double *K, double *out, int n) {
    double **a = rows_copy(K, n, n), **o = rows_copy(K, n, n);
    compute_K_inverse(a, o, n);
    for (int i = 0; i < n; i++) memcpy(out + (size_t)i * n, o[i], n * sizeof(double));
    rows_free(a, n); rows_free(o, n);
}

void ref_chol_and_det(const double *K, const double *y, int n, double *quad, double *logdet) {
    StdoutTo q(NULL);
    double **a = rows_copy(K, n, n);
    std::pair<double, double> p = compute_chol_and_det(a, const_cast<double *>(y), n);
    *quad = p.first; *logdet = p.second;
    rows_free(a, n);
}

}  // extern "C"
