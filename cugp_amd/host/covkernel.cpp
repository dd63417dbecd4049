// covkernel.cpp -- class Covsum over the C-ABI (see covkernel.h).  No arithmetic of the GP happens
// here: rows are packed into the contiguous layout the library takes and every method is one call.
#include "covkernel.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/cugp.h"

namespace {
void must(int rc, const char *what)
{
    if (rc != CUGP_OK) throw std::runtime_error(std::string(what) + ": " + cugp_last_error());
}
}  // namespace

Covsum::Covsum() : handle(nullptr), inputdatasize(0), numdim(0), loghyper{0, 0, 0}, have_data(false), device(0) {}

Covsum::Covsum(int n, int d) : Covsum(n, d, 0) {}

Covsum::Covsum(int n, int d, int dev)
    : handle(nullptr), inputdatasize(n), numdim(d), loghyper{0, 0, 0}, have_data(false), device(dev)
{
    must(cugp_create(n, d, dev, &handle), "Covsum");
}

Covsum::~Covsum()
{
    if (handle) cugp_destroy(handle);
}

// Pack the row pointers; upload only when the contents differ from what the GPU already holds.
void Covsum::bind(double **X, double *y)
{
    const size_t n = inputdatasize, d = numdim;
    std::vector<double> flat(n * d);
    for (size_t i = 0; i < n; i++) memcpy(&flat[i * d], X[i], d * sizeof(double));
    const bool same_x = have_data && flat == xflat;
    const bool same_y = have_data && (y == nullptr || memcmp(y, ycopy.data(), n * sizeof(double)) == 0);
    if (same_x && same_y) return;
    xflat.swap(flat);
    if (y) ycopy.assign(y, y + n);
    else if (ycopy.size() != n) ycopy.assign(n, 0.0);
    must(cugp_set_data(handle, xflat.data(), ycopy.data()), "cugp_set_data");
    have_data = true;
}

double Covsum::compute_loglikelihood(double **X, double *y)
{
    bind(X, y);
    double ll = 0, g[3];
    // one factorisation serves this call and the gradient call that cg_solve makes next
    must(cugp_loglik_grad(handle, &ll, g), "cugp_loglik_grad");
    return ll;
}

double *Covsum::compute_gradient_loghyperparam(double **X, double *y)
{
    static double ans[3];                       // covkernel.cpp:167: shared by every instance
    bind(X, y);
    must(cugp_grad(handle, ans), "cugp_grad");
    return ans;
}

void Covsum::compute_K_train(double **X, double **output)
{
    bind(X, nullptr);
    const size_t n = inputdatasize;
    std::vector<double> K(n * n);
    must(cugp_compute_K_train(handle, K.data()), "cugp_compute_K_train");
    for (size_t i = 0; i < n; i++) memcpy(output[i], &K[i * n], n * sizeof(double));
}

void Covsum::compute_k_test(double **X, double *xtest, double *output)
{
    bind(X, nullptr);
    must(cugp_compute_k_test(handle, xtest, 1, output), "cugp_compute_k_test");
}

void Covsum::compute_squared_dist(double **X, double c)
{
    bind(X, nullptr);
    sqdist.resize((size_t)inputdatasize * inputdatasize);
    must(cugp_compute_squared_dist(handle, c, sqdist.data()), "cugp_compute_squared_dist");
}

double *Covsum::get_loghyperparam() { return loghyper; }

void Covsum::set_loghyperparam(const double *initval)
{
    for (int i = 0; i < 3; i++) loghyper[i] = initval[i];
    must(cugp_set_loghyper(handle, loghyper), "cugp_set_loghyper");
}

void Covsum::set_loghyperparam(double *initval) { set_loghyperparam((const double *)initval); }

void Covsum::compute_test_means_and_variances(double **X, double *y, double **Xtest, double *tmeanvec,
                                              double *tvarvec, int numtest)
{
    bind(X, y);
    std::vector<double> xt((size_t)numtest * numdim);
    for (int i = 0; i < numtest; i++) memcpy(&xt[(size_t)i * numdim], Xtest[i], numdim * sizeof(double));
    must(cugp_predict(handle, xt.data(), numtest, tmeanvec, tvarvec), "cugp_predict");
}

void Covsum::cg_solve(double **X, double *y, bool verbose)
{
    bind(X, y);
    const int budget = 100;                     // covkernel.cpp:413
    std::vector<double> trace(4 * (4 * budget + 8));
    int nev = 0;
    must(cugp_cg_solve(handle, budget, trace.data(), 4 * budget + 8, &nev), "cugp_cg_solve");
    must(cugp_get_loghyper(handle, loghyper), "cugp_get_loghyper");
    if (verbose) {
        for (int i = 0; i < nev; i++)
            printf("eval %3d  hp = %lf, %lf, %lf  f = %.10g\n", i, trace[4 * i], trace[4 * i + 1], trace[4 * i + 2],
                   trace[4 * i + 3]);
        printf("\n\n PLEASE-SEE 3 : %lf, %lf, %lf\n\n", loghyper[0], loghyper[1], loghyper[2]);
    }
}

void Covsum::rprop_solve(double **X, double *y, bool verbose)
{
    bind(X, y);
    const int iters = 100;                      // covkernel.cpp:345
    std::vector<double> trace(4 * (2 * iters + 8));
    int nev = 0;
    must(cugp_rprop_solve(handle, iters, trace.data(), 2 * iters + 8, &nev), "cugp_rprop_solve");
    must(cugp_get_loghyper(handle, loghyper), "cugp_get_loghyper");
    if (verbose)
        for (int i = 1; i < nev; i += 2) printf("%d %.10g\n", i / 2, trace[4 * i + 3]);
}

double Covsum::get_negative_log_predprob(double *actual, double *predmean, double *predvar, int TS)
{
    double out = 0;
    must(cugp_nlpp(actual, predmean, predvar, TS, &out), "cugp_nlpp");
    return out;
}

int Covsum::get_param_dim() { return numdim; }   // covkernel.cpp:661-663 (returns D, as the reference does)
