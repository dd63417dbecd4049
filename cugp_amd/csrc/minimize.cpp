// minimize.cpp -- host-side optimisers of the GP objective (3 log-hyper-parameters).
//
// cugp_cg_minimize follows the reference's Rasmussen-style "minimize" exactly in its arithmetic
// and control flow (cpp_serial_gp/covkernel.cpp:405-647, identical in
// distributed_gp/distributed_ver1.cpp:13-232 and cuda_scalingdist/cg_solver.cpp:292-523):
// Polack-Ribiere conjugate gradients, cubic extrapolation / quadratic-or-cubic interpolation line
// search under the Wolfe-Powell conditions, a budget counted in objective evaluations, both f and
// grad taken at every probe, NaN/Inf probes answered by bisecting the step.
// cugp_rprop_minimize follows covkernel.cpp:337-402.
//
// cugp_cg_minimize_sparing is the same loop with an objective split in two (value, then gradient at the same
// point): the gradient -- two thirds of an evaluation on the GPU -- is only asked for where the line search
// can use it.  A probe with f > f0 (the value the search started from) fails the sufficient-decrease test
// whatever its slope, is never the best point, and enters the interpolation only through the quadratic fit that
// ignores its slope (covkernel.cpp:532-536,556-560,572-573): its gradient is never read.  Same for a NaN/Inf probe
// (bisected).  With an objective whose value does not depend on which half is called, the trajectory is the
// default one probe for probe (tests/test_host_logic.py).  Opt-in: the default entry points keep the
// reference's "both at every probe" (covkernel.cpp:500-501,585-586).
//
// Pure host code: no device calls.  The objective is a callback so the same loop drives one
// expert, the experts of one GPU, or an all-reduced sum over ranks.
#include <cfloat>
#include <cmath>

#include "../../include/cugp.h"

namespace {

struct Vec3 {
    double v[3];
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};

inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 axpy(const Vec3& x, const Vec3& s, double t) { return Vec3{{x[0] + s[0] * t, x[1] + s[1] * t, x[2] + s[2] * t}}; }
inline Vec3 neg(const Vec3& a) { return Vec3{{-a[0], -a[1], -a[2]}}; }
inline bool any_nan(const Vec3& a) { return std::isnan(a[0]) || std::isnan(a[1]) || std::isnan(a[2]); }

class Probe {   // evaluates the objective and keeps the optional trace
public:
    Probe(cugp_objective_fn fn, void* ctx, double* trace, int cap) : fn_(fn), ctx_(ctx), trace_(trace), cap_(cap) {}
    Probe(cugp_value_fn vf, cugp_gradient_fn gf, void* ctx, double* trace, int cap)
        : vf_(vf), gf_(gf), ctx_(ctx), trace_(trace), cap_(cap) {}
    // fref: the value the current line search started from (a probe above it never needs its gradient)
    void at(const Vec3& th, double& f, Vec3& g, double fref)
    {
        if (fn_) {
            fn_(ctx_, th.v, &f, g.v);
        } else {
            vf_(ctx_, th.v, &f);
            if (std::isnan(f) || std::isinf(f) || f > fref) g = Vec3{{0.0, 0.0, 0.0}};   // never read (see header)
            else { gf_(ctx_, th.v, g.v); ++grads_; }
        }
        if (trace_ && count_ < cap_) {
            double* r = trace_ + 4 * (long)count_;
            r[0] = th[0]; r[1] = th[1]; r[2] = th[2]; r[3] = f;
        }
        ++count_;
    }
    int count() const { return count_; }
    int grads() const { return fn_ ? count_ : grads_; }
private:
    cugp_objective_fn fn_ = nullptr;
    cugp_value_fn vf_ = nullptr;
    cugp_gradient_fn gf_ = nullptr;
    int grads_ = 0;
    void* ctx_;
    double* trace_;
    int cap_;
    int count_ = 0;
};

}  // namespace

static int cg_loop(Probe& probe, double theta[3], int budget);

extern "C" int cugp_cg_minimize(cugp_objective_fn fn, void* ctx, double theta[3], int budget, double* trace,
                                int trace_cap, int* nevals)
{
    if (!fn || !theta || budget < 0) return CUGP_ERR_INVALID;
    Probe probe(fn, ctx, trace, trace_cap);
    const int rc = cg_loop(probe, theta, budget);
    if (nevals) *nevals = probe.count();
    return rc;
}

extern "C" int cugp_cg_minimize_sparing(cugp_value_fn value, cugp_gradient_fn gradient, void* ctx, double theta[3],
                                        int budget, double* trace, int trace_cap, int* nevals, int* ngrads)
{
    if (!value || !gradient || !theta || budget < 0) return CUGP_ERR_INVALID;
    Probe probe(value, gradient, ctx, trace, trace_cap);
    const int rc = cg_loop(probe, theta, budget);
    if (nevals) *nevals = probe.count();
    if (ngrads) *ngrads = probe.grads();
    return rc;
}

static int cg_loop(Probe& probe, double theta[3], int budget)
{
    // constants: covkernel.cpp:407-411
    const double kInt = 0.1, kExt = 3.0, kRatio = 10, kSig = 0.1, kRho = kSig / 2;
    const int kMaxPerSearch = 20;
    const int n = budget;

    Vec3 X{{theta[0], theta[1], theta[2]}};
    Vec3 df0, df3, s;
    double f0;
    probe.at(X, f0, df0, INFINITY);          // covkernel.cpp:438-439
    s = neg(df0);
    df3 = df0;
    double d0 = -dot(s, s);
    double x3 = 1 / (1 - d0);                // initial step, covkernel.cpp:457
    double f3 = 0, d3 = 0, x2 = 0, x4 = 0, f2 = 0, f4 = 0, d2 = 0, d4 = 0;
    bool prev_failed = false;

    for (int i = 0; i < n; ++i) {
        Vec3 bestX = X, bestG = df0;
        double bestF = f0;
        unsigned left = (unsigned)(kMaxPerSearch < (n - i) ? kMaxPerSearch : (n - i));

        // (1) extrapolate until the Wolfe-Powell conditions say "far enough"
        for (;;) {
            x2 = 0; f2 = f0; d2 = d0; f3 = f0; df3 = df0;
            bool ok = false;
            while (!ok && left > 0) {
                --left; ++i;
                probe.at(axpy(X, s, x3), f3, df3, f0);
                if (!std::isnan(f3) && !std::isinf(f3) && !any_nan(df3)) ok = true;
                else x3 = (x2 + x3) / 2;     // non-PD covariance comes back NaN: halve the step
            }
            if (f3 < bestF) { bestX = axpy(X, s, x3); bestF = f3; bestG = df3; }
            d3 = dot(df3, s);
            if (d3 > kSig * d0 || f3 > f0 + x3 * kRho * d0 || left == 0) break;

            const double x1 = x2, f1 = f2, d1 = d2;
            x2 = x3; f2 = f3; d2 = d3;
            const double A = 6 * (f1 - f2) + 3 * (d2 + d1) * (x2 - x1);
            const double B = 3 * (f2 - f1) - (2 * d1 + d2) * (x2 - x1);
            x3 = x1 - d1 * (x2 - x1) * (x2 - x1) / (B + std::sqrt(B * B - A * d1 * (x2 - x1)));
            if (std::isnan(x3) || x3 < 0 || x3 > x2 * kExt) x3 = kExt * x2;
            else if (x3 < x2 + kInt * (x2 - x1)) x3 = x2 + kInt * (x2 - x1);
        }

        // (2) interpolate inside the bracket
        while ((std::fabs(d3) > -kSig * d0 || f3 > f0 + x3 * kRho * d0) && left > 0) {
            if (d3 > 0 || f3 > f0 + x3 * kRho * d0) { x4 = x3; f4 = f3; d4 = d3; }
            else { x2 = x3; f2 = f3; d2 = d3; }
            if (f4 > f0) {
                x3 = x2 - (0.5 * d2 * (x4 - x2) * (x4 - x2)) / (f4 - f2 - d2 * (x4 - x2));
            } else {
                const double A = 6 * (f2 - f4) / (x4 - x2) + 3 * (d4 + d2);
                const double B = 3 * (f4 - f2) - (2 * d2 + d4) * (x4 - x2);
                x3 = x2 + std::sqrt(B * B - A * d2 * (x4 - x2) * (x4 - x2) - B) / A;
            }
            if (std::isnan(x3) || std::isinf(x3)) x3 = (x2 + x4) / 2;
            const double hi = x4 - kInt * (x4 - x2), lo = x2 + kInt * (x4 - x2);
            x3 = std::fmax(std::fmin(x3, hi), lo);
            probe.at(axpy(X, s, x3), f3, df3, f0);
            if (f3 < bestF) { bestX = axpy(X, s, x3); bestF = f3; bestG = df3; }
            --left; ++i;
            d3 = dot(df3, s);
        }

        // (3) new direction (Polack-Ribiere) or fall back to steepest descent
        if (std::fabs(d3) < -kSig * d0 && f3 < f0 + x3 * kRho * d0) {
            X = axpy(X, s, x3);
            f0 = f3;
            const double beta = (dot(df3, df3) - dot(df0, df3)) / (dot(df0, df0));
            s = Vec3{{beta * s[0] - df3[0], beta * s[1] - df3[1], beta * s[2] - df3[2]}};
            df0 = df3;
            d3 = d0;
            d0 = dot(df0, s);
            if (d0 > 0) { s = neg(df0); d0 = -dot(s, s); }
            const double ratio = d3 / (d0 - DBL_MIN);
            x3 = x3 * (kRatio < ratio ? kRatio : ratio);
            prev_failed = false;
        } else {
            X = bestX; f0 = bestF; df0 = bestG;
            if (prev_failed || i >= n) break;
            s = neg(df0);
            d0 = -dot(s, s);
            x3 = 1 / (1 - d0);
            prev_failed = true;
        }
    }
    theta[0] = X[0]; theta[1] = X[1]; theta[2] = X[2];
    return CUGP_OK;
}

extern "C" int cugp_rprop_minimize(cugp_objective_fn fn, void* ctx, double theta[3], int iters, double* trace,
                                   int trace_cap, int* nevals)
{
    if (!fn || !theta || iters < 0) return CUGP_ERR_INVALID;
    // covkernel.cpp:339-345
    const double eps_stop = 0.0, delta0 = 0.1, delta_min = 1e-6, delta_max = 50, eta_minus = 0.5, eta_plus = 1.2;
    Probe probe(fn, ctx, trace, trace_cap);
    Vec3 delta{{delta0, delta0, delta0}}, prev{{0, 0, 0}};
    Vec3 p{{theta[0], theta[1], theta[2]}}, best_p = p;
    double best = -INFINITY;

    for (int it = 0; it < iters; ++it) {
        Vec3 g, gscratch;
        double fscratch, f;
        probe.at(p, fscratch, g, INFINITY);            // gradient at the current point (:369)
        for (int j = 0; j < 3; j++) prev[j] = prev[j] * g[j];
        for (int j = 0; j < 3; j++) {
            if (prev[j] > 0) {
                delta[j] = std::fmin(delta[j] * eta_plus, delta_max);
            } else if (prev[j] < 0) {
                delta[j] = std::fmax(delta[j] * eta_minus, delta_min);
                g[j] = 0;
            }
            const double sg = g[j] > 0 ? 1.0 : (g[j] < 0 ? -1.0 : 0.0);
            p[j] += -sg * delta[j];
        }
        prev = g;
        if (std::sqrt(dot(prev, prev)) < eps_stop) break;
        probe.at(p, f, gscratch, INFINITY);            // likelihood at the stepped point (:393)
        const double lik = -f;
        if (lik > best) { best = lik; best_p = p; }
    }
    theta[0] = best_p[0]; theta[1] = best_p[1]; theta[2] = best_p[2];
    if (nevals) *nevals = probe.count();
    return CUGP_OK;
}
