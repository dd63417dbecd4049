/*
 * gp_oracle.c -- TEST INFRASTRUCTURE ONLY (see gp_oracle.h).
 *
 * CPU restatement of the reference's serial GP path.  Operation ORDER follows
 * the reference wherever it affects rounding (truncated constants 1.83787 and
 * 6.283185, exp(2*theta), -val*0.5/ell_sq, unblocked right-looking Cholesky,
 * three factorisations per gradient, explicit inverse by N-column substitution).
 * Build with -O3 -ffp-contract=off (the reference's g++ -O3 emits no FMAs on
 * x86-64) -- oracle/Makefile does.  Each function cites the reference
 * file:line it follows (paths relative to the reference checkout).
 */
#include "gp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- helpers */

/* The reference allocates every matrix as n separately new'd rows
 * (covkernel.cpp:26-34, matrixops.cpp:119-126); keep that so cache/TLB
 * behaviour of the baseline timing matches. */
static double **rows_alloc(int n, int m)
{
    double **r = (double **)malloc((size_t)n * sizeof(double *));
    for (int i = 0; i < n; i++)
        r[i] = (double *)malloc((size_t)m * sizeof(double));
    return r;
}

static void rows_free(double **r, int n)
{
    if (!r) return;
    for (int i = 0; i < n; i++) free(r[i]);
    free(r);
}

static void rows_from_flat(double **r, const double *flat, int n, int m)
{
    for (int i = 0; i < n; i++) memcpy(r[i], flat + (size_t)i * m, (size_t)m * sizeof(double));
}

static void rows_to_flat(double **r, double *flat, int n, int m)
{
    for (int i = 0; i < n; i++) memcpy(flat + (size_t)i * m, r[i], (size_t)m * sizeof(double));
}

/* matrixops.cpp:220-230 */
static void vec_sub(const double *a, const double *b, double *c, int d)
{
    for (int i = 0; i < d; i++) c[i] = a[i] - b[i];
}

static double vec_dot(const double *a, const double *b, int d)
{
    double s = 0.0;
    for (int i = 0; i < d; i++) s += a[i] * b[i];
    return s;
}

/* ------------------------------------------------- matrixops.cpp: Cholesky */

/* matrixops.cpp:68-108 -- unblocked right-looking, column scaled then rank-1
 * update of the trailing lower triangle, strict upper zeroed at the end. */
static void chol_rows(double **in, double **out, int n)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) out[i][j] = in[i][j];

    for (int c = 0; c < n; c++) {
        out[c][c] = sqrt(out[c][c]);
        for (int r = c + 1; r < n; r++) out[r][c] = out[r][c] / out[c][c];
        for (int c2 = c + 1; c2 < n; c2++)
            for (int r2 = c2; r2 < n; r2++)
                out[r2][c2] = out[r2][c2] - out[r2][c] * out[c2][c];
    }
    for (int r = 0; r < n; r++)
        for (int c = r + 1; c < n; c++) out[r][c] = 0.0;
}

static void transpose_rows(double **in, double **out, int n)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) out[j][i] = in[i][j];
}

/* forward then backward vector substitution, matrixops.cpp:144-164 / :284-303 */
static void chol_solve_vec(double **L, double **U, const double *b, double *tmp, double *x, int n)
{
    for (int i = 0; i < n; i++) {
        tmp[i] = b[i];
        for (int j = 0; j < i; j++) tmp[i] -= L[i][j] * tmp[j];
        tmp[i] /= L[i][i];
    }
    for (int i = n - 1; i >= 0; i--) {
        x[i] = tmp[i];
        for (int j = i + 1; j < n; j++) x[i] -= U[i][j] * x[j];
        x[i] /= U[i][i];
    }
}

/* matrixops.cpp:113-185 (multiply_and_get_logdeterminant) */
static void chol_and_det_rows(double **K, const double *y, int n, double *quad, double *logdet)
{
    double **L = rows_alloc(n, n), **U = rows_alloc(n, n);
    double *x = (double *)malloc((size_t)n * sizeof(double));
    double *tmp = (double *)malloc((size_t)n * sizeof(double));
    double det = 0.0, prod = 0.0;

    chol_rows(K, L, n);
    for (int i = 0; i < n; i++) det += log(L[i][i]);
    det = 2 * det;
    transpose_rows(L, U, n);
    chol_solve_vec(L, U, y, tmp, x, n);
    for (int i = 0; i < n; i++) prod += y[i] * x[i];

    *quad = prod;
    *logdet = det;
    free(x); free(tmp);
    rows_free(L, n); rows_free(U, n);
}

/* matrixops.cpp:264-316 (vector_Kinvy_using_cholesky): factorises again */
static void Kinvy_rows(double **K, const double *y, double *ans, int n)
{
    double **L = rows_alloc(n, n), **U = rows_alloc(n, n);
    double *tmp = (double *)malloc((size_t)n * sizeof(double));
    chol_rows(K, L, n);
    transpose_rows(L, U, n);
    chol_solve_vec(L, U, y, tmp, ans, n);
    free(tmp);
    rows_free(L, n); rows_free(U, n);
}

/* matrixops.cpp:330-340 / :361-372: one RHS column at a time, no sparsity used */
static void fwd_subst_matrix(double **A, double **B, double **out, int n)
{
    for (int k = 0; k < n; k++)
        for (int i = 0; i < n; i++) {
            out[i][k] = B[i][k];
            for (int j = 0; j < i; j++) out[i][k] = out[i][k] - A[i][j] * out[j][k];
            out[i][k] = out[i][k] / A[i][i];
        }
}

static void bwd_subst_matrix(double **A, double **B, double **out, int n)
{
    for (int k = 0; k < n; k++)
        for (int i = n - 1; i >= 0; i--) {
            out[i][k] = B[i][k];
            for (int j = i + 1; j < n; j++) out[i][k] = out[i][k] - A[i][j] * out[j][k];
            out[i][k] = out[i][k] / A[i][i];
        }
}

/* matrixops.cpp:383-435 (compute_K_inverse): K^-1 = L^-T (L^-1 I) */
static void K_inverse_rows(double **K, double **out, int n)
{
    double **Lt = rows_alloc(n, n), **T = rows_alloc(n, n);
    double **I = rows_alloc(n, n), **L = rows_alloc(n, n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) I[i][j] = (i == j) ? 1.0 : 0.0;
    chol_rows(K, L, n);
    fwd_subst_matrix(L, I, T, n);
    transpose_rows(L, Lt, n);
    bwd_subst_matrix(Lt, T, out, n);
    rows_free(Lt, n); rows_free(T, n); rows_free(I, n); rows_free(L, n);
}

/* ---- flat-array entry points for the LA pieces ---- */

void oracle_get_cholesky(const double *in, double *out, int n)
{
    double **a = rows_alloc(n, n), **l = rows_alloc(n, n);
    rows_from_flat(a, in, n, n);
    chol_rows(a, l, n);
    rows_to_flat(l, out, n, n);
    rows_free(a, n); rows_free(l, n);
}

void oracle_chol_and_det(const double *K, const double *y, int n, double *quad, double *logdet)
{
    double **a = rows_alloc(n, n);
    rows_from_flat(a, K, n, n);
    chol_and_det_rows(a, y, n, quad, logdet);
    rows_free(a, n);
}

void oracle_K_inverse(const double *K, double *out, int n)
{
    double **a = rows_alloc(n, n), **o = rows_alloc(n, n);
    rows_from_flat(a, K, n, n);
    K_inverse_rows(a, o, n);
    rows_to_flat(o, out, n, n);
    rows_free(a, n); rows_free(o, n);
}

void oracle_Kinvy(const double *K, const double *y, double *ans, int n)
{
    double **a = rows_alloc(n, n);
    rows_from_flat(a, K, n, n);
    Kinvy_rows(a, y, ans, n);
    rows_free(a, n);
}

/* ------------------------------------------------------------------ Covsum */

struct oracle_gp {
    int n, d;
    double hp[3];
    /* covkernel.h:10-18: the same seven n x n scratch matrices */
    double **K, **S, **KS, **W, **AA, **Kinv;
    double *vec, *tvec;
    const double **Xrows; /* row views into the caller's flat X */
};

oracle_gp *oracle_gp_create(int n, int d)
{
    oracle_gp *gp = (oracle_gp *)calloc(1, sizeof(*gp));
    gp->n = n; gp->d = d;
    gp->K = rows_alloc(n, n);  gp->S = rows_alloc(n, n);  gp->KS = rows_alloc(n, n);
    gp->W = rows_alloc(n, n);  gp->AA = rows_alloc(n, n); gp->Kinv = rows_alloc(n, n);
    gp->vec = (double *)malloc((size_t)(n > d ? n : d) * sizeof(double));
    gp->tvec = (double *)malloc((size_t)(n > d ? n : d) * sizeof(double));
    gp->Xrows = (const double **)malloc((size_t)n * sizeof(double *));
    return gp;
}

void oracle_gp_destroy(oracle_gp *gp)
{
    if (!gp) return;
    rows_free(gp->K, gp->n);  rows_free(gp->S, gp->n);  rows_free(gp->KS, gp->n);
    rows_free(gp->W, gp->n);  rows_free(gp->AA, gp->n); rows_free(gp->Kinv, gp->n);
    free(gp->vec); free(gp->tvec); free((void *)gp->Xrows);
    free(gp);
}

void oracle_gp_set_loghyper(oracle_gp *gp, const double hp[3])
{
    for (int i = 0; i < 3; i++) gp->hp[i] = hp[i];
}

void oracle_gp_get_loghyper(const oracle_gp *gp, double hp[3])
{
    for (int i = 0; i < 3; i++) hp[i] = gp->hp[i];
}

static const double **xrows(oracle_gp *gp, const double *X)
{
    for (int i = 0; i < gp->n; i++) gp->Xrows[i] = X + (size_t)i * gp->d;
    return gp->Xrows;
}

/* covkernel.cpp:64-102 */
static void K_train_rows(oracle_gp *gp, const double **X, double **out)
{
    double ell_sq = exp(gp->hp[0] * 2);
    double signal_var = exp(gp->hp[1] * 2);
    double noise_var = exp(gp->hp[2] * 2);
    int n = gp->n;
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            vec_sub(X[i], X[j], gp->vec, gp->d);
            double val = vec_dot(gp->vec, gp->vec, gp->d);
            val = signal_var * exp(-val * 0.5 / ell_sq);
            out[i][j] = val;
            out[j][i] = val;
            if (i == j) out[i][j] += noise_var;
        }
}

void oracle_gp_K_train(oracle_gp *gp, const double *X, double *Kout)
{
    K_train_rows(gp, xrows(gp, X), gp->K);
    rows_to_flat(gp->K, Kout, gp->n, gp->n);
}

/* covkernel.cpp:105-116: no noise term on the test covariance */
static void k_test_vec(oracle_gp *gp, const double **X, const double *xt, double *out)
{
    double ell_sq = exp(gp->hp[0] * 2);
    double signal_var = exp(gp->hp[1] * 2);
    for (int i = 0; i < gp->n; i++) {
        vec_sub(X[i], xt, gp->tvec, gp->d);
        double val = vec_dot(gp->tvec, gp->tvec, gp->d);
        out[i] = signal_var * exp(-val * 0.5 / ell_sq);
    }
}

void oracle_gp_k_test(oracle_gp *gp, const double *X, const double *xt, double *out)
{
    k_test_vec(gp, xrows(gp, X), xt, out);
}

/* covkernel.cpp:130-157 */
static void sqdist_rows(oracle_gp *gp, const double **X, double c, double **S)
{
    int n = gp->n;
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            if (i == j) { S[i][j] = 0.0; continue; }
            vec_sub(X[i], X[j], gp->vec, gp->d);
            double val = vec_dot(gp->vec, gp->vec, gp->d) / c;
            S[i][j] = val;
            S[j][i] = val;
        }
}

void oracle_gp_sqdist(oracle_gp *gp, const double *X, double c, double *Sout)
{
    sqdist_rows(gp, xrows(gp, X), c, gp->S);
    rows_to_flat(gp->S, Sout, gp->n, gp->n);
}

/* covkernel.cpp:118-129 */
double oracle_gp_loglik(oracle_gp *gp, const double *X, const double *y)
{
    int n = gp->n;
    double quad, logdet;
    K_train_rows(gp, xrows(gp, X), gp->K);
    chol_and_det_rows(gp->K, y, n, &quad, &logdet);
    return -0.5 * (quad + logdet + n * 1.83787);
}

/* covkernel.cpp:162-263: gradient of the NEGATIVE log-likelihood */
void oracle_gp_grad(oracle_gp *gp, const double *X, const double *y, double g[3])
{
    int n = gp->n;
    const double **Xr = xrows(gp, X);
    double ell_sq = exp(gp->hp[0] * 2);
    double noise_var = exp(gp->hp[2] * 2);

    K_train_rows(gp, Xr, gp->K);
    sqdist_rows(gp, Xr, ell_sq, gp->S);
    for (int i = 0; i < n; i++)                      /* matrixops.cpp:437-448 */
        for (int j = 0; j < n; j++) gp->KS[i][j] = gp->K[i][j] * gp->S[i][j];

    K_inverse_rows(gp->K, gp->Kinv, n);              /* 2nd factorisation */
    Kinvy_rows(gp->K, y, gp->vec, n);                /* 3rd factorisation; vec = alpha */

    for (int i = 0; i < n; i++)                      /* matrixops.cpp:250-260 */
        for (int j = 0; j < n; j++) gp->AA[i][j] = gp->vec[i] * gp->vec[j];
    for (int i = 0; i < n; i++)                      /* matrixops.cpp:237-247 */
        for (int j = 0; j < n; j++) gp->W[i][j] = gp->Kinv[i][j] - gp->AA[i][j];

    double p1 = 0.0, p2 = 0.0, p3 = 0.0;             /* covkernel.cpp:244-254 */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double w = gp->W[i][j];
            p1 += w * gp->KS[i][j];
            p2 += w * 2.0 * gp->K[i][j];
            if (i == j) {
                p3 += w * noise_var * 2;
                p2 -= w * 2.0 * noise_var;
            }
        }
    g[0] = p1 / 2.0;
    g[1] = p2 / 2.0;
    g[2] = p3 / 2.0;
}

/* covkernel.cpp:277-323: one test point at a time; variance includes noise */
void oracle_gp_predict(oracle_gp *gp, const double *X, const double *y, const double *Xt,
                       int nt, double *mean, double *var)
{
    int n = gp->n;
    const double **Xr = xrows(gp, X);
    double signal_var = exp(gp->hp[1] * 2);
    double noise_var = exp(gp->hp[2] * 2);
    double *ks = (double *)malloc((size_t)n * sizeof(double));
    double *sv = (double *)malloc((size_t)n * sizeof(double));
    double *alpha = (double *)malloc((size_t)n * sizeof(double));

    K_train_rows(gp, Xr, gp->K);
    Kinvy_rows(gp->K, y, alpha, n);
    K_inverse_rows(gp->K, gp->Kinv, n);

    for (int t = 0; t < nt; t++) {
        k_test_vec(gp, Xr, Xt + (size_t)t * gp->d, ks);
        double m = 0.0;                                /* matrixops.cpp:52-58 */
        for (int i = 0; i < n; i++) m += ks[i] * alpha[i];
        mean[t] = m;
        var[t] = signal_var + noise_var;
        for (int k = 0; k < n; k++) {                  /* matrixops.cpp:25-37 */
            double s = 0.0;
            for (int i = 0; i < n; i++) s += ks[i] * gp->Kinv[i][k];
            sv[k] = s;
        }
        double q = 0.0;
        for (int i = 0; i < n; i++) q += sv[i] * ks[i];
        var[t] -= q;
    }
    free(ks); free(sv); free(alpha);
}

/* covkernel.cpp:649-659 (== BCM.cpp:34-42): 2*pi truncated to 6.283185 */
double oracle_nlpp(const double *actual, const double *mean, const double *var, int nt)
{
    double ans = 0.0;
    for (int i = 0; i < nt; i++) {
        double v = 0.5 * log(6.283185 * var[i]) + pow((mean[i] - actual[i]), 2) / (2 * var[i]);
        ans += v;
    }
    return ans / nt;
}

/* -------------------------------------------------------------- optimisers */

static double dot3(const double a[3], const double b[3])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

struct tracer { double *buf; int cap, n; };

static void eval_at(oracle_objective_fn fn, void *ctx, const double th[3], double *f, double g[3],
                    struct tracer *tr)
{
    fn(ctx, th, f, g);
    if (tr->buf && tr->n < tr->cap) {
        double *r = tr->buf + 4 * (size_t)tr->n;
        r[0] = th[0]; r[1] = th[1]; r[2] = th[2]; r[3] = *f;
    }
    tr->n++;
}

/* covkernel.cpp:405-647 (identical in distributed_ver1.cpp:13-232 and
 * cuda_scalingdist/cg_solver.cpp:292-523): Rasmussen-style minimize, Polack-
 * Ribiere CG with cubic extrapolation / interpolation line search.  `budget`
 * counts objective evaluations inside the loop; f and grad are both taken at
 * every probe. */
int oracle_cg_minimize(oracle_objective_fn fn, void *ctx, double theta[3], int budget,
                       double *trace, int trace_cap)
{
    const double INT = 0.1, EXT = 3.0, RATIO = 10, SIG = 0.1, RHO = SIG / 2;
    const int MAXEV = 20;
    const int n = budget;
    struct tracer tr = { trace, trace_cap, 0 };
    int ls_failed = 0;
    double X[3] = { theta[0], theta[1], theta[2] };
    double f0, df0[3], s[3], df3[3], probe[3];

    eval_at(fn, ctx, X, &f0, df0, &tr);
    for (int j = 0; j < 3; j++) { s[j] = -df0[j]; df3[j] = df0[j]; }
    double d0 = -dot3(s, s);
    double x3 = 1 / (1 - d0);
    double f3 = 0, d3 = 0, x2 = 0, x4 = 0, f2 = 0, f4 = 0, d2 = 0, d4 = 0;

    for (int i = 0; i < n; ++i) {
        double X0[3] = { X[0], X[1], X[2] };
        double F0 = f0;
        double dF0[3] = { df0[0], df0[1], df0[2] };
        unsigned int M = (unsigned int)(MAXEV < (n - i) ? MAXEV : (n - i));

        for (;;) { /* extrapolate until far enough */
            x2 = 0; f2 = f0; d2 = d0; f3 = f0;
            for (int j = 0; j < 3; j++) df3[j] = df0[j];
            int success = 0;
            while (!success && M > 0) {
                M--; i++;
                for (int j = 0; j < 3; j++) probe[j] = X[j] + s[j] * x3;
                eval_at(fn, ctx, probe, &f3, df3, &tr);
                int bad = isnan(df3[0]) || isnan(df3[1]) || isnan(df3[2]);
                if (!isnan(f3) && !isinf(f3) && !bad) success = 1;
                else x3 = (x2 + x3) / 2;     /* non-PD => NaN => bisect */
            }
            if (f3 < F0) {
                for (int j = 0; j < 3; j++) { X0[j] = X[j] + s[j] * x3; dF0[j] = df3[j]; }
                F0 = f3;
            }
            d3 = dot3(df3, s);
            if ((d3 > SIG * d0) || (f3 > f0 + x3 * RHO * d0) || M == 0) break;

            double x1 = x2, f1 = f2, d1 = d2;
            x2 = x3; f2 = f3; d2 = d3;
            double A = 6 * (f1 - f2) + 3 * (d2 + d1) * (x2 - x1);
            double B = 3 * (f2 - f1) - (2 * d1 + d2) * (x2 - x1);
            x3 = x1 - d1 * (x2 - x1) * (x2 - x1) / (B + sqrt(B * B - A * d1 * (x2 - x1)));
            if (isnan(x3) || x3 < 0 || x3 > x2 * EXT) x3 = EXT * x2;
            else if (x3 < x2 + INT * (x2 - x1)) x3 = x2 + INT * (x2 - x1);
        }

        while (((fabs(d3) > -SIG * d0) || (f3 > f0 + x3 * RHO * d0)) && (M > 0)) { /* interpolate */
            if ((d3 > 0) || (f3 > f0 + x3 * RHO * d0)) { x4 = x3; f4 = f3; d4 = d3; }
            else { x2 = x3; f2 = f3; d2 = d3; }

            if (f4 > f0)
                x3 = x2 - (0.5 * d2 * (x4 - x2) * (x4 - x2)) / (f4 - f2 - d2 * (x4 - x2));
            else {
                double A = 6 * (f2 - f4) / (x4 - x2) + 3 * (d4 + d2);
                double B = 3 * (f4 - f2) - (2 * d2 + d4) * (x4 - x2);
                x3 = x2 + sqrt(B * B - A * d2 * (x4 - x2) * (x4 - x2) - B) / A;
            }
            if (isnan(x3) || isinf(x3)) x3 = (x2 + x4) / 2;
            {
                double hi = x4 - INT * (x4 - x2), lo = x2 + INT * (x4 - x2);
                double t = x3 < hi ? x3 : hi;
                x3 = t > lo ? t : lo;
            }
            for (int j = 0; j < 3; j++) probe[j] = X[j] + s[j] * x3;
            eval_at(fn, ctx, probe, &f3, df3, &tr);
            if (f3 < F0) {
                for (int j = 0; j < 3; j++) { X0[j] = X[j] + s[j] * x3; dF0[j] = df3[j]; }
                F0 = f3;
            }
            M--; i++;
            d3 = dot3(df3, s);
        }

        if ((fabs(d3) < -SIG * d0) && (f3 < f0 + x3 * RHO * d0)) { /* line search succeeded */
            for (int j = 0; j < 3; j++) X[j] = X[j] + s[j] * x3;
            f0 = f3;
            double c = (dot3(df3, df3) - dot3(df0, df3)) / (dot3(df0, df0));
            for (int j = 0; j < 3; j++) s[j] = c * s[j] - df3[j];
            for (int j = 0; j < 3; j++) df0[j] = df3[j];
            d3 = d0; d0 = dot3(df0, s);
            if (d0 > 0) {
                for (int j = 0; j < 3; j++) s[j] = -df0[j];
                d0 = -dot3(s, s);
            }
            {
                double r = d3 / (d0 - DBL_MIN);
                x3 = x3 * (RATIO < r ? RATIO : r);
            }
            ls_failed = 0;
        } else {
            for (int j = 0; j < 3; j++) { X[j] = X0[j]; df0[j] = dF0[j]; }
            f0 = F0;
            if (ls_failed || i >= n) break;
            for (int j = 0; j < 3; j++) s[j] = -df0[j];
            d0 = -dot3(s, s);
            x3 = 1 / (1 - d0);
            ls_failed = 1;
        }
    }
    for (int j = 0; j < 3; j++) theta[j] = X[j];
    return tr.n;
}

/* covkernel.cpp:337-402 (rprop_solve): the gradient is taken at the current
 * parameters, then the likelihood at the stepped parameters. The callback is
 * therefore invoked twice per iteration (grad use, then f use). */
int oracle_rprop_minimize(oracle_objective_fn fn, void *ctx, double theta[3], int iters,
                          double *trace, int trace_cap)
{
    const double eps_stop = 0.0, Delta0 = 0.1, Deltamin = 1e-6, Deltamax = 50;
    const double etaminus = 0.5, etaplus = 1.2;
    struct tracer tr = { trace, trace_cap, 0 };
    double Delta[3] = { Delta0, Delta0, Delta0 };
    double grad_old[3] = { 0, 0, 0 };
    double params[3] = { theta[0], theta[1], theta[2] };
    double best_params[3] = { theta[0], theta[1], theta[2] };
    double best = -INFINITY; /* log(0) */

    for (int i = 0; i < iters; ++i) {
        double f, grad[3], fdummy, gdummy[3];
        eval_at(fn, ctx, params, &fdummy, grad, &tr);
        for (int j = 0; j < 3; j++) grad_old[j] = grad_old[j] * grad[j];
        for (int j = 0; j < 3; j++) {
            if (grad_old[j] > 0) {
                double t = Delta[j] * etaplus;
                Delta[j] = t < Deltamax ? t : Deltamax;
            } else if (grad_old[j] < 0) {
                double t = Delta[j] * etaminus;
                Delta[j] = t > Deltamin ? t : Deltamin;
                grad[j] = 0;
            }
            double sg = grad[j] > 0 ? 1.0 : (grad[j] < 0 ? -1.0 : 0.0);
            params[j] += -sg * Delta[j];
        }
        for (int j = 0; j < 3; j++) grad_old[j] = grad[j];
        if (sqrt(dot3(grad_old, grad_old)) < eps_stop) break;
        eval_at(fn, ctx, params, &f, gdummy, &tr);
        double lik = -f;
        if (lik > best) {
            best = lik;
            for (int j = 0; j < 3; j++) best_params[j] = params[j];
        }
    }
    for (int j = 0; j < 3; j++) theta[j] = best_params[j];
    return tr.n;
}

struct gp_ctx { oracle_gp *gp; const double *X, *y; };

static void gp_objective(void *c, const double th[3], double *f, double g[3])
{
    struct gp_ctx *x = (struct gp_ctx *)c;
    oracle_gp_set_loghyper(x->gp, th);
    *f = -1.0 * oracle_gp_loglik(x->gp, x->X, x->y);
    oracle_gp_grad(x->gp, x->X, x->y, g);
}

int oracle_gp_cg_solve(oracle_gp *gp, const double *X, const double *y, int budget,
                       double *trace, int trace_cap)
{
    struct gp_ctx c = { gp, X, y };
    double th[3];
    oracle_gp_get_loghyper(gp, th);
    int ne = oracle_cg_minimize(gp_objective, &c, th, budget, trace, trace_cap);
    oracle_gp_set_loghyper(gp, th);   /* covkernel.cpp:646 */
    return ne;
}

int oracle_gp_rprop_solve(oracle_gp *gp, const double *X, const double *y, int iters,
                          double *trace, int trace_cap)
{
    struct gp_ctx c = { gp, X, y };
    double th[3];
    oracle_gp_get_loghyper(gp, th);
    int ne = oracle_rprop_minimize(gp_objective, &c, th, iters, trace, trace_cap);
    oracle_gp_set_loghyper(gp, th);
    return ne;
}

/* --------------------------------------------------------------------- BCM */

struct oracle_bcm {
    const double *X, *y;
    int N, D, K;
    int *offset, *rows;
    oracle_gp **experts;
    double hp[3];
};

/* BCM.cpp:85-110: contiguous row ranges of floor(N/K), remainder to the last */
oracle_bcm *oracle_bcm_create(const double *X, const double *y, int N, int D, int K)
{
    oracle_bcm *b = (oracle_bcm *)calloc(1, sizeof(*b));
    b->X = X; b->y = y; b->N = N; b->D = D; b->K = K;
    b->offset = (int *)malloc((size_t)K * sizeof(int));
    b->rows = (int *)malloc((size_t)K * sizeof(int));
    b->experts = (oracle_gp **)malloc((size_t)K * sizeof(oracle_gp *));
    int part = N / K, start = 0, cur = part;
    for (int k = 0; k < K; k++) {
        if (k == K - 1) cur = N - start;
        b->offset[k] = start;
        b->rows[k] = cur;
        b->experts[k] = oracle_gp_create(cur, D);
        start += part;
    }
    return b;
}

void oracle_bcm_destroy(oracle_bcm *b)
{
    if (!b) return;
    for (int k = 0; k < b->K; k++) oracle_gp_destroy(b->experts[k]);
    free(b->experts); free(b->offset); free(b->rows); free(b);
}

int oracle_bcm_expert_rows(const oracle_bcm *b, int k, int *offset)
{
    if (offset) *offset = b->offset[k];
    return b->rows[k];
}

void oracle_bcm_set_loghyper(oracle_bcm *b, const double hp[3])
{
    for (int i = 0; i < 3; i++) b->hp[i] = hp[i];
    for (int k = 0; k < b->K; k++) oracle_gp_set_loghyper(b->experts[k], b->hp);
}

void oracle_bcm_get_loghyper(const oracle_bcm *b, double hp[3])
{
    for (int i = 0; i < 3; i++) hp[i] = b->hp[i];
}

/* BCM.cpp:182-198: plain sum over experts in index order */
double oracle_bcm_loglik(oracle_bcm *b, double *per_expert)
{
    double ans = 0.0;
    for (int k = 0; k < b->K; k++) {
        double v = oracle_gp_loglik(b->experts[k], b->X + (size_t)b->offset[k] * b->D,
                                    b->y + b->offset[k]);
        ans = ans + v;
        if (per_expert) per_expert[k] = v;
    }
    return ans;
}

/* BCM.cpp:153-180 */
void oracle_bcm_grad(oracle_bcm *b, double g[3])
{
    double acc[3] = { 0, 0, 0 }, gk[3];
    for (int k = 0; k < b->K; k++) {
        oracle_gp_grad(b->experts[k], b->X + (size_t)b->offset[k] * b->D, b->y + b->offset[k], gk);
        for (int i = 0; i < 3; i++) acc[i] = (k == 0) ? gk[i] : acc[i] + gk[i];
    }
    for (int i = 0; i < 3; i++) g[i] = acc[i];
}

/* BCM.cpp:45-62: product of experts, no prior-precision correction.
 * means/vars are K x nt row-major. */
void oracle_poe(const double *means, const double *vars, int K, int nt, double *mean, double *var)
{
    for (int i = 0; i < nt; i++) {
        double tv = 0.0, tm = 0.0;
        for (int e = 0; e < K; e++) {
            double inv = 1.0 / vars[(size_t)e * nt + i];
            tv += inv;
            tm += inv * means[(size_t)e * nt + i];
        }
        tv = 1.0 / tv;
        tm = tv * tm;
        mean[i] = tm;
        var[i] = tv;
    }
}

/* BCM.cpp:64-83 */
void oracle_bcm_predict(oracle_bcm *b, const double *Xt, int nt, double *mean, double *var)
{
    double *m = (double *)malloc((size_t)b->K * nt * sizeof(double));
    double *v = (double *)malloc((size_t)b->K * nt * sizeof(double));
    for (int k = 0; k < b->K; k++)
        oracle_gp_predict(b->experts[k], b->X + (size_t)b->offset[k] * b->D, b->y + b->offset[k],
                          Xt, nt, m + (size_t)k * nt, v + (size_t)k * nt);
    oracle_poe(m, v, b->K, nt, mean, var);
    free(m); free(v);
}

static void bcm_objective(void *c, const double th[3], double *f, double g[3])
{
    oracle_bcm *b = (oracle_bcm *)c;
    oracle_bcm_set_loghyper(b, th);
    *f = -1.0 * oracle_bcm_loglik(b, NULL);
    oracle_bcm_grad(b, g);
}

int oracle_bcm_cg_solve(oracle_bcm *b, int budget, double *trace, int trace_cap)
{
    double th[3] = { b->hp[0], b->hp[1], b->hp[2] };
    int ne = oracle_cg_minimize(bcm_objective, b, th, budget, trace, trace_cap);
    oracle_bcm_set_loghyper(b, th);
    return ne;
}
