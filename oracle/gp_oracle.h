/*
 * gp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64, single thread) of the reference's serial GP
 * path (cpp_serial_gp/covkernel.cpp + common/matrixops.cpp + distributed_gp/BCM.cpp).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * it, and only as the checker / reported CPU baseline -- never as a product
 * code path.  Parity is pinned: tests/test_oracle_golden.py checks every
 * function here against tests/golden/*.json, which were produced by the
 * reference's own sources compiled unmodified (oracle/Makefile -> oracle/_ref)
 * and against the reference's committed log cuda_bettersinglenode_ver2/REF.
 *
 * All matrices cross this interface as flat row-major arrays; inside, rows are
 * separately allocated (as the reference's `new double[n]` per row) so that the
 * memory behaviour -- and therefore the CPU-baseline timing -- is the same.
 */
#ifndef GP_ORACLE_H
#define GP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_gp oracle_gp;   /* mirrors class Covsum, covkernel.h:3-38 */
typedef struct oracle_bcm oracle_bcm; /* mirrors class BCM, distributed_gp/BCM.h:2-27 */

/* objective callback used by the optimisers: fills f = -LL and g = d(-LL)/dtheta at theta */
typedef void (*oracle_objective_fn)(void *ctx, const double theta[3], double *f, double g[3]);

/* ---- dense LA (common/matrixops.cpp) ---- */
void oracle_get_cholesky(const double *in, double *out, int n);          /* matrixops.cpp:68-108 */
void oracle_chol_and_det(const double *K, const double *y, int n,
                         double *quad, double *logdet);                  /* matrixops.cpp:113-185,232-234 */
void oracle_K_inverse(const double *K, double *out, int n);              /* matrixops.cpp:383-435 */
void oracle_Kinvy(const double *K, const double *y, double *ans, int n); /* matrixops.cpp:264-316 */

/* ---- Covsum (cpp_serial_gp/covkernel.cpp) ---- */
oracle_gp *oracle_gp_create(int n, int d);                               /* covkernel.cpp:14-37 */
void oracle_gp_destroy(oracle_gp *gp);
void oracle_gp_set_loghyper(oracle_gp *gp, const double hp[3]);          /* covkernel.cpp:270-274 */
void oracle_gp_get_loghyper(const oracle_gp *gp, double hp[3]);          /* covkernel.cpp:266-268 */
void oracle_gp_K_train(oracle_gp *gp, const double *X, double *Kout);    /* covkernel.cpp:64-102 */
void oracle_gp_k_test(oracle_gp *gp, const double *X, const double *xt, double *out); /* :105-116 */
void oracle_gp_sqdist(oracle_gp *gp, const double *X, double c, double *Sout);        /* :130-157 */
double oracle_gp_loglik(oracle_gp *gp, const double *X, const double *y);             /* :118-129 */
void oracle_gp_grad(oracle_gp *gp, const double *X, const double *y, double g[3]);    /* :162-263 */
void oracle_gp_predict(oracle_gp *gp, const double *X, const double *y, const double *Xt,
                       int nt, double *mean, double *var);                            /* :277-323 */
double oracle_nlpp(const double *actual, const double *mean, const double *var, int nt); /* :649-659 */

/* ---- optimisers (covkernel.cpp:405-647 cg_solve, :337-402 rprop_solve) ----
 * trace (may be NULL): up to trace_cap rows of 4 doubles [theta0,theta1,theta2,f]
 * -- one row per objective evaluation, in evaluation order.  Returns the
 * number of objective evaluations performed. */
int oracle_cg_minimize(oracle_objective_fn fn, void *ctx, double theta[3], int budget,
                       double *trace, int trace_cap);
int oracle_rprop_minimize(oracle_objective_fn fn, void *ctx, double theta[3], int iters,
                          double *trace, int trace_cap);
int oracle_gp_cg_solve(oracle_gp *gp, const double *X, const double *y, int budget,
                       double *trace, int trace_cap);
int oracle_gp_rprop_solve(oracle_gp *gp, const double *X, const double *y, int iters,
                          double *trace, int trace_cap);

/* ---- BCM / product of experts (distributed_gp/BCM.cpp) ---- */
oracle_bcm *oracle_bcm_create(const double *X, const double *y, int N, int D, int K); /* BCM.cpp:85-110 */
void oracle_bcm_destroy(oracle_bcm *b);
int oracle_bcm_expert_rows(const oracle_bcm *b, int k, int *offset);
void oracle_bcm_set_loghyper(oracle_bcm *b, const double hp[3]);                      /* BCM.cpp:123-130 */
void oracle_bcm_get_loghyper(const oracle_bcm *b, double hp[3]);
double oracle_bcm_loglik(oracle_bcm *b, double *per_expert /* K or NULL */);          /* BCM.cpp:182-198 */
void oracle_bcm_grad(oracle_bcm *b, double g[3]);                                     /* BCM.cpp:153-180 */
void oracle_bcm_predict(oracle_bcm *b, const double *Xt, int nt, double *mean, double *var); /* :64-83 */
void oracle_poe(const double *means, const double *vars, int K, int nt,
                double *mean, double *var);                                           /* BCM.cpp:45-62 */
int oracle_bcm_cg_solve(oracle_bcm *b, int budget, double *trace, int trace_cap);     /* distributed_ver1.cpp:13-232 */

#ifdef __cplusplus
}
#endif
#endif
