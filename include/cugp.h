/*
 * cugp.h -- C-ABI of the MI355X-native GP-regression hot path (libcugp.so).
 *
 * Drop-in boundary for the reference's GP objective.  Every entry
 * point names the reference interface it replaces (paths relative to the reference
 * checkout).  Two reference surfaces bind here:
 *   A. class Covsum            cpp_serial_gp/covkernel.h:3-38   (cugp_amd/host/covkernel.h wraps this ABI)
 *   B. the free functions over file-scope globals that main.cpp / cg_solver.cpp forward-declare,
 *      cuda_scalingdist/main.cpp:21-53                          (cugp_amd/host/gp_api.h wraps this ABI)
 * plus class BCM, distributed_gp/BCM.h:2-27.
 *
 * Conventions
 *   - plain pointers and sizes only; all arrays fp64; X is row-major n x d, caller-owned HOST memory
 *     unless a function says "device";
 *   - every function returns 0 (CUGP_OK) or a negative code and never exits the process
 *     (the reference exits on CUDA errors, cuda_scalingdist/cuda_gp.cu:98-107);
 *   - a covariance matrix that is not positive definite is NOT an error: results come back NaN,
 *     which the line search relies on (covkernel.cpp:509-524);
 *   - hyper-parameters are the reference's log-hyper vector [log l, log sigma_f, log sigma_n];
 *   - gradients are d(-LL)/d(theta), as Covsum::compute_gradient_loghyperparam returns them;
 *   - a handle may be used by one host thread at a time; all device work of a handle is ordered on
 *     its own HIP stream.
 */
#ifndef CUGP_H
#define CUGP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CUGP_OK 0
#define CUGP_ERR_INVALID (-1)   /* bad argument / call order */
#define CUGP_ERR_NOMEM (-2)     /* host or device allocation failed */
#define CUGP_ERR_DEVICE (-3)    /* HIP runtime error; see cugp_last_error() */
#define CUGP_ERR_NODEVICE (-4)  /* no gfx950 device visible */
#define CUGP_ERR_BUSY (-5)      /* an evaluation of this handle / group is still in flight: fetch it first */

typedef struct cugp_gp cugp_gp;    /* one expert: Covsum / the cuda_gp.cu global state */
typedef struct cugp_bcm cugp_bcm;  /* a set of experts resident on one GPU: class BCM */

int cugp_version(void);
/* hash of the sources this library was built from (cugp_amd/build.py: source_hash); measurements kept as files carry
 * it, so a reader can tell which library they belong to (bench.py: roofline.traffic) */
const char *cugp_build_id(void);
const char *cugp_last_error(void);          /* thread-local text of the last failure */
int cugp_device_count(int *count);

/* ---- lifetime: Covsum::Covsum(n,d) covkernel.cpp:14-37 ; setup(numtrain,dim) cuda_scalingdist/cuda_gp.cu:587 ;
 *      ~Covsum covkernel.cpp:39-61 ; destruct_cublas_cusoler cuda_gp.cu ---- */
int cugp_create(int n, int d, int device, cugp_gp **out);
/* the same with the matrices padded (by identity rows: no result changes) to at least npad_min rows -- the
 * experts of a BCM get one common padded size so that they can share launches (distributed_gp/BCM.cpp:85-110
 * gives the last expert the remainder rows) */
int cugp_create_padded(int n, int d, int device, int npad_min, cugp_gp **out);
int cugp_destroy(cugp_gp *gp);
int cugp_dims(const cugp_gp *gp, int *n, int *d, int *npad);
/* A gradient evaluation builds L^-1 and K^-1 block row by block row on three further streams while the
 * factorisation is still running (its tail leaves most of the chip idle).  On by default; cugp_bcm_create
 * turns it off when several experts share the device.  No reference counterpart (the reference calls
 * cusolverDnDpotrf, then inverts: cuda_scalingdist/cuda_gp.cu:647-708). */
int cugp_set_overlap(cugp_gp *gp, int enable);

/* ---- data: the X,y arguments of every Covsum method ; copy_training_data_to_GPU cuda_gp.cu:510-518 ---- */
int cugp_set_data(cugp_gp *gp, const double *X, const double *y);
int cugp_set_data_device(cugp_gp *gp, const double *dX, const double *dy);   /* device pointers, same device */

/* ---- hyper-parameters: set_loghyperparam / get_loghyperparam / set_loghyper_eigen covkernel.cpp:266-274,325-329 ;
 *      set_loghyper_eigen / get_loghyperparam cuda_gp.cu:960-975 ---- */
int cugp_set_loghyper(cugp_gp *gp, const double hp[3]);
int cugp_get_loghyper(const cugp_gp *gp, double hp[3]);

/* ---- objective ----
 * cugp_loglik       : Covsum::compute_loglikelihood covkernel.cpp:118-129 ; compute_log_likelihood cuda_gp.cu:838-855
 * cugp_loglik_grad  : the pair compute_loglikelihood + compute_gradient_loghyperparam (covkernel.cpp:162-263 ;
 *                     compute_gradient_log_hyperparams cuda_gp.cu:885-957) that cg_solve always calls together,
 *                     from ONE factorisation
 * cugp_grad         : gradient alone (same device work as cugp_loglik_grad) */
int cugp_loglik(cugp_gp *gp, double *ll);
int cugp_loglik_grad(cugp_gp *gp, double *ll, double g[3]);
int cugp_grad(cugp_gp *gp, double g[3]);
/* split form, so several experts can be in flight at once: enqueue on the handle's stream, then fetch */
int cugp_loglik_grad_enqueue(cugp_gp *gp, int want_grad);
int cugp_loglik_grad_fetch(cugp_gp *gp, double *ll, double g[3] /* may be NULL */);
/* y' K^-1 y and log|K| of the last evaluation: compute_chol_and_det matrixops.cpp:232-234 */
int cugp_last_quad_logdet(const cugp_gp *gp, double *quad, double *logdet);

/* ---- prediction: Covsum::compute_test_means_and_variances covkernel.cpp:277-323 (variance includes the
 *      noise term, :316) ; compute_k_test :105-116 ; get_negative_log_predprob :649-659 ; testing_phase
 *      cuda_src/cuda_gp.cu:2063 ---- */
int cugp_predict(cugp_gp *gp, const double *Xt, int nt, double *mean, double *var);
int cugp_nlpp(const double *actual, const double *mean, const double *var, int nt, double *nlpp);

/* ---- intermediates (parity tests; each copies device -> host) ----
 * cugp_compute_K_train : Covsum::compute_K_train covkernel.cpp:64-102 -> full symmetric n x n
 * cugp_compute_squared_dist : Covsum::compute_squared_dist covkernel.cpp:130-157 -> |xi-xj|^2 / c, zero diagonal
 * cugp_compute_k_test  : Covsum::compute_k_test covkernel.cpp:105-116 -> nt x n
 * cugp_get_cholesky    : get_cholesky matrixops.cpp:68-108 -> lower factor of the last evaluation, upper zeroed
 * cugp_get_K_inverse   : compute_K_inverse matrixops.cpp:383-435 -> full symmetric (needs a gradient evaluation)
 * cugp_get_alpha       : vector_Kinvy_using_cholesky matrixops.cpp:264-316 (needs a gradient evaluation) */
int cugp_compute_K_train(cugp_gp *gp, double *K);
int cugp_compute_squared_dist(cugp_gp *gp, double c, double *S);
int cugp_compute_k_test(cugp_gp *gp, const double *Xt, int nt, double *Ks);
int cugp_get_cholesky(cugp_gp *gp, double *L);
int cugp_get_K_inverse(cugp_gp *gp, double *Kinv);
int cugp_get_alpha(cugp_gp *gp, double *alpha);

/* ---- stand-alone dense LA on a caller matrix (common/matrixops.h:5-25) ----
 * cugp_potrf        : get_cholesky            (L lower, upper zeroed)
 * cugp_potri        : compute_K_inverse
 * cugp_chol_and_det : compute_chol_and_det    (y'K^-1y, log|K|)
 * cugp_potrs_vec    : vector_Kinvy_using_cholesky */
int cugp_potrf(int n, const double *K, double *L, int device);
int cugp_potri(int n, const double *K, double *Kinv, int device);
int cugp_chol_and_det(int n, const double *K, const double *y, double *quad, double *logdet, int device);
int cugp_potrs_vec(int n, const double *K, const double *y, double *x, int device);

/* ---- timing: per-phase HIP-event times of the last evaluation (profiling must be on) ----
 * phases: 0 kernel build, 1 Cholesky, 2 triangular inverse, 3 K^-1 product, 4 vectors+traces+finalize, 5 total;
 * cugp_get_kernel_stats: HIP-event time of every launch of the Cholesky trailing update (MFMA SYRK) in the
 * evaluations since the last reset: sum of durations (ms), launches, algorithmic flop */
int cugp_set_profiling(cugp_gp *gp, int level /* 0 off, 1 phases, 2 phases + HIP events around a rotating sample of the
                                                 MFMA kernels' launches, 3 phases + events around every such launch,
                                                 4 phases + every such launch timed by its OWN start / stop events
                                                 (hipExtLaunchKernelGGL; the start event still sits in front of the
                                                 dispatch gap and the evaluation runs 3 % slower), 5 phases + every
                                                 such launch and the covariance build timed by its own workgroups:
                                                 first start / last end on the chip's 100 MHz clock in a device
                                                 buffer -- no events, the untimed schedule */);
int cugp_get_phase_ms(cugp_gp *gp, double ms[6]);
int cugp_get_kernel_stats(cugp_gp *gp, double *sum_ms, long long *launches, double *flop, int reset);
/* the same per kernel, as rocprofv3 names them: kind 0 = k_syrk_step (near-window update + next diagonal block,
 * K = 128; timed one launch in 16, rotating), 1 = k_syrk_wide (the far trailing matrix once per panel, K = 128 * panel
 * width), 2 / 3 = k_trtri_border<4> / <2> (bordering steps of L^-1) and 4 / 5 = k_lauum<4> / <2> (shares of K^-1):
 * timed for every fourth block of inverse rows, 6 / 7 = k_trtri_level<4> / <2> (doubling inside a block of rows; timed
 * one launch in 16), 8 = k_trtri_block (a hand-over block's own inverse in one launch; every fourth block), 9 =
 * k_predict_gemm (W = Ks L^-T of cugp_predict; levels 3 to 5 only), 10 = k_build (level 5 only; its `flop` is BYTES: the
 * lower 64x64 tiles of K written once + X read).  The
 * sampling rates are those of level 2; levels 3 to 5 time every launch.  flop = algorithmic
 * (entries on or below the diagonal, a triangular k tile counted half), multiply + add */
int cugp_get_kernel_stats_kind(cugp_gp *gp, int kind, double *sum_ms, long long *launches, double *flop, int reset);
/* level 5 only: the durations of the same launches counted from the END of the launch directly in front of each on its
 * stream (kind 11 = k_trsm_inv64 -> [k_syrk_wide] -> k_syrk_step on the factorisation's stream), where rocprofv3
 * --kernel-trace puts the begin of an in-order dispatch -- the launch's wait for its first workgroup slot included;
 * launches without a stamped predecessor count from their first workgroup.  Read before a resetting call above. */
int cugp_get_kernel_stats_dispatch_ms(cugp_gp *gp, int kind, double *sum_ms);
void *cugp_get_stream(cugp_gp *gp);          /* hipStream_t of the handle */

/* ---- optimisers (host logic): Covsum::cg_solve covkernel.cpp:405-647 == cg_solve(BCM)
 *      distributed_gp/distributed_ver1.cpp:13-232 == cg_solve(char*) cuda_scalingdist/cg_solver.cpp:292-523 ;
 *      Covsum::rprop_solve covkernel.cpp:337-402 ----
 * fn fills f = -LL and g = d(-LL)/d(theta) at theta (both, at every probe, as the reference does).
 * trace (may be NULL): rows [theta0, theta1, theta2, f] per evaluation.  *nevals <- evaluations made. */
typedef void (*cugp_objective_fn)(void *ctx, const double theta[3], double *f, double g[3]);
int cugp_cg_minimize(cugp_objective_fn fn, void *ctx, double theta[3], int budget, double *trace, int trace_cap,
                     int *nevals);
int cugp_rprop_minimize(cugp_objective_fn fn, void *ctx, double theta[3], int iters, double *trace, int trace_cap,
                        int *nevals);
/* Opt-in, evaluation-sparing form of the same loop: the objective comes in two halves, the value
 * and -- only where the line search can use it -- the gradient at the point of the last value call.  A probe whose
 * value is above the line search's starting value (or NaN/Inf) never has its gradient read by covkernel.cpp:405-647,
 * so it is not computed: on the GPU that probe costs the factorisation only (N^3/3 instead of N^3).  *ngrads <-
 * gradient evaluations made.  The default entry points keep the reference's "both at every probe". */
typedef void (*cugp_value_fn)(void *ctx, const double theta[3], double *f);
typedef void (*cugp_gradient_fn)(void *ctx, const double theta[3], double g[3]);
int cugp_cg_minimize_sparing(cugp_value_fn value, cugp_gradient_fn gradient, void *ctx, double theta[3], int budget,
                             double *trace, int trace_cap, int *nevals, int *ngrads);
int cugp_cg_solve_sparing(cugp_gp *gp, int budget, double *trace, int trace_cap, int *nevals, int *ngrads);
int cugp_cg_solve(cugp_gp *gp, int budget, double *trace, int trace_cap, int *nevals);
int cugp_rprop_solve(cugp_gp *gp, int iters, double *trace, int trace_cap, int *nevals);

/* ---- BCM / product of experts on one GPU: class BCM distributed_gp/BCM.h:2-27 ----
 * The experts given to one cugp_bcm are the ones resident on this process's GPU; their evaluations run
 * concurrently on separate HIP streams.  Sums over the experts of OTHER GPUs are the caller's all-reduce
 * (cugp_amd.bcm does it over RCCL; the reference: cuda_scalingdist/cg_solver.cpp:72-213).
 * cugp_bcm_create_split : BCM::BCM(X,y,N,D,K) BCM.cpp:85-110 (contiguous floor(N/K) rows, remainder to the last)
 * cugp_bcm_loglik_grad  : get_BCM_loglikelihood BCM.cpp:182-198 + get_BCM_gradient_hyper :153-180 -> sums in
 *                         expert order; per_expert_ll may be NULL
 * cugp_bcm_predict_partial : per-expert compute_test_means_and_variances, reduced to the two PoE sums
 *                         sum_k 1/v_k and sum_k mu_k/v_k (BCM.cpp:45-62); cugp_poe_finish turns (all-reduced)
 *                         sums into mean / variance */
int cugp_bcm_create(int nexperts, const int *rows, int d, int device, cugp_bcm **out);
/* the same over several GPUs of ONE process: expert k on devices[k mod ndev] (chunk i -> worker i mod W,
 * cuda_scalingdist/cg_solver.cpp:93), all devices in flight at once, sums in expert order on the host -- the
 * C++ class BCM (cugp_amd/host/BCM.h) uses the whole node through this.  Listing a device twice is allowed. */
int cugp_bcm_create_multi(int ndev, const int *devices, int nexperts, const int *rows, int d, cugp_bcm **out);
int cugp_bcm_create_split_multi(const double *X, const double *y, int N, int D, int K, int ndev, const int *devices,
                                cugp_bcm **out);
int cugp_bcm_create_split(const double *X, const double *y, int N, int D, int K, int device, cugp_bcm **out);
int cugp_bcm_destroy(cugp_bcm *b);
int cugp_bcm_num_experts(const cugp_bcm *b, int *k);
int cugp_bcm_expert(cugp_bcm *b, int k, cugp_gp **gp);
int cugp_bcm_set_expert_data(cugp_bcm *b, int k, const double *X, const double *y);
int cugp_bcm_set_loghyper(cugp_bcm *b, const double hp[3]);      /* BCM::set_BCM_log_hyperparam BCM.cpp:123-130 */
int cugp_bcm_get_loghyper(const cugp_bcm *b, double hp[3]);
int cugp_bcm_loglik_grad(cugp_bcm *b, double *ll, double g[3], double *per_expert_ll);
/* rows[k][4] = {LL_k, dLL_k/dtheta (as gradients of -LL)} per expert of this device: the payload a multi-device
 * BCM sums across devices (the two gathers of cuda_scalingdist/cg_solver.cpp:72-213 in one buffer) */
int cugp_bcm_loglik_grad_rows(cugp_bcm *b, double *rows);
/* the same rows left in DEVICE memory for a collective that stays on the device (RCCL all-reduce): row slot[k] of
 * dev_rows ([.][4] doubles on the handle's device) receives local expert k's {LL, g}; single-device handles */
int cugp_bcm_loglik_grad_rows_device(cugp_bcm *b, double *dev_rows, const int *slot);
/* ---- the same exchange done by the library, one process per GPU (RCCL over xGMI) ----
 * replaces: the master's worker-by-worker collection of log-likelihoods and gradients over TCP,
 * cuda_scalingdist/cg_solver.cpp:72-213 (and the hyper-parameter broadcast :245-279, which is not needed: every rank
 * runs the same deterministic optimiser on identical sums).  Expert k lives on rank k mod W (cg_solver.cpp:93).
 * cugp_comm_unique_id: rank 0 obtains the 128-byte id and hands it to the other ranks by any means (the Python layer
 * broadcasts it through torch.distributed, tests/cpp/rccl_driver.cpp through a file); cugp_comm_create: collective
 * over all ranks (id == NULL with world == 1: no communicator, nothing to exchange).  RCCL is opened at run time
 * (librccl.so.1): CUGP_ERR_NODEVICE when it cannot be.
 * cugp_bcm_loglik_grad_allgather: evaluate this rank's experts (b; NULL on a rank that owns none) and gather
 * everybody's rows: rows_out[world * per][4], rank r's i-th expert (global expert r + i * world) in row r * per + i as
 * {LL, g[3]}, zeros in slots beyond a rank's experts; per >= the largest number of experts on a rank.  Evaluation,
 * ncclAllGather and the copy to the host are one in-order sequence on the evaluation's stream: one host wait. */
typedef struct cugp_comm cugp_comm;
int cugp_comm_unique_id(void *id, int bytes);        /* bytes must be 128 */
int cugp_comm_create(const void *id, int bytes, int rank, int world, int device, cugp_comm **out);
int cugp_comm_destroy(cugp_comm *c);
int cugp_bcm_loglik_grad_allgather(cugp_bcm *b, cugp_comm *c, int per, double *rows_out);
int cugp_bcm_predict_partial(cugp_bcm *b, const double *Xt, int nt, double *sum_prec, double *sum_prec_mean);
int cugp_poe_finish(const double *sum_prec, const double *sum_prec_mean, int nt, double *mean, double *var);
int cugp_bcm_predict(cugp_bcm *b, const double *Xt, int nt, double *mean, double *var); /* BCM.cpp:64-83 */
int cugp_bcm_cg_solve(cugp_bcm *b, int budget, double *trace, int trace_cap, int *nevals);

/* ---- test / bench hooks ---- */
int cugp_test_gemm_nt(int m, int n, int k, const double *A, const double *B, double *C, int device);
int cugp_mfma_peak_tflops(int device, double *tflops);
/* stand-alone LA timings on a device-built SPD matrix (ms, best of reps): op 0 Cholesky (cuda_src/
 * cholesky_cu_solver.cpp), 1 triangular inverse of the factor (tmi_cu_solver.cpp), 2 K^-1 from it, 3 all three,
 * 4 plain C = K K^T with uniform tiles (cublas_matrix_multiply.cpp) */
int cugp_bench_la(int op, int n, int device, int reps, double *ms);
/* the same, and for ops 0 and 3 log|K| taken from the factor the timed launches produced (NaN otherwise) */
int cugp_bench_la_check(int op, int n, int device, int reps, double *ms, double *logdet);
/* launch-shape thresholds (kernels.h TUNE_*), for A/B runs.  cugp_set_tuning changes the PROCESS DEFAULT of a key
 * (thread-safe: a lock); every handle carries its own copy and takes the defaults over when it next starts to enqueue,
 * so an evaluation in flight keeps the shapes it was enqueued with.  cugp_set_handle_tuning sets a key for ONE handle
 * (own != 0: later cugp_set_tuning calls no longer change it; own == 0: back to the default), called by the thread that
 * owns the handle like every other call on it; cugp_get_handle_tuning reads the value the handle's next enqueue uses.
 * (The reference has no counterpart: its launch shapes are compile-time constants, cuda_scalingdist/cuda_gp.cu:20-60.) */
int cugp_set_tuning(int key, int value);
int cugp_set_handle_tuning(cugp_gp *gp, int key, int value, int own);
int cugp_get_handle_tuning(cugp_gp *gp, int key, int *value);
/* the launches of step kb of the two-speed Cholesky with panels of P steps and a near window of about `near_tiles`
 * tiles (pure arithmetic, no device): out = {wide k0, wide k tiles, wide columns [a0,a1), step-launch width in tile
 * columns from kb+1}; tests/test_host_logic.py replays it: every tile sees every k exactly once, ascending */
int cugp_potrf_plan(int nt, int P, int near_tiles, int kb, int out[5]);
/* the same with the near window in sub-panels of S steps (tuning key 17); out[5] = first k tile of the step launch's
 * pass: it subtracts the k tiles [out[5], kb] from the columns [kb+1, kb+1+out[4]) */
int cugp_potrf_plan_sub(int nt, int P, int near_tiles, int S, int kb, int out[6]);

#ifdef __cplusplus
}
#endif
#endif /* CUGP_H */
