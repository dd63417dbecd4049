// group.h -- internal interface between the BCM layer (bcm.cpp) and the evaluation engine (cugp_capi.cpp):
// several experts of equal shape on one device evaluated by ONE sequence of launches (blockIdx.y = expert).
// Not part of the public boundary (include/cugp.h).
#pragma once
#include "../../include/cugp.h"

struct cugp_group;

extern "C" {

// experts must live on one device and agree in padded size and dimension (else CUGP_ERR_INVALID)
int cugp_group_create(cugp_gp* const* experts, int k, cugp_group** out);
void cugp_group_destroy(cugp_group* gr);

// Evaluate all experts at their (common) hyper-parameters; ll[k], g[3k..3k+2] (g may be null when !want_grad).
// Returns CUGP_ERR_INVALID without touching anything when the experts cannot be evaluated as a group right now
// (different hyper-parameters, profiling on, missing data): the caller then evaluates them one by one.
int cugp_group_eval(cugp_group* gr, int want_grad, double* ll, double* g);
// the same in two halves, so the groups of several devices are all in flight before the first result is read
int cugp_group_enqueue(cugp_group* gr, int want_grad);
int cugp_group_fetch(cugp_group* gr, double* ll, double* g);
// device copy of the results of the evaluation in flight ([k][8] doubles: LL, g0, g1, g2, ...) and the stream
// (hipStream_t) they are ordered on -- for a reduction that stays on the device (RCCL all-reduce)
int cugp_group_device_results(cugp_group* gr, const double** dout, void** stream);
// 4 doubles device -> device on `stream` (hipStream_t); the result row of the evaluation a single expert has in flight
int cugp_copy_device_row(double* dst, const double* src, void* stream);
// [count][4] <- {LL, g0, g1, g2} of every [8]-double result row of src (a group's device results), one 2D copy on `stream`
int cugp_pack_result_rows(double* dst, const double* src, int count, void* stream);
// error text for cugp_last_error from the other translation units; returns `code`
int cugp_internal_fail(int code, const char* what);
// halves of cugp_bcm_loglik_grad_allgather (comm.cpp; defined in bcm.cpp)
int cugp_bcm_enqueue_rows_packed(cugp_bcm* b, double* dsend, void** stream);
int cugp_bcm_finish_rows(cugp_bcm* b);
// device copy of a single expert's result row ([8] doubles, valid once its stream -- cugp_get_stream -- has run)
const double* cugp_result_row_device(cugp_gp* gp);
int cugp_copy_result_row(cugp_gp* gp, double* dst);
// the handle holds L^-1, K^-1, alpha for its current data and hyper-parameters (what a prediction needs)
int cugp_has_inverse(const cugp_gp* gp);
// prediction in two halves, so that the experts of a BCM are all in flight before the first result is read: enqueue
// builds the cross-covariance, its product with L^-T and the means / variances on the handle's stream and copies them
// into `host_mv` (2 * nt doubles, PINNED: mean then variance) behind it; fetch waits for that stream.  The handle must
// hold its inverse quantities (cugp_has_inverse), or enqueue evaluates them first like cugp_predict.
int cugp_predict_enqueue(cugp_gp* gp, const double* Xt, int nt, double* host_mv);
int cugp_predict_fetch(cugp_gp* gp);
}  // extern "C"
