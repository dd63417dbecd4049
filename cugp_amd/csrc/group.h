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
}  // extern "C"
