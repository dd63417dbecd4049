// bcm_solve.h -- the library's conjugate-gradient solver for a BCM (cugp_bcm_cg_solve: the Rasmussen minimize
// of distributed_gp/distributed_ver1.cpp:13-232 in libcugp, same constants and probe order).  The reference keeps
// this function in its *driver* (`void cg_solve(BCM pobj)`, by value); a driver that brings its own keeps working
// on the drop-in class, one that does not includes this header after BCM.h.
#ifndef CUGP_HOST_BCM_SOLVE_H
#define CUGP_HOST_BCM_SOLVE_H

#include "BCM.h"

void cugp_cg_solve(BCM &pobj);

#endif
