// gp_api_eigen.cpp -- the one symbol of surface B whose mangled name depends on the caller's Eigen:
//     void set_loghyper_eigen(Eigen::VectorXd initval)       (cuda_scalingdist/cuda_gp.cu:965-975,
//                                                              forward-declared at main.cpp:53 and cg_solver.cpp:17)
// Compile it beside the reference's drivers with the same Eigen they include:
//     g++ -I<dir that holds Eigen/> -DCUGP_EIGEN_DENSE='"Eigen/Dense"' -c gp_api_eigen.cpp
// Not part of libcugp_host.so (the library itself has no Eigen dependency).
#ifndef CUGP_EIGEN_DENSE
#define CUGP_EIGEN_DENSE "../cuda_src/Eigen/Dense"      /* what main.cpp:2 and cg_solver.cpp:5 include */
#endif
#include CUGP_EIGEN_DENSE

void set_loghyper(const double hp[3]);                  // libcugp_host.so (gp_api.h)

void set_loghyper_eigen(Eigen::VectorXd initval)
{
    const double t[3] = {initval[0], initval[1], initval[2]};
    set_loghyper(t);
}
