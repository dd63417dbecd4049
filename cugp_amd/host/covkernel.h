// covkernel.h -- drop-in for the reference's class Covsum (cpp_serial_gp/covkernel.h:3-38,
// distributed_gp/covkernel.h), implemented over the MI355X C-ABI (include/cugp.h).
//
// Same public methods, argument meaning and ownership as the reference: X is an array of row
// pointers, y a plain array, both caller-owned; compute_gradient_loghyperparam returns a pointer to
// a function-static double[3] (covkernel.cpp:167); get_loghyperparam returns an internal pointer.
// The header is C++98-clean: the reference's drivers only compile as -std=gnu++98 (unqualified isnan/isinf,
// distributed_ver1.cpp:98) and include this file under its reference name.
// Differences a caller can see:
//   * Eigen is optional: when the including translation unit can see the Eigen the reference's own header
//     includes ("./Eigen/Dense" in cpp_serial_gp/covkernel.h:2, "eigen3/Eigen/Dense" in distributed_gp/covkernel.h:2)
//     it is included here too -- the reference's drivers get Eigen::VectorXd through covkernel.h
//     (distributed_ver1.cpp:36) -- and set_loghyper_eigen(Eigen::VectorXd) is the reference's signature; otherwise
//     set_loghyper_eigen is a template over anything indexable with [0..2].  -DCUGP_HOST_NO_EIGEN switches the
//     detection off;
//   * compute_loglikelihood followed by compute_gradient_loghyperparam at the same hyper-parameters
//     (the order cg_solve uses, covkernel.cpp:500-501) costs ONE factorisation on the GPU, not three;
//   * compute_squared_dist fills an internal buffer exactly like the reference (tempmatrix2 is
//     private there); squared_dist() exposes it for tests;
//   * resource failures throw std::runtime_error (the reference has no error path); a covariance
//     that is not positive definite still comes back as NaN.
#ifndef CUGP_HOST_COVKERNEL_H
#define CUGP_HOST_COVKERNEL_H

#include <vector>

#if !defined(CUGP_HOST_NO_EIGEN) && !defined(CUGP_HOST_HAVE_EIGEN) && defined(__has_include)
#if __has_include("eigen3/Eigen/Dense")
#include "eigen3/Eigen/Dense"
#define CUGP_HOST_HAVE_EIGEN 1
#elif __has_include("./Eigen/Dense")
#include "./Eigen/Dense"
#define CUGP_HOST_HAVE_EIGEN 1
#elif __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define CUGP_HOST_HAVE_EIGEN 1
#endif
#endif

struct cugp_gp;

class Covsum {
private:
    cugp_gp *handle;
    int inputdatasize;            // number of training examples
    int numdim;                   // dimensionality of the problem
    double loghyper[3];
    std::vector<double> xflat, ycopy, sqdist;   // last uploaded data; |xi-xj|^2/c of compute_squared_dist
    bool have_data;
    int device;

    void bind(double **X, double *y);
    Covsum(const Covsum &);                  // not copyable (the reference never copies one either:
    Covsum &operator=(const Covsum &);       // BCM holds std::vector<Covsum *>, BCM.h:7)

public:
    Covsum();
    Covsum(int n, int d);
    Covsum(int n, int d, int device);
    ~Covsum();

    double compute_loglikelihood(double **X, double *y);
    double *compute_gradient_loghyperparam(double **X, double *y);
    void compute_K_train(double **X, double **output);
    void compute_k_test(double **X, double *xtest, double *output);
    void compute_squared_dist(double **X, double c);
    const std::vector<double> &squared_dist() const { return sqdist; }
    double *get_loghyperparam();
    void set_loghyperparam(double *initval);
    void set_loghyperparam(const double *initval);

    void compute_test_means_and_variances(double **X, double *y, double **Xtest, double *tmeanvec, double *tvarvec,
                                          int numtest);
#ifdef CUGP_HOST_HAVE_EIGEN
    void set_loghyper_eigen(Eigen::VectorXd initval)          // covkernel.h:33, covkernel.cpp:325-329
    {
        const double t[3] = {initval[0], initval[1], initval[2]};
        set_loghyperparam(t);
    }
#endif
    template <class Vec3>
    void set_loghyper_eigen(const Vec3 &v)
    {
        const double t[3] = {v[0], v[1], v[2]};
        set_loghyperparam(t);
    }
    void cg_solve(double **X, double *y, bool verbose = true);
    void rprop_solve(double **X, double *y, bool verbose = true);
    double get_negative_log_predprob(double *actual, double *predmean, double *predvar, int TS);
    int get_param_dim();

    cugp_gp *native() { return handle; }
};

#endif
