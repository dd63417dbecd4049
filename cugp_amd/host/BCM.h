// BCM.h -- drop-in for the reference's class BCM (distributed_gp/BCM.h:2-27) over the C-ABI.
// K experts on contiguous row ranges of one dataset (BCM.cpp:85-110).  The default constructor spreads them
// over ALL GPUs the process sees (expert k on GPU k mod G, the reference's chunk i -> worker i mod W,
// cuda_scalingdist/cg_solver.cpp:93): one host thread drives every GPU (cugp_bcm_create_multi), sums are taken in
// expert order, so the numbers do not depend on the GPU count.  One process per GPU with an RCCL all-reduce of
// the same per-expert rows: cugp_amd/bcm.py.
//
// Like the reference's header it expects covkernel.h to be included first (distributed_ver1.cpp:6,8) and is
// C++98-clean.  The reference's driver defines `void cg_solve(BCM pobj)` itself and passes the object BY VALUE
// (distributed_ver1.cpp:13,285; it works there because ~BCM is empty, BCM.cpp:112-122): copies of this class
// therefore share one device-side model (reference-counted), and hyper-parameters set through a copy are seen by
// every copy, which is what the reference's shallow copy does too (log_hyper_bcm is a shared pointer there).
// The library's own solver for a BCM is declared in bcm_solve.h (not here: an overload taking BCM& would make the
// reference driver's call cg_solve(poe) ambiguous).
#ifndef CUGP_HOST_BCM_H
#define CUGP_HOST_BCM_H

struct cugp_bcm;

class BCM {
private:
    struct Shared;                          // handle + hyper-parameters + reference count
    Shared *s;
    int num_experts, dim;

    void init(double **inp, double *out, int N, int D, int K, const int *devices, int ndev);

public:
    BCM(double **inp, double *out, int N, int D, int K);
    BCM(double **inp, double *out, int N, int D, int K, int device);                       // one given GPU
    BCM(double **inp, double *out, int N, int D, int K, const int *devices, int ndev);     // a given list of GPUs
    BCM(const BCM &o);
    BCM &operator=(const BCM &o);
    ~BCM();

    void set_BCM_log_hyperparam(double *hp);
    void get_BCM_log_hyperparam(double *hp);
    void get_BCM_gradient_hyper(double *g);
    double get_BCM_loglikelihood();
#ifdef CUGP_HOST_HAVE_EIGEN
    void set_BCM_loghyper_eigen(Eigen::VectorXd initval)       // BCM.h:22, BCM.cpp:123-130
    {
        double t[3] = {initval[0], initval[1], initval[2]};
        set_BCM_log_hyperparam(t);
    }
#endif
    template <class Vec3>
    void set_BCM_loghyper_eigen(const Vec3 &v)
    {
        double t[3] = {v[0], v[1], v[2]};
        set_BCM_log_hyperparam(t);
    }
    void get_loghyperparam(double *hp);
    void compute_BCM_test_means_and_var(double **Xtest, double *tmeanvec, double *tvarvec, int size);
    double get_BCM_negative_log_predprob(double *actual, double *predmean, double *predvar, int TS);

    cugp_bcm *native();
};

#endif
