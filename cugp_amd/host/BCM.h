// BCM.h -- drop-in for the reference's class BCM (distributed_gp/BCM.h:2-27) over the C-ABI.
// K experts on contiguous row ranges of one dataset (BCM.cpp:85-110).  The default constructor spreads them
// over ALL GPUs the process sees (expert k on GPU k mod G, the reference's chunk i -> worker i mod W,
// cuda_scalingdist/cg_solver.cpp:93): one host thread drives every GPU (cugp_bcm_create_multi), sums are taken in
// expert order, so the numbers do not depend on the GPU count.  One process per GPU with an RCCL all-reduce of
// the same per-expert rows: cugp_amd/bcm.py.
#ifndef CUGP_HOST_BCM_H
#define CUGP_HOST_BCM_H

struct cugp_bcm;

class BCM {
private:
    cugp_bcm *handle;
    int num_experts, dim;
    double log_hyper_bcm[3];

public:
    BCM(double **inp, double *out, int N, int D, int K);
    BCM(double **inp, double *out, int N, int D, int K, int device);                       // one given GPU
    BCM(double **inp, double *out, int N, int D, int K, const int *devices, int ndev);     // a given list of GPUs
    ~BCM();
    BCM(const BCM &) = delete;              // the reference passes BCM by value relying on an empty destructor
    BCM &operator=(const BCM &) = delete;   // (distributed_ver1.cpp:13,285); take it by reference instead

    void set_BCM_log_hyperparam(double *hp);
    void get_BCM_log_hyperparam(double *hp);
    void get_BCM_gradient_hyper(double *g);
    double get_BCM_loglikelihood();
    template <class Vec3>
    void set_BCM_loghyper_eigen(const Vec3 &v)
    {
        double t[3] = {v[0], v[1], v[2]};
        set_BCM_log_hyperparam(t);
    }
    void get_loghyperparam(double *hp);
    void compute_BCM_test_means_and_var(double **Xtest, double *tmeanvec, double *tvarvec, int size);
    double get_BCM_negative_log_predprob(double *actual, double *predmean, double *predvar, int TS);

    cugp_bcm *native() { return handle; }
};

void cg_solve(BCM &pobj);                  // distributed_gp/distributed_ver1.cpp:13

#endif
