// dropin_driver.cpp -- exercises the C++ drop-in layer the way the reference's drivers use their
// classes (cpp_serial_gp/serial_gp.cpp:33-72, distributed_gp/distributed_ver1.cpp:240-285,
// cuda_scalingdist/main.cpp:290-305) and prints one JSON object for the pytest side to check.
//   dropin_driver <input.txt> <labels.txt> <numtrain> <numexperts>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../cugp_amd/host/covkernel.h"
#include "../../cugp_amd/host/BCM.h"
#include "../../cugp_amd/host/bcm_solve.h"
#include "../../cugp_amd/host/gp_api.h"

// the reference's driver takes its BCM BY VALUE (distributed_gp/distributed_ver1.cpp:13,285): copies of the drop-in
// class share one device-side model
static double ll_by_value(BCM pobj, double *hp3)
{
    pobj.set_BCM_log_hyperparam(hp3);                    // seen by the caller's object too, as with the reference's shallow copy
    return pobj.get_BCM_loglikelihood();
}

static void pv(const char *k, const double *v, int n, bool last = false)
{
    printf("\"%s\": [", k);
    for (int i = 0; i < n; i++) printf("%.17g%s", v[i], i + 1 < n ? ", " : "");
    printf("]%s\n", last ? "" : ",");
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int ntrain = atoi(argv[3]), K = atoi(argv[4]);
    FILE *fi = fopen(argv[1], "r"), *fl = fopen(argv[2], "r");
    if (!fi || !fl) return 3;
    int n, dim;
    if (fscanf(fi, "%d%d", &n, &dim) != 2) return 4;
    double **X = new double *[n];
    double *y = new double[n];
    for (int i = 0; i < n; i++) {
        X[i] = new double[dim];
        for (int j = 0; j < dim; j++)
            if (fscanf(fi, "%lf", &X[i][j]) != 1) return 5;
    }
    for (int i = 0; i < n; i++)
        if (fscanf(fl, "%lf", &y[i]) != 1) return 6;
    fclose(fi);
    fclose(fl);
    const int ntest = n - ntrain;

    printf("{\n");
    {   // ---- surface A: class Covsum, as serial_gp.cpp uses it ----
        double inithypervalues[] = {1.5, 1.5, 1.5};
        Covsum kernelobj(ntrain, dim);
        kernelobj.set_loghyperparam(inithypervalues);
        double ll = kernelobj.compute_loglikelihood(X, y);
        double *grad = kernelobj.compute_gradient_loghyperparam(X, y);
        printf("\"ll\": %.17g,\n", ll);
        pv("grad", grad, 3);
        std::vector<double> tm(ntest), tv(ntest);
        kernelobj.compute_test_means_and_variances(X, y, X + ntrain, tm.data(), tv.data(), ntest);
        pv("pred_mean", tm.data(), ntest);
        pv("pred_var", tv.data(), ntest);
        printf("\"nlpp\": %.17g,\n", kernelobj.get_negative_log_predprob(y + ntrain, tm.data(), tv.data(), ntest));
        kernelobj.cg_solve(X, y, false);
        pv("cg_final_hp", kernelobj.get_loghyperparam(), 3);
        printf("\"cg_final_ll\": %.17g,\n", kernelobj.compute_loglikelihood(X, y));
        std::vector<double *> Krows(ntrain);
        std::vector<double> Kbuf((size_t)ntrain * ntrain);
        for (int i = 0; i < ntrain; i++) Krows[i] = &Kbuf[(size_t)i * ntrain];
        kernelobj.compute_K_train(X, Krows.data());
        pv("K_row5", Krows[5], ntrain);
        printf("\"param_dim\": %d,\n", kernelobj.get_param_dim());
        // the second optimiser of the class (serial_gp.cpp:70 keeps the call commented out beside cg_solve)
        kernelobj.set_loghyperparam(inithypervalues);
        kernelobj.rprop_solve(X, y, false);
        pv("rprop_final_hp", kernelobj.get_loghyperparam(), 3);
        printf("\"rprop_final_ll\": %.17g,\n", kernelobj.compute_loglikelihood(X, y));
        kernelobj.compute_squared_dist(X, 2.5);
        pv("sqdist_row3", kernelobj.squared_dist().data() + (size_t)3 * ntrain, ntrain);
    }
    {   // ---- class BCM, as distributed_ver1.cpp uses it ----
        double inithypervalues[] = {1.5, 1.5, 1.5};
        BCM poe(X, y, ntrain, dim, K);
        poe.set_BCM_log_hyperparam(inithypervalues);
        printf("\"bcm_ll\": %.17g,\n", poe.get_BCM_loglikelihood());
        double g[3];
        poe.get_BCM_gradient_hyper(g);
        pv("bcm_grad", g, 3);
        std::vector<double> tm(ntest), tv(ntest);
        poe.compute_BCM_test_means_and_var(X + ntrain, tm.data(), tv.data(), ntest);
        pv("bcm_pred_mean", tm.data(), ntest);
        pv("bcm_pred_var", tv.data(), ntest);
        {
            double hp2[] = {1.25, 1.0, 0.5};
            const double llv = ll_by_value(poe, hp2);     // copy made and destroyed: the model must survive it
            double seen[3];
            poe.get_loghyperparam(seen);
            pv("bcm_byvalue_hp_seen", seen, 3);
            printf("\"bcm_byvalue_ll\": %.17g,\n\"bcm_after_copy_ll\": %.17g,\n", llv, poe.get_BCM_loglikelihood());
            BCM second = poe;                             // copy construction + assignment
            BCM third(X, y, 64, dim, 2);
            third = second;
            printf("\"bcm_assigned_ll\": %.17g,\n", third.get_BCM_loglikelihood());
            poe.set_BCM_log_hyperparam(inithypervalues);
        }
        cugp_cg_solve(poe);
        double hp[3];
        poe.get_loghyperparam(hp);
        pv("bcm_cg_final_hp", hp, 3);
    }
    {   // ---- surface B: free functions, as cuda_scalingdist/main.cpp uses them ----
        setup(ntrain, dim);
        double init[3] = {1.5, 1.5, 1.5};
        set_loghyper_eigen(init);
        read_trainingdata_and_copy_to_GPU(argv[1], argv[2]);
        printf("\"api_ll\": %.17g,\n", compute_log_likelihood());
        double g[3];
        compute_gradient_log_hyperparams(g);
        pv("api_grad", g, 3);
        // the multi-node drivers' path (cuda_scalingdist/main.cpp:99-121, cg_solver.cpp:42-70): a chunk read into one of
        // the two host buffers setup() allocated, X_host / labels_host pointed at it, copied to the device
        read_trainingdata_into_dram(argv[1], argv[2], X_host_buffers[1], labels_host_buffers[1]);
        X_host = X_host_buffers[1];
        labels_host = labels_host_buffers[1];
        copy_training_data_to_GPU(X_host, labels_host);
        printf("\"api_ll_from_buffer\": %.17g,\n", compute_log_likelihood());
        cg_solve(argv[0]);
        pv("api_cg_final_hp", get_loghyperparam(), 3, true);
        printf("}\n");
        // cuda_src/main.cpp:197: rows [numtrain, numtrain+numtest) of the same file are the test set; prints NLPP
        set_loghyper_eigen(init);
        testing_phase(ntrain, ntest);
        destruct_cublas_cusoler();
    }
    return 0;
}
