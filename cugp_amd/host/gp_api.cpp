// gp_api.cpp -- surface B over the C-ABI (see gp_api.h).
#include "gp_api.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/cugp.h"

// the globals cuda_gp.cu defines and the drivers declare extern (cuda_scalingdist/cuda_gp.cu:25-28,
// main.cpp:61-66, cg_solver.cpp:31-34).  numtrain / dimensions / numchunks / worker_id / ... are the DRIVER's
// (main.cpp:14-16,40,55-59); this file keeps its own copies of the two sizes, like cuda_gp.cu's N and DIM.
double *X_host = nullptr, *labels_host = nullptr;
double *X_host_buffers[2] = {nullptr, nullptr}, *labels_host_buffers[2] = {nullptr, nullptr};

namespace {
cugp_gp *g_gp = nullptr;
double g_lh[3] = {0.5, 0.5, 0.5};            // cuda_gp.cu:437-440
int g_n = 0, g_d = 0;                        // cuda_gp.cu's N, DIM
double *g_owned[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // what setup allocated (the drivers re-point X_host)
std::vector<double> g_file_X, g_file_y;      // every row of the last file read whole (testing_phase addresses rows past N)

void must(int rc, const char *what)
{
    if (rc != CUGP_OK) throw std::runtime_error(std::string(what) + ": " + cugp_last_error());
}
}  // namespace

int read_matrix_file(const std::string &path, int dim, double **out, int *rows)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int h1 = 0, h2 = 0;
    if (fscanf(f, "%d%d", &h1, &h2) != 2) { fclose(f); return -2; }
    const int d = dim > 0 ? dim : h2;
    std::vector<double> v;
    double x;
    while (fscanf(f, "%lf", &x) == 1) v.push_back(x);
    fclose(f);
    const int n = (int)(v.size() / d);
    *out = (double *)malloc((size_t)n * d * sizeof(double));
    memcpy(*out, v.data(), (size_t)n * d * sizeof(double));
    *rows = n;
    return 0;
}

void setup(int n, int d)
{
    if (g_gp) cugp_destroy(g_gp);
    g_gp = nullptr;
    g_n = n;
    g_d = d;
    // setup_input_datastructures (cuda_gp.cu:423-448): the drivers read chunk files into these (background_reader,
    // cg_solver.cpp:42-70) and point X_host / labels_host at one of the two buffers (main.cpp:112-113)
    for (int i = 0; i < 6; i++) delete[] g_owned[i];
    X_host = g_owned[0] = new double[(size_t)n * d];
    labels_host = g_owned[1] = new double[n];
    for (int b = 0; b < 2; b++) {
        X_host_buffers[b] = g_owned[2 + b] = new double[(size_t)n * d];
        labels_host_buffers[b] = g_owned[4 + b] = new double[n];
    }
    must(cugp_create(n, d, 0, &g_gp), "setup");
    // setup_input_datastructures resets lh_host to 0.5 on EVERY call (cuda_gp.cu:437-440): a second model set up in one
    // process starts from the reference's initial hyper-parameters, not from what the previous model ended at
    g_lh[0] = g_lh[1] = g_lh[2] = 0.5;
    must(cugp_set_loghyper(g_gp, g_lh), "cugp_set_loghyper");
}

void read_trainingdata_into_dram(std::string inputfile, std::string labelfile, double *X_cur, double *labels_cur)
{
    // cuda_gp.cu:477-508: skip the header, read numtrain x dimensions values and numtrain labels
    FILE *fi = fopen(inputfile.c_str(), "r"), *fl = fopen(labelfile.c_str(), "r");
    if (!fi || !fl) throw std::runtime_error("Open input file failed: " + inputfile + " / " + labelfile);
    int t1, t2;
    if (fscanf(fi, "%d%d", &t1, &t2) != 2) throw std::runtime_error("bad header in " + inputfile);
    for (int i = 0; i < g_n * g_d; i++)
        if (fscanf(fi, "%lf", &X_cur[i]) != 1) throw std::runtime_error("short input file " + inputfile);
    for (int i = 0; i < g_n; i++)
        if (fscanf(fl, "%lf", &labels_cur[i]) != 1) throw std::runtime_error("short label file " + labelfile);
    fclose(fi);
    fclose(fl);
}

void copy_training_data_to_GPU(double *X_cur, double *labels_cur)
{
    must(cugp_set_data(g_gp, X_cur, labels_cur), "copy_training_data_to_GPU");
    must(cugp_set_loghyper(g_gp, g_lh), "cugp_set_loghyper");
}

void read_trainingdata_and_copy_to_GPU(std::string inputfile, std::string labelfile)
{
    // cuda_gp.cu:450-475: the first N rows go into X_host / labels_host and on to the device.  Every row of the
    // file is kept beside them so testing_phase can address rows past N (cuda_src/cuda_gp.cu:1992-2003).
    double *all = nullptr;
    int rows = 0;
    if (read_matrix_file(inputfile, g_d, &all, &rows) != 0) throw std::runtime_error("cannot read " + inputfile);
    g_file_X.assign(all, all + (size_t)rows * g_d);
    free(all);
    FILE *fl = fopen(labelfile.c_str(), "r");
    if (!fl) throw std::runtime_error("cannot read " + labelfile);
    g_file_y.clear();
    double x;
    while (fscanf(fl, "%lf", &x) == 1) g_file_y.push_back(x);
    fclose(fl);
    if (rows < g_n || (int)g_file_y.size() < g_n) throw std::runtime_error("file shorter than numtrain: " + inputfile);
    if (!X_host || !labels_host) throw std::runtime_error("read_trainingdata_and_copy_to_GPU before setup");
    memcpy(X_host, g_file_X.data(), (size_t)g_n * g_d * sizeof(double));
    memcpy(labels_host, g_file_y.data(), (size_t)g_n * sizeof(double));
    copy_training_data_to_GPU(X_host, labels_host);
}

void setup(int n, std::string inputfile, std::string labelfile)
{
    FILE *f = fopen(inputfile.c_str(), "r");
    int h1 = 0, h2 = 0;
    if (!f || fscanf(f, "%d%d", &h1, &h2) != 2) throw std::runtime_error("cannot read " + inputfile);
    fclose(f);
    setup(n, h2);
    read_trainingdata_and_copy_to_GPU(inputfile, labelfile);
}

double compute_log_likelihood()
{
    double ll = 0, g[3];
    must(cugp_loglik_grad(g_gp, &ll, g), "compute_log_likelihood");   // the gradient call that follows is free
    return ll;
}

void compute_gradient_log_hyperparams(double *localhp_grad) { must(cugp_grad(g_gp, localhp_grad), "compute_gradient_log_hyperparams"); }

double *get_loghyperparam()
{
    must(cugp_get_loghyper(g_gp, g_lh), "get_loghyperparam");
    return g_lh;
}

void set_loghyper(const double hp[3])
{
    for (int i = 0; i < 3; i++) g_lh[i] = hp[i];
    must(cugp_set_loghyper(g_gp, g_lh), "set_loghyper");
}

void cg_solve(char *)
{
    int nev = 0;
    must(cugp_cg_solve(g_gp, 100, nullptr, 0, &nev), "cg_solve");
    get_loghyperparam();
    printf("\n\n PLEASE-SEE 3 : %lf, %lf, %lf\n\n", g_lh[0], g_lh[1], g_lh[2]);
}

void testing_phase(int offset, int numtest)
{
    // cuda_src/cuda_gp.cu:1992-2061: test rows are rows [offset, offset+numtest) of the same file
    const int rows = (int)std::min(g_file_X.size() / (size_t)(g_d > 0 ? g_d : 1), g_file_y.size());
    if (offset < 0 || numtest <= 0 || offset + numtest > rows) throw std::runtime_error("testing_phase: rows not loaded");
    std::vector<double> m(numtest), v(numtest);
    must(cugp_predict(g_gp, g_file_X.data() + (size_t)offset * g_d, numtest, m.data(), v.data()), "testing_phase");
    double nlpp = 0;
    must(cugp_nlpp(g_file_y.data() + offset, m.data(), v.data(), numtest, &nlpp), "cugp_nlpp");
    printf("NLPP = %.12g\n", nlpp);
}

void destruct_cublas_cusoler()
{
    if (g_gp) cugp_destroy(g_gp);
    g_gp = nullptr;
}
