// gp_api.cpp -- surface B over the C-ABI (see gp_api.h).
#include "gp_api.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/cugp.h"

int numtrain = 0, dimensions = 0;
double *X_host = nullptr, *labels_host = nullptr;

namespace {
cugp_gp *g_gp = nullptr;
double g_lh[3] = {0, 0, 0};
int g_rows_in_file = 0;

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
    numtrain = n;
    dimensions = d;
    must(cugp_create(n, d, 0, &g_gp), "setup");
}

void read_trainingdata_into_dram(std::string inputfile, std::string labelfile, double *X_cur, double *labels_cur)
{
    // cuda_gp.cu:477-508: skip the header, read numtrain x dimensions values and numtrain labels
    FILE *fi = fopen(inputfile.c_str(), "r"), *fl = fopen(labelfile.c_str(), "r");
    if (!fi || !fl) throw std::runtime_error("Open input file failed: " + inputfile + " / " + labelfile);
    int t1, t2;
    if (fscanf(fi, "%d%d", &t1, &t2) != 2) throw std::runtime_error("bad header in " + inputfile);
    for (int i = 0; i < numtrain * dimensions; i++)
        if (fscanf(fi, "%lf", &X_cur[i]) != 1) throw std::runtime_error("short input file " + inputfile);
    for (int i = 0; i < numtrain; i++)
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
    // keeps every row of the file on the host so testing_phase can address rows past numtrain
    free(X_host);
    free(labels_host);
    int rows = 0, lrows = 0;
    if (read_matrix_file(inputfile, dimensions, &X_host, &rows) != 0) throw std::runtime_error("cannot read " + inputfile);
    FILE *fl = fopen(labelfile.c_str(), "r");
    if (!fl) throw std::runtime_error("cannot read " + labelfile);
    std::vector<double> lab;
    double x;
    while (fscanf(fl, "%lf", &x) == 1) lab.push_back(x);
    fclose(fl);
    lrows = (int)lab.size();
    if (rows < numtrain || lrows < numtrain) throw std::runtime_error("file shorter than numtrain: " + inputfile);
    labels_host = (double *)malloc(lab.size() * sizeof(double));
    memcpy(labels_host, lab.data(), lab.size() * sizeof(double));
    g_rows_in_file = rows < lrows ? rows : lrows;
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
    if (!X_host || offset + numtest > g_rows_in_file) throw std::runtime_error("testing_phase: rows not loaded");
    std::vector<double> m(numtest), v(numtest);
    must(cugp_predict(g_gp, X_host + (size_t)offset * dimensions, numtest, m.data(), v.data()), "testing_phase");
    double nlpp = 0;
    must(cugp_nlpp(labels_host + offset, m.data(), v.data(), numtest, &nlpp), "cugp_nlpp");
    printf("NLPP = %.12g\n", nlpp);
}

void destruct_cublas_cusoler()
{
    if (g_gp) cugp_destroy(g_gp);
    g_gp = nullptr;
}
