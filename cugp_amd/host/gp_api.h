// gp_api.h -- surface B: the free functions over file-scope state that the reference's GPU drivers
// forward-declare (cuda_scalingdist/main.cpp:21-53, cg_solver.cpp:14-40), over the C-ABI.
// The reference's drivers do not include a header for these: they forward-declare them, so every name here is a
// real symbol of libcugp_host.so with the reference's signature.  The one exception is
// `void set_loghyper_eigen(Eigen::VectorXd)` (main.cpp:53,221, cg_solver.cpp:17,262): its mangled name contains
// the maintainer's Eigen, so it lives in gp_api_eigen.cpp, which the maintainer compiles with that Eigen beside
// the drivers; callers without Eigen use set_loghyper(const double[3]) or the template below.
// One GP per process, as in the reference; not re-entrant.
#ifndef CUGP_HOST_GP_API_H
#define CUGP_HOST_GP_API_H

#include <string>

extern double *X_host, *labels_host;                         // cuda_gp.cu:25,27 (setup allocates N x DIM and N doubles)
extern double *X_host_buffers[2], *labels_host_buffers[2];   // cuda_gp.cu:26,28 (the background reader's two buffers)

void setup(int numtrain, int dimensions);                                        // cuda_gp.cu:587
void setup(int numtrain, std::string inputfile, std::string labelfile);          // cuda_src/cuda_gp.cu (older drivers)
void read_trainingdata_into_dram(std::string inputfile, std::string labelfile, double *X_cur, double *labels_cur);
void copy_training_data_to_GPU(double *X_cur, double *labels_cur);               // cuda_gp.cu:510-518
void read_trainingdata_and_copy_to_GPU(std::string inputfile, std::string labelfile);
double compute_log_likelihood();                                                 // cuda_gp.cu:838-855
void compute_gradient_log_hyperparams(double *localhp_grad);                     // cuda_gp.cu:885-957
double *get_loghyperparam();                                                     // cuda_gp.cu:960-963
void set_loghyper(const double hp[3]);
template <class Vec3>
void set_loghyper_eigen(const Vec3 &v)                                           // cuda_gp.cu:965-975
{
    const double t[3] = {v[0], v[1], v[2]};
    set_loghyper(t);
}
void cg_solve(char *hostname);                                                   // cg_solver.cpp:292 (single process)
void testing_phase(int offset, int numtest);                                     // cuda_src/cuda_gp.cu:2063
void destruct_cublas_cusoler();                                                  // releases the device state

// text data files of the reference: "N D" header line, then rows; labels one per line.  Header counts are
// not trusted (several shipped files disagree with their header): rows are counted from the data.
int read_matrix_file(const std::string &path, int dim, double **out, int *rows);

#endif
