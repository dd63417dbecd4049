// ozaki_probe.hip -- feasibility probe, NOT part of the product: an fp64-accurate NT product C = A * B^T computed as
// EXACT int8 slice products on the int8 matrix cores (Ozaki scheme I): every operand row is scaled by a power of two
// and cut into 8 signed slices of 6 / 7 bits (55 bits in all), the 36 slice pairs (p, q) with p + q <= 7 are multiplied
// by v_mfma_i32_32x32x32_i8 with exact int32 accumulation (one accumulator per p + q), and the accumulators are
// recombined in fp64.  MI355X has 64x the int8 matrix rate of its fp64 matrix rate; 36 int8 MFMAs of 32 cycles do the
// work of 32 fp64 MFMAs of 64 cycles: 1.78x at the issue level.  The probe measures what is left of that in a tile
// kernel (operand traffic per flop is 1.5x the fp64 kernel's: 32x32 outputs per wave) and the error against a
// long-double reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ozaki_probe.hip -o tools/bin/ozaki_probe
//   tools/bin/ozaki_probe [m=8192] [n=8192] [k=2048]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int NS = 8;                 // slices per operand
constexpr int KS = 32;                // k bytes per stage = one MFMA's K

// ---- scale: 2^e with |a| / 2^e < 1 for every a of the row (one wave per row) ----
__global__ __launch_bounds__(256) void k_rowscale(const double* __restrict__ A, int lda, int rows, int K, double* __restrict__ scale)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const double* a = A + (size_t)row * lda;
    double m = 0.0;
    for (int k = lane * 2; k < K; k += 128) {
        const d2 v = *(const d2*)(a + k);
        m = fmax(m, fmax(fabs(v[0]), fabs(v[1])));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    if (lane == 0) {
        int e = 0;
        if (m > 0.0) (void)frexp(m, &e);                 // m = f * 2^e, f in [0.5, 1)
        scale[row] = ldexp(1.0, e);
    }
}

// ---- slices: S[p][row][k] (int8), thread = 16 consecutive k of one row ----
__global__ __launch_bounds__(256) void k_slice(const double* __restrict__ A, int lda, int rows, int K, const double* __restrict__ scale,
                                               signed char* __restrict__ S)
{
    const int chunks = K / 16;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)rows * chunks) return;
    const int row = (int)(id / chunks), c = (int)(id % chunks);
    const double inv = 1.0 / scale[row];                  // exact: a power of two
    const double* a = A + (size_t)row * lda + c * 16;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const d2 v = *(const d2*)(a + i);
        x[i] = v[0] * inv * 64.0;                         // |x| < 64
        x[i + 1] = v[1] * inv * 64.0;
    }
    const size_t plane = (size_t)rows * K;
#pragma unroll
    for (int p = 0; p < NS; p++) {
        int w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const double s = rint(x[i]);                  // |s| <= 64
            x[i] = (x[i] - s) * 128.0;                    // exact; |x| <= 64 again
            w[i >> 2] |= ((int)s & 0xff) << (8 * (i & 3));
        }
        *(v4i*)(S + p * plane + (size_t)row * K + c * 16) = (v4i){w[0], w[1], w[2], w[3]};
    }
}

// ---- product: one workgroup = 64 x 64 outputs, 4 waves of 32 x 32; K staged 32 bytes deep through LDS ----
// LDS per stage and operand: [plane][row][32 bytes]; the two 16-byte halves of a row are swapped on odd (row >> 3) so
// that the 16 lanes of a ds_read_b128 group touch 16 different 16-byte slots.
constexpr int OPER = NS * 64 * KS;            // 16 KB
constexpr int STAGE = 2 * OPER;               // A + B
constexpr int GEMM_LDS = 2 * STAGE;           // double buffered: 64 KB

__device__ __forceinline__ int swz(int row, int half) { return row * KS + ((half ^ ((row >> 3) & 1)) << 4); }

__global__ __launch_bounds__(256, 2) void k_gemm_i8(const signed char* __restrict__ SA, const double* __restrict__ sa, int rowsA,
                                                    const signed char* __restrict__ SB, const double* __restrict__ sb, int rowsB,
                                                    int K, double* __restrict__ C, int ldc, int tiles_m)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
    const int ti = blockIdx.x % tiles_m, tj = blockIdx.x / tiles_m;
    const int i0 = ti * 64, j0 = tj * 64;
    const size_t planeA = (size_t)rowsA * K, planeB = (size_t)rowsB * K;
    // staging: per operand 8 planes x 64 rows x 2 halves = 1024 chunks of 16 B: 4 per thread
    const signed char* ga[4];
    const signed char* gb[4];
    int la[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int ch = t + 256 * q, p = ch >> 7, row = (ch >> 1) & 63, half = ch & 1;
        ga[q] = SA + p * planeA + (size_t)(i0 + row) * K + half * 16;
        gb[q] = SB + p * planeB + (size_t)(j0 + row) * K + half * 16;
        la[q] = p * 64 * KS + swz(row, half);
    }
    const int r = lane & 31, h = lane >> 5;
    const int fa = swz(wr * 32 + r, h), fb = OPER + swz(wc * 32 + r, h);
    v16i acc[NS];
#pragma unroll
    for (int d = 0; d < NS; d++) acc[d] = (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i ra[4], rb[4];
    const int nk = K / KS;
#pragma unroll
    for (int q = 0; q < 4; q++) { ra[q] = *(const v4i*)ga[q]; rb[q] = *(const v4i*)gb[q]; }
#pragma unroll
    for (int q = 0; q < 4; q++) { *(v4i*)(smem + la[q]) = ra[q]; *(v4i*)(smem + OPER + la[q]) = rb[q]; }
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const char* cur = smem + (kt & 1) * STAGE;
        char* nxt = smem + ((kt + 1) & 1) * STAGE;
        const bool more = kt + 1 < nk;
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; q++) { ra[q] = *(const v4i*)(ga[q] + (kt + 1) * KS); rb[q] = *(const v4i*)(gb[q] + (kt + 1) * KS); }
        }
        v4i a[NS], b[NS];
#pragma unroll
        for (int p = 0; p < NS; p++) {
            a[p] = *(const v4i*)(cur + fa + p * 64 * KS);
            b[p] = *(const v4i*)(cur + fb + p * 64 * KS);
        }
#pragma unroll
        for (int p = 0; p < NS; p++)
#pragma unroll
            for (int q = 0; q + p < NS; q++) acc[p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b[q], acc[p + q], 0, 0, 0);
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; q++) { *(v4i*)(nxt + la[q]) = ra[q]; *(v4i*)(nxt + OPER + la[q]) = rb[q]; }
        }
        __syncthreads();
    }
    // recombination: sum_d acc_d 2^(-12 - 7 d), Horner from the smallest term up; x row scale x column scale
    const double cs = sb[j0 + wc * 32 + r];
#pragma unroll
    for (int g = 0; g < 16; g++) {
        const int row = i0 + wr * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        double v = (double)acc[NS - 1][g];
#pragma unroll
        for (int d = NS - 2; d >= 0; d--) v = __builtin_fma(v, 0.0078125, (double)acc[d][g]);
        C[(size_t)row * ldc + j0 + wc * 32 + r] = v * (1.0 / 4096.0) * sa[row] * cs;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int m = argc > 1 ? atoi(argv[1]) : 8192, n = argc > 2 ? atoi(argv[2]) : 8192, k = argc > 3 ? atoi(argv[3]) : 2048;
    if (m % 64 || n % 64 || k % 32) { printf("m, n multiples of 64, k of 32\n"); return 2; }
    std::mt19937_64 rng(7);
    std::normal_distribution<double> G(0.0, 1.0);
    std::vector<double> A((size_t)m * k), B((size_t)n * k);
    for (double& v : A) v = G(rng) * exp(3.0 * G(rng));            // wide dynamic range inside a row
    for (double& v : B) v = G(rng) * exp(3.0 * G(rng));
    double *dA, *dB, *dC, *sa, *sb;
    signed char *SA, *SB;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dC, (size_t)m * n * 8));
    CK(hipMalloc(&sa, m * 8)); CK(hipMalloc(&sb, n * 8));
    CK(hipMalloc(&SA, (size_t)NS * m * k)); CK(hipMalloc(&SB, (size_t)NS * n * k));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k_gemm_i8, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS));
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    auto slice = [&]() {
        hipLaunchKernelGGL(k_rowscale, dim3((m + 3) / 4), dim3(256), 0, 0, dA, k, m, k, sa);
        hipLaunchKernelGGL(k_rowscale, dim3((n + 3) / 4), dim3(256), 0, 0, dB, k, n, k, sb);
        hipLaunchKernelGGL(k_slice, dim3((unsigned)(((long)m * (k / 16) + 255) / 256)), dim3(256), 0, 0, dA, k, m, k, sa, SA);
        hipLaunchKernelGGL(k_slice, dim3((unsigned)(((long)n * (k / 16) + 255) / 256)), dim3(256), 0, 0, dB, k, n, k, sb, SB);
    };
    auto gemm = [&]() {
        hipLaunchKernelGGL(k_gemm_i8, dim3((m / 64) * (n / 64)), dim3(256), GEMM_LDS, 0, SA, sa, m, SB, sb, n, k, dC, n, m / 64);
    };
    slice(); gemm();
    CK(hipDeviceSynchronize());
    const int reps = 5;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) slice();
    hipEventRecord(e1);
    for (int i = 0; i < reps; i++) gemm();
    hipEventRecord(e2);
    CK(hipDeviceSynchronize());
    float ms_s, ms_g;
    hipEventElapsedTime(&ms_s, e0, e1); hipEventElapsedTime(&ms_g, e1, e2);
    ms_s /= reps; ms_g /= reps;
    const double flop = 2.0 * m * (double)n * k;
    printf("m %d n %d k %d: slicing both operands %.3f ms, product %.3f ms = %.1f TF/s fp64-equivalent (with the slicing: %.1f); fp64 MFMA spec 78.6\n",
           m, n, k, ms_s, ms_g, flop / (ms_g * 1e-3) / 1e12, flop / ((ms_g + ms_s) * 1e-3) / 1e12);
    // error against long double on a sample of entries
    std::vector<double> C((size_t)m * n);
    CK(hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0, worst_f64 = 0.0;
    std::uniform_int_distribution<int> Ui(0, m - 1), Uj(0, n - 1);
    for (int s = 0; s < 4000; s++) {
        const int i = s < 64 ? s : Ui(rng), j = s < 64 ? (s * 37) % n : Uj(rng);
        long double ref = 0.0L, mag = 0.0L;
        double f64 = 0.0;
        for (int q = 0; q < k; q++) {
            const long double pq = (long double)A[(size_t)i * k + q] * (long double)B[(size_t)j * k + q];
            ref += pq; mag += fabsl(pq);
            f64 = fma(A[(size_t)i * k + q], B[(size_t)j * k + q], f64);
        }
        worst = fmax(worst, (double)(fabsl((long double)C[(size_t)i * n + j] - ref) / mag));
        worst_f64 = fmax(worst_f64, (double)(fabsl((long double)f64 - ref) / mag));
    }
    printf("max |C - ref| / sum |a b| over 4000 entries: int8-slice product %.3e, a sequential fp64 fma chain %.3e (2^-53 = 1.1e-16)\n", worst, worst_f64);
    return 0;
}
