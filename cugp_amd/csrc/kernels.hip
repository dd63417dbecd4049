// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the GP hot path.
//
// Path (reference file:line each kernel replaces):
//   k_build        SE+noise covariance          cpp_serial_gp/covkernel.cpp:64-102, cuda_scalingdist/cuda_gp.cu:232-281
//   potf2 / trsm / syrk_step / syrk_wide   blocked right-looking Cholesky (near window per step, far columns once per
//                  panel)   common/matrixops.cpp:68-108, cpp_matrixalgebra/blocked_cholesky.cpp:221-262,
//                  cuda_src/cuda_gp.cu:1237-1308
//   trtri_*        L^-1 by recursive doubling   common/matrixops.cpp:330-340, cuda_src/cuda_gp.cu:1854-1915 (TMI)
//   lauum          K^-1 = L^-T L^-1             common/matrixops.cpp:383-435, cuda_src/cuda_gp.cu:119-136
//   trmv / trace / finalize   alpha, y'K^-1 y, log|K|, gradient traces   covkernel.cpp:118-129,162-263
//   kcross / predict_*        predictive mean / variance                 covkernel.cpp:105-116,277-323
//
// Everything dense runs through ONE fp64 MFMA tile product (v_mfma_f64_16x16x4_f64):
//   C[128x128] = sum_k A[i][k] * B[j][k]   ("NT": both operands row-major with k contiguous)
// 4 waves per workgroup in a 2x2 grid, 64x64 per wave = 4x4 MFMA tiles (64 fp64 accumulators
// per lane), K staged 16 deep through LDS, double-buffered, global loads for stage t+1 in
// flight while stage t is multiplied.  LDS image: [k-pair plane, padded by 16 B][row][2 doubles]
// -- fragment reads are 256 contiguous bytes per 32 lanes (conflict-free ds_read_b64) at one
// per-lane base + immediate offsets (no address arithmetic in the loop: VALU issue costs MFMA
// issue on gfx950), staging writes (8 lanes = 8 planes of one row) land on 8 different 16-B slots
// (two rows of a 16-lane group overlap in 7 of them: 2-way, the ~5 % SQ_LDS_BANK_CONFLICT of the tile kernels;
// a 32-byte pad removes the overlap and changes no timing: the K loop does not wait for these writes).
//
// Matrices are npad x npad row-major with npad = ceil(n/128)*128; the padding of K is the
// identity, so every kernel works on whole tiles and the factor, inverse, log-determinant and
// traces of the leading n x n block are unchanged.
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

namespace cugp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// A pointer read out of the experts' table (ExpertPtrs) is a generic pointer to the compiler: every access through it
// became a FLAT instruction, and flat loads count against lgkmcnt as well as vmcnt -- each wait for an LDS fragment
// read also waited for the global loads meant to stay in flight behind the MFMAs.  Round-tripping it through the
// global address space lets the address-space inference make it (and the select with the kernel argument) global.
template <class T>
__device__ __forceinline__ T* GP(T* p)
{
    // (through an integer: a plain generic -> global -> generic cast pair is folded away before the inference runs)
    return (T*)(__attribute__((address_space(1))) T*)(unsigned long long)p;
}

// Diagnostic build only (tools/wgtimes.hip, -DCUGP_WGTIMES): every workgroup of the kernels below leaves
// {kind | param << 8, start, end (s_memrealtime, 100 MHz), shader cycles lived (s_memtime)} in a side buffer -- when the workgroups of a launch were
// dispatched and how long each ran, i.e. whether a launch that took long beside other streams WAITED for workgroup
// slots or RAN slowly.  Nothing of it exists in the product build.
#ifdef CUGP_WGTIMES
constexpr unsigned WGT_CAP = 1u << 19;
__device__ unsigned long long g_wgt[4 * WGT_CAP];
__device__ unsigned g_wgt_n;
struct WgTimer {
    unsigned long long t0, c0, tag;
    __device__ __forceinline__ WgTimer(int kind, int param) : t0(__builtin_amdgcn_s_memrealtime()), c0(__builtin_amdgcn_s_memtime()), tag((unsigned long long)kind | (unsigned long long)param << 8) {}
    __device__ __forceinline__ ~WgTimer()
    {
        if (threadIdx.x == 0) {
            const unsigned i = atomicAdd(&g_wgt_n, 1u);
            if (i < WGT_CAP) {
                g_wgt[4 * i] = tag; g_wgt[4 * i + 1] = t0; g_wgt[4 * i + 2] = __builtin_amdgcn_s_memrealtime();
                g_wgt[4 * i + 3] = __builtin_amdgcn_s_memtime() - c0;          // shader cycles the workgroup lived
            }
        }
    }
};
#define WGT(name, kind, param) WgTimer name(kind, param)
void wgt_reset() { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgt_n), &z, sizeof z); }
unsigned wgt_fetch(unsigned long long* out, unsigned cap)
{
    unsigned n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wgt_n), sizeof n);
    if (n > WGT_CAP) n = WGT_CAP;
    if (n > cap) n = cap;
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgt), (size_t)n * 32);
    return n;
}
#else
#define WGT(name, kind, param)
#endif
enum { WGT_TRSM = 0, WGT_DIAGUPD = 1, WGT_POTF2 = 2, WGT_STEPTILE = 3, WGT_BORDER = 4, WGT_LAUUM = 5, WGT_LEVEL = 6,
       WGT_TRTRI_DIAG = 7, WGT_WIDE = 8 };

// Profiling level 5 (cugp_capi.cpp TimedLaunch): a timed launch carries a slot in a device buffer and every workgroup
// of it leaves its start in slot[0] (atomic min) and its end in slot[STAMP_STRIDE] (atomic max), on the chip-wide 100 MHz
// clock (s_memrealtime): launch duration = first workgroup's first instruction .. last workgroup's last, with NOTHING
// added to the stream -- the schedule is the untimed one (an event pair around a launch costs ~5 us of device time and
// brackets the dispatch gap in front of it; hipExtLaunchKernelGGL's own start / stop events still slow the evaluation
// by 3 %).  What rocprofv3 adds to this figure is the dispatch's ramp and drain, 1-3 us per launch.  slot == nullptr:
// one scalar compare per workgroup.
struct LaunchStamp {
    unsigned long long* p;
    // every: stamp one workgroup in `every` (and the last 64 of the grid): a launch of thousands of SHORT workgroups
    // (k_build: 8256 of ~2 us) would otherwise spend its time queueing 16k atomics on two words (+25 %)
    __device__ __forceinline__ explicit LaunchStamp(unsigned long long* p_, unsigned every = 1)
        : p((p_ && (every <= 1 || blockIdx.x % every == 0 || blockIdx.x + 64 >= gridDim.x)) ? p_ : nullptr)
    {
        // (workgroups are dispatched in index order: the launch's first instruction is one of the first workgroups')
        if (p && threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y == 0)
            __hip_atomic_fetch_min(p, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ ~LaunchStamp()
    {
        if (p && threadIdx.x == 0)
            __hip_atomic_fetch_max(p + STAMP_STRIDE, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// ------------------------------------------------------------------------------------------
// fp64 MFMA tile product
// ------------------------------------------------------------------------------------------
constexpr int BK = 16;                       // k depth per LDS stage
// geometry of a tile product whose waves hold WM x WM MFMA tiles each (block edge 32*WM: 128 or 64)
template <int WM> struct Geo {
    static constexpr int BT = 32 * WM;               // block tile edge
    static constexpr int PLANE = BT * 16 + 16;       // bytes per k-pair plane (BT rows x 2 doubles) + 16 B pad:
                                                     // 8 staging lanes (8 planes of one row) hit 8 different 16-B slots
    static constexpr int OPER = (BK / 2) * PLANE;    // bytes per operand per stage
    static constexpr int STAGE = 2 * OPER;           // A + B
    static constexpr int LDS = 2 * STAGE;            // double buffered: 66048 B (WM=4, dynamic LDS) / 33280 B (WM=2)
};
constexpr int GEMM_LDS = Geo<4>::LDS;

// diagonal-block kernels work on 16x16 micro tiles (one MFMA tile)
constexpr int MT = 16;                             // micro tile (one MFMA tile)
constexpr int MTS = MT * (MT + 1);                 // doubles per LDS micro tile, rows padded to 17
constexpr int NMT = TILE / MT;                     // 8 micro tiles per edge
constexpr int NLT = NMT * (NMT + 1) / 2;           // 36 lower micro tiles
constexpr int POTF2_LDS = (NLT * MTS + TILE) * 8;  // tiles + 1/L_ii  = 79360 B

// B-operand rows are stored permuted inside each wave's column range so that MFMA tiles n = 2p, 2p+1
// of a wave produce ADJACENT output columns in one lane (16-byte global accesses in the epilogue)
// while the fragment reads stay 256 contiguous bytes: tile row R -> LDS position bpos(R).
template <int WM>
__device__ __forceinline__ int bpos(int R)
{
    const int q = R & (16 * WM - 1);
    return (R - q) + (((q >> 5) * 2 + (q & 1)) * 16) + ((q & 31) >> 1);
}

// threadIdx.x behind an optimisation barrier: inside a persistent tile loop (k_syrk_wide) the compiler would
// otherwise hoist every per-thread address of the tile product out of the loop and keep them all live across
// the accumulator load / store, which overflows the 256-register budget of 2 workgroups per CU into scratch.
__device__ __forceinline__ int opaque_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// One K stage (16 deep) of MFMAs out of LDS buffer `cur`.  Measured on gfx950: every VALU instruction a wave
// issues between fp64 MFMAs costs MFMA issue time (pure MFMA stream 74 TF/s, +1 VALU per MFMA 62, +4: 55),
// so the loop body carries NO address arithmetic: fragment addresses are one per-lane base + immediates.
template <int WM>
__device__ __forceinline__ void tile_stage_mma(const char* __restrict__ cur, int abase, int bbase,
                                               d4 (&acc)[WM][WM])
{
    typedef Geo<WM> G;
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
        double a[WM], b[WM];
#pragma unroll
        for (int m = 0; m < WM; m++) {
            a[m] = *(const double*)(cur + abase + kk * 2 * G::PLANE + m * 256);
            b[m] = *(const double*)(cur + bbase + kk * 2 * G::PLANE + m * 256);
        }
#pragma unroll
        for (int m = 0; m < WM; m++)
#pragma unroll
            for (int n = 0; n < WM; n++)
                acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
}

// acc[m][n] += (NEGA ? -1 : 1) * A(i0.., kbeg..kend) * B(j0.., kbeg..kend)^T ; Ag -> A[i0][0], Bg -> B[j0][0];
// kbeg, kend multiples of BK.  All 256 threads must call (barriers inside).  WM (deduced from acc) is the
// number of 16x16 MFMA tiles per wave per dimension: 4 -> 128x128 block, 2 -> 64x64 block.
// LDS image per operand and stage: [k-pair plane (padded by 16 B)][row][2 doubles].
#ifdef CUGP_TILE_STAMPS   // diagnostic build only (tools/gemm_k_bench.hip): where a tile's time goes
__device__ unsigned long long g_tile_stamps[64];
__device__ int g_tile_stamp_on;                          // stamps inside tile_nt only while this is set
#define TILE_STAMP(i)                                                              \
    do {                                                                           \
        if (threadIdx.x == 0 && blockIdx.x == 200) {                               \
            g_tile_stamps[i] = __builtin_readcyclecounter();                       \
            g_tile_stamps[32 + i] = __builtin_amdgcn_s_memrealtime();              \
        }                                                                          \
    } while (0)
#else
#define TILE_STAMP(i)
#endif

template <bool NEGA, int WM>
__device__ __forceinline__ void tile_nt(const double* __restrict__ Ag, int lda, const double* __restrict__ Bg,
                                        int ldb, int kbeg, int kend, d4 (&acc)[WM][WM], char* smem)
{
    typedef Geo<WM> G;
    const int t = opaque_tid();
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // staging map: WM chunks of 16 B per operand per thread; 8 consecutive lanes cover one row's 128 B
    const int srow = t >> 3, skp = t & 7;                 // + 32 rows per q
    const double* ag = Ag + (size_t)srow * lda + 2 * skp;
    const double* bg = Bg + (size_t)srow * ldb + 2 * skp;
    d2 ra[WM], rb[WM];
    int wa[WM], wb[WM];                                   // LDS byte offsets of my staging chunks
#pragma unroll
    for (int q = 0; q < WM; q++) {
        wa[q] = skp * G::PLANE + (srow + 32 * q) * 16;
        wb[q] = G::OPER + skp * G::PLANE + bpos<WM>(srow + 32 * q) * 16;
    }
    // fragment bases: lane (fr = lane & 15, fk = lane >> 4) reads row fr (+16 m) of k-pair plane (2 kk + fk/2)
    const int fr = lane & 15, fk = lane >> 4;
    const int abase = (fk >> 1) * G::PLANE + (wr * 16 * WM + fr) * 16 + (fk & 1) * 8;
    const int bbase = G::OPER + (fk >> 1) * G::PLANE + (wc * 16 * WM + fr) * 16 + (fk & 1) * 8;

    const int nk = (kend - kbeg) / BK;
    if (nk <= 0) return;

#pragma unroll
    for (int q = 0; q < WM; q++) {
        ra[q] = *(const d2*)(ag + (size_t)(32 * q) * lda + kbeg);
        rb[q] = *(const d2*)(bg + (size_t)(32 * q) * ldb + kbeg);
    }
#pragma unroll
    for (int q = 0; q < WM; q++) {
        *(d2*)(smem + wa[q]) = NEGA ? -ra[q] : ra[q];
        *(d2*)(smem + wb[q]) = rb[q];
    }
    __syncthreads();
#ifdef CUGP_TILE_STAMPS
    if (g_tile_stamp_on) TILE_STAMP(1);
#endif

    // two stages per trip so both LDS buffers are compile-time offsets (no address arithmetic in the loop)
#define CUGP_STAGE(HALF, MORE, KNEXT)                                                              \
    do {                                                                                           \
        const char* cur_ = smem + (HALF) * G::STAGE;                                               \
        char* nxt_ = smem + (1 - (HALF)) * G::STAGE;                                               \
        const bool more_ = (MORE);                                                                 \
        if (more_) {                                                                               \
            const int k_ = (KNEXT);                                                                \
            _Pragma("unroll") for (int q = 0; q < WM; q++) {                                       \
                ra[q] = *(const d2*)(ag + (size_t)(32 * q) * lda + k_);                            \
                rb[q] = *(const d2*)(bg + (size_t)(32 * q) * ldb + k_);                            \
            }                                                                                      \
        }                                                                                          \
        tile_stage_mma<WM>(cur_, abase, bbase, acc);                                               \
        if (more_) {                                                                               \
            _Pragma("unroll") for (int q = 0; q < WM; q++) {                                       \
                *(d2*)(nxt_ + wa[q]) = NEGA ? -ra[q] : ra[q];                                      \
                *(d2*)(nxt_ + wb[q]) = rb[q];                                                      \
            }                                                                                      \
        }                                                                                          \
        __syncthreads();                                                                           \
    } while (0)

    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        CUGP_STAGE(0, true, kbeg + (kt + 1) * BK);
        CUGP_STAGE(1, kt + 2 < nk, kbeg + (kt + 2) * BK);
    }
    if (kt < nk) CUGP_STAGE(0, false, 0);
#undef CUGP_STAGE
}

template <int WM>
__device__ __forceinline__ void acc_zero(d4 (&acc)[WM][WM])
{
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int n = 0; n < WM; n++) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
}

// accumulator element (m,n,r) of this lane is C[row][col]: f64 16x16x4 C/D map (col = lane&15,
// row = (lane>>4) + 4*r inside a 16x16 tile) composed with the B-row permutation above, so tiles
// n = 2p and 2p+1 hold columns 2*(lane&15) and 2*(lane&15)+1 of the 32-column group p.
#define ACC_ROW(m, r) (wr * 16 * WM + (m) * 16 + (lane >> 4) + 4 * (r))
#define ACC_COL2(np) (wc * 16 * WM + (np) * 32 + 2 * (lane & 15))

// acc = C   (the K loop then accumulates straight onto it: no read-modify-write epilogue).  The product form of rounds
// 1-4; since round 5 the library's kernels use tile_accum_store below (tools/wide_bench.hip keeps both side by side).
// STREAM: the tile is touched once per launch -> non-temporal accesses keep the shared operand panels in L2
template <bool STREAM = false, int WM>
__device__ __forceinline__ void tile_load(const double* __restrict__ C, int ldc, d4 (&acc)[WM][WM])
{
    const int tid = opaque_tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++) {
                const d2* src = (const d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
                const d2 v = STREAM ? __builtin_nontemporal_load(src) : *src;
                acc[m][2 * np][r] = v[0];
                acc[m][2 * np + 1][r] = v[1];
            }
}

// C = alpha * acc
template <bool STREAM = false, int WM>
__device__ __forceinline__ void tile_store(double* __restrict__ C, int ldc, const d4 (&acc)[WM][WM], double alpha)
{
    const int tid = opaque_tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++)
            {
                d2* dst = (d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
                const d2 v = (d2){alpha * acc[m][2 * np][r], alpha * acc[m][2 * np + 1][r]};
                if (STREAM) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
}

// C += SIGN * acc, the C tile read in the EPILOGUE (round 5).  The accumulators start from zero and the K loop runs
// first; then one row group m of the wave at a time -- 16 doubles per lane in flight -- is read, updated and written.
// The product form (acc = C in front of the K loop, tile_load) held the first MFMA back until all 128 KB of the tile
// had arrived, in a burst with the launch's other workgroups and with the first operand stage; here the K loop starts
// at once and the C traffic of a workgroup overlaps the K loop of the other workgroup on its CU.  Measured
// (tools/wide_bench.hip, trailing-update form, 1596 tiles): K = 256 52.8 -> 54.7 TF/s, K = 512 56.4 -> 59.3, K = 1024
// 59.0 -> 61.4, K = 2048 54.4 -> 56.3, K = 128 44.6 -> 45.2; with neither read nor write 61.4 / 62.5 / 62.9 / 56.9 /
// 59.2.  Plain accesses: non-temporal ones lost 1-3 TF/s in this form.  One more rounding per entry and pass than the
// product form (the sum is formed first, then added), for every kernel alike.
template <int SIGN, int WM>
__device__ __forceinline__ void tile_accum_store(double* __restrict__ C, int ldc, const d4 (&acc)[WM][WM])
{
    const int tid = opaque_tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < WM; m++) {
        d2 v[4][WM / 2];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++)
                v[r][np] = *(const d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int np = 0; np < WM / 2; np++) {
                d2* dst = (d2*)(C + (size_t)ACC_ROW(m, r) * ldc + ACC_COL2(np));
                if (SIGN > 0) *dst = (d2){v[r][np][0] + acc[m][2 * np][r], v[r][np][1] + acc[m][2 * np + 1][r]};
                else *dst = (d2){v[r][np][0] - acc[m][2 * np][r], v[r][np][1] - acc[m][2 * np + 1][r]};
            }
    }
}

// Ct[col][row] = alpha*acc  (transposed store).  The tile is turned through LDS (free after the K loop)
// in two halves of 16*WM rows so that global stores are whole runs of that many doubles; the LDS image
// [col][16*WM rows] is XOR-swizzled on the row index so the accumulator-layout writes do not pile onto
// one bank.  All 256 threads must call.
template <int WM>
__device__ __forceinline__ void tile_store_t(double* __restrict__ Ct, int ldc, const d4 (&acc)[WM][WM], double alpha,
                                             char* smem)
{
    constexpr int RH = 16 * WM, BT = 32 * WM;            // rows per half, block edge
    constexpr int LPC = RH / 2, CPP = 256 / LPC;         // lanes per column-row, column-rows per pass
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    double* sd = (double*)smem;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        __syncthreads();
        if (wr == h) {
#pragma unroll
            for (int m = 0; m < WM; m++)
#pragma unroll
                for (int n = 0; n < WM; n++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int col = ACC_COL2(n >> 1) + (n & 1);
                        const int rr = m * 16 + (lane >> 4) + 4 * r;
                        sd[col * RH + (rr ^ ((col >> 1) & 15))] = alpha * acc[m][n][r];
                    }
        }
        __syncthreads();
        const int rr = (t % LPC) * 2;
#pragma unroll
        for (int pass = 0; pass < BT / CPP; pass++) {
            const int col = pass * CPP + t / LPC;
            const int sw = (col >> 1) & 15;
            const d2 v = (d2){sd[col * RH + (rr ^ sw)], sd[col * RH + ((rr + 1) ^ sw)]};
            *(d2*)(Ct + (size_t)col * ldc + h * RH + rr) = v;
        }
    }
}

// lower-triangular tile index: idx -> (ti >= tj)
__device__ __forceinline__ void tri_index(int idx, int& ti, int& tj)
{
    int r = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= idx) r++;
    while (r * (r + 1) / 2 > idx) r--;
    ti = r;
    tj = idx - r * (r + 1) / 2;
}

// Tiles (ti >= tj) of the tile columns [0, wcol) of a lower triangle, row by row: row r holds min(r + 1, wcol)
// tiles.  idx -> (ti, tj), both relative to the region's first column.  wcol >= the triangle's size gives the
// whole triangle (tri_index).
__device__ __forceinline__ void trap_index(int idx, int wcol, int& ti, int& tj)
{
    const int head = wcol * (wcol + 1) / 2;
    if (idx < head) { tri_index(idx, ti, tj); return; }
    const int rem = idx - head;
    ti = wcol + rem / wcol;
    tj = rem % wcol;
}

// ---- K^-1 (lower tiles, diagonal tiles complete) = U * U^T, U = L^-T upper, taken one block of
// inverse rows [a, a+w) at a time:  Kinv(ti,tj) (+)= sum_{k in [max(ti,a), a+w)} U[ti][k] U[tj][k]^T
// for tj <= ti < a+w.  a = 0, w = nt is the whole product in one launch.
template <int WM>
__device__ __forceinline__ void lauum_tile(const double* __restrict__ U, double* __restrict__ Kinv, int ld, int a, int w,
                                           int tile, int sub, char* smem)
{
    constexpr int SUB = 4 / WM, BT = 32 * WM;
    int ti, tj;
    tri_index(tile, ti, tj);                           // ascending ti = longest k ranges first
    const int si = (sub / SUB) * BT, sj = (sub % SUB) * BT;
    double* C = Kinv + (size_t)(ti * TILE + si) * ld + tj * TILE + sj;
    d4 acc[WM][WM];
    acc_zero(acc);
    tile_nt<false>(U + (size_t)(ti * TILE + si) * ld, ld, U + (size_t)(tj * TILE + sj) * ld, ld,
                   (ti < a ? a : ti) * TILE, (a + w) * TILE, acc, smem);
    if (ti < a) tile_accum_store<1>(C, ld, acc);       // rows of earlier blocks: add this block's share
    else tile_store(C, ld, acc, 1.0);
}

// nfull (WM = 4 only): the first nfull tiles as 128x128 workgroups, the rest -- a last, partly empty round of 512
// workgroup slots that would take a whole tile time -- as four 64x64 workgroups each (split_round below)
template <int WM>
__global__ __launch_bounds__(256, 2) void k_lauum(const double* __restrict__ U, double* __restrict__ Kinv, int ld,
                                                  int a, int w, int nfull, const ExpertPtrs* __restrict__ bt,
                                                  unsigned long long* stamp)
{
    LaunchStamp stamp_(stamp);
    if (bt) { U = GP(bt[blockIdx.y].U); Kinv = GP(bt[blockIdx.y].Kinv); }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WGT(wgt_, WGT_LAUUM, a);
    if (WM == 2) {
        lauum_tile<2>(U, Kinv, ld, a, w, blockIdx.x >> 2, blockIdx.x & 3, smem);
    } else if ((int)blockIdx.x < nfull) {
        lauum_tile<4>(U, Kinv, ld, a, w, blockIdx.x, 0, smem);
    } else {
        const int y = blockIdx.x - nfull;
        lauum_tile<2>(U, Kinv, ld, a, w, nfull + (y >> 2), y & 3, smem);
    }
}

// ---- triangular inverse.  [A 0; C B]^-1 = [TA 0; -TB C TA, TB] with A = tiles [.., b0), B = tiles [b0, ..):
// step 1: Wt(tj in A, ti in B) = sum_{k in A, k >= tj} U[tj][k] * L[ti][k]   -> scratch in T's upper tiles
//         (k tiles [kbeg, kend) of it; `accumulate` adds onto what earlier launches left there)
// step 2: T(ti in B, tj in A) = -sum_{k in B, k <= ti} T[ti][k] * Wt[tj][k]   and U(tj,ti) = transpose
template <int WM>
__device__ __forceinline__ void trtri_tile(const double* __restrict__ L, double* __restrict__ T,
                                           double* __restrict__ U, int ld, int tj, int ti, int step, int kbeg,
                                           int kend, bool accumulate, int sub, char* smem)
{
    constexpr int SUB = 4 / WM;                          // output sub-tiles per 128-tile edge (1 or 2)
    constexpr int BT = 32 * WM;
    const int si = (sub / SUB) * BT, sj = (sub % SUB) * BT;     // offsets of my sub-tile inside the 128-tile
    d4 acc[WM][WM];
    if (step == 1) {
        // Wt(tj, ti) rows in A's tile tj, columns in B's tile ti
        double* W = T + (size_t)(tj * TILE + si) * ld + ti * TILE + sj;
        acc_zero(acc);
        tile_nt<false>(U + (size_t)(tj * TILE + si) * ld, ld, L + (size_t)(ti * TILE + sj) * ld, ld, kbeg * TILE,
                       kend * TILE, acc, smem);
        if (accumulate) tile_accum_store<1>(W, ld, acc);
        else tile_store(W, ld, acc, 1.0);
    } else {
        acc_zero(acc);
        tile_nt<true>(T + (size_t)(ti * TILE + si) * ld, ld, T + (size_t)(tj * TILE + sj) * ld, ld, kbeg * TILE,
                      kend * TILE, acc, smem);
        tile_store(T + (size_t)(ti * TILE + si) * ld + tj * TILE + sj, ld, acc, 1.0);
        tile_store_t(U + (size_t)(tj * TILE + sj) * ld + ti * TILE + si, ld, acc, 1.0, smem);
    }
}

// one tile (blk) / sub-tile (sub) of one level of recursive doubling over nt tiles: all pairs of s-tile blocks
template <int WM>
__device__ __forceinline__ void level_item(const double* __restrict__ L, double* __restrict__ T, double* __restrict__ U,
                                           int ld, int nt, int s, int step, int blk, int sub, char* smem)
{
    const int npairs = (nt + 2 * s - 1) / (2 * s);      // last one may have a short (or empty) B
    int p = blk / (s * s);
    if (p > npairs - 1) p = npairs - 1;
    int rem = blk - p * s * s;
    const int a0 = 2 * p * s;                            // A = [a0, a0+s), B = [a0+s, min(a0+2s, nt))
    const int b0 = a0 + s;
    int sb = nt - b0;
    if (sb > s) sb = s;
    // tile in A (column block), tile in B (row block); longest k range first in both steps
    // (step 1: k from tj to the end of A -> ja ascending; step 2: k from b0 to ti -> ib descending)
    const int ja = (step == 1) ? rem / sb : rem % s;
    const int ib = (step == 1) ? rem % sb : sb - 1 - rem / s;
    const int tj = a0 + ja, ti = b0 + ib;
    trtri_tile<WM>(L, T, U, ld, tj, ti, step, step == 1 ? tj : b0, step == 1 ? b0 : ti + 1, false, sub, smem);
}

// tiles |A| x |B| of one level over nt tiles (pairs p = 0.. : A = [2ps, 2ps+s), B = [2ps+s, min(2ps+2s, nt)))
__host__ __device__ inline int level_tiles(int nt, int s)
{
    int tiles = 0;
    for (int a0 = 0; a0 + s < nt; a0 += 2 * s) {
        int sb = nt - (a0 + s);
        if (sb > s) sb = s;
        tiles += s * sb;
    }
    return tiles;
}

// one level of recursive doubling: all pairs of s-tile blocks at once
template <int WM>
__global__ __launch_bounds__(256, 2) void k_trtri_level(const double* __restrict__ L, double* __restrict__ T,
                                                        double* __restrict__ U, int ld, int nt, int s, int step,
                                                        size_t off, const ExpertPtrs* __restrict__ bt,
                                                        unsigned long long* stamp)
{
    LaunchStamp stamp_(stamp);
    // off: element offset of the diagonal sub-matrix the level works on (a block of inverse rows)
    if (bt) { L = GP(bt[blockIdx.y].A); T = GP(bt[blockIdx.y].T); U = GP(bt[blockIdx.y].U); }
    L += off; T += off; U += off;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WGT(wgt_, WGT_LEVEL, s * 2 + step);
    constexpr int SUB = 4 / WM;
    level_item<WM>(L, T, U, ld, nt, s, step, blockIdx.x / (SUB * SUB), blockIdx.x % (SUB * SUB), smem);
}

// bordering: rows B = [a, a+w) of the inverse from the finished leading block A = [0, a) and B's own
// inverse (the same two steps with an unbalanced split; lets the inverse follow the factorisation
// block row by block row).  Step 1 comes in launches over k chunks [c0, c1) of A, ascending, each adding
// onto the last; the host issues chunk [c0, c1) for ALL rows below it as soon as those inverse rows are
// final (uniform K, and most of the work is done before the rows' own turn comes).
template <int WM>
__device__ __forceinline__ void border_tile(const double* __restrict__ L, double* __restrict__ T, double* __restrict__ U,
                                            int ld, int a, int w, int step, int c0, int c1, int blk, int sub, char* smem)
{
    if (step == 1) {
        const int tj = blk / w, ti = a + blk % w;                       // tj < c1
        trtri_tile<WM>(L, T, U, ld, tj, ti, 1, tj > c0 ? tj : c0, c1, tj < c0, sub, smem);
    } else {
        const int tj = blk % a, ti = a + w - 1 - blk / a;               // longest k range first
        trtri_tile<WM>(L, T, U, ld, tj, ti, 2, a, ti + 1, false, sub, smem);
    }
}

// nfull (WM = 4 only): tiles beyond it run as four 64x64 workgroups each (the last, partly empty round; split_round)
template <int WM>
__global__ __launch_bounds__(256, 2) void k_trtri_border(const double* __restrict__ L, double* __restrict__ T,
                                                         double* __restrict__ U, int ld, int a, int w, int step,
                                                         int c0, int c1, int nfull, const ExpertPtrs* __restrict__ bt,
                                                         unsigned long long* stamp)
{
    LaunchStamp stamp_(stamp);
    if (bt) { L = GP(bt[blockIdx.y].A); T = GP(bt[blockIdx.y].T); U = GP(bt[blockIdx.y].U); }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WGT(wgt_, WGT_BORDER, a * 4 + step);
    if (WM == 2) {
        border_tile<2>(L, T, U, ld, a, w, step, c0, c1, blockIdx.x >> 2, blockIdx.x & 3, smem);
    } else if ((int)blockIdx.x < nfull) {
        border_tile<4>(L, T, U, ld, a, w, step, c0, c1, blockIdx.x, 0, smem);
    } else {
        const int y = blockIdx.x - nfull;
        border_tile<2>(L, T, U, ld, a, w, step, c0, c1, nfull + (y >> 2), y & 3, smem);
    }
}

// ---- prediction: W[t][i] = sum_{k <= i} Ks[t][k] * T[i][k] ----
// T = L^-1 is lower triangular: row tile i needs k < (i + 1) * tile only, so the work per output tile grows linearly
// with i.  Rounds 1-4 gave every 128x128 output tile a workgroup of its own: 512 workgroups for 1000 test points at
// N = 8192, one round of the chip, whose duration is the LONGEST tile's (K = 8192) while the average is half of it --
// 45 TF/s = 0.58 of peak.  Round 5: 64x64 output tiles in PAIRS (row tile p with row tile n64 - 1 - p, the long one
// first): every workgroup does K = (n64 + 1) * 64 in all, 1024 equal workgroups, four per CU.  The same per-element
// sums in the same order (the k range of a tile ends at its own diagonal; what the 128-tile form added beyond it were
// exact zeros of T): bit-identical results.
__global__ __launch_bounds__(256, 2) void k_predict_gemm(const double* __restrict__ Ks, const double* __restrict__ T,
                                                         double* __restrict__ W, int ld, int ntt64, int n64,
                                                         unsigned long long* stamp)
{
    LaunchStamp stamp_(stamp);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tt = blockIdx.x % ntt64, p = blockIdx.x / ntt64;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        const int ti = h == 0 ? n64 - 1 - p : p;
        d4 acc[2][2];
        acc_zero(acc);
        tile_nt<false>(Ks + (size_t)tt * 64 * ld, ld, T + (size_t)ti * 64 * ld, ld, 0, (ti + 1) * 64, acc, smem);
        tile_store(W + (size_t)tt * 64 * ld + ti * 64, ld, acc, 1.0);
    }
}

// ---- plain NT product for tests ----
__global__ __launch_bounds__(256, 2) void k_test_gemm(const double* __restrict__ A, const double* __restrict__ B,
                                                      double* __restrict__ C, int n, int k, int mt)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ti = blockIdx.x % mt, tj = blockIdx.x / mt;
    d4 acc[4][4];
    acc_zero(acc);
    tile_nt<false>(A + (size_t)ti * TILE * k, k, B + (size_t)tj * TILE * k, k, 0, k, acc, smem);
    tile_store(C + (size_t)ti * TILE * n + tj * TILE, n, acc, 1.0);
}

__global__ __launch_bounds__(256) void k_mfma_peak(double* sink, int iters)
{
    d4 acc[8];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// ------------------------------------------------------------------------------------------
// panel solve with the two 64x64 diagonal inverses (T00, T11) of the factor block: per 16-row strip
//   X0 = A0 T00^T ;  Z1 = A1 - X0 L10^T ;  X1 = Z1 T11^T
// One workgroup of EIGHT waves per strip.  Everything is kept transposed, so a result in C/D layout is the next
// product's B operand (k = (lane>>4) + 4*reg) and passes from wave to wave through LDS as it stands.
// Round 3: the three 64x64 matrices and the strip come into LDS with coalesced 16-byte loads (L10 and T11 wait in
// registers until the buffer is free) and the MFMA operands are read from there.  The strip used to load every
// operand from global memory in MFMA layout -- 16 rows x 32 bytes per instruction, 68 such loads per lane, 96 KB of
// inverses per strip: ~280 MB of L2 sector traffic for a 63-tile panel, which took the launch 17-20 us (round-3
// timeline) on the factorisation's serial path; and each of the three dependent MFMA chains (<= 16 long) ran on one
// wave per SIMD at 138 cycles per MFMA.  Now wave (w, h) takes the k sub-ranges r in {2h, 2h+1} of column tile w:
// chains of <= 8, the two waves of a SIMD (w, w+4) interleaving at the pipe's full rate; the two halves meet in LDS.
// ------------------------------------------------------------------------------------------
constexpr int TRSM_TS = 68;                          // row stride (doubles) of the 64x64 matrix in LDS: rows 4 doubles apart mod 32
constexpr int TRSM_SS = 132;                         // ... of the 16 x 128 strip
constexpr int TRSM_LDS = (64 * TRSM_TS + MT * TRSM_SS + 3 * 4 * 4 * 64) * 8;   // + half sums, X0^T, Z1^T: 76288 B

// ---- forward substitution L z = y folded into the factorisation's own launches (round 5; LL-only evaluations) ----
// z_kb = L_kk^-1 w_kb needs only block kb of the factor and the running right-hand side w; w_i -= L(i,kb) z_kb for the
// rows below needs only the solved column kb.  Both ride in launches the factorisation makes anyway: ONE extra
// workgroup of the panel solve of column kb computes z_kb from the block's two 64x64 inverses (zblock_solve: z0 = T00
// w0, z1 = T11 (w1 - L10 z0)), and nt - kb - 1 extra workgroups of the step launch of column kb apply it to w
// (vec_update: one wave per row, the per-row order of k_trsv_update).  What used to follow the factorisation as 2 nt
// small launches (0.56 ms of a 5.5 ms log-likelihood at 8192 rows, 0.10 of 0.54 ms at 1500) is one 1-workgroup launch
// for the last block.
__device__ __forceinline__ void zblock_solve(const double* __restrict__ A, const double* __restrict__ d64, int ld, int kb,
                                             const double* __restrict__ w, double* __restrict__ z, double* __restrict__ sm)
{
    // 512 threads: row r = t >> 3 of a 64-row half, column group cg = t & 7 (8 columns each), partial sums over the 8
    // lanes of a row in a fixed order
    const int t = threadIdx.x, r = t >> 3, cg = t & 7, k0 = kb * TILE;
    double* w0 = sm;                                     // w_kb: [0, 64) upper half, [64, 128) lower half
    double* z0 = sm + 128;
    double* tt = sm + 192;
    const double* T00 = d64 + (size_t)kb * 8192;
    const double* T11 = T00 + 4096;
    const double* L10 = A + (size_t)(k0 + 64) * ld + k0;
    // every operand this thread will touch is requested before the first wait: one memory latency for the three
    // dependent products (requested phase by phase the workgroup took ~12 us and held the panel-solve launch -- on the
    // factorisation's chain -- 5 us longer than its strips)
    const int cend = (r | 15) + 1;
    double a0[8], a1[8], a2[8];
    const double wmine = t < TILE ? w[k0 + t] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const int c = cg * 8 + q;
        const d2 v0 = *(const d2*)(T00 + r * 64 + c), v1 = *(const d2*)(L10 + (size_t)r * ld + c), v2 = *(const d2*)(T11 + r * 64 + c);
        // (the 64x64 inverses hold their lower 16x16 micro tiles only: what lies beyond the row's own micro tile is
        //  never written -- selected away, not multiplied by zero; inside a diagonal micro tile the entries above the
        //  diagonal are exact zeros)
        a0[q] = c < cend ? v0[0] : 0.0; a0[q + 1] = c + 1 < cend ? v0[1] : 0.0;
        a1[q] = v1[0]; a1[q + 1] = v1[1];
        a2[q] = c < cend ? v2[0] : 0.0; a2[q + 1] = c + 1 < cend ? v2[1] : 0.0;
    }
    if (t < TILE) w0[t] = wmine;
    __syncthreads();
    auto row_sum = [&](double s) {
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        return s;
    };
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) s = __builtin_fma(a0[q], w0[cg * 8 + q], s);
    s = row_sum(s);
    if (cg == 0) z0[r] = s;
    __syncthreads();
    s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) s = __builtin_fma(a1[q], z0[cg * 8 + q], s);
    s = row_sum(s);
    if (cg == 0) tt[r] = w0[64 + r] - s;
    __syncthreads();
    s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) s = __builtin_fma(a2[q], tt[cg * 8 + q], s);
    s = row_sum(s);
    if (cg == 0) { z[k0 + r] = z0[r]; z[k0 + 64 + r] = s; }
}

// w[row] -= L[row][k0 .. k0 + 128) . z[k0 ..) for the 128 rows of tile row `ti` (one wave per row, 32 rows per wave of a
// 256-thread workgroup): k_trsv_update's arithmetic
__device__ __forceinline__ void vec_update(const double* __restrict__ A, int ld, int kb, int ti, const double* __restrict__ z,
                                           double* __restrict__ w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k0 = kb * TILE;
    const d2 xv = *(const d2*)(z + k0 + lane * 2);
    const double* a = A + (size_t)(ti * TILE + wave * 32) * ld + k0 + lane * 2;
    // all 32 rows of the wave requested before the first is reduced (row by row the loop was one memory latency per
    // row: ~45 us per workgroup, longer than the diagonal block the step launch hides)
    d2 v[32];
#pragma unroll
    for (int q = 0; q < 32; q++) v[q] = *(const d2*)(a + (size_t)q * ld);
    double mine = 0.0;                                   // lane q ends up with row q's sum
#pragma unroll
    for (int q = 0; q < 32; q++) {
        double sum = v[q][0] * xv[0] + v[q][1] * xv[1];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
        const double tot = __shfl(sum, 0, 64);
        if (lane == q) mine = tot;
    }
    if (lane < 32) {
        double* wr = w + ti * TILE + wave * 32 + lane;
        *wr -= mine;
    }
}

// zv / wv (when given): the forward substitution rides along -- workgroup `nstrips` of the launch computes z_kb
__global__ __launch_bounds__(512) void k_trsm_inv64(double* __restrict__ A, const double* __restrict__ d64,
                                                    int ld, int kb, const ExpertPtrs* __restrict__ bt,
                                                    unsigned long long* stamp, int nstrips, double* __restrict__ zvec,
                                                    const double* __restrict__ wvec)
{
    LaunchStamp stamp_(stamp);
    if (bt) {
        A = GP(bt[blockIdx.y].A); d64 = GP(bt[blockIdx.y].d64);
        if (zvec) { zvec = GP(bt[blockIdx.y].z); wvec = GP(bt[blockIdx.y].w); }
    }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // (zvec given: workgroup 0 -- dispatched first -- is the vector workgroup, the strips follow)
    const int strip = zvec ? (int)blockIdx.x - 1 : (int)blockIdx.x;
    if (strip < 0) {
        zblock_solve(A, d64, ld, kb, wvec, zvec, sm);
        return;
    }
    double* Tm = sm;                                    // T00, then -L10, then T11
    double* Sb = Tm + 64 * TRSM_TS;                     // the strip: A in, X out (row-major)
    double* Sc = Sb + MT * TRSM_SS;                     // half sums of the waves h = 1: [w][r][lane]
    double* X0 = Sc + 4 * 4 * 64;                       // X0^T tiles as B operands: [kt][r][lane]
    double* Z1 = X0 + 4 * 4 * 64;
    WGT(wgt_, WGT_TRSM, kb);
    __builtin_amdgcn_s_setprio(3);                      // on the factorisation's serial chain (see k_syrk_step)
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int w = wv & 3, h = wv >> 2, c = lane & 15, g = lane >> 4;
    const int k0 = kb * TILE;
    double* Ag = A + (size_t)(k0 + TILE + strip * MT) * ld + k0;
    const double* T00 = d64 + (size_t)kb * 8192;
    const double* T11 = T00 + 4096;
    const double* L10 = A + (size_t)(k0 + 64) * ld + k0;
    // coalesced requests for everything the strip will read: 4 chunks of 16 B per thread and 64x64 matrix, 2 of the strip
    d2 rt[4], rl[4], r1[4], ra[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int q = t + 512 * i, row = q >> 5, cp = (q & 31) * 2;
        rt[i] = *(const d2*)(T00 + row * 64 + cp);
        rl[i] = *(const d2*)(L10 + (size_t)row * ld + cp);
        r1[i] = *(const d2*)(T11 + row * 64 + cp);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int q = t + 512 * i, row = q >> 6, cp = (q & 63) * 2;
        ra[i] = *(const d2*)(Ag + (size_t)row * ld + cp);
    }
    auto fill_tm = [&](const d2 (&r)[4], double sign) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int q = t + 512 * i, row = q >> 5, cp = (q & 31) * 2;
            *(d2*)(Tm + row * TRSM_TS + cp) = sign * r[i];
        }
    };
    fill_tm(rt, 1.0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int q = t + 512 * i, row = q >> 6, cp = (q & 63) * 2;
        *(d2*)(Sb + row * TRSM_SS + cp) = ra[i];
    }
    __syncthreads();
    const d4 zero4 = (d4){0.0, 0.0, 0.0, 0.0};
    const double* tm = Tm + (w * MT + c) * TRSM_TS + g + 8 * h;        // my row of the matrix, my k sub-range
    // acc += sum over the k tiles kt < nkt of M[w][kt] (from Tm) times the B tiles bop[kt][r][lane], r in {2h, 2h+1}
    auto chain = [&](d4 acc, const double* bop, int nkt) {
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
            if (kt < nkt) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tm[kt * MT], bop[(kt * 4 + 2 * h) * 64 + lane], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tm[kt * MT + 4], bop[(kt * 4 + 2 * h + 1) * 64 + lane], acc, 0, 0, 0);
            }
        return acc;
    };
    // phase 1: X0^T[w] = sum_{kt <= w} T00[w][kt] A0^T[kt]   (B operand straight from the strip: A0[row c][k])
    d4 x = zero4;
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
        if (kt <= w) {
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(tm[kt * MT], Sb[c * TRSM_SS + kt * MT + g + 8 * h], x, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(tm[kt * MT + 4], Sb[c * TRSM_SS + kt * MT + g + 8 * h + 4], x, 0, 0, 0);
        }
    if (h == 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) Sc[(w * 4 + r) * 64 + lane] = x[r];
    }
    __syncthreads();                                    // T00 and A0 have been read; the half sums are in LDS
    d4 z = zero4;
    if (h == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double v = x[r] + Sc[(w * 4 + r) * 64 + lane];
            X0[(w * 4 + r) * 64 + lane] = v;
            Sb[c * TRSM_SS + w * MT + g + 4 * r] = v;                   // X0 is final: into the strip, over A0
            z[r] = Sb[c * TRSM_SS + 64 + w * MT + g + 4 * r];           // A1^T[w]
        }
    }
    fill_tm(rl, -1.0);
    __syncthreads();
    // phase 2: Z1^T[w] = A1^T[w] - sum_kt L10[w][kt] X0^T[kt]
    z = chain(z, X0, 4);
    if (h == 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) Sc[(w * 4 + r) * 64 + lane] = z[r];
    }
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) Z1[(w * 4 + r) * 64 + lane] = z[r] + Sc[(w * 4 + r) * 64 + lane];
    }
    fill_tm(r1, 1.0);
    __syncthreads();
    // phase 3: X1^T[w] = sum_{kt <= w} T11[w][kt] Z1^T[kt]
    d4 x1 = chain(zero4, Z1, w + 1);
    if (h == 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) Sc[(w * 4 + r) * 64 + lane] = x1[r];
    }
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) Sb[c * TRSM_SS + 64 + w * MT + g + 4 * r] = x1[r] + Sc[(w * 4 + r) * 64 + lane];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {                       // the solved strip back to global memory, whole rows
        const int q = t + 512 * i, row = q >> 6, cp = (q & 63) * 2;
        *(d2*)(Ag + (size_t)row * ld + cp) = *(const d2*)(Sb + row * TRSM_SS + cp);
    }
}

// ------------------------------------------------------------------------------------------
// SE covariance build: 64x64 tile per 256-thread workgroup, 4x4 outputs per thread, X tiles in LDS
// ------------------------------------------------------------------------------------------
constexpr int KT = 64;      // kernel-build tile
constexpr int DC = 16;      // feature chunk staged per pass

// the 4 columns of a thread's 4x4 micro-tile inside the 64-column tile: two adjacent pairs, 32 apart, so that the 16
// lanes of a row make one 256-byte run per 16-byte access (columns 4 tx + b made two half-used runs of 512 bytes)
__device__ __forceinline__ int col4(int tx, int b) { return (b >> 1) * 32 + tx * 2 + (b & 1); }

// squared distances of a 4x4 micro-tile, accumulated over d in index order without FMA
// contraction so that the value matches the reference's sub / mul / add sequence bit for bit
__device__ __forceinline__ void sqdist_4x4(const double* __restrict__ X, const double* __restrict__ Y, int nx,
                                           int ny, int d, int i0, int j0, double (&xs)[KT][DC + 1],
                                           double (&ys)[KT][DC + 1], double (&acc)[4][4])
{
#pragma clang fp contract(off)
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
    for (int c0 = 0; c0 < d; c0 += DC) {
        const int dc = (d - c0 < DC) ? (d - c0) : DC;
        __syncthreads();
        for (int e = t; e < KT * dc; e += 256) {
            int r = e / dc, c = e - r * dc;
            xs[r][c] = (i0 + r < nx) ? X[(size_t)(i0 + r) * d + c0 + c] : 0.0;
            ys[r][c] = (j0 + r < ny) ? Y[(size_t)(j0 + r) * d + c0 + c] : 0.0;
        }
        __syncthreads();
        for (int c = 0; c < dc; c++) {
            double xv[4], yv[4];
#pragma unroll
            for (int a = 0; a < 4; a++) { xv[a] = xs[ty * 4 + a][c]; yv[a] = ys[col4(tx, a)][c]; }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    double df = xv[a] - yv[b];
                    acc[a][b] = acc[a][b] + df * df;
                }
        }
    }
}

// a / b for many a and one b: with y = RN(1/b) from one real division, q0 = RN(a y), the exact remainder a - b q0 by
// FMA and q = RN(q0 + rem y) give the correctly rounded quotient (Markstein; 2e7 random pairs identical to a / b) in
// three instructions instead of the ~18 of a full IEEE division per matrix entry.  Only while b and 1/b are far
// from the ends of the exponent range (the optimisers do walk l^2 = exp(2 theta) to infinity: a / inf must stay 0,
// 0 * inf is NaN): DivBy::y == 0 selects the real division (uniform over the launch).
struct DivBy { double b, y; };
__device__ __forceinline__ DivBy div_prepare(double b)
{
    return DivBy{b, (b > 1e-100 && b < 1e100) ? 1.0 / b : 0.0};
}
__device__ __forceinline__ double div_by(double a, const DivBy& d)
{
    if (d.y == 0.0) return a / d.b;
    const double q0 = a * d.y;
    const double rem = __builtin_fma(-q0, d.b, a);
    return __builtin_fma(rem, d.y, q0);
}

// hd (when given): hyper-scalars resident in device memory -- a captured graph of the evaluation is replayed
// with new hyper-parameters by refreshing that one buffer instead of every kernel's arguments
__global__ __launch_bounds__(256) void k_build(const double* __restrict__ X, int n, int d, int npad,
                                               HyperScalars h_arg, const HyperScalars* __restrict__ hd,
                                               double* __restrict__ K, int full, unsigned* __restrict__ tickets,
                                               const ExpertPtrs* __restrict__ bt, unsigned long long* stamp)
{
    LaunchStamp stamp_(stamp, 16);
    if (bt) {
        X = GP(bt[blockIdx.y].X); n = bt[blockIdx.y].n; K = GP(bt[blockIdx.y].A);
        if (tickets) tickets = GP(bt[blockIdx.y].tickets);
    }
    // tickets (when given): the factorisation's per-step arrival counters, zeroed here instead of by a memset node in
    // front of the factorisation (one launch boundary less on a chain that small matrices are bound by)
    // (2 per tile row: [0, nt) the step tickets of k_syrk_step, [nt, 2 nt) the stage counters of k_trtri_block; [2 nt]: the
    //  arrival counter of k_trace's fused finalize)
    if (tickets && blockIdx.x == 0)
        for (int i = threadIdx.x; i < ticket_count(npad / TILE); i += 256) tickets[i] = 0u;
    const HyperScalars h = hd ? *hd : h_arg;
    __shared__ double xs[KT][DC + 1], ys[KT][DC + 1];
    int ti, tj;
    tri_index(blockIdx.x, ti, tj);
    const int i0 = ti * KT, j0 = tj * KT;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    double d2v[4][4];
    sqdist_4x4(X, X, n, n, d, i0, j0, xs, ys, d2v);
    double out[4][4];
    const DivBy dl = div_prepare(h.ell_sq);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = i0 + ty * 4 + a, j = j0 + col4(tx, b);
            const bool in = i < n && j < n;
            double v;
            if (full == 2) {
                v = (i == j) ? 0.0 : div_by(d2v[a][b], dl);    // covkernel.cpp:143-151 (squared distance / c)
            } else {
                v = h.signal_var * exp(div_by(-d2v[a][b] * 0.5, dl));   // covkernel.cpp:89
                if (i == j) v += h.noise_var;                            // covkernel.cpp:93-94
            }
            out[a][b] = in ? v : ((i == j) ? 1.0 : 0.0);                // identity padding
        }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        double* p = K + (size_t)(i0 + ty * 4 + a) * npad + j0 + tx * 2;
        *(d2*)p = (d2){out[a][0], out[a][1]};
        *(d2*)(p + 32) = (d2){out[a][2], out[a][3]};
    }
    if (full && ti != tj) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            double* p = K + (size_t)(j0 + col4(tx, b)) * npad + i0 + ty * 4;
            *(d2*)p = (d2){out[0][b], out[1][b]};
            *(d2*)(p + 2) = (d2){out[2][b], out[3][b]};
        }
    }
}

// Ks[t][i] = sf2 * exp(-0.5 |xt_t - x_i|^2 / l^2) (no noise, covkernel.cpp:105-116); zero padding
__global__ __launch_bounds__(256) void k_cross(const double* __restrict__ X, int n, int d, int npad,
                                               const double* __restrict__ Xt, int nt, int ntpad, HyperScalars h,
                                               double* __restrict__ Ks)
{
    __shared__ double xs[KT][DC + 1], ys[KT][DC + 1];
    const int tiles_i = npad / KT;
    const int tt = blockIdx.x / tiles_i, ti = blockIdx.x % tiles_i;
    const int t0 = tt * KT, i0 = ti * KT;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    double d2v[4][4];
    // the reference subtracts X[i] - xtest (covkernel.cpp:112); squares are sign-independent but keep the order
    sqdist_4x4(Xt, X, nt, n, d, t0, i0, xs, ys, d2v);
    const DivBy dl = div_prepare(h.ell_sq);
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int tr = t0 + ty * 4 + a;
        double o[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = i0 + col4(tx, b);
            o[b] = (tr < nt && i < n) ? h.signal_var * exp(div_by(-d2v[a][b] * 0.5, dl)) : 0.0;
        }
        double* p = Ks + (size_t)tr * npad + i0 + tx * 2;
        *(d2*)p = (d2){o[0], o[1]};
        *(d2*)(p + 32) = (d2){o[2], o[3]};
    }
}

// ------------------------------------------------------------------------------------------
// diagonal block: 128x128 Cholesky in LDS (one workgroup), 16-wide inner blocks
// ------------------------------------------------------------------------------------------
constexpr int TRTRI_LDS = NLT * MTS * 8;             // the 36 lower micro tiles

// ---- helpers for the diagonal block ------------------------------------------------------

__device__ __forceinline__ int mt_off(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * MTS; }

// broadcast of one lane's double through SGPRs (v_readlane x2); `src` must be wave-uniform
__device__ __forceinline__ double readlane_f64(double v, int src)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffull), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// 1/sqrt(x): v_rsq_f64 seed + two Newton steps; x <= 0 or NaN gives NaN/Inf that flows on
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const double e = __builtin_fma(-x * y, y, 1.0);
        y = __builtin_fma(0.5 * y, e, y);
    }
    return y;
}

// workgroup barrier that orders LDS traffic only: global stores in flight are NOT waited for (a __syncthreads
// would be: ~2k cycles at every barrier once the factored diagonal tiles go straight to global memory)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- panel factor: Cholesky of one 16x16 micro tile held in LDS (rows padded to 17) together with everything that
// hangs on it, by up to three waves side by side.  Lane l < 16 of every wave owns row l of the diagonal tile in
// registers (each wave factors the tile for itself: no hand-over between waves inside the pivot loop), right-looking.
// Every instruction of the loop runs on all 64 lanes, so lanes 16..63 carry OTHER ROWS through the same elimination
// for free: rows of the column block below the diagonal tile (48 per wave) come out as x = a L_jj^-T, and 16 lanes
// that start from the rows of the IDENTITY come out as the columns of L_jj^-1.
// One wave issues an fp64 VALU instruction per ~5.5-6.5 cycles and a dependent one per ~8.3, so the loop -- the
// latency floor of the whole factorisation -- is written for a short dependent chain and few instructions (round 3;
// measured by tools/panel_bench.hip: 16 pivots in 2.8k cycles, 175 per pivot; the round-2 loop: 4.2k, 265 per pivot):
//   * the pivot recurrence is in LDL^T form: the rank-1 update is  r[j] -= (m_i / d) m_j  with the UNSCALED column m,
//     which is final when the pivot starts -- its trip through LDS (one ds_write, uniform 16-byte reads back by all
//     64 lanes) starts at once and hides behind the rsq chain (the Cholesky form scales the column by 1/sqrt(d)
//     first, so that trip started at the END of the chain and its update had to wait a pivot);
//   * 1/sqrt(d): v_rsq_f64 seed (2^-24) + ONE third-order step y0 (1 + e/2 + 3e^2/8), e = 1 - d y0^2 (error
//     5/16 e^3 ~ 3e-24: below rounding); the stored factor is x = m y as before, m / d = x y;
//   * the chain per pivot: rsq, 5 polish ops, x, x y, ONE fma on column c+1, ONE readlane pair for the next pivot
//     (m_(c+1), the other operand of that fma, came back from LDS a pivot ago);
//   * software-pipelined by hand (one wave issues in order): the next pivot's v_rsq goes out first, the bulk of the
//     previous pivot's update (columns >= c+1) fills its latency and the gaps between the polish ops;
//   * every lane parks its column entry with ONE unmasked ds_write per pivot: rows of the diagonal tile into the
//     wave's column buffer, passenger rows into dead slots of their own LDS row (element 0, rewritten at the end, and
//     the pad element 16: the +128-byte immediate that alternates the two column buffers stays inside the 17-double
//     row), identity rows into parking slots of their own.  Spare lanes repeat rows (same values to the same
//     addresses) instead of being masked off.
// Two calls with a workgroup barrier between them: panel_load (everybody reads the diagonal tile) and panel_factor
// (the owner's rows go straight to global memory, the inverse replaces the diagonal tile in LDS).
struct PanelLanes {
    double* myrow;      // LDS row this lane loads from (rows of the factor: also where it stores to)
    double* wslot;      // where the lane parks its column entry at every pivot
    double* icol;       // identity rows: column `idx` of the diagonal tile in LDS (row stride MT + 1), else unused
    int kind;           // 0 row of the diagonal tile, 1 row of the column block below, 2 identity row
};

// zz: 64 doubles of LDS: [0, 32) a delta vector (1.0 at index 15: identity row i reads its 16 entries from zz + 15 - i,
// no selects), [32, 64) the parking slots of the identity rows.
// ident: this wave carries the 16 identity rows in its first spare lanes (the caller picks the wave with >= 16 of them):
// row i comes out as column i of L_jj^-1 -- the 16x16 inverse the 64x64 inverses are built from -- for free, every
// instruction of the pivot loop runs on all 64 lanes anyway.  The remaining spare lanes repeat rows (block rows, or the
// identity rows when the wave holds no block rows): same values to the same addresses.
__device__ __forceinline__ void panel_load(double* __restrict__ sm, int jb, int nrows, int q0, bool ident,
                                            double* colbuf, double* zz, PanelLanes& pl, double (&r)[MT])
{
    const int lane = threadIdx.x & 63;
    int nv = nrows - q0;                                       // block rows this wave holds: 0, 16, 32 or 48
    nv = nv < 0 ? 0 : (nv > 48 ? 48 : nv);
    int l = lane - MT;                                         // lanes >= 16: position among the wave's 48 passenger rows
    double* diagtile = sm + mt_off(jb, jb);
    if (lane < MT) {
        pl.kind = 0;
        pl.myrow = diagtile + lane * (MT + 1);
        pl.wslot = colbuf + lane;
        pl.icol = diagtile;
    } else {
        bool isid = ident && l >= nv && l < nv + MT;
        if (!isid && l >= nv) {                                // spare lane: repeat a block row, or an identity row
            if (nv > 0) { l -= nv; if (l >= nv) l -= nv; if (l >= nv) l -= nv; }
            else isid = true;
        }
        if (isid) {
            const int idx = (l - nv) & (MT - 1);
            pl.kind = 2;
            pl.myrow = zz + (MT - 1) - idx;
            pl.wslot = zz + 2 * MT + idx;
            pl.icol = diagtile + idx;
        } else {
            const int q = q0 + l;
            pl.kind = 1;
            pl.myrow = sm + mt_off(jb + 1 + (q >> 4), jb) + (q & 15) * (MT + 1);
            pl.wslot = pl.myrow;
            pl.icol = diagtile;
        }
    }
#pragma unroll
    for (int c = 0; c < MT; c++) r[c] = pl.myrow[c];
}

// arrive / arrive_target (when arrive_target > 0: several waves factor this panel): every factoring wave counts itself
// in *arrive once its copy of the diagonal tile is in registers; the wave with the identity rows overwrites the tile
// in LDS with the inverse only after it has seen all of them (a bounded wait that is over long before it is reached:
// the store comes 16 pivots after the loads) -- no workgroup barrier between panel_load and panel_factor.
__device__ __forceinline__ void panel_factor(const PanelLanes& pl, double (&r)[MT], double* __restrict__ gdiag, int ld,
                                              double* __restrict__ rinv, const double* colbuf, unsigned* arrive,
                                              unsigned arrive_target)
{
    // (colbuf is written through pl.wslot: no __restrict__)
    // Software-pipelined by hand, one wave issues in order: the next pivot's v_rsq goes out FIRST, the bulk of the
    // previous pivot's rank-1 update (columns >= c+1, operands long since back from LDS) fills its latency, then the
    // polish and the one fma + readlane pair that give the next pivot; as soon as column c+1 is final its LDS trip
    // and the broadcast of m_(c+2) start.  sched_barriers keep the compiler from sinking the bulk in front of the rsq.
    const int lane = threadIdx.x & 63;
    double piv = readlane_f64(r[0], 0);
    if (arrive_target > 0) {
        // (r[15], the last value panel_load asked for, is in its register: LDS returns a wave's reads in order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    double ys[MT];
    d2 mm[2][MT / 2];                                  // uniform column entries m_j of the pivot in flight / the one before
    double ltp = 0.0;                                  // -(m / d) of the previous pivot
    pl.wslot[0] = r[0];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int c2 = 0; c2 < MT; c2 += 2) mm[0][c2 / 2] = *(const d2*)(colbuf + c2);
#define CUGP_SB __builtin_amdgcn_sched_barrier(0)
    // k-th fma of the previous pivot's bulk update (columns c+1 .. 15, column c+1 first: the chain needs it)
#define CUGP_BULK(k)                                                                                       \
    do {                                                                                                   \
        if (c >= 1 && c + 1 + (k) < MT) {                                                                  \
            const int j_ = c + 1 + (k);                                                                    \
            r[j_] = __builtin_fma(ltp, mm[(c - 1) & 1][j_ / 2][j_ & 1], r[j_]);                            \
            CUGP_SB;                                                                                       \
        }                                                                                                  \
    } while (0)
#pragma unroll
    for (int c = 0; c < MT; c++) {
        const int nb = c >= 1 ? MT - 1 - c : 0;        // fmas of the previous pivot's bulk
        const int pre = nb > 7 ? nb - 7 : (nb < 3 ? nb : 3);   // in the shadow of the rsq; the rest one per chain op
        const double y0 = __builtin_amdgcn_rsq(piv);
        CUGP_SB;
#pragma unroll
        for (int k = 0; k < pre; k++) CUGP_BULK(k);
        const double t = -piv * y0;
        CUGP_SB;
        CUGP_BULK(pre + 0);
        const double e = __builtin_fma(t, y0, 1.0);
        CUGP_SB;
        CUGP_BULK(pre + 1);
        const double h = __builtin_fma(e, 0.375, 0.5);
        CUGP_SB;
        CUGP_BULK(pre + 2);
        const double qq = e * h;
        CUGP_SB;
        CUGP_BULK(pre + 3);
        const double y = __builtin_fma(y0, qq, y0);
        CUGP_SB;
        CUGP_BULK(pre + 4);
        const double x = r[c] * y;
        CUGP_SB;
        CUGP_BULK(pre + 5);
        const double lt = -x * y;                      // -(m / d)
        CUGP_SB;
        CUGP_BULK(pre + 6);
        ys[c] = y;
        if (c + 1 < MT) {
            r[c + 1] = __builtin_fma(lt, mm[c & 1][(c + 1) / 2][(c + 1) & 1], r[c + 1]);   // m_(c+1): back from LDS a pivot ago
            CUGP_SB;
            piv = readlane_f64(r[c + 1], c + 1);       // next pivot
            pl.wslot[((c + 1) & 1) * MT] = r[c + 1];   // column c+1 is final: on its way
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int c2 = (c + 2) & ~1; c2 < MT; c2 += 2) mm[(c + 1) & 1][c2 / 2] = *(const d2*)(colbuf + ((c + 1) & 1) * MT + c2);
        }
        r[c] = x;
        ltp = lt;
        CUGP_SB;
    }
#undef CUGP_BULK
#undef CUGP_SB
    if (pl.kind == 1) {
#pragma unroll
        for (int c = 0; c < MT; c++) pl.myrow[c] = r[c];
    } else if (pl.kind == 2) {
        // my row is column idx of L_jj^-1: the inverse replaces the diagonal tile in LDS (the factor itself goes to
        // global memory from the owner's registers)
        // (unbounded: the waves counted are co-resident waves of this workgroup, each of which counts itself right
        //  after its loads -- a bounded wait that fell through would overwrite the tile under a wave still loading it
        //  and give a silently wrong factor)
        if (arrive_target > 0)
            while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < arrive_target)
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int c = 0; c < MT; c++) pl.icol[c * (MT + 1)] = r[c];
    } else if (pl.kind == 0 && gdiag) {                                  // the owner wave: the factor's diagonal tile is final
        double* g = gdiag + (size_t)lane * ld;
        // (entries above the diagonal of a diagonal micro tile are never read on the device, and cugp_get_cholesky
        //  zeroes the strict upper triangle on the host: no masking)
#pragma unroll
        for (int c = 0; c < MT; c += 2) *(d2*)(g + c) = (d2){r[c], r[c + 1]};
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < MT; c += 2) *(d2*)(rinv + c) = (d2){ys[c], ys[c + 1]};
        }
    }
}

// C(bi,bj) -= sum_{p in [p0, p1)} X(bi,p) X(bj,p)^T on LDS micro tiles, one wave, NT tiles (bi, bi+1, .. of one
// column) side by side: every finished panel the tiles still lack in ONE pass over them (4 MFMAs per tile and panel,
// two accumulation chains each; the next panel's operands are requested before the current one's MFMAs, whose issue
// time (~550 cycles per tile from one wave) covers the LDS latency).  A tile alone pays ~400 cycles of prologue and
// epilogue (operand latency, the wait for its last MFMA before the store); two side by side share them -- the first
// tile's sums and stores issue while the second's MFMAs are still in flight.
template <int NT, int NP>   // NP = p1 - p0 panels, unrolled (as a rolled loop the compiler carried the accumulators in
                            // VGPRs and copied all of them to the MFMA's AGPRs and back on every trip)
__device__ __forceinline__ void micro_update_np(double* __restrict__ sm, int bi, int bj, int p0)
{
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    double* C[NT];
    const double* Xi[NT];
    const double* Xj = sm + mt_off(bj, p0) + c * (MT + 1) + g;          // tiles (b, p), (b, p+1) are MTS doubles apart
    d4 acc[NT], acc2[NT];
    double a[NP][NT][4], b[NP][4];
#pragma unroll
    for (int s = 0; s < 4; s++) b[0][s] = Xj[4 * s];
#pragma unroll
    for (int n = 0; n < NT; n++) {
        C[n] = sm + mt_off(bi + n, bj);
        Xi[n] = sm + mt_off(bi + n, p0) + c * (MT + 1) + g;
#pragma unroll
        for (int s = 0; s < 4; s++) a[0][n][s] = -Xi[n][4 * s];
#pragma unroll
        for (int r = 0; r < 4; r++) acc[n][r] = C[n][(g + 4 * r) * (MT + 1) + c];
        acc2[n] = (d4){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        if (p + 1 < NP) {                                                // next panel's operands on their way
#pragma unroll
            for (int s = 0; s < 4; s++) b[p + 1][s] = Xj[(p + 1) * MTS + 4 * s];
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < 4; s++) a[p + 1][n][s] = -Xi[n][(p + 1) * MTS + 4 * s];
        }
#pragma unroll
        for (int n = 0; n < NT; n++) {
            acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p][n][0], b[p][0], acc[n], 0, 0, 0);
            acc2[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p][n][1], b[p][1], acc2[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < NT; n++) {
            acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p][n][2], b[p][2], acc[n], 0, 0, 0);
            acc2[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p][n][3], b[p][3], acc2[n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) C[n][(g + 4 * r) * (MT + 1) + c] = acc[n][r] + acc2[n][r];
}

template <int NT>
__device__ __forceinline__ void micro_update_multi(double* __restrict__ sm, int bi, int bj, int p0, int p1)
{
    switch (p1 - p0) {
    case 1: micro_update_np<NT, 1>(sm, bi, bj, p0); break;
    case 2: micro_update_np<NT, 2>(sm, bi, bj, p0); break;
    case 3: micro_update_np<NT, 3>(sm, bi, bj, p0); break;
    case 4: micro_update_np<NT, 4>(sm, bi, bj, p0); break;
    case 5: micro_update_np<NT, 5>(sm, bi, bj, p0); break;
    case 6: micro_update_np<NT, 6>(sm, bi, bj, p0); break;
    default: break;
    }
}

// The same for NB tiles at once (tile n of the step = the n-th of (bj = jb+1.., bi = bj..7) in that order): all
// operands of the batch are requested before the first MFMA and the NB x 4 MFMAs are independent, so the LDS
// latency and the MFMA latency of one tile hide behind the others (one tile at a time took ~1100 cycles, 4 MFMAs
// of 64).  Tiles past the end of the step (n >= ntiles) are skipped wave-uniformly.
template <int NB>
__device__ __forceinline__ void micro_update_batch(double* __restrict__ sm, int jb, int n0, int stride, int nend)
{
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int mm = NMT - 1 - jb;                       // tiles per edge of the trailing part
    double* C[NB];
    const double *Xi[NB], *Xj[NB];
    bool on[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int n = __builtin_amdgcn_readfirstlane(n0 + b * stride);
        on[b] = n < nend;
        // n -> (column cj, row ri) of the trailing lower triangle, column by column: column cj holds mm - cj tiles
        int cj = 0, rem = on[b] ? n : 0;
        while (rem >= mm - cj) { rem -= mm - cj; cj++; }
        const int bj = jb + 1 + cj, bi = bj + rem;
        C[b] = sm + mt_off(bi, bj);
        Xi[b] = sm + mt_off(bi, jb);
        Xj[b] = sm + mt_off(bj, jb);
    }
    d4 acc[NB], acc2[NB];
    double xa[NB][4], xb[NB][4];
#pragma unroll
    for (int b = 0; b < NB; b++) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc[b][r] = C[b][(g + 4 * r) * (MT + 1) + c];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            xa[b][s] = -Xi[b][c * (MT + 1) + 4 * s + g];
            xb[b][s] = Xj[b][c * (MT + 1) + 4 * s + g];
        }
        acc2[b] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    // every operand of the batch is on its way before the first MFMA waits for one (left alone the scheduler
    // interleaves loads and MFMAs tile by tile with a full LDS wait in front of each)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; s += 2)
#pragma unroll
        for (int b = 0; b < NB; b++) {
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[b][s], xb[b][s], acc[b], 0, 0, 0);
            acc2[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[b][s + 1], xb[b][s + 1], acc2[b], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NB; b++)
        if (on[b]) {
#pragma unroll
            for (int r = 0; r < 4; r++) C[b][(g + 4 * r) * (MT + 1) + c] = acc[b][r] + acc2[b][r];
        }
}

// tiles n = n0, n0 + stride, ... < nend of step jb's rank-16 update by the calling wave, four at a time
__device__ __forceinline__ void micro_update_run(double* __restrict__ sm, int jb, int n0, int stride, int nend)
{
    for (int n = n0; n < nend; n += 4 * stride) {
        if (n + 2 * stride < nend) micro_update_batch<4>(sm, jb, n, stride, nend);
        else if (n + stride < nend) micro_update_batch<2>(sm, jb, n, stride, nend);
        else micro_update_batch<1>(sm, jb, n, stride, nend);
    }
}

// ------------------------------------------------------------------------------------------
// diagonal block: 128x128 Cholesky by one workgroup.  The lower triangle lives in LDS as 36
// padded 16x16 micro tiles (78 KiB: fits beside one resident MFMA workgroup on the CU).
// Per 16-wide inner step: (A) micro_factor of the diagonal micro tile in registers (one wave),
// (B) the rows below by per-row forward substitution (L_jj broadcast from LDS), (C) rank-16
// update of the remaining micro tiles with MFMA -- wave 0 takes the next diagonal tile first and
// factors it while waves 1..3 finish the update (look-ahead inside the block).
// Outputs: L (lower) back into A, the 16x16 diagonal inverses (d16, used by the panel solve and
// the inverse), and this block's share of log|K|.
// ------------------------------------------------------------------------------------------
#ifdef CUGP_STAMPS   // diagnostic build only (tools/chain_bench.hip): cycle stamps of wave 0 into a side buffer
__device__ unsigned long long g_stamps[128];
#define WSTAMP(i)                                                                  \
    do {                                                                           \
        if ((threadIdx.x & 63) == 0) {                                             \
            unsigned long long t_;                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            g_stamps[i] = t_;                                                      \
        }                                                                          \
    } while (0)
#define STAMP(i)                                                                   \
    do {                                                                           \
        if (threadIdx.x == 0) {                                                    \
            unsigned long long t_;                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            g_stamps[i] = t_;                                                      \
        }                                                                          \
    } while (0)
#else
#define STAMP(i)
#define WSTAMP(i)
#endif

// one MFMA 16x16x16 product on LDS micro tiles, "NN": acc += A(tile a)[row][k] * B(tile b)[k][col]
__device__ __forceinline__ d4 micro_mma_nn(const double* __restrict__ a, const double* __restrict__ b, d4 acc)
{
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; s++)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c * (MT + 1) + 4 * s + g], b[(4 * s + g) * (MT + 1) + c], acc, 0,
                                                   0, 0);
    return acc;
}

// acc -= A(tile a)[row][k] * W[k][col] where W is a previous MFMA result held in registers (C/D layout):
// register r of lane (col, g) is W[g + 4r][col], i.e. already the B operand for k = g + 4r
__device__ __forceinline__ d4 micro_mma_acc_b(const double* __restrict__ a, d4 w, d4 acc)
{
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[c * (MT + 1) + g + 4 * r], w[r], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ void micro_store(double* __restrict__ tile, d4 v)
{
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) tile[(g + 4 * r) * (MT + 1) + c] = v[r];
}

// Which finished panels the helper waves (the ones not factoring) bring into which micro tiles while panel q is being
// factored.  Column q+1 gets ALL panels older than q in one pass over its tiles, one phase before it is needed (panel q
// itself follows in U1, by every wave, as soon as it exists): 4q MFMAs per tile, at most 24 per helper wave -- the
// ~3.3k cycles one wave needs to issue them fit under the ~3.5k cycles of the 16 pivots, in every phase.  (The
// right-looking order this replaces gave the two helper waves of the first phases 21 and 15 tiles: 9k and 7k cycles.)
// Two range tasks per (phase, helper): rows [bi0, bi1] of column bj take panels [p0, p1); 15 bits each.
__device__ __forceinline__ unsigned helper_tasks(int q, int h)
{
#define CUGP_T(bi0, bi1, bj, p0, p1) ((unsigned)((bi0) | (bi1) << 3 | (bj) << 6 | (p0) << 9 | (p1) << 12))
    switch (q * 4 + h) {
    case 1 * 4 + 0: return CUGP_T(3, 7, 2, 0, 1);                                      // 20 MFMAs (the only helper; tile (2,2) rode in U1(0))
    case 2 * 4 + 0: return CUGP_T(3, 5, 3, 0, 2);                                      // 24
    case 2 * 4 + 1: return CUGP_T(6, 7, 3, 0, 2) | CUGP_T(7, 7, 5, 0, 2) << 16;        // 16 + 8 (ahead of phase 4)
    case 3 * 4 + 0: return CUGP_T(4, 5, 4, 0, 3);                                      // 24
    case 3 * 4 + 1: return CUGP_T(6, 7, 4, 0, 3);                                      // 24
    case 4 * 4 + 0: return CUGP_T(5, 5, 5, 0, 4) | CUGP_T(7, 7, 5, 2, 4) << 16;        // 16 + 8
    case 4 * 4 + 1: return CUGP_T(6, 6, 5, 0, 4);                                      // 16
    case 5 * 4 + 0: return CUGP_T(6, 6, 6, 0, 5);                                      // 20
    case 5 * 4 + 1: return CUGP_T(7, 7, 6, 0, 5);                                      // 20
    case 6 * 4 + 0: return CUGP_T(7, 7, 7, 0, 6);                                      // 24
    default: return 0u;
    }
#undef CUGP_T
}

template <bool HANDED = false>   // HANDED: the block was written by other workgroups of this launch (agent-scope loads)
__device__ __forceinline__ void potf2_body(double* __restrict__ Ab, int ld, double* __restrict__ d16blk,
                                           double* __restrict__ d64blk, double* __restrict__ logdet_out,
                                           double* __restrict__ sm, double* __restrict__ red)
{
    double* rinv = sm + NLT * MTS;                          // 1 / L_ii, 128 entries
    double* colbuf = red;                                   // pivot columns: 2 x 16 doubles for each of the 4 waves
    double* zz = red + TILE;                                // delta vector + parking slots of the identity rows (panel_load)
    unsigned* arrive = (unsigned*)(zz + 2 * MT - 1);        // factoring waves that hold their copy of the diagonal tile (panel_factor)
    unsigned arrived = 0;                                   // ... expected once every wave of the phases so far has counted itself
    (void)d16blk;                                           // (16x16 inverses live in LDS only)
    // (the wave index through readfirstlane: the compiler then knows that everything decided by it -- which tiles
    //  a wave updates, which rows it factors -- is wave-uniform and keeps that control flow on the scalar unit)
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    __builtin_amdgcn_s_setprio(3);                          // this workgroup is the critical path of the step

    STAMP(0);
    {   // load the lower micro tiles; thread t = element (t>>4, t&15) of every tile.  All 36 loads in flight before
        // the first LDS store (fully unrolled; as a rolled loop they went out a few at a time: 3.9k cycles)
        const int r = t >> 4, c = t & 15;
        double v[NLT];
#pragma unroll
        for (int bi = 0; bi < NMT; bi++)
#pragma unroll
            for (int bj = 0; bj <= bi; bj++) {
                const double* src = Ab + (size_t)(bi * MT + r) * ld + bj * MT + c;
                v[bi * (bi + 1) / 2 + bj] = HANDED ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
            }
#pragma unroll
        for (int q = 0; q < NLT; q++) sm[q * MTS + r * (MT + 1) + c] = v[q];
        if (t < 2 * MT) zz[t] = t == MT - 1 ? 1.0 : 0.0;    // (zz[31], never read as a double, is the arrival counter: 0)
    }
    __syncthreads();
    STAMP(1);
    // Phase q = 0..7:  P(q) panel factor of micro-tile column q -- the diagonal tile, the rows below it (48 per wave)
    // and 16 identity rows that come out as the tile's 16x16 inverse (panel_load / panel_factor) -- by waves 0..iw,
    // iw = rows / 48: the last of them is the one with >= 16 spare lanes (or holds the identity rows alone).  Beside
    // it the other waves run H(q): the older panels into column q+1 (helper_tasks).  Then U1(q), every wave: panel q
    // into column q+1, and on to P(q+1).
    // (barriers in this loop order LDS traffic only -- lds_barrier: the diagonal tiles' global stores drain behind them)
    PanelLanes pl;
    double pr[MT];
    double* gblk = Ab;                                       // global home of the 128x128 block (row stride ld)
    const d4 zero4 = (d4){0.0, 0.0, 0.0, 0.0};
    d4 hw0 = zero4, hw1 = zero4;                            // products a helper wave carries across a barrier
    // one off-diagonal micro tile of the factor back to global memory, by one wave (its LDS home is about to be
    // overwritten by a piece of an inverse, or the wave has nothing else to do)
    auto store_tile = [&](int bi, int bj) {
        const int lane = t & 63, r = lane >> 3, c = (lane & 7) * 2;             // two adjacent entries per lane: 16-byte stores
        const double* src = sm + mt_off(bi, bj) + r * (MT + 1) + c;
        double* dst = Ab + (size_t)(bi * MT + r) * ld + bj * MT + c;
#pragma unroll
        for (int k = 0; k < 2; k++)
            *(d2*)(dst + (size_t)(8 * k) * ld) = (d2){src[8 * k * (MT + 1)], src[8 * k * (MT + 1) + 1]};
    };
    // 16 -> 32: T21 = -T_B (L21 T_A) for the pair of diagonal micro tiles a = 2p, b = 2p + 1, in place of tile (b,a)
    auto pair_double = [&](int p2) {
        const int a = 2 * p2, b = 2 * p2 + 1;
        store_tile(b, a);
        const d4 w = micro_mma_nn(sm + mt_off(b, a), sm + mt_off(a, a), zero4);
        const d4 t21 = micro_mma_acc_b(sm + mt_off(b, b), w, zero4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        micro_store(sm + mt_off(b, a), t21);
    };
    // 32 -> 64 of 64-block h, column bj of its lower-left 2x2 micro tiles, first half:
    // W(kb', bj) = sum_{jb' >= bj} C(kb', jb') T_A(jb', bj)  (reads only)
    auto block_w = [&](int h, int bj, d4& w0, d4& w1) {
        const int a0 = 4 * h, b0 = 4 * h + 2;
        w0 = zero4; w1 = zero4;
        for (int jp = bj; jp < 2; jp++) {
            w0 = micro_mma_nn(sm + mt_off(b0, a0 + jp), sm + mt_off(a0 + jp, a0 + bj), w0);
            w1 = micro_mma_nn(sm + mt_off(b0 + 1, a0 + jp), sm + mt_off(a0 + jp, a0 + bj), w1);
        }
    };
    // ... second half: T(b0 + i, a0 + bj) = -sum_{k <= i} T_B(i, k) W(k, bj)  (t0, t1 still to be stored)
    auto block_t = [&](int h, const d4& w0, const d4& w1, d4& t0, d4& t1) {
        const int b0 = 4 * h + 2;
        t0 = micro_mma_acc_b(sm + mt_off(b0, b0), w0, zero4);
        t1 = micro_mma_acc_b(sm + mt_off(b0 + 1, b0), w0, zero4);
        t1 = micro_mma_acc_b(sm + mt_off(b0 + 1, b0 + 1), w1, t1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto block_store = [&](int h, int bj, const d4& t0, const d4& t1) {
        micro_store(sm + mt_off(4 * h + 2, 4 * h + bj), t0);
        micro_store(sm + mt_off(4 * h + 3, 4 * h + bj), t1);
    };
    // one 64x64 inverse to global, row-major (micro tiles above the diagonal are never read), by nw waves from wave w0
    auto store_d64 = [&](int h, int w0, int nw) {
        const int u = t - w0 * 64, nth = nw * 64;
#pragma unroll 5
        for (int q = 0; q < 10; q++) {
            const int bi = q < 1 ? 0 : (q < 3 ? 1 : (q < 6 ? 2 : 3)), bj = q - bi * (bi + 1) / 2;
            for (int e = u; e < 128; e += nth) {                                // 128 pairs of adjacent entries per tile
                const int r = e >> 3, c = (e & 7) * 2;
                const double* src = sm + mt_off(4 * h + bi, 4 * h + bj) + r * (MT + 1) + c;
                *(d2*)(d64blk + (size_t)h * 4096 + (bi * MT + r) * 64 + bj * MT + c) = (d2){src[0], src[1]};
            }
        }
    };
    for (int q = 0; q < NMT; q++) {
        const int rows = (NMT - 1 - q) * MT, iw = rows / 48;
        if (wave <= iw) {
            panel_load(sm, q, rows, wave * 48, wave == iw, colbuf + wave * 2 * MT, zz, pl, pr);
            double* gd = wave == 0 ? gblk + (size_t)q * MT * ld + q * MT : nullptr;
            panel_factor(pl, pr, gd, ld, rinv + q * MT, colbuf + wave * 2 * MT, arrive, iw > 0 ? arrived + iw + 1 : 0u);
        } else {
            const int h = wave - iw - 1;
            unsigned code = helper_tasks(q, h);
            for (; code & 0x7fffu; code >>= 16) {
                const int bi1 = code >> 3 & 7, bj = code >> 6 & 7, p0 = code >> 9 & 7, p1 = code >> 12 & 7;
                int bi = code & 7;
                for (; bi + 1 <= bi1; bi += 2) micro_update_multi<2>(sm, bi, bj, p0, p1);
                if (bi <= bi1) micro_update_multi<1>(sm, bi, bj, p0, p1);
            }
            // The spare time of the helpers in the last phases goes to what used to follow the factorisation: the
            // finished columns of the factor back to global memory, and the doubling of the 16x16 inverses
            // (16 -> 32 -> 64) as far as the finished diagonal tiles allow.  A tile is stored before a piece of an
            // inverse takes its place in LDS; every such piece is read only from the next phase on (barriers between).
            if (q == 4 && h == 1) pair_double(0);
            if (q == 5) {                                           // columns 0..3 of the factor are final
                if (h == 2) {
#pragma nounroll
                    for (int bj = 0; bj < 2; bj++)
#pragma unroll 3
                        for (int bi = 2; bi < NMT; bi++) store_tile(bi, bj);    // ((1,0) went out with its pair in phase 4)
                    pair_double(1);                                 // (stores (3,2) itself)
                } else {
                    const int bj = 2 + h;                           // h = 0: column 2 below (3,2); h = 1: column 3
#pragma unroll 4
                    for (int bi = 4; bi < NMT; bi++) store_tile(bi, bj);
                }
            }
            if (q == 6 && h >= 1) {                                 // 64-block 0: column bj = h - 1
                block_w(0, h - 1, hw0, hw1);
                d4 t0, t1;
                block_t(0, hw0, hw1, t0, t1);
                // column 0 at once (wave h = 2 reads only column 1 of these tiles); column 1 after the barrier
                // (wave h = 1 reads it)
                if (h == 1) block_store(0, 0, t0, t1);
                else { hw0 = t0; hw1 = t1; }
            }
            if (q == 7) {
                if (h == 0) { pair_double(2); block_w(1, 0, hw0, hw1); }      // (4,5), then W(., 0) of 64-block 1
                if (h == 1) { block_w(1, 1, hw0, hw1); store_d64(0, 2, 1); }
                if (h == 2) {
                    store_tile(6, 4); store_tile(7, 4); store_tile(6, 5); store_tile(7, 5); store_tile(7, 6);
                    hw0 = micro_mma_nn(sm + mt_off(7, 6), sm + mt_off(6, 6), zero4);   // first half of the pair (6,7)
                    // log-determinant share of the first seven diagonal tiles: sum_i log(1/L_ii), i < 112, fixed order
                    const int lane = t & 63;
                    double v = log(rinv[lane]) + (lane < 48 ? log(rinv[lane + 64]) : 0.0);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                    if (lane == 0) colbuf[3 * 2 * MT] = v;          // (this wave's column buffer is idle: it never factors)
                }
            }
        }
        if (iw > 0) arrived += iw + 1;
        WSTAMP(64 + q * 4 + wave);                         // (diagnostic build: when each wave reaches the barrier)
        lds_barrier();
        STAMP(2 + 2 * q);
        if (q == 6 && wave == 3) block_store(0, 1, hw0, hw1);
        if (q + 1 < NMT) {
            // U1: panel q into column q+1 = tiles 0 .. m-1 of its update (phase 0: and tile (2,2), for the one wave
            // that would otherwise hold a single tile)
            micro_update_run(sm, q, wave, 4, NMT - 1 - q + (q == 0 ? 1 : 0));
            lds_barrier();
        }
        STAMP(3 + 2 * q);
    }
    lds_barrier();                                          // the last tile's inverse is in LDS
    STAMP(30);
    // what is left of the inverses: second half of the pair (6,7), then of 64-block 1 (its W came from phase 7)
    if (wave == 3) {
        const d4 t21 = micro_mma_acc_b(sm + mt_off(7, 7), hw0, zero4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        micro_store(sm + mt_off(7, 6), t21);
    }
    // log-determinant share of this block: sum_i log L_ii = -sum_i log(1/L_ii); the last tile's 16 terms join the
    // partial sum of phase 7 (fixed order)
    if (wave == 0) {
        const int lane = t & 63;
        double v = lane < MT ? log(rinv[TILE - MT + lane]) : 0.0;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) *logdet_out = -(colbuf[3 * 2 * MT] + v);
    }
    lds_barrier();
    STAMP(31);
    if (wave == 1 || wave == 2) {
        d4 t0, t1;
        block_t(1, hw0, hw1, t0, t1);
        block_store(1, wave - 1, t0, t1);
    }
    lds_barrier();
    STAMP(32);
    store_d64(1, 0, 4);
    STAMP(33);
}

__global__ __launch_bounds__(256) void k_potf2(double* __restrict__ A, int ld, int kb, double* __restrict__ d16,
                                               double* __restrict__ d64, double* __restrict__ logdet_part,
                                               const ExpertPtrs* __restrict__ bt)
{
    if (bt) { A = GP(bt[blockIdx.y].A); d16 = GP(bt[blockIdx.y].d16); d64 = GP(bt[blockIdx.y].d64); logdet_part = GP(bt[blockIdx.y].logdet); }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[TILE + 4 * MT];
    potf2_body(A + (size_t)kb * TILE * ld + kb * TILE, ld, d16 + (size_t)kb * NMT * (MT * MT),
               d64 + (size_t)kb * 8192, logdet_part + kb, sm, red);
}

// one 16x16 micro tile of A(kb+1,kb+1) -= sum_{k in [ks, kb]} L(kb+1,k) L(kb+1,k)^T by ONE workgroup, operands straight
// from L2: the four waves take a quarter of the k range each (8 MFMAs per k tile in two chains; one wave issues an fp64
// MFMA per ~138 cycles, so the whole K = 128 on one wave was 32 x 138 = 4.4k cycles of the chain), partial sums through
// LDS, added in a fixed order by wave 0, which stores the tile write-through (the consumer is another workgroup).
// NK = kb + 1 - ks k tiles (sub-panelled near window, enqueue_potrf: the tile receives the whole sub-panel so far in
// this one pass); every operand of the wave is requested before its first MFMA.
template <int NK>
__device__ __forceinline__ void diag_update_tile_nk(double* __restrict__ A, int ld, int kb, int mtile, double* __restrict__ part)
{
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bi, bj;
    tri_index(mtile, bi, bj);
    const int c = lane & 15, g = lane >> 4;
    const int k0 = (kb + 1 - NK) * TILE + 32 * NK * w, i0 = (kb + 1) * TILE;
    const double* Li = A + (size_t)(i0 + bi * MT + c) * ld + k0 + g;
    const double* Lj = A + (size_t)(i0 + bj * MT + c) * ld + k0 + g;
    double* C = A + (size_t)(i0 + bi * MT + g) * ld + i0 + bj * MT + c;
    double la[8 * NK], lb[8 * NK];                      // all operands in flight before the first MFMA
#pragma unroll
    for (int s = 0; s < 8 * NK; s++) { la[s] = -Li[4 * s]; lb[s] = Lj[4 * s]; }
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0}, acc2 = acc;
    if (w == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = C[(size_t)(4 * r) * ld];
    }
#pragma unroll
    for (int s = 0; s < 8 * NK; s += 2) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[s], lb[s], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[s + 1], lb[s + 1], acc2, 0, 0, 0);
    }
    acc += acc2;
    if (w > 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) part[((w - 1) * 4 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double v = ((acc[r] + part[r * 64 + lane]) + part[(4 + r) * 64 + lane]) + part[(8 + r) * 64 + lane];
            __hip_atomic_store(C + (size_t)(4 * r) * ld, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ void diag_update_tile(double* __restrict__ A, int ld, int kb, int ks, int mtile, double* __restrict__ part)
{
    switch (kb + 1 - ks) {                              // (uniform over the launch)
    case 1: diag_update_tile_nk<1>(A, ld, kb, mtile, part); break;
    case 2: diag_update_tile_nk<2>(A, ld, kb, mtile, part); break;
    case 3: diag_update_tile_nk<3>(A, ld, kb, mtile, part); break;
    default: diag_update_tile_nk<4>(A, ld, kb, mtile, part); break;
    }
}

// ------------------------------------------------------------------------------------------
// One launch per factorisation step kb: the trailing update A22 -= L21 L21^T AND, inside it, the
// factorisation of the NEXT diagonal block.  Workgroups 0..35 (dispatched first) update one micro tile of
// tile (kb+1,kb+1) each, straight from L2; the last of them to finish (a ticket at agent scope, no spinning)
// goes on to factor that block (potf2_body) while the other workgroups of the launch run the MFMA tile
// products of the rest of the trailing matrix.  The latency-bound diagonal block therefore never waits for
// a free CU slot and needs no second stream.
// (Round 3 also ran the panel solve inside this launch -- strips first, the diagonal workgroups and the tile
//  products waiting on per-row counters: every tile workgroup then needs an agent-scope acquire (per-XCD L2s
//  are not coherent) and the hand-offs cost what the kernel boundary did: 41.7 vs 43.5 us per chain-bound
//  step, Cholesky 5.86 vs 5.52 ms at N=8192.  Not kept.)
// ------------------------------------------------------------------------------------------
constexpr int NDIAGWG = NLT;                          // 36 workgroups, one micro tile of the next diagonal block each
constexpr int STEP_LDS = POTF2_LDS;                   // >= GEMM_LDS (66048); two such workgroups still fit one CU
static_assert(POTF2_LDS >= GEMM_LDS, "the fused step kernel sizes its LDS for both roles");

// wcol (two-speed form, enqueue_potrf): the launch updates only the tile columns [kb+1, kb+1+wcol) -- the near
// window -- and the far columns are brought up to date once per panel by k_syrk_wide with K = P*128.
// wcol >= the trailing size: the classic full update.
// ks (sub-panelled near window): the launch subtracts the k tiles [ks, kb] in one pass, K = (kb + 1 - ks) * 128 --
// the columns it touches have not seen any of them yet (plan_step, cugp_capi.cpp).  ks = kb: one k tile per step.
__global__ __launch_bounds__(256, 2) void k_syrk_step(double* __restrict__ A, int ld, int kb,
                                                      double* __restrict__ d16, double* __restrict__ d64,
                                                      double* __restrict__ logdet_part,
                                                      unsigned* __restrict__ tickets, int nfull, int wcol,
                                                      int ks, const ExpertPtrs* __restrict__ bt,
                                                      unsigned long long* stamp, int vec0, const double* __restrict__ zv,
                                                      double* __restrict__ wv)
{
    LaunchStamp stamp_(stamp);
    // batched: the EXPERT is the fast grid index, so the diagonal-block workgroups of all experts are
    // dispatched before any tile product (the serial chain of every expert starts at launch)
    const int bid = bt ? blockIdx.y : blockIdx.x;
    if (bt) {
        const ExpertPtrs& e = bt[blockIdx.x];
        A = GP(e.A); d16 = GP(e.d16); d64 = GP(e.d64); logdet_part = GP(e.logdet); tickets = GP(e.tickets);
        if (zv) { zv = GP(e.z); wv = GP(e.w); }
    }
    // workgroups from vec0 on (only launched when zv is given): the forward substitution's update of the running
    // right-hand side with the column this step consumes, one workgroup per tile row below it
    if (zv && bid >= vec0) {
        vec_update(A, ld, kb, kb + 1 + (bid - vec0), zv, wv);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double red[TILE + 4 * MT];
    __shared__ unsigned s_ticket;
    if (bid < NDIAGWG) {
        // the factorisation's serial chain: win instruction issue over the product waves sharing the SIMD
        // (this launch's own tiles and the inverse blocks running on the other streams)
        __builtin_amdgcn_s_setprio(3);
        WGT(wgt_, WGT_DIAGUPD, kb);
        diag_update_tile(A, ld, kb, ks, bid, sm);
        // publish: wave 0 drains its (agent-scope, write-through) tile stores, then ONE of its lanes draws the
        // ticket (relaxed: the tile is in memory before the ticket, and the last arriver reads the tiles with
        // agent-scope loads that bypass its L1 -- the hand-off form of the CDNA guide's Guideline 16, R1)
        // This form leans on gfx950 behaviour the HIP memory model does not promise: vmcnt counts stores as well as
        // loads (no separate store counter) and sc1 stores / loads are coherent across the XCDs' L2s.  EVERY read of
        // the handed-over block must be one of the agent-scope loads of potf2_body<true>.  The library is built for
        // gfx950 only (cugp_amd/build.py); another target must not compile this silently:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_syrk_step's fence-free ticket hand-off is validated on gfx950 only: use an acq_rel ticket + agent acquire fence elsewhere"
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0)
            s_ticket = __hip_atomic_fetch_add(&tickets[kb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != NDIAGWG - 1) return;           // not the last arriver
        // (the last arriver reads the 36 micro tiles with agent-scope loads: potf2_body<true>, no acquire fence)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int kn = kb + 1;
        WGT(wgt2_, WGT_POTF2, kb);
        potf2_body<true>(A + (size_t)kn * TILE * ld + kn * TILE, ld, d16 + (size_t)kn * NMT * (MT * MT),
                   d64 + (size_t)kn * 8192, logdet_part + kn, sm, red);
        return;
    }
    // tile 0 = (kb+1,kb+1) is the diagonal one above.  Odd steps walk the tiles backwards so the tiles
    // written last by step kb (still in the 256 MiB Infinity Cache) are the first read by step kb+1.
    // Regular tiles: `nfull` of them as 128x128 workgroups; the rest (a partial last round that would
    // leave most of the chip idle for a whole tile time) as four 64x64 workgroups each.
    __builtin_amdgcn_s_setprio(1);                      // ahead of the inverse-block products (priority 0)
    WGT(wgt_, WGT_STEPTILE, kb);
    const int x = bid - NDIAGWG;
    if (x < nfull) {
        // Workgroups are dealt round-robin over the 8 XCDs (private L2 each): give every XCD one contiguous
        // run of the tile list, so neighbouring tiles (same panel rows) share an L2 (bijective for any count).
        // Odd steps walk backwards: the tiles written last by step kb are the first read by step kb+1.
        int ti, tj;
        const int xg = x & 7, xq = nfull >> 3, xr = nfull & 7;
        const int tlin = (xg < xr ? xg * (xq + 1) : xr * (xq + 1) + (xg - xr) * xq) + (x >> 3) + 1;
        trap_index((kb & 1) ? nfull + 1 - tlin : tlin, wcol, ti, tj);   // tile 0 = (kb+1,kb+1) is the diagonal one
        const int i0 = (kb + 1 + ti) * TILE, j0 = (kb + 1 + tj) * TILE;
        d4 acc[4][4];
        acc_zero(acc);
        tile_nt<false>(A + (size_t)i0 * ld, ld, A + (size_t)j0 * ld, ld, ks * TILE, (kb + 1) * TILE, acc, (char*)sm);
        tile_accum_store<-1>(A + (size_t)i0 * ld + j0, ld, acc);
    } else {
        int ti, tj;
        const int y = x - nfull;
        trap_index(nfull + 1 + (y >> 2), wcol, ti, tj);
        const int i0 = (kb + 1 + ti) * TILE + ((y >> 1) & 1) * 64, j0 = (kb + 1 + tj) * TILE + (y & 1) * 64;
        d4 acc[2][2];
        acc_zero(acc);
        tile_nt<false>(A + (size_t)i0 * ld, ld, A + (size_t)j0 * ld, ld, ks * TILE, (kb + 1) * TILE, acc, (char*)sm);
        tile_accum_store<-1>(A + (size_t)i0 * ld + j0, ld, acc);
    }
}

// ------------------------------------------------------------------------------------------
// Wide trailing update (look-ahead Cholesky): A(ti,tj) -= sum_{k in [k0, k0+kw)} L(ti,k) L(tj,k)^T for the tile
// columns tj in [ca, cb), ti >= tj -- ONE pass over the C tiles with K = kw*128 (the tile product costs a fixed
// ~12 us per C tile + 30.5 us per 128 of K: 50 TF/s at K=128, 66 at K=512 when the tiles fill whole rounds of
// the 512 workgroup slots).  One workgroup per tile; a small last round runs as 64x64 quarters.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_syrk_wide(double* __restrict__ A, int ld, int k0, int kw, int ca,
                                                      int cb, int ntiles, int nfull, int rev,
                                                      const ExpertPtrs* __restrict__ bt, unsigned long long* stamp)
{
    LaunchStamp stamp_(stamp);
    if (bt) A = GP(bt[blockIdx.y].A);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __builtin_amdgcn_s_setprio(1);
    WGT(wgt_, WGT_WIDE, k0);
    if ((int)blockIdx.x >= nfull) {
        // the last, partly empty round of the launch as 64x64 quarters (split_round)
        const int y = blockIdx.x - nfull;
        int ti, tj;
        trap_index(nfull + (y >> 2), cb - ca, ti, tj);
        const int i0 = (ca + ti) * TILE + ((y >> 1) & 1) * 64, j0 = (ca + tj) * TILE + (y & 1) * 64;
        d4 acc[2][2];
        acc_zero(acc);
        tile_nt<false>(A + (size_t)i0 * ld, ld, A + (size_t)j0 * ld, ld, k0 * TILE, (k0 + kw) * TILE, acc, smem);
        tile_accum_store<-1>(A + (size_t)i0 * ld + j0, ld, acc);
        return;
    }
    // every XCD (workgroups are dealt round-robin over the 8 of them) walks one contiguous run of the row-major
    // tile list: neighbours share panel rows in its L2
    const int xg = blockIdx.x & 7;
    const int xq = nfull >> 3, xr = nfull & 7;
    const int tlin = (xg < xr ? xg * (xq + 1) : xr * (xq + 1) + (xg - xr) * xq) + (blockIdx.x >> 3);
    int ti, tj;
    trap_index(rev ? nfull - 1 - tlin : tlin, cb - ca, ti, tj);
    const int i0 = (ca + ti) * TILE, j0 = (ca + tj) * TILE;
    d4 acc[4][4];
    acc_zero(acc);
    tile_nt<false>(A + (size_t)i0 * ld, ld, A + (size_t)j0 * ld, ld, k0 * TILE, (k0 + kw) * TILE, acc, smem);
    tile_accum_store<-1>(A + (size_t)i0 * ld + j0, ld, acc);
}

// inverse of a 128x128 diagonal factor block from the two 64x64 inverses potf2 left in d64 (one doubling
// step, T10 = -T11 (L10 T00), on 16x16 micro tiles: wave = micro-tile column of T10); writes T (lower, zeros
// above) and U = T^T (upper, zeros below).  blockIdx.x = block offset from kb.  ~6 us (the blocked
// substitution from the 16x16 inverses it replaces took 92 us -- 10 % of a 1500-row evaluation).
__device__ __forceinline__ void trtri_diag_body(const double* __restrict__ A, int ld, int b,
                                                const double* __restrict__ d64, double* __restrict__ T,
                                                double* __restrict__ U, double* __restrict__ sm)
{
    const int t = threadIdx.x, wave = t >> 6;
    const double* Ab = A + (size_t)b * TILE * ld + b * TILE;
    const double* T00 = d64 + (size_t)b * 8192;
    const double* T11 = T00 + 4096;
    {
        // all 36 loads in flight before the first LDS store (fully unrolled: which of the three sources a micro
        // tile comes from is then a compile-time choice; as a rolled loop with the branches inside it the loads
        // went out one at a time and the kernel took 25 us, nearly all of it this prologue)
        const int r = t >> 4, c = t & 15;
        double v[NLT];
#pragma unroll
        for (int bi = 0; bi < NMT; bi++)
#pragma unroll
            for (int bj = 0; bj <= bi; bj++) {
                const int q = bi * (bi + 1) / 2 + bj;
                if (bi < 4) v[q] = T00[(bi * MT + r) * 64 + bj * MT + c];
                else if (bj >= 4) v[q] = T11[((bi - 4) * MT + r) * 64 + (bj - 4) * MT + c];
                else v[q] = Ab[(size_t)(bi * MT + r) * ld + bj * MT + c];            // L10
            }
#pragma unroll
        for (int q = 0; q < NLT; q++) sm[q * MTS + r * (MT + 1) + c] = v[q];
    }
    __syncthreads();
    {
        const int bj = wave;                                 // my micro-tile column of T10
        const d4 zero4 = (d4){0.0, 0.0, 0.0, 0.0};
        d4 w[4];                                             // W(4 + q, bj) = sum_{jp >= bj} L10(4 + q, jp) T00(jp, bj)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            w[q] = zero4;
            for (int jp = bj; jp < 4; jp++) w[q] = micro_mma_nn(sm + mt_off(4 + q, jp), sm + mt_off(jp, bj), w[q]);
        }
        __syncthreads();                                     // every wave has read its L10 tiles
#pragma unroll
        for (int q = 0; q < 4; q++) {                        // T10(4 + q, bj) = -sum_{k <= q} T11(4 + q, 4 + k) W(4 + k, bj)
            d4 x = zero4;
#pragma unroll
            for (int k = 0; k <= q; k++) x = micro_mma_acc_b(sm + mt_off(4 + q, 4 + k), w[k], x);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            micro_store(sm + mt_off(4 + q, bj), x);
        }
    }
    __syncthreads();
    double* Tb = T + (size_t)b * TILE * ld + b * TILE;
    double* Ub = U + (size_t)b * TILE * ld + b * TILE;
#pragma unroll 8
    for (int e = t; e < TILE * TILE; e += 256) {
        const int r = e >> 7, c = e & 127, br = r >> 4, bc = c >> 4;
        // T(r,c) lives in micro tile (br,bc) when bc <= br (diagonal micro tiles are zero above their diagonal)
        Tb[(size_t)r * ld + c] = bc <= br ? sm[mt_off(br, bc) + (r & 15) * (MT + 1) + (c & 15)] : 0.0;
        Ub[(size_t)r * ld + c] = br <= bc ? sm[mt_off(bc, br) + (c & 15) * (MT + 1) + (r & 15)] : 0.0;   // T(c,r)
    }
}

__global__ __launch_bounds__(256) void k_trtri_diag(const double* __restrict__ A, int ld, int kb,
                                                    const double* __restrict__ d64, double* __restrict__ T,
                                                    double* __restrict__ U, const ExpertPtrs* __restrict__ bt)
{
    if (bt) { A = GP(bt[blockIdx.y].A); d64 = GP(bt[blockIdx.y].d64); T = GP(bt[blockIdx.y].T); U = GP(bt[blockIdx.y].U); }
    extern __shared__ __attribute__((aligned(16))) double sm[];       // 36 lower micro tiles, as potf2_body
    WGT(wgt_, WGT_TRTRI_DIAG, kb);
    trtri_diag_body(A, ld, kb + blockIdx.x, d64, T, U, sm);
}

// ------------------------------------------------------------------------------------------
// The whole inverse of ONE hand-over block of rows [a, a + wb) -- the 128x128 inverses of its diagonal tiles and every
// level of the recursive doubling inside the block -- in ONE launch: what k_trtri_diag + 2 log2(wb) launches of
// k_trtri_level<2> did (at N = 8192, wb = 4: five launches of 2-16 workgroups, 15-45 us apiece while the big products of
// the other streams hold the chip; 80 of the evaluation's 260 launches).  A small persistent grid (<= 64 workgroups)
// walks the stages; between two stages every workgroup arrives at a monotonic agent-scope counter and waits for the
// others (release fence -> add ... poll -> acquire fence: the safe forms of the CDNA guide's barrier-counter row).
// The same tile code (trtri_diag_body, level_item<2>) in the same per-element order: bit-identical to the launches it
// replaces.  The grid is far below one workgroup per CU, the workgroups of a launch are dispatched in order, and what
// holds the other slots always ends without waiting for this launch -- so the waiting is deadlock-free; it is BOUNDED
// all the same (~seconds), and a wait that ran out poisons the block's log-determinant share (the evaluation comes
// back NaN instead of hanging the device).
// ------------------------------------------------------------------------------------------
constexpr int TRTRI_BLOCK_LDS = TRTRI_LDS > Geo<2>::LDS ? TRTRI_LDS : Geo<2>::LDS;

__device__ __forceinline__ bool stage_barrier(unsigned* __restrict__ ctr, unsigned target, unsigned spin_cap)
{
    __shared__ int s_ok;
    // every wave drains its own stores of the stage (the barrier's own wait is lgkmcnt only), THEN the workgroup
    // meets, then one lane writes this XCD's L2 back (all of the workgroup's stores are in it by now) and arrives
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the compiler may drop the wait behind the write-back)
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 0;
        for (unsigned spin = 0; spin < spin_cap; spin++) {
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

__global__ __launch_bounds__(256, 2) void k_trtri_block(const double* __restrict__ L, const double* __restrict__ d64,
                                                        double* __restrict__ T, double* __restrict__ U, int ld, int a,
                                                        int wb, unsigned* __restrict__ ctr, double* __restrict__ poison,
                                                        int ctr_off, unsigned spin_cap, double* __restrict__ hstat,
                                                        const ExpertPtrs* __restrict__ bt, unsigned long long* stamp)
{
    LaunchStamp stamp_(stamp);
    if (bt) {
        const ExpertPtrs& e = bt[blockIdx.y];
        L = GP(e.A); d64 = GP(e.d64); T = GP(e.T); U = GP(e.U); ctr = GP(e.tickets) + ctr_off; poison = GP(e.logdet);
        if (hstat) hstat += (size_t)blockIdx.y * 8;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WGT(wgt_, WGT_TRTRI_DIAG, a);
    const int G = gridDim.x, wg = blockIdx.x;
    for (int b = wg; b < wb; b += G) {
        if (b != wg) __syncthreads();                    // (the body's LDS image is reused)
        trtri_diag_body(L, ld, a + b, d64, T, U, (double*)smem);
    }
    const size_t off = (size_t)a * TILE * ld + (size_t)a * TILE;
    const double* Lb = L + off;
    double* Tb = T + off;
    double* Ub = U + off;
    unsigned stage = 0;
    bool ok = true;
    for (int s = 1; s < wb; s *= 2)
        for (int step = 1; step <= 2; step++) {
            ok = stage_barrier(ctr, (unsigned)G * ++stage, spin_cap) && ok;
            const int items = level_tiles(wb, s) * 4;
            for (int it = wg; it < items; it += G) {
                if (it != wg) __syncthreads();           // (the transposed store of the item before still reads its LDS image)
                level_item<2>(Lb, Tb, Ub, ld, wb, s, step, it >> 2, it & 3, smem);
            }
        }
    // a wait that ran out: this block's numbers are not to be trusted -- NaN into its log-determinant share (the
    // evaluation's values come back NaN) and the handle's pinned status word (its fetch returns CUGP_ERR_DEVICE)
    if (!ok && threadIdx.x == 0) {
        poison[a] = __builtin_nan("");
        if (hstat) hstat[6] = 1.0;
    }
}

// ------------------------------------------------------------------------------------------
// vector kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// z[i] = sum_{k < (ti+1)*128} T[i][k] x[k]  (one wave per row; the diagonal tile is zero above the diagonal)
__global__ __launch_bounds__(256) void k_trmv_lower(const double* __restrict__ T, int ld, int npad,
                                                    const double* __restrict__ x, double* __restrict__ z,
                                                    const ExpertPtrs* __restrict__ bt)
{
    if (bt) { T = GP(bt[blockIdx.y].T); x = GP(bt[blockIdx.y].y); z = GP(bt[blockIdx.y].z); }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= npad) return;
    const int kend = (row / TILE + 1) * TILE;
    const double* tr = T + (size_t)row * ld;
    double s = 0.0;
    for (int k = lane * 2; k < kend; k += 128) {
        d2 v = *(const d2*)(tr + k), xv = *(const d2*)(x + k);
        s += v[0] * xv[0] + v[1] * xv[1];
    }
    s = wave_sum(s);
    if (lane == 0) z[row] = s;
}

// a[i] = sum_{k >= ti*128} U[i][k] x[k]
__global__ __launch_bounds__(256) void k_trmv_upper(const double* __restrict__ U, int ld, int npad,
                                                    const double* __restrict__ x, double* __restrict__ a,
                                                    const ExpertPtrs* __restrict__ bt)
{
    if (bt) { U = GP(bt[blockIdx.y].U); x = GP(bt[blockIdx.y].z); a = GP(bt[blockIdx.y].alpha); }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= npad) return;
    const int kbeg = (row / TILE) * TILE;
    const double* ur = U + (size_t)row * ld;
    double s = 0.0;
    for (int k = kbeg + lane * 2; k < npad; k += 128) {
        d2 v = *(const d2*)(ur + k), xv = *(const d2*)(x + k);
        s += v[0] * xv[0] + v[1] * xv[1];
    }
    s = wave_sum(s);
    if (lane == 0) a[row] = s;
}

// w = y for every expert of a batched launch (the forward substitution consumes w)
__global__ __launch_bounds__(256) void k_copy_y_to_w(int npad, const ExpertPtrs* __restrict__ bt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < npad) GP(bt[blockIdx.y].w)[i] = GP(bt[blockIdx.y].y)[i];
}

// blocked forward substitution L z = y (LL-only path): step kb = (1) z_kb = T_kk w_kb, (2) w[rows below] -= L21 z_kb
__global__ __launch_bounds__(256) void k_trsv_diag(const double* __restrict__ T, int ld, int kb,
                                                   const double* __restrict__ w, double* __restrict__ z,
                                                   const ExpertPtrs* __restrict__ bt)
{
    if (bt) { T = GP(bt[blockIdx.y].T); w = GP(bt[blockIdx.y].w); z = GP(bt[blockIdx.y].z); }
    // z_kb = T_kk w_kb (128x128, lower): two threads per row, 64 columns each, 16-byte loads
    __shared__ double ws[TILE];
    const int k0 = kb * TILE, t = threadIdx.x;
    if (t < TILE) ws[t] = w[k0 + t];
    __syncthreads();
    const int r = t >> 1, h = t & 1;
    const double* tr = T + (size_t)(k0 + r) * ld + k0 + h * 64;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int c = 0; c < 64; c += 2) {
        const d2 v = *(const d2*)(tr + c);
        s0 = __builtin_fma(v[0], ws[h * 64 + c], s0);
        s1 = __builtin_fma(v[1], ws[h * 64 + c + 1], s1);
    }
    double sum = s0 + s1;
    sum += __shfl_xor(sum, 1, 64);
    if (h == 0) z[k0 + r] = sum;
}

__global__ __launch_bounds__(256) void k_trsv_update(const double* __restrict__ A, int ld, int kb, int npad,
                                                     const double* __restrict__ z, double* __restrict__ w,
                                                     const ExpertPtrs* __restrict__ bt)
{
    if (bt) { A = GP(bt[blockIdx.y].A); z = GP(bt[blockIdx.y].z); w = GP(bt[blockIdx.y].w); }
    const int k0 = kb * TILE;
    const int row = k0 + TILE + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= npad) return;
    const double* ar = A + (size_t)row * ld + k0;
    d2 v = *(const d2*)(ar + lane * 2), xv = *(const d2*)(z + k0 + lane * 2);
    double s = wave_sum(v[0] * xv[0] + v[1] * xv[1]);
    if (lane == 0) w[row] -= s;
}

// Final sums of an evaluation by ONE workgroup of NTHR threads, in the order 1024 threads would take them whatever NTHR
// is (so the stand-alone k_finalize, 1024 threads, and the last block of k_trace, 256 threads, give the same bits):
// virtual thread v = 0..1023 sums the entries i = v, v + 1024, ... ascending, then the halving tree red[v] += red[v + w],
// w = 512 .. 1.  A real thread takes the virtual threads t, t + NTHR, ... and the tree levels above NTHR in registers.
// HANDED: the trace partials were written by other workgroups of this launch -- agent-scope loads.
// red: 5 * NTHR / ... doubles of LDS: [5][NTHR].
constexpr int FIN_THREADS = 1024;     // one workgroup; at N = 8192 it sums 3 x 8256 trace partials and 8192 squares (23 us with 256 threads)
template <int NTHR, bool HANDED>
__device__ __forceinline__ void finalize_sums(const double* __restrict__ z, int npad, int n,
                                              const double* __restrict__ logdet_part, int nt,
                                              const double* __restrict__ part, int nblocks, HyperScalars h,
                                              double* __restrict__ out, double* __restrict__ hout, double* red)
{
    constexpr int V = FIN_THREADS / NTHR;              // virtual threads per real thread
    const int t = threadIdx.x;
    double acc[5][V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int v = t + j * NTHR;
        double q = 0.0, ld = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int i = v; i < npad; i += FIN_THREADS) q += z[i] * z[i];
        for (int i = v; i < nt; i += FIN_THREADS) ld += logdet_part[i];
        if (part)
            for (int i0 = v; i0 < nblocks; i0 += 4 * FIN_THREADS) {
                // four entries' loads in flight before the first is added (agent-scope loads one by one cost a memory
                // latency each: 40 us for the 8256 partials of an 8192-row matrix); the sums are taken in order
                double p[4][3];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * FIN_THREADS;
                    const double* pp = part + (size_t)(i < nblocks ? i : v) * 3;
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        p[u][c] = HANDED ? __hip_atomic_load(pp + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pp[c];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i0 + u * FIN_THREADS < nblocks) { s0 += p[u][0]; s1 += p[u][1]; s2 += p[u][2]; }
            }
        acc[0][j] = q; acc[1][j] = ld; acc[2][j] = s0; acc[3][j] = s1; acc[4][j] = s2;
    }
    // tree levels w >= NTHR: red[v] += red[v + w] pairs virtual threads of the SAME real thread (v and v + w differ by a multiple of NTHR)
#pragma unroll
    for (int w = V / 2; w > 0; w >>= 1)
#pragma unroll
        for (int c = 0; c < 5; c++)
#pragma unroll
            for (int j = 0; j < w; j++) acc[c][j] += acc[c][j + w];
    __syncthreads();                                   // (red may alias LDS the caller used before)
#pragma unroll
    for (int c = 0; c < 5; c++) red[c * NTHR + t] = acc[c][0];
    __syncthreads();
    for (int w = NTHR / 2; w > 0; w >>= 1) {
        if (t < w)
            for (int c = 0; c < 5; c++) red[c * NTHR + t] += red[c * NTHR + t + w];
        __syncthreads();
    }
    if (t == 0) {
        const double quad = red[0], logdet = 2 * red[NTHR];
        out[0] = -0.5 * (quad + logdet + n * 1.83787);
        if (part) {
            const double s1 = red[2 * NTHR], s2 = red[3 * NTHR], s3 = red[4 * NTHR];
            out[1] = s1 / 2.0;
            out[2] = (2.0 * s2 - 2.0 * h.noise_var * s3) / 2.0;
            out[3] = (2.0 * h.noise_var * s3) / 2.0;
        }
        out[4] = quad;
        out[5] = logdet;
        if (hout)
            for (int i = 0; i < 6; i++) hout[i] = out[i];
    }
}

// what the last block of k_trace needs to finish the evaluation (out == nullptr: no fused finalize)
struct FinalizeArgs {
    const double* z; const double* logdet_part; int nt; double* out; double* hout; unsigned* ticket;
};

// gradient traces, fused: for every lower 64x64 tile recompute k(xi,xj) and |xi-xj|^2/l^2, read K^-1 once,
// W = K^-1 - alpha alpha^T, accumulate  s1 = sum W*K*S, s2 = sum W*K, s3 = sum_i W_ii  (off-diagonal tiles x2)
__global__ __launch_bounds__(256) void k_trace(const double* __restrict__ X, int n, int d, int npad,
                                               HyperScalars h_arg, const HyperScalars* __restrict__ hd,
                                               const double* __restrict__ Kinv, const double* __restrict__ alpha,
                                               double* __restrict__ part, const ExpertPtrs* __restrict__ bt,
                                               FinalizeArgs fin)
{
    if (bt) {
        const ExpertPtrs& e = bt[blockIdx.y];
        X = GP(e.X); n = e.n; Kinv = GP(e.Kinv); alpha = GP(e.alpha); part = GP(e.part);
        if (fin.out) {
            fin.z = GP(e.z); fin.logdet_part = GP(e.logdet); fin.out = GP(e.out); fin.ticket = GP(e.tickets) + 2 * fin.nt;
            if (fin.hout) fin.hout += (size_t)blockIdx.y * 8;
        }
    }
    const HyperScalars h = hd ? *hd : h_arg;
    // (one buffer: the two X tiles of the squared distances, later the [5][256] sums of the fused finalize)
    __shared__ double lds[2 * KT * (DC + 1)];
    static_assert(2 * KT * (DC + 1) >= 5 * 256, "the fused finalize reduces in the X tiles' LDS");
    double (&xs)[KT][DC + 1] = *reinterpret_cast<double (*)[KT][DC + 1]>(lds);
    double (&ys)[KT][DC + 1] = *reinterpret_cast<double (*)[KT][DC + 1]>(lds + KT * (DC + 1));
    __shared__ double red[3][4];
    int ti, tj;
    tri_index(blockIdx.x, ti, tj);
    const int i0 = ti * KT, j0 = tj * KT;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    double d2v[4][4];
    sqdist_4x4(X, X, n, n, d, i0, j0, xs, ys, d2v);
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const DivBy dl = div_prepare(h.ell_sq);
    double aj[4];
#pragma unroll
    for (int b = 0; b < 4; b++) aj[b] = alpha[j0 + col4(tx, b)];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int i = i0 + ty * 4 + a;
        const double ai = alpha[i];
        const double* kr = Kinv + (size_t)i * npad + j0 + tx * 2;
        d2 k01 = *(const d2*)kr, k23 = *(const d2*)(kr + 32);
        const double kv[4] = {k01[0], k01[1], k23[0], k23[1]};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int j = j0 + col4(tx, b);
            if (i < n && j < n && (ti != tj || j <= i)) {
                const double w = kv[b] - ai * aj[b];
                double kse = h.signal_var * exp(div_by(-d2v[a][b] * 0.5, dl));
                const double sd = div_by(d2v[a][b], dl);
                if (i == j) {
                    kse += h.noise_var;
                    s1 += w * (kse * sd);
                    s2 += w * kse;
                    s3 += w;
                } else {
                    s1 += 2.0 * (w * (kse * sd));
                    s2 += 2.0 * (w * kse);
                }
            }
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if ((t & 63) == 0) { red[0][t >> 6] = s1; red[1][t >> 6] = s2; red[2][t >> 6] = s3; }
    __syncthreads();
    if (!fin.out) {
        if (t < 3) part[(size_t)blockIdx.x * 3 + t] = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
        return;
    }
    // Fused finalize (round 6): the partial sums leave write-through, the block draws a ticket, and the LAST block of the
    // launch (of this expert) goes on to the final sums and the scalar formulas -- what k_finalize did as one more launch
    // behind this one (a kernel boundary and a 1-workgroup launch on the serial tail of every evaluation: ~9 us of a
    // 590-us evaluation at 1500 rows).  The hand-off is the step kernel's (k_syrk_step): write-through (agent-scope)
    // stores by lanes of wave 0, that wave's vmcnt(0), one relaxed agent-scope ticket; the last arriver reads every
    // partial with agent-scope loads.  Same sums in the same order as k_finalize: identical bits.
    __shared__ unsigned s_ticket;
    if (t < 3) __hip_atomic_store(part + (size_t)blockIdx.x * 3 + t, (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) s_ticket = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t == 0) *fin.ticket = 0u;                          // (ready for the next evaluation; nobody else touches it any more)
    finalize_sums<256, true>(fin.z, npad, n, fin.logdet_part, fin.nt, part, (int)gridDim.x, h, fin.out, fin.hout, lds);
}

// single workgroup: deterministic final sums and the scalar formulas
//   LL = -0.5 (z'z + 2 sum log L_ii + n * 1.83787)                     covkernel.cpp:127
//   g0 = s1/2, g1 = (2 s2 - 2 sn2 s3)/2, g2 = (2 sn2 s3)/2              covkernel.cpp:244-261
__global__ __launch_bounds__(FIN_THREADS) void k_finalize(const double* __restrict__ z, int npad, int n,
                                                  const double* __restrict__ logdet_part, int nt,
                                                  const double* __restrict__ part, int nblocks,
                                                  HyperScalars h_arg, const HyperScalars* __restrict__ hd,
                                                  double* __restrict__ out, double* __restrict__ hout,
                                                  const ExpertPtrs* __restrict__ bt)
{
    // hout: the same 6 results straight into the caller's pinned host buffer ([expert][8]) -- visible to the host
    // when the launch has completed, no copy node (and its ~10-us boundary) behind the evaluation
    if (bt) {
        const ExpertPtrs& e = bt[blockIdx.y];
        z = GP(e.z); n = e.n; logdet_part = GP(e.logdet); out = GP(e.out);
        if (part) part = GP(e.part);
        if (hout) hout += (size_t)blockIdx.y * 8;
    }
    const HyperScalars h = hd ? *hd : h_arg;
    __shared__ double red[5 * FIN_THREADS];
    finalize_sums<FIN_THREADS, false>(z, npad, n, logdet_part, nt, part, nblocks, h, out, hout, red);
}

// mean[t] = Ks[t] . alpha ; var[t] = sf2 + sn2 - |W[t]|^2        covkernel.cpp:314-319
__global__ __launch_bounds__(256) void k_predict_finish(const double* __restrict__ Ks, const double* __restrict__ W,
                                                        const double* __restrict__ alpha, int n, int npad,
                                                        int ntest, HyperScalars h, double* __restrict__ mean,
                                                        double* __restrict__ var)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= ntest) return;
    const double* kr = Ks + (size_t)row * npad;
    const double* wr = W + (size_t)row * npad;
    double m = 0.0, q = 0.0;
    for (int k = lane * 2; k < npad; k += 128) {
        d2 kv = *(const d2*)(kr + k), av = *(const d2*)(alpha + k), wv = *(const d2*)(wr + k);
        m += kv[0] * av[0] + kv[1] * av[1];
        q += wv[0] * wv[0] + wv[1] * wv[1];
    }
    m = wave_sum(m); q = wave_sum(q);
    if (lane == 0) { mean[row] = m; var[row] = h.signal_var + h.noise_var - q; }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// Per-launch timing without extra packets on the stream (profiling level 4, cugp_capi.cpp TimedLaunch): the next launch
// of a timed kernel on this thread goes out through hipExtLaunchKernelGGL with a start / stop event pair, which take
// the dispatch's OWN begin / end timestamps -- what rocprofv3 --kernel-trace reports for it.  (An event pair recorded
// around the launch, level 3, brackets the ~6 us between the record in front of it and its first workgroup as well
// and costs ~5 us of device time per pair.)
thread_local hipEvent_t t_ev0 = nullptr, t_ev1 = nullptr;
thread_local unsigned long long* t_stamp = nullptr;        // level 5: the next timed launch's slot (LaunchStamp)
void time_next_launch(hipEvent_t start, hipEvent_t stop) { t_ev0 = start; t_ev1 = stop; }
void stamp_next_launch(unsigned long long* slot) { t_stamp = slot; }
bool timing_pending() { return t_ev0 != nullptr || t_stamp != nullptr; }
static inline unsigned long long* take_stamp() { unsigned long long* p = t_stamp; t_stamp = nullptr; return p; }
#define CUGP_LAUNCH(kernel, grid, block, lds, stream, ...)                                          \
    do {                                                                                            \
        if (t_ev0) {                                                                                \
            hipEvent_t e0_ = t_ev0, e1_ = t_ev1;                                                    \
            t_ev0 = t_ev1 = nullptr;                                                                \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, e0_, e1_, 0, __VA_ARGS__);      \
        } else {                                                                                    \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                      \
        }                                                                                           \
    } while (0)

const int g_tune_init[TUNE_COUNT] = {768, 1200, 384, -1, 511, 1, 1, 1 << 20, 16, 500, 32, 1, 2100, 256, 1536, 0, 1 << 21, 1, 1};   // defaults chosen by interleaved A/B runs (tools/ab.py)
thread_local const int* t_tune = g_tune_init;

static inline int tri_count(int n) { return n * (n + 1) / 2; }

void launch_kbuild(const double* X, int n, int d, int npad, HyperScalars h, double* K, bool full, hipStream_t s,
                   const HyperScalars* hd, Batch bt, unsigned* tickets)
{
    hipLaunchKernelGGL(k_build, dim3(tri_count(npad / KT), bt.count), dim3(256), 0, s, X, n, d, npad, h, hd, K,
                       full ? 1 : 0, tickets, bt.tab, take_stamp());
}

void launch_sqdist(const double* X, int n, int d, int npad, double c, double* S, hipStream_t s)
{
    HyperScalars h{c, 0.0, 0.0};
    hipLaunchKernelGGL(k_build, dim3(tri_count(npad / KT)), dim3(256), 0, s, X, n, d, npad, h,
                       (const HyperScalars*)nullptr, S, 2, (unsigned*)nullptr, (const ExpertPtrs*)nullptr,
                       (unsigned long long*)nullptr);
}

void launch_kcross(const double* X, int n, int d, int npad, const double* Xt, int nt, int ntpad, HyperScalars h,
                   double* Ks, hipStream_t s)
{
    hipLaunchKernelGGL(k_cross, dim3((ntpad / KT) * (npad / KT)), dim3(256), 0, s, X, n, d, npad, Xt, nt, ntpad, h,
                       Ks);
}

// hipFuncAttributeMaxDynamicSharedMemorySize applies to the CURRENT device only: one flag per device, and a
// failure is kept (the launch that follows would be rejected) and reported by prepare_kernels().
static unsigned long long g_attr_done = 0;              // bit per device ordinal (< 64)
static hipError_t g_attr_err = hipSuccess;
static void set_big_lds()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;
    if (g_attr_done >> dev & 1ull) return;
    hipError_t e = hipSuccess;
    auto attr = [&](const void* f, int bytes) {
        const hipError_t r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (r != hipSuccess && e == hipSuccess) e = r;
    };
    attr((const void*)k_potf2, POTF2_LDS);
    attr((const void*)k_syrk_step, STEP_LDS);
    attr((const void*)k_syrk_wide, GEMM_LDS);
    const void* gemm4[] = {(const void*)k_trtri_level<4>, (const void*)k_trtri_border<4>, (const void*)k_lauum<4>,
                           (const void*)k_test_gemm};
    for (const void* f : gemm4) attr(f, GEMM_LDS);
    attr((const void*)k_trtri_diag, TRTRI_LDS);
    attr((const void*)k_trtri_block, TRTRI_BLOCK_LDS);
    attr((const void*)k_trsm_inv64, TRSM_LDS);
    if (e != hipSuccess) { g_attr_err = e; return; }
    g_attr_done |= 1ull << dev;
}

int prepare_kernels() { set_big_lds(); return (int)g_attr_err; }

void launch_potf2(double* A, int ld, int kb, double* d16, double* d64, double* logdet_part, hipStream_t s, Batch bt)
{
    set_big_lds();
    hipLaunchKernelGGL(k_potf2, dim3(1, bt.count), dim3(256), POTF2_LDS, s, A, ld, kb, d16, d64, logdet_part, bt.tab);
}

void launch_trsm_inv64(double* A, const double* d64, int ld, int kb, int nt, hipStream_t s, Batch bt, double* zv,
                       const double* wv)
{
    const int nstrips = (nt - kb - 1) * (TILE / MT);
    if (nstrips <= 0 && !zv) return;                      // (kb = nt - 1 with zv: the last block's z alone)
    set_big_lds();
    hipLaunchKernelGGL(k_trsm_inv64, dim3(nstrips + (zv ? 1 : 0), bt.count), dim3(512), TRSM_LDS, s, A, d64, ld, kb, bt.tab,
                       take_stamp(), nstrips, zv, wv);
}

void launch_trtri_diag(const double* A, int ld, int kb, int nblocks, const double* d64, double* T, double* U,
                       hipStream_t s, Batch bt)
{
    set_big_lds();
    hipLaunchKernelGGL(k_trtri_diag, dim3(nblocks, bt.count), dim3(256), TRTRI_LDS, s, A, ld, kb, d64, T, U, bt.tab);
}

int launch_trtri_block(const double* L, const double* d64, double* T, double* U, int ld, int a, int wb, unsigned* ctr,
                       double* poison, int ctr_off, hipStream_t s, Batch bt, int gcap, double* hstat)
{
    if (wb <= 0) return 0;
    set_big_lds();
    int G = wb;                                           // workgroups: the widest stage, one item each (<= 64)
    for (int sl = 1; sl < wb; sl *= 2) {
        const int items = level_tiles(wb, sl) * 4;
        if (items > G) G = items;
    }
    if (G > TRTRI_BLOCK_MAXWG) G = TRTRI_BLOCK_MAXWG;
    if (gcap >= 1 && G > gcap) G = gcap;                  // (every workgroup count gives the same bits: a stage's items are independent)
    CUGP_LAUNCH(k_trtri_block, dim3(G, bt.count), dim3(256), TRTRI_BLOCK_LDS, s, L, d64, T, U, ld, a, wb, ctr,
                       poison, ctr_off, (unsigned)tune(TUNE_BARRIER_SPIN), hstat, bt.tab, take_stamp());
    return G;
}

// A launch of `tiles` uniform-ish 128x128 tiles fills the 512 workgroup slots round by round; a last round that is
// mostly empty still costs a whole tile time (134 us at K = 512).  When the remainder is small its tiles run as
// four 64x64 workgroups each instead: -> number of tiles launched whole (the rest are split).
static int split_round(int tiles, int count)
{
    const int slots = 512 / (count > 0 ? count : 1) > 0 ? 512 / (count > 0 ? count : 1) : 1;
    const int rem = tiles % slots;
    if (tiles < slots || rem == 0 || rem > tune(TUNE_SPLIT_REM_MAX)) return tiles;
    return tiles - rem;
}

static inline int trap_count(int m, int wcol)          // tiles (ti >= tj) of the first wcol columns of an m-triangle
{
    return wcol >= m ? tri_count(m) : tri_count(wcol) + (m - wcol) * wcol;
}

void launch_syrk_step(double* A, int ld, int kb, int nt, double* d16, double* d64, double* logdet_part,
                      unsigned* tickets, hipStream_t s, Batch bt, int wcol, int ks, const double* zv, double* wv)
{
    const int m = nt - kb - 1;
    if (m <= 0) return;
    set_big_lds();
    if (wcol <= 0 || wcol > m) wcol = m;
    // tiles beyond the last full round of 512 workgroup slots run as 64x64 quarters when that round
    // would be less than three quarters full
    const int ntl = trap_count(m, wcol) - 1;
    int nfull = ntl;
    if (ntl >= 512 && (ntl % 512) <= tune(TUNE_SYRK_REM_MAX)) nfull = ntl - ntl % 512;
    // Few tiles (the chain-bound tail of the factorisation): a 128x128 workgroup alone on its CU issues one MFMA
    // per ~138 cycles (one wave per SIMD: half the pipe's rate) and takes ~40 us -- as long as the diagonal block
    // inside this launch.  As 64x64 quarters the same tiles take ~12 us per round of 512 workgroups.
    if ((long long)ntl * bt.count * 4 <= tune(TUNE_STEP_QUARTER_MAX)) nfull = 0;
    const int vec0 = NDIAGWG + nfull + 4 * (ntl - nfull);
    const unsigned nwg = vec0 + (zv ? m : 0);
    if (ks < 0 || ks > kb) ks = kb;
    if (kb + 1 - ks > SUBPANEL_MAX) return;               // (plan_step never asks for more)
    CUGP_LAUNCH(k_syrk_step, bt.tab ? dim3(bt.count, nwg) : dim3(nwg), dim3(256), STEP_LDS, s, A, ld, kb, d16,
                       d64, logdet_part, tickets, nfull, wcol, ks, bt.tab, take_stamp(), vec0, zv, wv);
}

// tile columns [ca, cb) (rows >= column) -= L(., k0..k0+kw) L(., k0..k0+kw)^T; returns the number of tiles
int launch_syrk_wide(double* A, int ld, int nt, int k0, int kw, int ca, int cb, int rev, hipStream_t s, Batch bt)
{
    if (cb > nt) cb = nt;
    if (ca >= cb || kw <= 0) return 0;
    set_big_lds();
    const int ntiles = trap_count(nt - ca, cb - ca);
    const int nfull = split_round(ntiles, bt.count);
    CUGP_LAUNCH(k_syrk_wide, dim3(nfull + 4 * (ntiles - nfull), bt.count), dim3(256), GEMM_LDS, s, A, ld, k0, kw, ca,
                       cb, ntiles, nfull, rev, bt.tab, take_stamp());
    return ntiles;
}

int launch_trtri_level(const double* L, double* T, double* U, int ld, int nt, int s, int step, hipStream_t st,
                       Batch bt, size_t off)
{
    // pairs p = 0.. : A = [2ps, 2ps+s), B = [2ps+s, min(2ps+2s, nt)); count tiles |A| x |B|
    int tiles = 0;
    for (int a0 = 0; a0 + s < nt; a0 += 2 * s) {
        int sb = nt - (a0 + s);
        if (sb > s) sb = s;
        tiles += s * sb;
    }
    if (tiles <= 0) return 0;
    // few 128-tiles cannot fill 512 workgroup slots: use 64x64 output tiles (4x the parallelism) there
    set_big_lds();
    if (tiles * bt.count <= tune(TUNE_TRTRI_WM2_MAX)) {  // (a batched launch fills the chip with fewer tiles each)
        CUGP_LAUNCH(k_trtri_level<2>, dim3(tiles * 4, bt.count), dim3(256), Geo<2>::LDS, st, L, T, U, ld, nt, s,
                           step, off, bt.tab, take_stamp());
        return 2;
    }
    CUGP_LAUNCH(k_trtri_level<4>, dim3(tiles, bt.count), dim3(256), GEMM_LDS, st, L, T, U, ld, nt, s, step,
                       off, bt.tab, take_stamp());
    return 4;
}

// step 1 of the bordering, spread over time: add the k tiles [c0, c1) (a block of inverse rows that just
// became final) to Wt(tj < c1, ti in [ra, ra+rw)) for ALL rows below the block
int launch_trtri_border1(const double* L, double* T, double* U, int ld, int ra, int rw, int c0, int c1,
                         hipStream_t st, Batch bt)
{
    const int tiles = c1 * rw;
    if (tiles <= 0) return 0;
    set_big_lds();
    if (tiles * bt.count <= tune(TUNE_BORDER_WM2_MAX)) {
        CUGP_LAUNCH(k_trtri_border<2>, dim3(tiles * 4, bt.count), dim3(256), Geo<2>::LDS, st, L, T, U, ld, ra, rw,
                           1, c0, c1, 0, bt.tab, take_stamp());
        return 2;
    }
    const int nfull = split_round(tiles, bt.count);
    CUGP_LAUNCH(k_trtri_border<4>, dim3(nfull + 4 * (tiles - nfull), bt.count), dim3(256), GEMM_LDS, st, L, T,
                       U, ld, ra, rw, 1, c0, c1, nfull, bt.tab, take_stamp());
    return 4;
}

// step 2: rows [a, a+w) of the inverse from their finished Wt and the block's own inverse
int launch_trtri_border2(const double* L, double* T, double* U, int ld, int a, int w, hipStream_t st, Batch bt)
{
    const int tiles = a * w;
    if (tiles <= 0) return 0;
    set_big_lds();
    if (tiles * bt.count <= tune(TUNE_TRTRI_WM2_MAX)) {
        CUGP_LAUNCH(k_trtri_border<2>, dim3(tiles * 4, bt.count), dim3(256), Geo<2>::LDS, st, L, T, U, ld, a, w, 2,
                           0, 0, 0, bt.tab, take_stamp());
        return 2;
    }
    const int nfull = split_round(tiles, bt.count);
    CUGP_LAUNCH(k_trtri_border<4>, dim3(nfull + 4 * (tiles - nfull), bt.count), dim3(256), GEMM_LDS, st, L, T,
                       U, ld, a, w, 2, 0, 0, nfull, bt.tab, take_stamp());
    return 4;
}

int launch_lauum(const double* U, double* Kinv, int ld, int a, int w, hipStream_t s, Batch bt)
{
    set_big_lds();
    const int tiles = tri_count(a + w);
    if (tiles * bt.count <= tune(TUNE_LAUUM_WM2_MAX)) {
        CUGP_LAUNCH(k_lauum<2>, dim3(tiles * 4, bt.count), dim3(256), Geo<2>::LDS, s, U, Kinv, ld, a, w, 0, bt.tab, take_stamp());
        return 2;
    } else {
        // (a whole-matrix product, a = 0, has k ranges from 1 to a+w tiles, longest first: its tail is short tiles
        //  already; the split is for the block-wise calls, whose tiles all take w k tiles)
        const int nfull = a > 0 ? split_round(tiles, bt.count) : tiles;
        CUGP_LAUNCH(k_lauum<4>, dim3(nfull + 4 * (tiles - nfull), bt.count), dim3(256), GEMM_LDS, s, U, Kinv, ld,
                           a, w, nfull, bt.tab, take_stamp());
        return 4;
    }
}

void launch_predict_gemm(const double* Ks, const double* T, double* W, int ld, int ntt, int nt, hipStream_t s)
{
    set_big_lds();
    // (ntt, nt in 128-row tiles; the kernel works on 64-row tiles in pairs)
    CUGP_LAUNCH(k_predict_gemm, dim3(2 * ntt * nt), dim3(256), Geo<2>::LDS, s, Ks, T, W, ld, 2 * ntt, 2 * nt, take_stamp());
}

void launch_predict_finish(const double* Ks, const double* W, const double* alpha, int n, int npad, int ntest,
                           HyperScalars h, double* mean, double* var, hipStream_t s)
{
    hipLaunchKernelGGL(k_predict_finish, dim3((ntest + 3) / 4), dim3(256), 0, s, Ks, W, alpha, n, npad, ntest, h,
                       mean, var);
}

void launch_trmv_lower(const double* T, int ld, int npad, const double* x, double* z, hipStream_t s, Batch bt)
{
    hipLaunchKernelGGL(k_trmv_lower, dim3(npad / 4, bt.count), dim3(256), 0, s, T, ld, npad, x, z, bt.tab);
}

void launch_trmv_upper(const double* U, int ld, int npad, const double* x, double* a, hipStream_t s, Batch bt)
{
    hipLaunchKernelGGL(k_trmv_upper, dim3(npad / 4, bt.count), dim3(256), 0, s, U, ld, npad, x, a, bt.tab);
}

void launch_copy_y_to_w(int npad, hipStream_t s, Batch bt)
{
    hipLaunchKernelGGL(k_copy_y_to_w, dim3((npad + 255) / 256, bt.count), dim3(256), 0, s, npad, bt.tab);
}

void launch_trsv_lower(const double* A, const double* T, int ld, int nt, const double* y, double* z, hipStream_t s,
                       Batch bt)
{
    // y is consumed as the running right-hand side w (caller passes a scratch copy)
    double* w = const_cast<double*>(y);
    const int npad = nt * TILE;
    for (int kb = 0; kb < nt; kb++) {
        hipLaunchKernelGGL(k_trsv_diag, dim3(1, bt.count), dim3(256), 0, s, T, ld, kb, w, z, bt.tab);
        const int rows = npad - (kb + 1) * TILE;
        if (rows > 0)
            hipLaunchKernelGGL(k_trsv_update, dim3(rows / 4, bt.count), dim3(256), 0, s, A, ld, kb, npad, z, w, bt.tab);
    }
}

int trace_num_blocks(int npad) { return tri_count(npad / KT); }

void launch_trace(const double* X, int n, int d, int npad, HyperScalars h, const double* Kinv, const double* alpha,
                  double* part, hipStream_t s, const HyperScalars* hd, Batch bt, const double* z, const double* logdet_part,
                  double* out, double* hout, unsigned* ticket)
{
    // the last block takes the final sums where the launch is small (its 256 threads against k_finalize's 1024: at 8192
    // rows -- 8256 blocks -- the separate launch is as fast and keeps the blocks' stores plain); TUNE_FINALIZE_FUSE_MAX
    const int nblocks = tri_count(npad / KT);
    const bool fuse = out != nullptr && nblocks <= tune(TUNE_FINALIZE_FUSE_MAX);
    const FinalizeArgs fin{z, logdet_part, npad / TILE, fuse ? out : nullptr, hout, ticket};
    hipLaunchKernelGGL(k_trace, dim3(nblocks, bt.count), dim3(256), 0, s, X, n, d, npad, h, hd, Kinv,
                       alpha, part, bt.tab, fin);
    if (out != nullptr && !fuse)
        launch_finalize(z, npad, n, logdet_part, npad / TILE, part, nblocks, h, out, hout, s, hd, bt);
}

void launch_finalize(const double* z, int npad, int n, const double* logdet_part, int nt, const double* part,
                     int nblocks, HyperScalars h, double* out, double* hout, hipStream_t s, const HyperScalars* hd,
                     Batch bt)
{
    hipLaunchKernelGGL(k_finalize, dim3(1, bt.count), dim3(FIN_THREADS), 0, s, z, npad, n, logdet_part, nt, part, nblocks, h,
                       hd, out, hout, bt.tab);
}

void launch_test_gemm_nt(const double* A, const double* B, double* C, int m, int n, int k, hipStream_t s)
{
    set_big_lds();
    hipLaunchKernelGGL(k_test_gemm, dim3((m / TILE) * (n / TILE)), dim3(256), GEMM_LDS, s, A, B, C, n, k, m / TILE);
}

void launch_mfma_peak(double* sink, int blocks, int iters, hipStream_t s)
{
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, s, sink, iters);
}

}  // namespace cugp
