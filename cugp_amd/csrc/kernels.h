// kernels.h -- internal launch interface between the C-ABI (cugp_capi.cpp) and
// the gfx950 kernels (kernels.hip).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>

namespace cugp {

constexpr int TILE = 128;          // tile edge of every fp64 MFMA product and of the padded leading dimension

// launch-shape thresholds, adjustable at run time for A/B tuning.  Every handle carries its own copy (cugp_gp::tune):
// cugp_set_tuning changes the process defaults under a lock, a handle takes them over when it next enqueues (keys set
// with cugp_set_handle_tuning stay as set), and the launchers below read the copy of the handle whose API call is running
// on this thread (t_tune) -- so handles driven from different threads never see each other's settings change mid-call.
enum { TUNE_LAUUM_WM2_MAX = 0,   // K^-1 product: use 64x64 tiles when there are at most this many 128-tiles
       TUNE_TRTRI_WM2_MAX = 1,   // inverse level: same rule
       TUNE_SYRK_REM_MAX = 2,    // trailing update: split the last partial round into 64x64 quarters when it has at most this many tiles
       TUNE_PIPE_BLOCK = 3,      // inverse rows (tiles) handed to the other streams at a time while the factorisation runs; 0 = after it, < 0 = about nt/16 (1 up to 8 tiles)
       TUNE_BORDER_WM2_MAX = 4,  // bordering step 1 (uniform K): 64x64 tiles when there are at most this many 128-tiles
       TUNE_GRAPHS = 5,          // replay single-stream evaluations as a captured HIP graph (1) or launch by launch (0)
       TUNE_GROUP_OVERLAP = 6,   // grouped experts: inverse blocks on the other streams beside the factorisation (1) or after it (0)
       TUNE_GROUP_MAX_TILES = 7, // experts up to this many tiles share launches (default: all; with the inverse beside the factorisation grouping won at every size tried: 4 x 6000 rows 18.9 -> 18.3 ms, 2 x 8192 rows 23.6 -> 21.9 ms)
       TUNE_PANEL = 8,           // two-speed Cholesky: steps per panel (far columns get K = 128*this in one pass per panel); 1 = classic
       TUNE_NEAR_TILES = 9,      // ... tiles in the near window (updated every step, K = 128) at a panel's first step
       TUNE_PANEL_MIN_NT = 10,   // ... only from this many tiles on (small matrices are bound by the chain alone)
       TUNE_LAUUM_STREAM = 11,   // the K^-1 share of an inverse block on its own stream beside the next block's bordering: 0 never, 1 expert groups only, 2 always
       TUNE_FINALIZE_FUSE_MAX = 12, // gradient evaluations: the last block of k_trace takes the final sums (no k_finalize launch) when the trace launch has at most this many blocks; 0 = always the separate launch
       TUNE_SPLIT_REM_MAX = 13,  // uniform-K launches (block-wise K^-1 share, bordering, wide update): a last round of at most this many tiles runs as 64x64 quarters
       TUNE_STEP_QUARTER_MAX = 14, // step kernel: launches of at most this many 64x64 workgroups run ALL their tiles as quarters (chain-bound tail)
       TUNE_STREAM_PRIO = 15,    // read when a handle is created: bit 0 = the factorisation's stream at the highest priority, bit 1 = the inverse streams at the lowest (default 0: prioritised streams serialised grouped experts in round 3)
       TUNE_BARRIER_SPIN = 16,   // polls a workgroup of k_trtri_block spends at a stage barrier before it gives up (the evaluation then fails with CUGP_ERR_DEVICE instead of hanging); 0 = give up at once (test hook)
       TUNE_SUBPANEL = 17,       // near window in sub-panels of this many steps (1, 2 or 4; must divide the panel): the step launch of a sub-panel's last step
                                 // updates the window with K = 128*this in ONE pass over its C tiles, the steps before it only the next column (left-looking inside the sub-panel)
       TUNE_ZFUSE = 18,          // LL-only evaluations: the forward substitution L z = y inside the factorisation's launches (1) or as 2 nt launches behind it (0)
       TUNE_COUNT = 19 };
extern const int g_tune_init[TUNE_COUNT];     // built-in defaults
extern thread_local const int* t_tune;        // the tuning the launchers on this thread read (a handle's copy, or the built-in defaults)
inline int tune(int key) { return t_tune[key]; }

struct HyperScalars;

// Batched launches: the experts of a BCM on one device have the same shapes, so one launch can serve all of
// them -- blockIdx.y picks the expert and the kernel takes its buffers from a device-resident table instead
// of its pointer arguments.  (16 experts x ~45 launches per evaluation from 16 streams were bounded by the
// command processor, not by the CUs.)
struct ExpertPtrs {
    double *A, *T, *U, *Kinv, *d16, *d64, *logdet, *y, *z, *alpha, *w, *part, *out;
    const double* X;
    unsigned* tickets;
    int n;
};
struct Batch {
    const ExpertPtrs* tab = nullptr;   // nullptr: one expert, buffers from the arguments
    int count = 1;
};

struct HyperScalars {              // exp(2*theta) evaluated on the host, as the reference does (covkernel.cpp:65-67)
    double ell_sq, signal_var, noise_var;
};

// ---- SE covariance (N1) ----
// lower 64x64 tiles of K (+ mirror when `full`), padding rows/cols >= n set to identity
// hd (optional, also below): read the hyper-scalars from device memory instead of the argument
void launch_kbuild(const double* X, int n, int d, int npad, HyperScalars h, double* K, bool full,
                   hipStream_t s, const HyperScalars* hd = nullptr, Batch bt = {}, unsigned* tickets = nullptr);
                   // tickets: the factorisation's arrival counters -- 2 * npad/128 per expert ([0, nt) the step tickets of
                   // k_syrk_step, [nt, 2 nt) the stage counters of k_trtri_block) --, zeroed by the launch when given
// S[i][j] = |x_i - x_j|^2 / c, zero diagonal, full symmetric (N2, covkernel.cpp:130-157)
void launch_sqdist(const double* X, int n, int d, int npad, double c, double* S, hipStream_t s);
// Ks[t][i] = sf2 * exp(-0.5*|x_i - xt_t|^2 / l^2), row-major nt_pad x npad (pad = 0)   (N12)
void launch_kcross(const double* X, int n, int d, int npad, const double* Xt, int nt, int ntpad,
                   HyperScalars h, double* Ks, hipStream_t s);

// ---- blocked right-looking Cholesky (N4) on the lower triangle of A (npad x npad, ld = npad) ----
// d16: 16x16 diagonal inverses [nt][8][256]; d64: the two 64x64 diagonal inverses of each block [nt][2][4096]
void launch_potf2(double* A, int ld, int kb, double* d16, double* d64, double* logdet_part, hipStream_t s,
                  Batch bt = {});
// zv / wv (when given): the forward substitution L z = y rides along -- one more workgroup computes z_kb = L_kk^-1 w_kb
// from the block's 64x64 inverses (w: the running right-hand side, y at first); kb = nt - 1 launches that workgroup alone
void launch_trsm_inv64(double* A, const double* d64, int ld, int kb, int nt, hipStream_t s, Batch bt = {},
                       double* zv = nullptr, const double* wv = nullptr);   // 3-phase, 64x64 inverses
void launch_trtri_diag(const double* A, int ld, int kb, int nblocks, const double* d64, double* T, double* U,
                       hipStream_t s, Batch bt = {});
// inverse of the hand-over block of rows [a, a + wb) in one launch: diagonal-tile inverses + every doubling level inside
// the block (k_trtri_block; wb <= TRTRI_BLOCK_MAX_TILES).  ctr: the block's stage counter (zero before the launch; batched:
// tickets[ctr_off] of every expert); poison: log-determinant shares (entry a becomes NaN if a stage wait ran out)
// gcap: most workgroups the launch may hold at its stage barriers (the caller's share of the device's budget for
// barrier grids, cugp_capi.cpp: barrier_cap); hstat: pinned host word ([expert][8] doubles, entry 6) that is set when a
// stage wait ran out -- the evaluation's fetch then returns CUGP_ERR_DEVICE
constexpr int TRTRI_BLOCK_MAX_TILES = 16;
constexpr int TRTRI_BLOCK_MAXWG = 64;
int launch_trtri_block(const double* L, const double* d64, double* T, double* U, int ld, int a, int wb, unsigned* ctr,
                       double* poison, int ctr_off, hipStream_t s, Batch bt = {}, int gcap = TRTRI_BLOCK_MAXWG,
                       double* hstat = nullptr);
// trailing update of step kb fused with the factorisation of diagonal block kb+1 (tickets[kb] must be 0)
// wcol > 0: only the tile columns [kb+1, kb+1+wcol) (two-speed form: the near window)
// ks: first k tile of the pass (sub-panels: the k tiles [ks, kb], at most SUBPANEL_MAX of them); < 0: kb alone
constexpr int SUBPANEL_MAX = 4;
void launch_syrk_step(double* A, int ld, int kb, int nt, double* d16, double* d64, double* logdet_part,
                      unsigned* tickets, hipStream_t s, Batch bt = {}, int wcol = 0, int ks = -1,
                      const double* zv = nullptr, double* wv = nullptr);
                      // zv / wv (when given): nt - kb - 1 more workgroups apply w_i -= L(i,kb) z_kb to the rows below
// wide trailing update: tile columns [ca, cb) (rows >= column) -= L(., k tiles [k0, k0+kw)) L(.)^T; returns tiles
int launch_syrk_wide(double* A, int ld, int nt, int k0, int kw, int ca, int cb, int rev, hipStream_t s, Batch bt = {});

// ---- triangular inverse by recursive doubling (N7) and K^-1 = U U^T (N8) ----
// off: element offset of the diagonal sub-matrix (nt tiles) the level works on inside L, T, U
// (these four return the edge of the output tiles they launched with, in 32s: 4 = 128x128, 2 = 64x64, 0 = nothing)
int launch_trtri_level(const double* L, double* T, double* U, int ld, int nt, int s, int step, hipStream_t st,
                       Batch bt = {}, size_t off = 0);
// bordering step 1, one k chunk: Wt(tj < c1, ti in [ra, ra+rw)) (+)= sum_{k in [max(tj,c0), c1)} U[tj][k] L[ti][k]
int launch_trtri_border1(const double* L, double* T, double* U, int ld, int ra, int rw, int c0, int c1,
                         hipStream_t st, Batch bt = {});
// bordering step 2: rows [a, a+w) of the inverse from their finished Wt and the block's own inverse
int launch_trtri_border2(const double* L, double* T, double* U, int ld, int a, int w, hipStream_t st,
                         Batch bt = {});
// Kinv(lower tiles < a+w) (+)= contribution of inverse rows [a, a+w); a = 0, w = nt: the whole product
int launch_lauum(const double* U, double* Kinv, int ld, int a, int w, hipStream_t s, Batch bt = {});

// ---- prediction products ----
// W[t][i] = sum_{k<=i} Ks[t][k] T[i][k]   (nt_pad x npad, row-major)
void launch_predict_gemm(const double* Ks, const double* T, double* W, int ld, int ntt, int nt, hipStream_t s);
void launch_predict_finish(const double* Ks, const double* W, const double* alpha, int n, int npad, int ntest,
                           HyperScalars h, double* mean, double* var, hipStream_t s);

// ---- vector kernels ----
void launch_trmv_lower(const double* T, int ld, int npad, const double* x, double* z, hipStream_t s,
                       Batch bt = {});                                                                 // z = T x
void launch_trmv_upper(const double* U, int ld, int npad, const double* x, double* a, hipStream_t s,
                       Batch bt = {});                                                                 // a = U x
void launch_trsv_lower(const double* A, const double* T, int ld, int nt, const double* y, double* z,
                       hipStream_t s, Batch bt = {});                                                  // L z = y
void launch_copy_y_to_w(int npad, hipStream_t s, Batch bt);                                            // batched only
// gradient traces (N10+N11 fused): partial sums per block into part[3*nblocks]
int trace_num_blocks(int npad);
// out (when given): the launch also FINISHES the evaluation -- its last block (an arrival ticket, zero before the launch)
// takes the final sums and writes out[0..5] (and hout) as launch_finalize would: z, logdet_part as there; batched:
// taken from the experts' table, ticket = tickets[2 nt] of every expert
void launch_trace(const double* X, int n, int d, int npad, HyperScalars h, const double* Kinv,
                  const double* alpha, double* part, hipStream_t s, const HyperScalars* hd = nullptr,
                  Batch bt = {}, const double* z = nullptr, const double* logdet_part = nullptr, double* out = nullptr,
                  double* hout = nullptr, unsigned* ticket = nullptr);
// arrival counters per expert: [0, nt) step tickets, [nt, 2 nt) stage counters of k_trtri_block, [2 nt] k_trace's fused finalize
constexpr int ticket_count(int nt) { return 2 * nt + 1; }
// out[0..3] = LL, g0, g1, g2  (LL only when part == nullptr)
void launch_finalize(const double* z, int npad, int n, const double* logdet_part, int nt, const double* part,
                     int nblocks, HyperScalars h, double* out, double* hout, hipStream_t s,
                     const HyperScalars* hd = nullptr, Batch bt = {});   // hout: pinned host copy of the results ([expert][8]) or null
// profiling level 4: the NEXT launch of a timed kernel (trailing updates, inverse products, k_trtri_block,
// k_predict_gemm) on this thread carries these events as the dispatch's own start / stop (hipExtLaunchKernelGGL)
void time_next_launch(hipEvent_t start, hipEvent_t stop);
// profiling level 5: the next launch of a timed kernel (the same set, and k_build) on this thread leaves its first
// workgroup's start in slot[0] and its last workgroup's end in slot[STAMP_STRIDE] (s_memrealtime, 100 MHz ticks); the
// caller has set slot[0] = ~0 and slot[STAMP_STRIDE] = 0
constexpr int STAMP_STRIDE = 2048;
void stamp_next_launch(unsigned long long* slot);
bool timing_pending();   // still armed: the launch it was meant for did not happen
// per-device function attributes (dynamic LDS sizes) for the current device; the launchers do it lazily, a
// stream capture must not.  Returns the hipError_t of a failed hipFuncSetAttribute (0 = fine).
int prepare_kernels();

// test hook: C[m x n] = A[m x k] * B[n x k]^T on the MFMA tile path (all multiples of 128 / 16)
void launch_test_gemm_nt(const double* A, const double* B, double* C, int m, int n, int k, hipStream_t s);
// peak probe: `iters` dependent-free fp64 MFMAs per wave; returns nothing, timed by the caller
void launch_mfma_peak(double* sink, int blocks, int iters, hipStream_t s);

}  // namespace cugp
