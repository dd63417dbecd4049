// cugp_capi.cpp -- the C-ABI (include/cugp.h) over the gfx950 kernels.
//
// One cugp_gp owns every device buffer of one expert (the reference keeps them as file-scope
// globals, cuda_scalingdist/cuda_gp.cu:20-95, or as Covsum members, covkernel.h:6-18) and one HIP
// stream on which all of its work is ordered.  Layout in HBM (npad = ceil(n/128)*128, row-major):
//   X     n x d              inputs as given
//   y     npad               labels, zero padded
//   A     npad x npad        K (lower tiles), overwritten by its Cholesky factor L
//   T     npad x npad        L^-1 (lower tiles; strictly-upper tiles are scratch of the inverse)
//   U     npad x npad        L^-T (upper tiles) -- the row-major operand of K^-1 = U U^T
//   Kinv  npad x npad        K^-1 (lower tiles, diagonal tiles complete)
// T, U, Kinv are allocated on first use (gradient / prediction), A on first evaluation.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cugp.h"
#include "group.h"
#include "kernels.h"

using namespace cugp;

namespace {

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIPCHK(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? CUGP_ERR_NOMEM : CUGP_ERR_DEVICE, #call, e_); \
    } while (0)

constexpr int NPHASE = 6;
constexpr int MAX_KEV = 4096;   // per-launch event pairs kept between resets
static_assert(MAX_KEV / 2 == cugp::STAMP_STRIDE, "one stamp slot per timed launch");
constexpr int GRAPH_MAX_TILES = 24;   // evaluations up to 3072 rows are replayed as captured graphs
constexpr int PROF_STRIDE = 16; // profiling level 2 times every 16th step launch, rotating (an event pair costs ~5 us of device time);
                                // level 3 times EVERY launch of the MFMA kernels (bench.py's profiled pass: averages comparable with rocprofv3's)
// kinds of timed launches (cugp_get_kernel_stats_kind): the kernels as rocprofv3 names them
enum { KIND_STEP = 0, KIND_WIDE = 1, KIND_BORDER4 = 2, KIND_BORDER2 = 3, KIND_LAUUM4 = 4, KIND_LAUUM2 = 5,
       KIND_LEVEL4 = 6, KIND_LEVEL2 = 7, KIND_BLOCK = 8, KIND_PREDICT = 9, KIND_BUILD = 10, KIND_TRSM = 11, KIND_COUNT = 12 };

// One hardware queue per stream up to 16 (the runtime default is 4): the experts of a BCM on one device each
// drive their own stream, and 16 experts on 4 queues serialise (5.8 ms vs 4.6 ms per evaluation of 16 x 1500
// rows).  Only a default: an existing setting wins, and it has no effect once the HIP runtime is initialised.
struct QueueDefault {
    QueueDefault() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} g_queue_default;

}  // namespace

// set on the lead expert while a group evaluation is being enqueued: every launch then serves all experts
struct GroupCtx {
    Batch bt;                       // device table of the experts' buffers, expert count
    unsigned* tickets = nullptr;    // the experts' step tickets and block-inverse stage counters, contiguous [k][2 nt]
    double* dout = nullptr;         // results [k][8] on the device ...
    double* hout = nullptr;         // ... and pinned
    bool overlap = false;           // hand the group's inverse blocks to the lead expert's other streams
    int gcap = 0;                   // barrier workgroups per expert of ONE batched k_trtri_block launch (the experts' smallest share; halved with the overlap)
};

struct cugp_gp {
    int n = 0, d = 0, npad = 0, nt = 0, device = 0;
    hipStream_t stream = nullptr;   // all device work of the handle is ordered here ...
    hipStream_t aux = nullptr;      // ... except the inverse blocks that run beside the factorisation (fork/join by events):
    hipStream_t aux2 = nullptr;     // aux = the large products, aux2 = each block's own small inverse,
    hipStream_t lq = nullptr;       // lq = each block's share of K^-1 (beside the next block's bordering)
    std::vector<hipEvent_t> bev;    // fork events, one per inverse block; last: "Wt of the last block rows complete" (aux -> main)
    std::vector<hipEvent_t> oev;    // "block's own inverse done" events (aux2 -> aux); last block: z, alpha done (aux2 -> main)
    std::vector<hipEvent_t> lev;    // "block's inverse rows final" events (aux -> lq / main -> aux2); last: K^-1 shares so far (-> main)
    double *dX = nullptr, *dy = nullptr, *dA = nullptr, *dT = nullptr, *dU = nullptr, *dKinv = nullptr;
    double *dz = nullptr, *dalpha = nullptr, *dw = nullptr, *d16 = nullptr, *dlogdet = nullptr, *dpart = nullptr, *d64 = nullptr;
    double* dout = nullptr;
    unsigned* dtickets = nullptr;  // [0, nt): one arrival counter per factorisation step (k_syrk_step); [nt, 2 nt): the stage
                                   // counter of the hand-over block that starts at that tile row (k_trtri_block)
    double* hout = nullptr;        // pinned, 8 doubles
    int nblocks_trace = 0;
    double hp[3] = {0, 0, 0};
    bool have_data = false;
    bool factor_valid = false;     // A holds L for (data, hp)
    bool inverse_valid = false;    // T, U, Kinv, alpha hold the inverse quantities for (data, hp)
    bool pending = false, pending_grad = false;
    bool zfuse = false;            // record_eval, LL-only: the forward substitution L z = y rides in the factorisation's launches
    bool vec_early = false;        // record_eval: z = L^-1 y, alpha = L^-T z go behind the last block's bordering
    bool vec_done = false;         // ... and were enqueued there
    const GroupCtx* grp = nullptr;  // non-null only inside cugp_group_eval
    double* pred_buf = nullptr;     // prediction scratch (test inputs, cross-covariance, its product with L^-T, results)
    size_t pred_cap = 0;            // ... in doubles, grow-only
    bool overlap = true;           // hand inverse blocks to the other streams while the factorisation runs
    // a single-stream evaluation is captured once as a HIP graph and replayed: one launch call instead of
    // ~60 (1500 rows) -- with 16 experts on a GPU the host's launch rate was the bound, not the device
    HyperScalars* dhs = nullptr;   // device copy of exp(2*theta): what the kernels of a captured evaluation read
    HyperScalars* hhs = nullptr;   // pinned staging for it (refreshed by a copy node at the head of the graph)
    hipGraphExec_t gexec[2] = {nullptr, nullptr};   // [0] log-likelihood only, [1] with gradient
    unsigned gepoch[2] = {0, 0};   // cfg_epoch the graph was captured under
    // launch-shape tuning of THIS handle (kernels.h TUNE_*): the process defaults as of the last sync_tuning(), except the
    // keys set for this handle alone (cugp_set_handle_tuning)
    int tune[TUNE_COUNT] = {};
    bool tune_own[TUNE_COUNT] = {};
    unsigned cfg_epoch = 1;        // bumped when a launch shape of this handle changed: captured graphs carry launch shapes
    bool counted = false;          // this handle is in g_live[device] (the budget of barrier grids, barrier_cap)
    int bar_quota = -1;            // barrier workgroups reserved for this handle from the device's pool; < 0: not yet asked for
    double last_ll = NAN, last_g[3] = {NAN, NAN, NAN}, last_quad = NAN, last_logdet = NAN;
    // profiling
    int prof = 0;
    hipEvent_t pev[NPHASE + 1] = {};
    bool pev_valid = false;
    std::vector<hipEvent_t> kev;   // start/stop pairs around MFMA launches (profiling level 2)
    std::vector<int> kev_kind;     // per pair: KIND_* below
    std::vector<double> kev_flopv; // per pair: algorithmic flop of the launch
    int kev_used = 0;
    unsigned long long* dstamps = nullptr;   // level 5: [0, STAMP_STRIDE) first-workgroup starts, [STAMP_STRIDE, 2 STAMP_STRIDE) last-workgroup ends
    std::vector<unsigned long long> hstamps;
    unsigned eval_seq = 0;         // factorisations enqueued so far (rotates the launches that get timed)
    double kst_ms[KIND_COUNT] = {}, kst_flop[KIND_COUNT] = {};   // folded sums per kernel kind
    double kst_disp_ms[KIND_COUNT] = {};   // level 5: the same launches from the END of the launch in front of them on their stream
                                           // (= where rocprofv3 puts a back-to-back dispatch's begin) to their own end
    long long kst_launches[KIND_COUNT] = {};
    std::vector<int> kev_prev;     // level 5, per pair: the pair of the launch directly in front of it on the same stream, or -1
    int last_main = -1;            // ... the main stream's latest timed launch while the factorisation is being enqueued
};

namespace {

HyperScalars scalars(const cugp_gp* g)
{
    // covkernel.cpp:65-67 -- exp(2*theta) on the host
    return HyperScalars{std::exp(g->hp[0] * 2), std::exp(g->hp[1] * 2), std::exp(g->hp[2] * 2)};
}

int use_device(const cugp_gp* g)
{
    HIPCHK(hipSetDevice(g->device));
    return CUGP_OK;
}

int ensure(double** p, size_t count)
{
    if (*p) return CUGP_OK;
    HIPCHK(hipMalloc((void**)p, count * sizeof(double)));
    return CUGP_OK;
}

int ensure_factor_bufs(cugp_gp* g)
{
    const size_t nn = (size_t)g->npad * g->npad;
    int rc;
    if ((rc = ensure(&g->dA, nn))) return rc;
    if ((rc = ensure(&g->dT, nn))) return rc;   // diagonal-block inverses live in T's diagonal tiles
    if ((rc = ensure(&g->dU, nn))) return rc;
    return CUGP_OK;
}

int ensure_inverse_bufs(cugp_gp* g)
{
    return ensure(&g->dKinv, (size_t)g->npad * g->npad);
}

void drain_kernel_events(cugp_gp* g)
{
    // fold finished per-launch timings into the running sums (everything is on or joined to g->stream, already synchronised)
    const bool stamps = g->prof >= 5 && g->dstamps && g->kev_used > 0;
    if (stamps && hipMemcpy(g->hstamps.data(), g->dstamps, (size_t)2 * STAMP_STRIDE * sizeof(unsigned long long),
                            hipMemcpyDeviceToHost) != hipSuccess) { g->kev_used = 0; return; }
    for (int i = 0; i + 1 < g->kev_used; i += 2) {
        float ms = 0;
        if (stamps) {
            const unsigned long long t0 = g->hstamps[i / 2], t1 = g->hstamps[STAMP_STRIDE + i / 2];
            if (t1 <= t0) continue;                              // (the launch never ran)
            const int kd = g->kev_kind[i / 2];
            g->kst_ms[kd] += (double)(t1 - t0) * 1e-5;           // 100 MHz ticks -> ms
            g->kst_launches[kd] += 1;
            g->kst_flop[kd] += g->kev_flopv[i / 2];
            // dispatch begin .. end as rocprofv3 sees an in-order launch behind another: from the end of the launch in
            // front of it (its workgroups may then still wait for slots held by the other streams' tiles)
            const int pv = g->kev_prev[i / 2];
            const unsigned long long pe = pv >= 0 ? g->hstamps[STAMP_STRIDE + pv] : 0ull;
            g->kst_disp_ms[kd] += (double)(t1 - ((pe > 0 && pe < t1 && pe <= t0) ? pe : t0)) * 1e-5;
            continue;
        }
        if (hipEventElapsedTime(&ms, g->kev[i], g->kev[i + 1]) == hipSuccess) {
            const int kd = g->kev_kind[i / 2];
            g->kst_ms[kd] += ms;
            g->kst_launches[kd] += 1;
            g->kst_flop[kd] += g->kev_flopv[i / 2];
        }
    }
    g->kev_used = 0;
}

Batch B(const cugp_gp* g) { return g->grp ? g->grp->bt : Batch{}; }

// pinned host buffer the evaluation's results land in: the group's ([expert][8]) or the handle's.  Entries 0..5: LL, the
// three gradients, y'K^-1y, log|K| (k_finalize); entry 6: set by a kernel whose bounded wait ran out (k_trtri_block)
double* host_out(const cugp_gp* g) { return g->grp ? g->grp->hout : g->hout; }

// ---- tuning: process defaults (cugp_set_tuning) under a lock, one copy per handle ----
std::mutex g_tune_mu;
int g_tune_default[TUNE_COUNT];
bool g_tune_default_init = false;

void tune_defaults_locked()
{
    if (!g_tune_default_init) {
        for (int k = 0; k < TUNE_COUNT; k++) g_tune_default[k] = g_tune_init[k];
        // CUGP_TUNE="key=value,key=value": process defaults from the environment (A/B runs of whole test suites)
        if (const char* e = getenv("CUGP_TUNE")) {
            while (*e) {
                char* end = nullptr;
                const long k = strtol(e, &end, 10);
                if (end == e || *end != '=') break;
                e = end + 1;
                const long v = strtol(e, &end, 10);
                if (end == e) break;
                if (k >= 0 && k < TUNE_COUNT) g_tune_default[k] = (int)v;
                e = *end == ',' ? end + 1 : end;
                if (*end != ',' ) break;
            }
        }
        g_tune_default_init = true;
    }
}

// Take the process defaults over into the handle (keys it owns excepted); called where an API call starts to
// enqueue.  A changed launch shape invalidates the handle's captured graphs.
void sync_tuning(cugp_gp* g)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    tune_defaults_locked();
    bool changed = false;
    for (int k = 0; k < TUNE_COUNT; k++)
        if (!g->tune_own[k] && g->tune[k] != g_tune_default[k]) {
            if (k != TUNE_GRAPHS) changed = true;
            g->tune[k] = g_tune_default[k];
        }
    if (changed) g->cfg_epoch++;
}

// While an API call enqueues for handle g the launchers of kernels.hip read g's tuning (thread-local pointer).
struct TuneScope {
    const int* prev;
    explicit TuneScope(cugp_gp* g) : prev(t_tune) { sync_tuning(g); t_tune = g->tune; }
    ~TuneScope() { t_tune = prev; }
};

// ---- budget of barrier grids ----
// k_trtri_block is a grid whose workgroups wait for each other: every launch of it needs all G of its workgroups
// resident at once.  One launch is safe by construction (G <= 64, far below the 512 slots; workgroups of a launch are
// dispatched in order), but launches of DIFFERENT handles interleave on the device: 9 handles x 64 workgroups exceed the
// slots, every launch is partly resident and each waits for workgroups that cannot be dispatched (a stall of seconds,
// then NaN results).  So the handles of a device share a pool of 384 barrier workgroups, and a handle's share is
// RESERVED: taken from the pool at the handle's first barrier launch -- 384 / (live handles on the device at that
// moment), at most 64, at most what is left -- kept unchanged for the handle's lifetime and returned when it is destroyed.
// A fixed share is what a captured graph may bake in: graphs captured while few handles were alive keep their grids, and
// handles created later get what is left of the pool (none: their block inverses go launch by launch, k_trtri_diag +
// k_trtri_level, which wait for nothing -- the same bits).  A handle whose inverse blocks run beside its factorisation can
// have TWO barrier launches in flight (the block in front of the last on aux2, the last block's on the main stream):
// each takes half the share.  The sum over everything in flight therefore stays <= 384 of the 512 slots whatever the
// interleaving, so some launch always gets the workgroups it is missing.
// (Per process: handles of other processes on the same GPU are not seen.)
constexpr int BARRIER_POOL = 384;
std::atomic<int> g_live[64];
std::atomic<int> g_bar_used[64];
int barrier_share(cugp_gp* g)
{
    if (g->bar_quota >= 0) return g->bar_quota;
    const int dev = g->device >= 0 && g->device < 64 ? g->device : 63;
    int live = g_live[dev].load(std::memory_order_relaxed);
    if (live < 1) live = 1;
    int want = live > 24 ? 0 : BARRIER_POOL / live;
    if (want > TRTRI_BLOCK_MAXWG) want = TRTRI_BLOCK_MAXWG;
    int used = g_bar_used[dev].load(std::memory_order_relaxed), grant;
    do {
        grant = want < BARRIER_POOL - used ? want : BARRIER_POOL - used;
        if (grant < 8) grant = 0;                              // (not worth a barrier grid: launch by launch)
    } while (grant > 0 && !g_bar_used[dev].compare_exchange_weak(used, used + grant, std::memory_order_relaxed));
    g->bar_quota = grant;
    return grant;
}
void barrier_release(cugp_gp* g)
{
    if (g->bar_quota > 0) g_bar_used[g->device >= 0 && g->device < 64 ? g->device : 63].fetch_sub(g->bar_quota, std::memory_order_relaxed);
    g->bar_quota = -1;
}
// workgroups ONE barrier launch of this handle may hold; 0: none (launch by launch)
int barrier_cap(cugp_gp* g)
{
    if (g->grp) return g->grp->gcap;                           // (the group's: cugp_group_enqueue)
    const int share = barrier_share(g);
    return g->overlap && share >= 16 ? share / 2 : share;
}

// block rows per hand-over to the other streams: about a sixteenth of the matrix, at least 2 tiles (interleaved
// A/B at 16, 32, 64 and 79 tiles: 2, 2-3, 4, 5 were the best), single tiles up to 8 tiles (600 and 1000 rows: -9 %
// and -3 % against 2), or as tuned; 0 = no hand-over (everything on the main stream after the factorisation)
int pipe_block(const cugp_gp* g, bool with_inverse)
{
    const bool overlap = g->grp ? g->grp->overlap : g->overlap;
    int w = (with_inverse && overlap) ? g->tune[TUNE_PIPE_BLOCK] : 0;
    if (w < 0) w = g->nt <= 8 ? 1 : ((g->nt + 8) / 16 > 2 ? (g->nt + 8) / 16 : 2);
    return w >= g->nt ? 0 : w;
}

// one timed launch: event pair + bookkeeping.  Levels 2 and 3 record an event in front of and behind the launch on its
// stream; level 4 hands the pair to the launch itself (hipExtLaunchKernelGGL: the dispatch's own begin / end, no extra
// packets on the stream -- the schedule of the timed pass, and the durations rocprofv3 --kernel-trace reports)
// level 5: no events at all -- the launch's own workgroups stamp a slot of g->dstamps (kernels.hip LaunchStamp)
struct TimedLaunch {
    cugp_gp* g; hipStream_t s; bool on, ext;
    // chain: this launch goes out on the main stream directly behind the main stream's latest timed launch
    TimedLaunch(cugp_gp* g_, hipStream_t s_, bool want, bool chain = false)
        : g(g_), s(s_), on(want && g_->kev_used + 2 <= (int)g_->kev.size()), ext(g_->prof >= 4)
    {
        if (!on) { if (chain) g->last_main = -1; return; }       // an untimed launch breaks the chain
        g->kev_prev[g->kev_used / 2] = chain ? g->last_main : -1;
        if (chain) g->last_main = g->kev_used / 2;
        if (g->prof >= 5) stamp_next_launch(g->dstamps + g->kev_used / 2);
        else if (ext) time_next_launch(g->kev[g->kev_used], g->kev[g->kev_used + 1]);
        else if (hipEventRecord(g->kev[g->kev_used], s) != hipSuccess) on = false;
    }
    ~TimedLaunch() { if (on && ext && timing_pending()) { time_next_launch(nullptr, nullptr); stamp_next_launch(nullptr); } }   // the launch did not happen
    void done(int kind, double flop)
    {
        if (!on) return;
        if (ext) { if (timing_pending()) return; }            // (armed but not consumed: nothing was launched)
        else if (hipEventRecord(g->kev[g->kev_used + 1], s) != hipSuccess) return;
        g->kev_kind[g->kev_used / 2] = kind;
        g->kev_flopv[g->kev_used / 2] = flop;
        g->kev_used += 2;
    }
};

// is this launch one of the timed ones?  level 2: one in `every`, rotating with `phase`; levels 3 and 4: all
bool sampled(const cugp_gp* g, int phase, int every) { return g->prof >= 3 || (g->prof == 2 && phase % every == 0); }

// algorithmic flop of the inverse's tile products (multiply + add; a k tile that is triangular counts half).
// border step 1, one chunk: Wt(tj < c1, ti in [ra, ra+rw)) (+)= sum_{k in [max(tj,c0), c1)} U[tj][k] L[ti][k]
double border1_flop(int rw, int c0, int c1)
{
    double kt = 0;
    for (int tj = 0; tj < c1; tj++) kt += (c1 - (tj > c0 ? tj : c0)) - (tj >= c0 ? 0.5 : 0.0);   // U[tj][tj] is triangular
    return kt * rw * 2.0 * TILE * TILE * TILE;
}
// border step 2: T(ti in [a, a+w), tj < a) = -sum_{k in [a, ti]} T[ti][k] Wt[tj][k]   (T[ti][ti] triangular)
double border2_flop(int a, int w)
{
    double kt = 0;
    for (int i = 0; i < w; i++) kt += i + 0.5;
    return kt * a * 2.0 * TILE * TILE * TILE;
}
// K^-1 share of inverse rows [a, a+w): Kinv(ti,tj) (+)= sum_{k in [max(ti,a), a+w)} U[ti][k] U[tj][k]^T, tj <= ti < a+w
double lauum_flop(int a, int w)
{
    double kt = 0;
    for (int ti = 0; ti < a + w; ti++) {
        const double k = (a + w) - (ti > a ? ti : a);
        kt += (ti + 1) * k - (ti >= a ? 0.5 * (ti + 1) : 0.0) - (ti >= a ? 0.5 * k : 0.0);   // triangular k tile; diagonal output tile
    }
    return kt * 2.0 * TILE * TILE * TILE;
}
// one level of recursive doubling over nt tiles, blocks of s: both steps move |A| x |B| tiles with k ranges up to s
double level_flop(int nt, int s, int step)
{
    double kt = 0;
    for (int a0 = 0; a0 + s < nt; a0 += 2 * s) {
        int sb = nt - (a0 + s);
        if (sb > s) sb = s;
        if (step == 1) for (int ja = 0; ja < s; ja++) kt += (s - ja - 0.5) * sb;       // k from tj to the end of A
        else for (int ib = 0; ib < sb; ib++) kt += (ib + 0.5) * s;                     // k from b0 to ti
    }
    return kt * 2.0 * TILE * TILE * TILE;
}

// The block's own inverse: T and U of the diagonal block of rows [a, a + wb), on stream o.  One launch (k_trtri_block:
// diagonal-tile inverses + every doubling level, stage counter tickets[nt + a]) for hand-over blocks; the whole-matrix
// form (wb > TRTRI_BLOCK_MAX_TILES: large levels want 128x128 tiles and fill the chip) stays launch by launch.
int enqueue_block_own_inverse(cugp_gp* g, int a, int wb, hipStream_t o)
{
    const int ld = g->npad;
    const int gcap = barrier_cap(g);                         // this handle's share of the device's barrier workgroups; 0: none
    if (wb <= TRTRI_BLOCK_MAX_TILES && gcap > 0) {
        unsigned* base = g->grp ? g->grp->tickets : g->dtickets;
        TimedLaunch tl(g, o, sampled(g, (a / (wb > 0 ? wb : 1)) + (int)g->eval_seq, 4));
        launch_trtri_block(g->dA, g->d64, g->dT, g->dU, ld, a, wb, base + g->nt + a, g->dlogdet, g->nt + a, o, B(g), gcap,
                           host_out(g));
        double fl = 0;
        for (int s = 1; s < wb; s *= 2) fl += level_flop(wb, s, 1) + level_flop(wb, s, 2);
        tl.done(KIND_BLOCK, fl);
        return CUGP_OK;
    }
    const size_t off = (size_t)a * TILE * ld + (size_t)a * TILE;
    launch_trtri_diag(g->dA, ld, a, wb, g->d64, g->dT, g->dU, o, B(g));
    for (int s = 1; s < wb; s *= 2)
        for (int step = 1; step <= 2; step++) {
            TimedLaunch tl(g, o, sampled(g, a + s + step + (int)g->eval_seq, 16));  // small launches: one in sixteen
            const int wm = launch_trtri_level(g->dA, g->dT, g->dU, ld, wb, s, step, o, B(g), off);
            if (wm) tl.done(wm == 4 ? KIND_LEVEL4 : KIND_LEVEL2, level_flop(wb, s, step));
        }
    return CUGP_OK;
}

// Inverse quantities of block rows [a, b): T and U = T^T (diagonal-tile inverses, doubling inside the block,
// bordering against the finished rows [0, a)) and the block's share of K^-1 = T^T T (when Kinv is wanted).
// The block's own inverse is a chain of small launches; on its own stream `xs` (when given) it runs beside
// the large products of the previous block instead of in front of this block's.  The K^-1 share only needs
// the block's rows of U: on its own stream `lq` (when given) it runs beside the NEXT block's bordering -- two
// independent large launches in flight fill each other's last, partly empty round of workgroups.
int enqueue_inverse_block(cugp_gp* g, int a, int b, bool kinv, hipStream_t x, hipStream_t xs, hipEvent_t own_done,
                          hipStream_t lq = nullptr, hipEvent_t rows_final = nullptr, bool before_last = false)
{
    const int ld = g->npad, wb = b - a;
    hipStream_t o = xs ? xs : x;
    int rc0;
    if ((rc0 = enqueue_block_own_inverse(g, a, wb, o))) return rc0;
    if (xs) {
        HIPCHK(hipEventRecord(own_done, xs));
        HIPCHK(hipStreamWaitEvent(x, own_done, 0));
    }
    // rows [a, b): their Wt was accumulated chunk by chunk while the earlier blocks became final
    const bool timed2 = sampled(g, (a / (wb > 0 ? wb : 1)) + (int)g->eval_seq, 4);   // large launches: every fourth block
    if (a > 0) {
        TimedLaunch tl(g, x, timed2);
        const int wm = launch_trtri_border2(g->dA, g->dT, g->dU, ld, a, wb, x, B(g));
        if (wm) tl.done(wm == 4 ? KIND_BORDER4 : KIND_BORDER2, border2_flop(a, wb));
    }
    auto kinv_share = [&](hipStream_t st) {
        TimedLaunch tl(g, st, timed2);
        const int wm = launch_lauum(g->dU, g->dKinv, ld, a, wb, st, B(g));
        tl.done(wm == 4 ? KIND_LAUUM4 : KIND_LAUUM2, lauum_flop(a, wb));
    };
    if (kinv && lq) {
        HIPCHK(hipEventRecord(rows_final, x));
        HIPCHK(hipStreamWaitEvent(lq, rows_final, 0));
        kinv_share(lq);
        if (before_last) HIPCHK(hipEventRecord(g->lev.back(), lq));      // the shares of K^-1 so far (enqueue_last_block)
    }
    // ... and these rows, now final, go into the Wt of every row below them
    if (b < g->nt) {
        TimedLaunch tl(g, x, timed2);
        const int wm = launch_trtri_border1(g->dA, g->dT, g->dU, ld, b, g->nt - b, a, b, x, B(g));
        if (wm) tl.done(wm == 4 ? KIND_BORDER4 : KIND_BORDER2, border1_flop(g->nt - b, a, b));
    }
    if (before_last) HIPCHK(hipEventRecord(g->bev.back(), x));           // Wt of the last rows is complete
    if (b == g->nt && g->vec_early) {
        // L^-1 is complete: z = L^-1 y and alpha = L^-T z run BESIDE the last share of K^-1 (~130 us at N = 8192)
        // instead of after it on the main stream: on the stream of the block's own inverse (idle by now) when the
        // share follows on this one
        hipStream_t vs = x;
        if (kinv && !lq && xs && rows_final) {
            HIPCHK(hipEventRecord(rows_final, x));
            HIPCHK(hipStreamWaitEvent(xs, rows_final, 0));
            vs = xs;
        }
        launch_trmv_lower(g->dT, g->npad, g->npad, g->dy, g->dz, vs, B(g));
        launch_trmv_upper(g->dU, g->npad, g->npad, g->dz, g->dalpha, vs, B(g));
        g->vec_done = true;
        if (vs != x) {
            HIPCHK(hipEventRecord(own_done, xs));
            if (kinv && !lq) kinv_share(x);
            HIPCHK(hipStreamWaitEvent(x, own_done, 0));             // x's last event covers both again
            return CUGP_OK;
        }
    }
    if (kinv && !lq) {
        kinv_share(x);
        if (before_last) HIPCHK(hipEventRecord(g->lev.back(), x));
    }
    return CUGP_OK;
}

// The LAST block of inverse rows [a, nt) on the MAIN stream: behind the factorisation nothing else is left there, and
// the block's chain -- diagonal inverses, doubling, bordering, its share of K^-1, then the traces -- crosses no stream
// boundary any more (every hand-over cost ~10 us of the evaluation's serial tail: 3 of them are 5 % of a 1500-row
// evaluation).  The other streams are waited for exactly where their results are needed: Wt complete (bev.back(), recorded
// behind the previous block's border1 launches) before the bordering, the earlier shares of K^-1 (lev.back()) before this
// one -- so the bordering no longer queues behind the previous block's share.  z = L^-1 y and alpha = L^-T z run beside
// the share on aux2.
int enqueue_last_block(cugp_gp* g, int a, int idx)
{
    const int nt = g->nt, ld = g->npad, wb = nt - a;
    hipStream_t m = g->stream;
    int rc0;
    if ((rc0 = enqueue_block_own_inverse(g, a, wb, m))) return rc0;
    const bool timed2 = sampled(g, (a / (wb > 0 ? wb : 1)) + (int)g->eval_seq, 4);
    HIPCHK(hipStreamWaitEvent(m, g->bev.back(), 0));
    {
        TimedLaunch tl(g, m, timed2);
        const int wm = launch_trtri_border2(g->dA, g->dT, g->dU, ld, a, wb, m, B(g));
        if (wm) tl.done(wm == 4 ? KIND_BORDER4 : KIND_BORDER2, border2_flop(a, wb));
    }
    // z = L^-1 y, alpha = L^-T z: beside the share on aux2 when they are long enough to be worth two hand-overs
    // (~130 us at 8192 rows, ~90 us for 16 x 1500 rows against ~10 us per hand-over; 14 us at 1500 rows: in line there)
    const bool beside = g->vec_early && nt * (g->grp ? g->grp->bt.count : 1) > 24;
    if (g->vec_early) {
        hipStream_t vs = m;
        if (beside) {
            HIPCHK(hipEventRecord(g->lev[idx], m));
            HIPCHK(hipStreamWaitEvent(g->aux2, g->lev[idx], 0));
            vs = g->aux2;
        }
        launch_trmv_lower(g->dT, g->npad, g->npad, g->dy, g->dz, vs, B(g));
        launch_trmv_upper(g->dU, g->npad, g->npad, g->dz, g->dalpha, vs, B(g));
        if (beside) HIPCHK(hipEventRecord(g->oev[idx], g->aux2));
        g->vec_done = true;
    }
    HIPCHK(hipStreamWaitEvent(m, g->lev.back(), 0));
    {
        TimedLaunch tl(g, m, timed2);
        const int wm = launch_lauum(g->dU, g->dKinv, ld, a, wb, m, B(g));
        tl.done(wm == 4 ? KIND_LAUUM4 : KIND_LAUUM2, lauum_flop(a, wb));
    }
    if (beside) HIPCHK(hipStreamWaitEvent(m, g->oev[idx], 0));
    return CUGP_OK;
}

int phase_mark(cugp_gp* g, int i);
int fetch_eval(cugp_gp* g);

// level 5: the slots the timed launches enqueued from here on stamp (the sums of the ones before were drained by the
// fetch that synchronised them): starts to ~0, ends to 0, on the handle's stream in front of those launches
int reset_stamps(cugp_gp* g)
{
    if (g->prof < 5 || !g->dstamps) return CUGP_OK;
    HIPCHK(hipMemsetAsync(g->dstamps, 0xff, (size_t)STAMP_STRIDE * sizeof(unsigned long long), g->stream));
    HIPCHK(hipMemsetAsync(g->dstamps + STAMP_STRIDE, 0, (size_t)STAMP_STRIDE * sizeof(unsigned long long), g->stream));
    return CUGP_OK;
}


// The shares of K^-1 on a stream of their own (beside the next block's bordering) or behind their block's bordering:
// for a group of experts two large launches in flight fill each other's partly empty last rounds (-3...-6 % per
// evaluation at 16x1500, 8x3000, 4x6000); for a single matrix the extra hand-over costs what it gains or more
// (1500: +1.9 %, 6000: +3.3 %, 8192 / 10000: +-0.3 %, 16384: +0.7 %).  TUNE_LAUUM_STREAM: 0 never, 1 groups, 2 always.
bool kinv_stream(const cugp_gp* g)
{
    const int v = g->tune[TUNE_LAUUM_STREAM];
    return v >= 2 || (v == 1 && g->grp != nullptr);
}

// (the fork event bev[idx] was recorded by the caller, on the factorisation's stream behind the panel solve that made
//  the rows final)
int fork_inverse_block(cugp_gp* g, int a, int b, int idx, bool before_last)
{
    HIPCHK(hipStreamWaitEvent(g->aux, g->bev[idx], 0));
    HIPCHK(hipStreamWaitEvent(g->aux2, g->bev[idx], 0));
    const bool own = kinv_stream(g);
    return enqueue_inverse_block(g, a, b, true, g->aux, g->aux2, g->oev[idx], own ? g->lq : nullptr, g->lev[idx],
                                 before_last);
}

// What step kb of the two-speed factorisation launches (pure column arithmetic, shared with the test hook
// cugp_potrf_plan so the schedule can be replayed on a CPU: every tile must see every k exactly once, in order).
//
// Steps are grouped in panels of P.  The step launch (k_syrk_step, K = 128, with the next diagonal block
// factored inside it) updates only the NEAR window: the tile columns [kb+1, F_p), F_p fixed per panel and chosen
// so that the window holds about `near` tiles at the panel's first step -- enough to fill the chip for the
// ~45 us the diagonal block needs, so the latency-bound chain stays hidden exactly as in the classic form.
// The FAR columns [F_p, nt) get the whole panel at once, K = P*128, from k_syrk_wide when the panel's last
// column is solved (one pass over their C tiles instead of P: the tile product costs a fixed ~12 us per C tile
// + 30.5 us per 128 of K).  F_p never decreases, so a column is either inside every later window or was
// brought up to date by every earlier wide pass.  Once the window reaches the last column (small trailing
// matrices, where the chain is the bound anyway) this IS the classic right-looking form.
struct StepPlan {
    int wide_k0, wide_kw;      // k tiles of the wide update issued at this step (after its panel solve) ...
    int wa0, wa1;              // ... over the tile columns [wa0, wa1); empty: none
    int wcol;                  // step launch: columns [kb+1, kb+1+wcol) ...
    int ks;                    // ... receive the k tiles [ks, kb] in one pass (sub-panels; ks = kb: one k tile per step)
};

static int tri_tiles(int n) { return n * (n + 1) / 2; }

// far boundary of panel p: first tile column NOT in the near window
int far_boundary(int nt, int P, int near, int p)
{
    int F = 0;
    for (int q = 0; q <= p; q++) {                                    // monotone over the panels
        const int k0 = q * P, m = nt - k0 - 1;                        // trailing size at the panel's first step
        int wc = P;                                                   // at least the panel itself and one more column
        while (wc < m && (wc >= m ? tri_tiles(m) : tri_tiles(wc) + (m - wc) * wc) < near) wc++;
        int f = k0 + 1 + wc;
        if (f < (q + 1) * P + 1) f = (q + 1) * P + 1;
        if (f > nt) f = nt;
        // the far pass costs whole rounds of 512 tile slots (134 us each at K = 512): move the boundary up to three
        // columns out if that makes its tri(nt - f) tiles end close to a full round
        {
            int best = f;
            double bw = 2.0;
            for (int c = f; c <= f + 3 && c < nt; c++) {
                const int tiles = tri_tiles(nt - c), rem = tiles % 512;
                const double waste = rem == 0 ? 0.0 : (512.0 - rem) / 512.0 / (tiles / 512 + 1);   // idle share of the pass
                if (waste < bw - 0.02) { bw = waste; best = c; }
            }
            f = best;
        }
        if (f > F) F = f;
    }
    return F;
}

// Sub-panels (S > 1 steps, aligned from step 0; S divides P): the near window is right-looking from sub-panel to
// sub-panel and left-looking inside one.  The step launch of a sub-panel's LAST step updates the whole window with the
// sub-panel's S k tiles in one pass (K = 128 S: the fixed cost of a pass over a C tile -- 256 KB of C traffic, prologue,
// dispatch -- is paid once per S steps; K = 128 ran at 45 TF/s, 256 at 55, 512 at 59); the steps before it update only
// the column the chain needs next, with the k tiles of the sub-panel so far (K = 128 .. 128 (S - 1)).  A column inside
// sub-panel [s0, s0 + S) has then seen every k < s0 (earlier window and wide passes) and [s0, column) (its own step).
StepPlan plan_step(int nt, int P, int near, int kb, int S = 1)
{
    StepPlan sp{0, 0, 0, 0, nt - kb - 1, kb};
    if (S < 1 || S > SUBPANEL_MAX || (P > 1 && P % S != 0)) S = 1;
    int F = nt;
    if (P > 1) {
        const int p = kb / P, i = kb % P;
        F = far_boundary(nt, P, near, p);
        if (i == P - 1 && F < nt) {
            sp.wide_k0 = p * P;
            sp.wide_kw = P;
            sp.wa0 = F;
            sp.wa1 = nt;
        }
    }
    sp.ks = kb - kb % S;
    sp.wcol = (kb % S == S - 1) ? F - (kb + 1) : 1;
    return sp;
}

// panel width of the look-ahead factorisation for this handle (1 = classic right-looking, K = 128 per pass)
int panel_width(const cugp_gp* g)
{
    int P = g->tune[TUNE_PANEL];
    if (P < 2 || g->nt < g->tune[TUNE_PANEL_MIN_NT] || g->nt < 3 * P) return 1;
    return P > 32 ? 32 : P;
}

// algorithmic flop of a trailing update over the tile columns [ca, cb) of an nt-tile matrix with kw k tiles:
// entries on or below the diagonal only (a diagonal tile counts half), multiply + add
double trailing_flop(int nt, int ca, int cb, int kw)
{
    if (cb > nt) cb = nt;
    double entries = 0;
    for (int c = ca; c < cb; c++) entries += ((double)(nt - c) - 0.5) * TILE * TILE;
    return entries * kw * TILE * 2.0;
}

// Blocked right-looking Cholesky of A (lower).  Handle's stream, two launches per step:
//   panel solve(k)  ->  [trailing update(k) + factorisation of diagonal block k+1] in ONE launch
// (k_syrk_step: the latency-bound diagonal block runs inside the update launch), and in the two-speed form
// (panels of P > 1 steps, plan_step above; cpp_matrixalgebra/blocked_cholesky.cpp:221-262 is the b = 2 ancestor)
// one k_syrk_wide pass per panel over the far columns.  Every tile sees its updates in a fixed order whatever
// the timing: results are reproducible.
// with_inverse: L^-1 and K^-1 are built block row by block row on further streams as the rows of L become final.
int enqueue_continue(cugp_gp* g);

// zeroed_tickets: the arrival counters a launch already in front of this on the stream has zeroed (record_eval: the
// covariance build, launch_kbuild's `tickets` argument) -- it must be the very buffer the step launches count in, or
// the counters are cleared here (a stale counter means a diagonal block factored twice or never).
int enqueue_potrf(cugp_gp* g, bool with_inverse, bool mark = false, const unsigned* zeroed_tickets = nullptr)
{
    int rc;
    const int nt = g->nt, ld = g->npad;
    const int w = pipe_block(g, with_inverse);
    const int P = panel_width(g);
    const int near = g->tune[TUNE_NEAR_TILES];
    const int S = g->tune[TUNE_SUBPANEL];
    hipStream_t m = g->stream;                              // the whole factorisation is ordered on the handle's stream
    int nblk = 0, done = 0;                                 // blocks forked so far, block rows handed over
    struct Fork { int a, b; };
    std::vector<Fork> forks;
    const bool chain_first = nt * (g->grp ? g->grp->bt.count : 1) <= 24;
    // (Round 2 handed the first nt - 2w block rows of small matrices over in ONE late block: every hand-over cost the
    //  main stream a bubble the ~50-us chain steps could not afford.  With the round-3 chain -- ~32 us per step -- the
    //  fine-grained hand-over wins at every size again: 1500 rows 0.77 -> 0.65 ms, 2 x 1500 rows 0.80 -> 0.73 ms.)
    g->eval_seq++;
    const unsigned* my_tickets = g->grp ? g->grp->tickets : g->dtickets;
    if (zeroed_tickets && zeroed_tickets == my_tickets) {}   // (the covariance build in front of it did that)
    else if (g->grp) HIPCHK(hipMemsetAsync(g->grp->tickets, 0, (size_t)g->grp->bt.count * ticket_count(nt) * sizeof(unsigned), m));
    else HIPCHK(hipMemsetAsync(g->dtickets, 0, (size_t)ticket_count(nt) * sizeof(unsigned), m));
    launch_potf2(g->dA, ld, 0, g->d16, g->d64, g->dlogdet, m, B(g));
    const bool zf = g->zfuse && !with_inverse;              // (LL-only: z = L^-1 y inside the panel-solve and step launches)
    g->last_main = -1;
    const bool l5 = g->prof >= 5;                           // level 5: every launch of the factorisation's stream is stamped
    for (int kb = 0; kb + 1 < nt; kb++) {
        {
            TimedLaunch tl(g, m, l5, true);
            launch_trsm_inv64(g->dA, g->d64, ld, kb, nt, m, B(g), zf ? g->dz : nullptr, g->dw);
            tl.done(KIND_TRSM, 2.0 * (nt - kb - 1) * TILE * (double)TILE * TILE / 2);      // (triangular solve: half a product)
        }
        // block rows < kb+1 of L are final, and so are the columns <= kb of every row below them
        const int b = kb + 1;
        const bool hand_over = w > 0 && b - done >= w;
        if (hand_over) HIPCHK(hipEventRecord(g->bev[nblk], m));
        const StepPlan sp = plan_step(nt, P, near, kb, S);
        if (sp.wa1 > sp.wa0) {
            // panel p is factored (its last panel solve is enqueued): the far columns get the panel's K = P*128 in
            // one pass, before the next panel's first step widens the near window into them
            TimedLaunch tl(g, m, g->prof >= 2, true);
            launch_syrk_wide(g->dA, ld, nt, sp.wide_k0, sp.wide_kw, sp.wa0, sp.wa1, (kb / P) & 1, m, B(g));
            tl.done(KIND_WIDE, trailing_flop(nt, sp.wa0, sp.wa1, sp.wide_kw));
        }
        // level 2 times a rotating eighth of the step launches (every step is sampled once in 8 evaluations):
        // an event pair around every launch costs several percent of the evaluation
        TimedLaunch tl(g, m, sampled(g, kb + (int)g->eval_seq, PROF_STRIDE), true);
        // (look-ahead form: plain stores -- the panel solve that follows reads the column at once, and reading
        //  freshly non-temporally stored tiles took it 50 us instead of 16)
        launch_syrk_step(g->dA, ld, kb, nt, g->d16, g->d64, g->dlogdet, g->dtickets, m, B(g), sp.wcol, sp.ks,
                         zf ? g->dz : nullptr, g->dw);
        tl.done(KIND_STEP, trailing_flop(nt, kb + 1, kb + 1 + sp.wcol, kb + 1 - sp.ks));
        if (hand_over) {
            // The block's own 6-8 launches take the host 15-35 us.  Where a chain step is shorter than that (small
            // matrices: ~35 us per step) they are enqueued behind the WHOLE chain of the factorisation -- enqueued at
            // the hand-over they left the main stream dry at every block, and the host is far ahead of the device
            // again before the first block is due.  Long steps (many tiles, or a group of experts): right here,
            // behind the step launch, so that the inverse streams start as early as the device allows even under a
            // slow host (a profiler doubles the host's cost per launch).
            if (chain_first) forks.push_back({done, b});
            else if ((rc = fork_inverse_block(g, done, b, nblk, nt - b <= w))) return rc;    // (true: the next block is the last)
            done = b;
            nblk++;
        }
    }
    if (mark && (rc = phase_mark(g, 2))) return rc;         // end of the factorisation on the main stream
    for (size_t i = 0; i < forks.size(); i++)
        if ((rc = fork_inverse_block(g, forks[i].a, forks[i].b, (int)i, nt - forks[i].b <= w))) return rc;
    if (w > 0) {
        if ((rc = enqueue_last_block(g, done, nblk))) return rc;      // waits for everything the other streams still do
    } else if (with_inverse) {
        if ((rc = enqueue_inverse_block(g, 0, nt, true, m, nullptr, nullptr))) return rc;
    } else if (zf) {
        // the last block's share of z (its diagonal block was factored inside the last step launch); nothing else of the
        // inverse is needed for a log-likelihood
        launch_trsm_inv64(g->dA, g->d64, ld, nt - 1, nt, m, B(g), g->dz, g->dw);
    } else {
        // inverses of all diagonal factor blocks at once (off the factorisation's critical path)
        launch_trtri_diag(g->dA, ld, 0, nt, g->d64, g->dT, g->dU, m, B(g));
    }
    HIPCHK(hipGetLastError());
    return CUGP_OK;
}

int enqueue_trtri(cugp_gp* g)
{
    for (int s = 1; s < g->nt; s *= 2) {
        launch_trtri_level(g->dA, g->dT, g->dU, g->npad, g->nt, s, 1, g->stream);
        launch_trtri_level(g->dA, g->dT, g->dU, g->npad, g->nt, s, 2, g->stream);
    }
    HIPCHK(hipGetLastError());
    return CUGP_OK;
}

int phase_mark(cugp_gp* g, int i)
{
    if (g->prof >= 1) HIPCHK(hipEventRecord(g->pev[i], g->stream));
    return CUGP_OK;
}

// K build + factorisation (+ inverse quantities, traces when want_grad); results land in dout.
// hd: the kernels read the hyper-scalars from device memory (a copy node refreshes it) -- the captured form.
int record_eval(cugp_gp* g, bool want_grad, const HyperScalars* hd)
{
    int rc;
    const HyperScalars h = scalars(g);
    hipStream_t s = g->stream;
    if (hd) HIPCHK(hipMemcpyAsync(g->dhs, g->hhs, sizeof(HyperScalars), hipMemcpyHostToDevice, s));
    if ((rc = phase_mark(g, 0))) return rc;
    unsigned* const tickets = g->grp ? g->grp->tickets : g->dtickets;
    if ((rc = reset_stamps(g))) return rc;
    {
        TimedLaunch tl(g, s, g->prof >= 5);
        launch_kbuild(g->dX, g->n, g->d, g->npad, h, g->dA, false, s, hd, B(g), tickets);   // also zeroes the step tickets
        // (bytes, not flop: the lower 64x64 tiles of K written once + X read)
        tl.done(KIND_BUILD, (double)trace_num_blocks(g->npad) * 64 * 64 * 8 + (double)g->n * g->d * 8);
    }
    if ((rc = phase_mark(g, 1))) return rc;
    g->vec_early = want_grad;
    g->vec_done = false;
    g->zfuse = !want_grad && g->tune[TUNE_ZFUSE] != 0;
    if (g->zfuse) {                                            // the running right-hand side of the forward substitution: w = y
        if (g->grp) launch_copy_y_to_w(g->npad, s, B(g));
        else HIPCHK(hipMemcpyAsync(g->dw, g->dy, (size_t)g->npad * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    rc = enqueue_potrf(g, want_grad, true, tickets);           // + L^-1 and K^-1, block rows at a time beside it
    g->vec_early = false;
    const bool zfused = g->zfuse;
    g->zfuse = false;
    if (rc) return rc;
    if (want_grad) {
        if ((rc = phase_mark(g, 3))) return rc;              // "trtri" phase = what is left of the inverse blocks
        if ((rc = phase_mark(g, 4))) return rc;
        if (!g->vec_done) {
            launch_trmv_lower(g->dT, g->npad, g->npad, g->dy, g->dz, s, B(g));        // z = L^-1 y
            launch_trmv_upper(g->dU, g->npad, g->npad, g->dz, g->dalpha, s, B(g));    // alpha = L^-T z
        }
        // traces and the final sums in ONE launch: the last block of k_trace finishes the evaluation (kernels.hip)
        launch_trace(g->dX, g->n, g->d, g->npad, h, g->dKinv, g->dalpha, g->dpart, s, hd, B(g), g->dz, g->dlogdet, g->dout,
                     host_out(g), tickets + 2 * g->nt);
    } else {
        if ((rc = phase_mark(g, 3))) return rc;
        if ((rc = phase_mark(g, 4))) return rc;
        if (!zfused) {                                         // (TUNE_ZFUSE = 0: the substitution as 2 nt launches behind the factorisation)
            if (g->grp) launch_copy_y_to_w(g->npad, s, B(g));
            else HIPCHK(hipMemcpyAsync(g->dw, g->dy, (size_t)g->npad * sizeof(double), hipMemcpyDeviceToDevice, s));
            launch_trsv_lower(g->dA, g->dT, g->npad, g->nt, g->dw, g->dz, s, B(g));   // L z = y
        }
        launch_finalize(g->dz, g->npad, g->n, g->dlogdet, g->nt, nullptr, 0, h, g->dout, host_out(g), s, hd, B(g));
    }
    if ((rc = phase_mark(g, 5))) return rc;
    HIPCHK(hipGetLastError());
    // (k_finalize wrote the results into the pinned host buffer itself)
    return CUGP_OK;
}

int enqueue_eval(cugp_gp* g, bool want_grad)
{
    int rc;
    if (!g->have_data) return fail(CUGP_ERR_INVALID, "no training data set (cugp_set_data)");
    if ((rc = fetch_eval(g))) return rc;                      // one evaluation in flight per handle
    TuneScope ts(g);
    if ((rc = use_device(g))) return rc;
    if ((rc = ensure_factor_bufs(g))) return rc;
    if (want_grad && (rc = ensure_inverse_bufs(g))) return rc;
    if (const int pe = prepare_kernels())                     // per-device launch attributes (dynamic LDS sizes)
        return fail(CUGP_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)", (hipError_t)pe);
    g->factor_valid = g->inverse_valid = false;

    // Replay a captured graph when the evaluation is a single-stream sequence (no hand-over to the other
    // streams) and nothing is being timed; otherwise enqueue the launches one by one.
    // (measured: 16 x 1500 rows 4.7 -> 4.1 ms, 2 x 1500 rows 1.25 -> 1.18 ms; nothing to gain above ~3000 rows)
    const bool graph = g->tune[TUNE_GRAPHS] != 0 && g->prof == 0 && g->nt <= GRAPH_MAX_TILES &&
                       pipe_block(g, want_grad) == 0;
    if (!graph) {
        if ((rc = record_eval(g, want_grad, nullptr))) return rc;
    } else {
        const int gi = want_grad ? 1 : 0;
        *g->hhs = scalars(g);
        if (!g->gexec[gi] || g->gepoch[gi] != g->cfg_epoch) {
            if (g->gexec[gi]) (void)hipGraphExecDestroy(g->gexec[gi]);
            g->gexec[gi] = nullptr;
            prepare_kernels();
            hipGraph_t graph_obj = nullptr;
            HIPCHK(hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal));
            rc = record_eval(g, want_grad, g->dhs);
            const hipError_t e = hipStreamEndCapture(g->stream, &graph_obj);   // always leave capture mode
            if (rc || e != hipSuccess) {
                if (graph_obj) (void)hipGraphDestroy(graph_obj);
                return rc ? rc : fail(CUGP_ERR_DEVICE, "hipStreamEndCapture", e);
            }
            const hipError_t ei = hipGraphInstantiate(&g->gexec[gi], graph_obj, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph_obj);
            if (ei != hipSuccess) { g->gexec[gi] = nullptr; return fail(CUGP_ERR_DEVICE, "hipGraphInstantiate", ei); }
            g->gepoch[gi] = g->cfg_epoch;
        }
        HIPCHK(hipGraphLaunch(g->gexec[gi], g->stream));
    }
    g->pending = true;
    g->pending_grad = want_grad;
    g->pev_valid = g->prof >= 1;
    return CUGP_OK;
}

// Gradient quantities from a factor that is already valid (a log-likelihood-only evaluation at the same data and
// hyper-parameters came first: Covsum::compute_loglikelihood then compute_gradient_loghyperparam, or the value and
// gradient halves of the evaluation-sparing line search): L^-1, K^-1, alpha, traces -- no rebuild of K, no second
// factorisation.  One stream (the factorisation it could have run beside is over).
int enqueue_continue(cugp_gp* g)
{
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (!g->factor_valid) return fail(CUGP_ERR_INVALID, "no valid factor to continue from");
    TuneScope ts(g);
    if ((rc = use_device(g))) return rc;
    if ((rc = ensure_inverse_bufs(g))) return rc;
    if (const int pe = prepare_kernels())
        return fail(CUGP_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)", (hipError_t)pe);
    const HyperScalars h = scalars(g);
    hipStream_t s = g->stream;
    g->inverse_valid = false;
    HIPCHK(hipMemsetAsync(g->dtickets + g->nt, 0, (size_t)(g->nt + 1) * sizeof(unsigned), s));     // k_trtri_block's stage counters, k_trace's arrival counter
    if ((rc = reset_stamps(g))) return rc;
    if ((rc = enqueue_inverse_block(g, 0, g->nt, true, s, nullptr, nullptr))) return rc;
    launch_trmv_lower(g->dT, g->npad, g->npad, g->dy, g->dz, s);
    launch_trmv_upper(g->dU, g->npad, g->npad, g->dz, g->dalpha, s);
    launch_trace(g->dX, g->n, g->d, g->npad, h, g->dKinv, g->dalpha, g->dpart, s, nullptr, {}, g->dz, g->dlogdet, g->dout, g->hout,
                 g->dtickets + 2 * g->nt);
    HIPCHK(hipGetLastError());
    g->pending = true;
    g->pending_grad = true;
    g->pev_valid = false;
    return CUGP_OK;
}

// After a synchronise that follows ANY launch sequence with k_trtri_block in it: did one of its bounded stage waits
// run out?  (pinned status word, entry 6 of the handle's result row.)  Reads and clears it.
bool barrier_expired(cugp_gp* g)
{
    if (g->hout[6] == 0.0) return false;
    g->hout[6] = 0.0;
    return true;
}

int fetch_eval(cugp_gp* g)
{
    if (!g->pending) return CUGP_OK;
    int rc;
    if ((rc = use_device(g))) return rc;
    HIPCHK(hipStreamSynchronize(g->stream));
    g->pending = false;
    if (barrier_expired(g)) {
        // a bounded wait inside a kernel ran out (k_trtri_block's stage barrier): nothing of this evaluation is to be
        // trusted, and nothing of it is kept -- the next call starts from the covariance build
        g->factor_valid = g->inverse_valid = false;
        g->last_ll = g->last_quad = g->last_logdet = NAN;
        g->last_g[0] = g->last_g[1] = g->last_g[2] = NAN;
        if (g->prof >= 2) drain_kernel_events(g);
        return fail(CUGP_ERR_DEVICE, "a stage barrier of k_trtri_block ran out of polls (the evaluation was abandoned; results NaN)");
    }
    g->last_ll = g->hout[0];
    g->last_quad = g->hout[4];
    g->last_logdet = g->hout[5];
    g->factor_valid = true;
    if (g->pending_grad) {
        for (int i = 0; i < 3; i++) g->last_g[i] = g->hout[1 + i];
        g->inverse_valid = true;
    }
    if (g->prof >= 2) drain_kernel_events(g);
    return CUGP_OK;
}

}  // namespace

extern "C" {

int cugp_version(void) { return 100; }

#ifndef CUGP_BUILD_ID
#define CUGP_BUILD_ID "unknown"
#endif
const char* cugp_build_id(void) { return CUGP_BUILD_ID; }

const char* cugp_last_error(void) { return g_err.c_str(); }

int cugp_device_count(int* count)
{
    if (!count) return CUGP_ERR_INVALID;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail(CUGP_ERR_NODEVICE, "hipGetDeviceCount", e); }
    *count = c;
    return CUGP_OK;
}

int cugp_create(int n, int d, int device, cugp_gp** out) { return cugp_create_padded(n, d, device, 0, out); }

int cugp_create_padded(int n, int d, int device, int npad_min, cugp_gp** out)
{
    if (!out || n <= 0 || d <= 0) return fail(CUGP_ERR_INVALID, "cugp_create: n, d must be positive");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        return fail(CUGP_ERR_NODEVICE, "no HIP device visible (libcugp has no CPU fallback)");
    if (device < 0 || device >= cnt) return fail(CUGP_ERR_INVALID, "cugp_create: device index out of range");
    cugp_gp* g = new (std::nothrow) cugp_gp;
    if (!g) return fail(CUGP_ERR_NOMEM, "host allocation");
    g->n = n; g->d = d; g->device = device;
    g->nt = ((n > npad_min ? n : npad_min) + TILE - 1) / TILE;
    g->npad = g->nt * TILE;
    g->nblocks_trace = trace_num_blocks(g->npad);
    *out = nullptr;
    {   // the process defaults as they stand now
        std::lock_guard<std::mutex> lk(g_tune_mu);
        tune_defaults_locked();
        for (int k = 0; k < TUNE_COUNT; k++) g->tune[k] = g_tune_default[k];
    }
    g_live[device < 64 ? device : 63].fetch_add(1, std::memory_order_relaxed);
    g->counted = true;
    hipError_t e = hipSetDevice(device);
    // default priority everywhere: prioritised streams share few hardware queues, which serialises the
    // experts of a BCM evaluated on one device
    // (TUNE_STREAM_PRIO, an A/B hook: the factorisation's stream at the highest and / or the inverse streams at the
    //  lowest priority)
    int plo = 0, phi = 0;
    if (e == hipSuccess && g->tune[TUNE_STREAM_PRIO] != 0) e = hipDeviceGetStreamPriorityRange(&plo, &phi);
    const int pmain = (g->tune[TUNE_STREAM_PRIO] & 1) ? phi : 0, pinv = (g->tune[TUNE_STREAM_PRIO] & 2) ? plo : 0;
    auto mkstream = [&](hipStream_t* s, int prio) {
        return g->tune[TUNE_STREAM_PRIO] != 0 ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio)
                                             : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    };
    if (e == hipSuccess) e = mkstream(&g->stream, pmain);
    if (e == hipSuccess) e = mkstream(&g->aux, pinv);
    if (e == hipSuccess) e = mkstream(&g->aux2, pinv);
    if (e == hipSuccess) e = mkstream(&g->lq, pinv);
    for (std::vector<hipEvent_t>* v : {&g->bev, &g->oev, &g->lev}) {
        v->assign((size_t)g->nt + 2, nullptr);
        for (size_t i = 0; i < v->size() && e == hipSuccess; i++) e = hipEventCreateWithFlags(&(*v)[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&g->dX, (size_t)n * d * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->dy, (size_t)g->npad * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->dz, (size_t)g->npad * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->dalpha, (size_t)g->npad * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->dw, (size_t)g->npad * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d16, (size_t)g->nt * 8 * 256 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d64, (size_t)g->nt * 8192 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->dlogdet, (size_t)g->nt * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->dpart, (size_t)g->nblocks_trace * 3 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->dout, 8 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->dtickets, (size_t)ticket_count(g->nt) * sizeof(unsigned));
    if (e == hipSuccess) e = hipHostMalloc((void**)&g->hout, 8 * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) memset(g->hout, 0, 8 * sizeof(double));   // (entry 6 is the status word fetch_eval reads)
    if (e == hipSuccess) e = hipHostMalloc((void**)&g->hhs, sizeof(HyperScalars), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&g->dhs, sizeof(HyperScalars));
    for (int i = 0; i <= NPHASE && e == hipSuccess; i++) e = hipEventCreate(&g->pev[i]);
    if (e != hipSuccess) {
        int code = fail(e == hipErrorOutOfMemory ? CUGP_ERR_NOMEM : CUGP_ERR_DEVICE, "cugp_create", e);
        cugp_destroy(g);
        return code;
    }
    *out = g;
    return CUGP_OK;
}

int cugp_destroy(cugp_gp* g)
{
    if (!g) return CUGP_OK;
    barrier_release(g);
    if (g->counted) g_live[g->device >= 0 && g->device < 64 ? g->device : 63].fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->aux) (void)hipStreamSynchronize(g->aux);
    if (g->aux2) (void)hipStreamSynchronize(g->aux2);
    if (g->lq) (void)hipStreamSynchronize(g->lq);
    double* bufs[] = {g->dX, g->dy, g->dA, g->dT, g->dU, g->dKinv, g->dz, g->dalpha, g->dw, g->d16, g->dlogdet,
                      g->dpart, g->dout, g->d64};
    for (double* p : bufs)
        if (p) (void)hipFree(p);
    if (g->dtickets) (void)hipFree(g->dtickets);
    if (g->dstamps) (void)hipFree(g->dstamps);
    if (g->pred_buf) (void)hipFree(g->pred_buf);
    if (g->hout) (void)hipHostFree(g->hout);
    if (g->hhs) (void)hipHostFree(g->hhs);
    if (g->dhs) (void)hipFree(g->dhs);
    for (hipGraphExec_t x : g->gexec)
        if (x) (void)hipGraphExecDestroy(x);
    for (int i = 0; i <= NPHASE; i++)
        if (g->pev[i]) (void)hipEventDestroy(g->pev[i]);
    for (hipEvent_t e : g->kev) (void)hipEventDestroy(e);
    for (std::vector<hipEvent_t>* v : {&g->bev, &g->oev, &g->lev})
        for (hipEvent_t e : *v)
            if (e) (void)hipEventDestroy(e);
    if (g->aux) (void)hipStreamDestroy(g->aux);
    if (g->aux2) (void)hipStreamDestroy(g->aux2);
    if (g->lq) (void)hipStreamDestroy(g->lq);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    return CUGP_OK;
}

int cugp_set_overlap(cugp_gp* g, int enable)
{
    if (!g) return CUGP_ERR_INVALID;
    g->overlap = enable != 0;
    return CUGP_OK;                                          // (a captured graph is only used without hand-over)
}

int cugp_dims(const cugp_gp* g, int* n, int* d, int* npad)
{
    if (!g) return CUGP_ERR_INVALID;
    if (n) *n = g->n;
    if (d) *d = g->d;
    if (npad) *npad = g->npad;
    return CUGP_OK;
}

static int set_data_common(cugp_gp* g, const double* X, const double* y, hipMemcpyKind kind)
{
    if (!g || !X || !y) return fail(CUGP_ERR_INVALID, "cugp_set_data: null argument");
    int rc;
    if ((rc = use_device(g))) return rc;
    if ((rc = fetch_eval(g))) return rc;
    HIPCHK(hipMemcpyAsync(g->dX, X, (size_t)g->n * g->d * sizeof(double), kind, g->stream));
    HIPCHK(hipMemsetAsync(g->dy, 0, (size_t)g->npad * sizeof(double), g->stream));
    HIPCHK(hipMemcpyAsync(g->dy, y, (size_t)g->n * sizeof(double), kind, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    g->have_data = true;
    g->factor_valid = g->inverse_valid = false;
    return CUGP_OK;
}

int cugp_set_data(cugp_gp* g, const double* X, const double* y) { return set_data_common(g, X, y, hipMemcpyHostToDevice); }

int cugp_set_data_device(cugp_gp* g, const double* dX, const double* dy)
{
    return set_data_common(g, dX, dy, hipMemcpyDeviceToDevice);
}

int cugp_set_loghyper(cugp_gp* g, const double hp[3])
{
    if (!g || !hp) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (hp[0] != g->hp[0] || hp[1] != g->hp[1] || hp[2] != g->hp[2] || std::isnan(hp[0] + hp[1] + hp[2]))
        g->factor_valid = g->inverse_valid = false;
    for (int i = 0; i < 3; i++) g->hp[i] = hp[i];
    return CUGP_OK;
}

int cugp_get_loghyper(const cugp_gp* g, double hp[3])
{
    if (!g || !hp) return CUGP_ERR_INVALID;
    for (int i = 0; i < 3; i++) hp[i] = g->hp[i];
    return CUGP_OK;
}

int cugp_loglik_grad_enqueue(cugp_gp* g, int want_grad)
{
    if (!g) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    return enqueue_eval(g, want_grad != 0);
}

int cugp_loglik_grad_fetch(cugp_gp* g, double* ll, double gr[3])
{
    if (!g) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (ll) *ll = g->last_ll;
    if (gr)
        for (int i = 0; i < 3; i++) gr[i] = g->last_g[i];
    return CUGP_OK;
}

int cugp_loglik(cugp_gp* g, double* ll)
{
    if (!g || !ll) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (!g->factor_valid) {
        if ((rc = enqueue_eval(g, false))) return rc;
        if ((rc = fetch_eval(g))) return rc;
    }
    *ll = g->last_ll;
    return CUGP_OK;
}

int cugp_loglik_grad(cugp_gp* g, double* ll, double gr[3])
{
    if (!g) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (!g->inverse_valid) {
        // a valid factor (cugp_loglik came first at the same point) is continued, not recomputed
        if ((rc = (g->factor_valid && g->prof == 0) ? enqueue_continue(g) : enqueue_eval(g, true))) return rc;
        if ((rc = fetch_eval(g))) return rc;
    }
    if (ll) *ll = g->last_ll;
    if (gr)
        for (int i = 0; i < 3; i++) gr[i] = g->last_g[i];
    return CUGP_OK;
}

int cugp_grad(cugp_gp* g, double gr[3]) { return cugp_loglik_grad(g, nullptr, gr); }

int cugp_last_quad_logdet(const cugp_gp* g, double* quad, double* logdet)
{
    if (!g || !g->factor_valid) return fail(CUGP_ERR_INVALID, "no evaluation available");
    if (quad) *quad = g->last_quad;
    if (logdet) *logdet = g->last_logdet;
    return CUGP_OK;
}

// ---------------------------------------------------------------- prediction
int cugp_nlpp(const double* actual, const double* mean, const double* var, int nt, double* nlpp)
{
    if (!actual || !mean || !var || !nlpp || nt <= 0) return CUGP_ERR_INVALID;
    double acc = 0.0;   // covkernel.cpp:649-659, 2*pi truncated to 6.283185
    for (int i = 0; i < nt; i++)
        acc += 0.5 * std::log(6.283185 * var[i]) + std::pow((mean[i] - actual[i]), 2) / (2 * var[i]);
    *nlpp = acc / nt;
    return CUGP_OK;
}

static int predict_device(cugp_gp* g, const double* Xt, int nt, double** dmean_out, double** dvar_out,
                          std::vector<double*>& to_free)
{
    int rc;
    if ((rc = cugp_loglik_grad(g, nullptr, nullptr))) return rc;     // factor, T, alpha for the current hp
    if ((rc = use_device(g))) return rc;                             // (a BCM over several devices predicts expert by expert)
    TuneScope ts(g);
    const int ntpad = ((nt + TILE - 1) / TILE) * TILE;
    const HyperScalars h = scalars(g);
    // prediction scratch lives with the handle and only grows (the allocations cost more than the kernels
    // for a few hundred test points)
    (void)to_free;
    const size_t nxt = (((size_t)nt * g->d + 15) / 16) * 16, nks = (size_t)ntpad * g->npad, nv = (size_t)ntpad;
    const size_t need = nxt + 2 * nks + 2 * nv;
    if (need > g->pred_cap) {
        HIPCHK(hipStreamSynchronize(g->stream));
        if (g->pred_buf) (void)hipFree(g->pred_buf);
        g->pred_buf = nullptr;
        g->pred_cap = 0;
        HIPCHK(hipMalloc((void**)&g->pred_buf, need * sizeof(double)));
        g->pred_cap = need;
    }
    double* dXt = g->pred_buf;
    double* dKs = dXt + nxt;
    double* dW = dKs + nks;
    double* dm = dW + nks;
    double* dv = dm + nv;
    HIPCHK(hipMemcpyAsync(dXt, Xt, (size_t)nt * g->d * sizeof(double), hipMemcpyHostToDevice, g->stream));
    if ((rc = reset_stamps(g))) return rc;
    launch_kcross(g->dX, g->n, g->d, g->npad, dXt, nt, ntpad, h, dKs, g->stream);
    {
        // W = Ks L^-T: test tile tt, row tile ti sums k <= ti (the diagonal k tile of T is triangular: counted half)
        TimedLaunch tl(g, g->stream, g->prof >= 3);
        launch_predict_gemm(dKs, g->dT, dW, g->npad, ntpad / TILE, g->nt, g->stream);
        double kt = 0;
        for (int ti = 0; ti < g->nt; ti++) kt += ti + 0.5;
        tl.done(KIND_PREDICT, kt * (ntpad / TILE) * 2.0 * TILE * TILE * TILE);
    }
    launch_predict_finish(dKs, dW, g->dalpha, g->n, g->npad, nt, h, dm, dv, g->stream);
    HIPCHK(hipGetLastError());
    *dmean_out = dm;
    *dvar_out = dv;
    return CUGP_OK;
}

int cugp_predict(cugp_gp* g, const double* Xt, int nt, double* mean, double* var)
{
    if (!g || !Xt || !mean || !var || nt <= 0) return fail(CUGP_ERR_INVALID, "cugp_predict: bad argument");
    std::vector<double*> tmp;
    double *dm = nullptr, *dv = nullptr;
    int rc = predict_device(g, Xt, nt, &dm, &dv, tmp);
    if (rc == CUGP_OK) {
        hipError_t e = hipMemcpyAsync(mean, dm, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(var, dv, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) rc = fail(CUGP_ERR_DEVICE, "cugp_predict copy", e);
        else if (g->prof >= 2) drain_kernel_events(g);
    }
    (void)hipStreamSynchronize(g->stream);
    for (double* p : tmp) (void)hipFree(p);
    return rc;
}

int cugp_has_inverse(const cugp_gp* g) { return g && !g->pending && g->inverse_valid ? 1 : 0; }

int cugp_predict_enqueue(cugp_gp* g, const double* Xt, int nt, double* host_mv)
{
    if (!g || !Xt || !host_mv || nt <= 0) return fail(CUGP_ERR_INVALID, "cugp_predict_enqueue: bad argument");
    std::vector<double*> tmp;
    double *dm = nullptr, *dv = nullptr;
    int rc = predict_device(g, Xt, nt, &dm, &dv, tmp);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(host_mv, dm, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipMemcpyAsync(host_mv + nt, dv, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    return CUGP_OK;
}

int cugp_predict_fetch(cugp_gp* g)
{
    if (!g) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = use_device(g))) return rc;
    HIPCHK(hipStreamSynchronize(g->stream));
    if (g->prof >= 2) drain_kernel_events(g);
    return CUGP_OK;
}

// ---------------------------------------------------------------- intermediates
int cugp_compute_K_train(cugp_gp* g, double* K)
{
    if (!g || !K) return CUGP_ERR_INVALID;
    if (!g->have_data) return fail(CUGP_ERR_INVALID, "no training data set");
    int rc;
    if ((rc = use_device(g))) return rc;
    if ((rc = fetch_eval(g))) return rc;
    if ((rc = ensure(&g->dA, (size_t)g->npad * g->npad))) return rc;
    g->factor_valid = g->inverse_valid = false;
    TuneScope ts(g);
    launch_kbuild(g->dX, g->n, g->d, g->npad, scalars(g), g->dA, true, g->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy2DAsync(K, (size_t)g->n * sizeof(double), g->dA, (size_t)g->npad * sizeof(double),
                            (size_t)g->n * sizeof(double), g->n, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    return CUGP_OK;
}

int cugp_compute_squared_dist(cugp_gp* g, double c, double* S)
{
    if (!g || !S) return CUGP_ERR_INVALID;
    if (!g->have_data) return fail(CUGP_ERR_INVALID, "no training data set");
    int rc;
    if ((rc = use_device(g))) return rc;
    if ((rc = fetch_eval(g))) return rc;
    if ((rc = ensure(&g->dA, (size_t)g->npad * g->npad))) return rc;
    g->factor_valid = g->inverse_valid = false;
    launch_sqdist(g->dX, g->n, g->d, g->npad, c, g->dA, g->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy2DAsync(S, (size_t)g->n * sizeof(double), g->dA, (size_t)g->npad * sizeof(double),
                            (size_t)g->n * sizeof(double), g->n, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    return CUGP_OK;
}

int cugp_compute_k_test(cugp_gp* g, const double* Xt, int nt, double* Ks)
{
    if (!g || !Xt || !Ks || nt <= 0) return CUGP_ERR_INVALID;
    if (!g->have_data) return fail(CUGP_ERR_INVALID, "no training data set");
    int rc;
    if ((rc = use_device(g))) return rc;
    const int ntpad = ((nt + TILE - 1) / TILE) * TILE;
    double *dXt = nullptr, *dKs = nullptr;
    HIPCHK(hipMalloc((void**)&dXt, (size_t)nt * g->d * sizeof(double)));
    hipError_t e = hipMalloc((void**)&dKs, (size_t)ntpad * g->npad * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(dXt, Xt, (size_t)nt * g->d * sizeof(double), hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) {
        TuneScope ts(g);
        launch_kcross(g->dX, g->n, g->d, g->npad, dXt, nt, ntpad, scalars(g), dKs, g->stream);
        e = hipMemcpy2DAsync(Ks, (size_t)g->n * sizeof(double), dKs, (size_t)g->npad * sizeof(double),
                             (size_t)g->n * sizeof(double), nt, hipMemcpyDeviceToHost, g->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    (void)hipFree(dXt);
    if (dKs) (void)hipFree(dKs);
    if (e != hipSuccess) return fail(CUGP_ERR_DEVICE, "cugp_compute_k_test", e);
    return CUGP_OK;
}

static int copy_square(cugp_gp* g, const double* dsrc, double* dst)
{
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)g->n * sizeof(double), dsrc, (size_t)g->npad * sizeof(double),
                            (size_t)g->n * sizeof(double), g->n, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    return CUGP_OK;
}

int cugp_get_cholesky(cugp_gp* g, double* L)
{
    if (!g || !L) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (!g->factor_valid) return fail(CUGP_ERR_INVALID, "cugp_get_cholesky: evaluate the likelihood first");
    if ((rc = use_device(g))) return rc;
    if ((rc = copy_square(g, g->dA, L))) return rc;
    for (int i = 0; i < g->n; i++)                       // matrixops.cpp:100-107: strict upper zeroed
        for (int j = i + 1; j < g->n; j++) L[(size_t)i * g->n + j] = 0.0;
    return CUGP_OK;
}

int cugp_get_K_inverse(cugp_gp* g, double* Kinv)
{
    if (!g || !Kinv) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (!g->inverse_valid) return fail(CUGP_ERR_INVALID, "cugp_get_K_inverse: evaluate the gradient first");
    if ((rc = use_device(g))) return rc;
    if ((rc = copy_square(g, g->dKinv, Kinv))) return rc;
    for (int i = 0; i < g->n; i++)
        for (int j = i + 1; j < g->n; j++) Kinv[(size_t)i * g->n + j] = Kinv[(size_t)j * g->n + i];
    return CUGP_OK;
}

int cugp_get_alpha(cugp_gp* g, double* alpha)
{
    if (!g || !alpha) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (!g->inverse_valid) return fail(CUGP_ERR_INVALID, "cugp_get_alpha: evaluate the gradient first");
    if ((rc = use_device(g))) return rc;
    HIPCHK(hipMemcpyAsync(alpha, g->dalpha, (size_t)g->n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    return CUGP_OK;
}

// ---------------------------------------------------------------- stand-alone LA
namespace {

// a handle whose A holds a caller matrix (identity padded) instead of a kernel build
int la_handle(int n, const double* K, const double* y, int device, cugp_gp** out)
{
    int rc = cugp_create(n, 1, device, out);
    if (rc) return rc;
    cugp_gp* g = *out;
    if ((rc = ensure_factor_bufs(g)) || (rc = ensure_inverse_bufs(g))) { cugp_destroy(g); return rc; }
    std::vector<double> pad((size_t)g->npad * g->npad, 0.0);
    for (int i = 0; i < g->npad; i++) {
        if (i < n) memcpy(&pad[(size_t)i * g->npad], K + (size_t)i * n, (size_t)n * sizeof(double));
        else pad[(size_t)i * g->npad + i] = 1.0;
    }
    hipError_t e = hipMemcpy(g->dA, pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(g->dy, 0, (size_t)g->npad * sizeof(double));
    if (e == hipSuccess && y) e = hipMemcpy(g->dy, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { cugp_destroy(g); return fail(CUGP_ERR_DEVICE, "la upload", e); }
    return CUGP_OK;
}

int la_factor_inverse(cugp_gp* g, bool inverse)
{
    int rc;
    TuneScope ts(g);
    if ((rc = enqueue_potrf(g, inverse))) return rc;
    HIPCHK(hipStreamSynchronize(g->stream));
    if (barrier_expired(g)) return fail(CUGP_ERR_DEVICE, "a stage barrier of k_trtri_block ran out of polls");
    g->factor_valid = true;
    g->inverse_valid = inverse;
    return CUGP_OK;
}

}  // namespace

int cugp_potrf(int n, const double* K, double* L, int device)
{
    if (n <= 0 || !K || !L) return CUGP_ERR_INVALID;
    cugp_gp* g = nullptr;
    int rc = la_handle(n, K, nullptr, device, &g);
    if (rc) return rc;
    if (!(rc = la_factor_inverse(g, false))) rc = cugp_get_cholesky(g, L);
    cugp_destroy(g);
    return rc;
}

int cugp_potri(int n, const double* K, double* Kinv, int device)
{
    if (n <= 0 || !K || !Kinv) return CUGP_ERR_INVALID;
    cugp_gp* g = nullptr;
    int rc = la_handle(n, K, nullptr, device, &g);
    if (rc) return rc;
    if (!(rc = la_factor_inverse(g, true))) rc = cugp_get_K_inverse(g, Kinv);
    cugp_destroy(g);
    return rc;
}

static int la_solve(int n, const double* K, const double* y, double* x, double* quad, double* logdet, int device)
{
    cugp_gp* g = nullptr;
    int rc = la_handle(n, K, y, device, &g);
    if (rc) return rc;
    TuneScope ts(g);
    if (!(rc = enqueue_potrf(g, false)) && !(rc = enqueue_trtri(g))) {
        launch_trmv_lower(g->dT, g->npad, g->npad, g->dy, g->dz, g->stream);
        launch_trmv_upper(g->dU, g->npad, g->npad, g->dz, g->dalpha, g->stream);
        launch_finalize(g->dz, g->npad, g->n, g->dlogdet, g->nt, nullptr, 0, scalars(g), g->dout, g->hout, g->stream);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && x)
            e = hipMemcpyAsync(x, g->dalpha, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) rc = fail(CUGP_ERR_DEVICE, "la_solve", e);
        else if (barrier_expired(g)) rc = fail(CUGP_ERR_DEVICE, "a stage barrier of k_trtri_block ran out of polls");
        else {
            if (quad) *quad = g->hout[4];
            if (logdet) *logdet = g->hout[5];
        }
    }
    cugp_destroy(g);
    return rc;
}

int cugp_chol_and_det(int n, const double* K, const double* y, double* quad, double* logdet, int device)
{
    if (n <= 0 || !K || !y || !quad || !logdet) return CUGP_ERR_INVALID;
    return la_solve(n, K, y, nullptr, quad, logdet, device);
}

int cugp_potrs_vec(int n, const double* K, const double* y, double* x, int device)
{
    if (n <= 0 || !K || !y || !x) return CUGP_ERR_INVALID;
    return la_solve(n, K, y, x, nullptr, nullptr, device);
}

// ---------------------------------------------------------------- timing
int cugp_set_profiling(cugp_gp* g, int level)
{
    if (!g) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = use_device(g))) return rc;
    if ((rc = fetch_eval(g))) return rc;
    g->prof = level;
    if (level >= 2 && g->kev.empty()) {
        g->kev.resize(MAX_KEV);
        g->kev_kind.assign(MAX_KEV / 2, 0);
        g->kev_prev.assign(MAX_KEV / 2, -1);
        g->kev_flopv.assign(MAX_KEV / 2, 0.0);
        for (auto& e : g->kev) HIPCHK(hipEventCreate(&e));
    }
    if (level >= 5 && !g->dstamps) {
        HIPCHK(hipMalloc((void**)&g->dstamps, (size_t)2 * STAMP_STRIDE * sizeof(unsigned long long)));
        g->hstamps.assign((size_t)2 * STAMP_STRIDE, 0ull);
    }
    return CUGP_OK;
}

int cugp_get_phase_ms(cugp_gp* g, double ms[6])
{
    if (!g || !ms) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (!g->pev_valid) return fail(CUGP_ERR_INVALID, "profiling was off for the last evaluation");
    if ((rc = use_device(g))) return rc;
    for (int i = 0; i < 5; i++) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, g->pev[i], g->pev[i + 1]));
        ms[i] = t;
    }
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, g->pev[0], g->pev[5]));
    ms[5] = t;
    return CUGP_OK;
}

int cugp_get_kernel_stats_kind(cugp_gp* g, int kind, double* sum_ms, long long* launches, double* flop, int reset)
{
    if (!g || kind < 0 || kind >= KIND_COUNT) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    if (sum_ms) *sum_ms = g->kst_ms[kind];
    if (launches) *launches = g->kst_launches[kind];
    if (flop) *flop = g->kst_flop[kind];
    if (reset) { g->kst_ms[kind] = 0; g->kst_launches[kind] = 0; g->kst_flop[kind] = 0; g->kst_disp_ms[kind] = 0; }
    return CUGP_OK;
}

// level 5 only: the same launches' durations counted from the END of the launch directly in front of each on its
// stream (the chain of the factorisation's stream: panel solve -> [far update] -> step launch -> ...), which is where
// rocprofv3 --kernel-trace puts the begin of an in-order dispatch behind another; launches without such a
// predecessor count from their first workgroup's start.  Read it BEFORE a resetting cugp_get_kernel_stats_kind.
int cugp_get_kernel_stats_dispatch_ms(cugp_gp* g, int kind, double* sum_ms)
{
    if (!g || !sum_ms || kind < 0 || kind >= KIND_COUNT) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    *sum_ms = g->kst_disp_ms[kind];
    return CUGP_OK;
}

int cugp_get_kernel_stats(cugp_gp* g, double* sum_ms, long long* launches, double* flop, int reset)
{
    return cugp_get_kernel_stats_kind(g, 0, sum_ms, launches, flop, reset);
}

void* cugp_get_stream(cugp_gp* g) { return g ? (void*)g->stream : nullptr; }
const double* cugp_result_row_device(cugp_gp* g) { return g ? g->dout : nullptr; }

// ---------------------------------------------------------------- optimiser glue
namespace {
void gp_objective(void* ctx, const double th[3], double* f, double gr[3])
{
    cugp_gp* g = (cugp_gp*)ctx;
    double ll = NAN;
    cugp_set_loghyper(g, th);
    if (cugp_loglik_grad(g, &ll, gr) != CUGP_OK) { ll = NAN; gr[0] = gr[1] = gr[2] = NAN; }
    *f = -1.0 * ll;
}
}  // namespace

int cugp_cg_solve(cugp_gp* g, int budget, double* trace, int trace_cap, int* nevals)
{
    if (!g) return CUGP_ERR_INVALID;
    double th[3] = {g->hp[0], g->hp[1], g->hp[2]};
    int rc = cugp_cg_minimize(gp_objective, g, th, budget, trace, trace_cap, nevals);
    if (rc) return rc;
    return cugp_set_loghyper(g, th);       // covkernel.cpp:646
}

namespace {
void gp_value(void* ctx, const double th[3], double* f)
{
    cugp_gp* g = (cugp_gp*)ctx;
    double ll = NAN;
    cugp_set_loghyper(g, th);
    if (cugp_loglik(g, &ll) != CUGP_OK) ll = NAN;
    *f = -1.0 * ll;
}
void gp_gradient(void* ctx, const double th[3], double gr[3])
{
    cugp_gp* g = (cugp_gp*)ctx;
    cugp_set_loghyper(g, th);                   // unchanged point: the factor of the value call stays valid
    if (cugp_grad(g, gr) != CUGP_OK) gr[0] = gr[1] = gr[2] = NAN;
}
}  // namespace

int cugp_cg_solve_sparing(cugp_gp* g, int budget, double* trace, int trace_cap, int* nevals, int* ngrads)
{
    if (!g) return CUGP_ERR_INVALID;
    double th[3] = {g->hp[0], g->hp[1], g->hp[2]};
    int rc = cugp_cg_minimize_sparing(gp_value, gp_gradient, g, th, budget, trace, trace_cap, nevals, ngrads);
    if (rc) return rc;
    return cugp_set_loghyper(g, th);
}

int cugp_rprop_solve(cugp_gp* g, int iters, double* trace, int trace_cap, int* nevals)
{
    if (!g) return CUGP_ERR_INVALID;
    double th[3] = {g->hp[0], g->hp[1], g->hp[2]};
    int rc = cugp_rprop_minimize(gp_objective, g, th, iters, trace, trace_cap, nevals);
    if (rc) return rc;
    return cugp_set_loghyper(g, th);
}

// ---------------------------------------------------------------- test hooks
// Process default of one tuning key.  Thread-safe (a lock); a handle takes the defaults over when it next starts to
// enqueue -- an evaluation already enqueued keeps the shapes it was enqueued with -- except for the keys set for that
// handle alone with cugp_set_handle_tuning.
int cugp_set_tuning(int key, int value)
{
    if (key < 0 || key >= TUNE_COUNT) return CUGP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_tune_mu);
    tune_defaults_locked();
    g_tune_default[key] = value;
    return CUGP_OK;
}

// One key for ONE handle (no other handle, and no later cugp_set_tuning, changes it); own = 0 hands the key back to the
// process default.  The handle's owner thread calls it, like every other call on the handle.
int cugp_set_handle_tuning(cugp_gp* g, int key, int value, int own)
{
    if (!g || key < 0 || key >= TUNE_COUNT) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = fetch_eval(g))) return rc;
    g->tune_own[key] = own != 0;
    if (own && g->tune[key] != value) {
        g->tune[key] = value;
        if (key != TUNE_GRAPHS) g->cfg_epoch++;
    }
    return CUGP_OK;
}

int cugp_get_handle_tuning(cugp_gp* g, int key, int* value)
{
    if (!g || !value || key < 0 || key >= TUNE_COUNT) return CUGP_ERR_INVALID;
    sync_tuning(g);
    *value = g->tune[key];
    return CUGP_OK;
}

int cugp_potrf_plan(int nt, int P, int near, int kb, int out[5])
{
    int v[6];
    const int rc = cugp_potrf_plan_sub(nt, P, near, 1, kb, v);
    if (rc || !out) return rc ? rc : CUGP_ERR_INVALID;
    for (int i = 0; i < 5; i++) out[i] = v[i];
    return CUGP_OK;
}

int cugp_potrf_plan_sub(int nt, int P, int near, int S, int kb, int out[6])
{
    if (!out || nt <= 1 || kb < 0 || kb + 1 >= nt || P < 1 || S < 1) return CUGP_ERR_INVALID;
    const StepPlan sp = plan_step(nt, P, near, kb, S);
    const int v[6] = {sp.wide_k0, sp.wide_kw, sp.wa0, sp.wa1, sp.wcol, sp.ks};
    for (int i = 0; i < 6; i++) out[i] = v[i];
    return CUGP_OK;
}

int cugp_test_gemm_nt(int m, int n, int k, const double* A, const double* B, double* C, int device)
{
    if (m <= 0 || n <= 0 || k <= 0 || m % TILE || n % TILE || k % 16 || !A || !B || !C)
        return fail(CUGP_ERR_INVALID, "cugp_test_gemm_nt: m,n multiples of 128 and k of 16");
    HIPCHK(hipSetDevice(device));
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    HIPCHK(hipMalloc((void**)&dA, (size_t)m * k * sizeof(double)));
    HIPCHK(hipMalloc((void**)&dB, (size_t)n * k * sizeof(double)));
    HIPCHK(hipMalloc((void**)&dC, (size_t)m * n * sizeof(double)));
    HIPCHK(hipMemcpy(dA, A, (size_t)m * k * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB, B, (size_t)n * k * sizeof(double), hipMemcpyHostToDevice));
    launch_test_gemm_nt(dA, dB, dC, m, n, k, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(C, dC, (size_t)m * n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    return CUGP_OK;
}

// stand-alone dense-LA timings on a synthetic SPD matrix built on the device (no host traffic in the timed
// region): the like-for-like counterparts of the reference's library probes cuda_src/cholesky_cu_solver.cpp
// (cusolverDnDpotrf), tmi_cu_solver.cpp (cublasDtrsm vs identity = triangular inverse) and
// cublas_matrix_multiply.cpp (cublasDgemm).  op: 0 Cholesky, 1 triangular inverse of the factor,
// 2 K^-1 = L^-T L^-1 from the triangular inverse, 3 the three together.  ms = best of `reps`.
int cugp_bench_la(int op, int n, int device, int reps, double* ms)
{
    return cugp_bench_la_check(op, n, device, reps, ms, nullptr);
}

// the same, and (ops 0 and 3) the log-determinant of the matrix from the factor it timed, so a test can tell
// that the timed launches produced the right factor
int cugp_bench_la_check(int op, int n, int device, int reps, double* ms, double* logdet)
{
    if (!ms || n <= 0 || op < 0 || op > 4 || reps <= 0) return CUGP_ERR_INVALID;
    cugp_gp* g = nullptr;
    const int d = 4;
    int rc = cugp_create(n, d, device, &g);
    if (rc) return rc;
    std::vector<double> X((size_t)n * d), y(n, 0.0);
    unsigned long long st = 88172645463325252ull;                 // xorshift: reproducible inputs
    for (double& v : X) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (double)(st >> 11) / 9007199254740992.0 * 6.0 - 3.0; }
    const double hp[3] = {0.0, 0.0, -1.0};
    if ((rc = cugp_set_data(g, X.data(), y.data())) || (rc = cugp_set_loghyper(g, hp)) ||
        (rc = ensure_factor_bufs(g)) || (rc = ensure_inverse_bufs(g))) { cugp_destroy(g); return rc; }
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 1e300;
    TuneScope ts(g);
    for (int r = 0; r < reps + 1 && e == hipSuccess && rc == CUGP_OK; r++) {
        launch_kbuild(g->dX, g->n, g->d, g->npad, scalars(g), g->dA, op == 4, g->stream);
        if (op == 1 || op == 2) rc = enqueue_potrf(g, false);
        if (op == 2 && !rc) rc = enqueue_trtri(g);
        if (rc) break;
        e = hipEventRecord(e0, g->stream);
        if (op == 4) launch_test_gemm_nt(g->dA, g->dA, g->dKinv, g->npad, g->npad, g->npad, g->stream);   // uniform tiles
        if (op == 0) rc = enqueue_potrf(g, false);
        if (op == 1) rc = enqueue_trtri(g);
        if (op == 2) launch_lauum(g->dU, g->dKinv, g->npad, 0, g->nt, g->stream);
        if (op == 3) rc = enqueue_potrf(g, true);                              // as an evaluation runs them
        if (e == hipSuccess) e = hipEventRecord(e1, g->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float t = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
        if (r > 0 && t < best) best = t;                          // first round warms up
        if (e == hipSuccess && barrier_expired(g)) rc = fail(CUGP_ERR_DEVICE, "a stage barrier of k_trtri_block ran out of polls");
    }
    if (logdet && rc == CUGP_OK && e == hipSuccess) {
        *logdet = NAN;
        if (op == 0 || op == 3) {                                     // 2 * sum of the diagonal blocks' shares
            std::vector<double> part(g->nt);
            e = hipMemcpy(part.data(), g->dlogdet, (size_t)g->nt * sizeof(double), hipMemcpyDeviceToHost);
            double acc = 0.0;
            for (double v : part) acc += v;
            *logdet = 2.0 * acc;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    cugp_destroy(g);
    if (rc) return rc;
    if (e != hipSuccess) return fail(CUGP_ERR_DEVICE, "cugp_bench_la", e);
    *ms = best;
    return CUGP_OK;
}

int cugp_mfma_peak_tflops(int device, double* tflops)
{
    if (!tflops) return CUGP_ERR_INVALID;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const int blocks = prop.multiProcessorCount * 2;      // 8 waves per CU = 2 per SIMD
    const int iters = 20000;
    double* sink = nullptr;
    HIPCHK(hipMalloc((void**)&sink, (size_t)blocks * 256 * sizeof(double)));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    launch_mfma_peak(sink, blocks, 2000, nullptr);
    HIPCHK(hipEventRecord(a, nullptr));
    launch_mfma_peak(sink, blocks, iters, nullptr);
    HIPCHK(hipEventRecord(b, nullptr));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    const double flop = (double)blocks * 4 /*waves*/ * iters * 8.0 * 2048.0;
    *tflops = flop / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipFree(sink);
    return CUGP_OK;
}

// ---------------------------------------------------------------- grouped evaluation (internal, group.h)
struct cugp_group {
    std::vector<cugp_gp*> experts;
    GroupCtx ctx;
    ExpertPtrs* dtab = nullptr;
    bool tab_valid = false;
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    unsigned gepoch[2] = {0, 0};
    bool pending = false, pending_grad = false;   // an evaluation is enqueued and not yet fetched
};

int cugp_group_create(cugp_gp* const* experts, int k, cugp_group** out)
{
    if (!experts || k <= 0 || !out) return CUGP_ERR_INVALID;
    for (int i = 0; i < k; i++)
        if (!experts[i] || experts[i]->npad != experts[0]->npad || experts[i]->d != experts[0]->d ||
            experts[i]->device != experts[0]->device)
            return CUGP_ERR_INVALID;
    cugp_group* gr = new (std::nothrow) cugp_group;
    if (!gr) return fail(CUGP_ERR_NOMEM, "host allocation");
    gr->experts.assign(experts, experts + k);
    const int nt = experts[0]->nt;
    hipError_t e = hipSetDevice(experts[0]->device);
    if (e == hipSuccess) e = hipMalloc((void**)&gr->dtab, (size_t)k * sizeof(ExpertPtrs));
    if (e == hipSuccess) e = hipMalloc((void**)&gr->ctx.tickets, (size_t)k * ticket_count(nt) * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void**)&gr->ctx.dout, (size_t)k * 8 * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void**)&gr->ctx.hout, (size_t)k * 8 * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) memset(gr->ctx.hout, 0, (size_t)k * 8 * sizeof(double));   // (entry 6 of a row: status word)
    if (e != hipSuccess) {
        cugp_group_destroy(gr);
        return fail(e == hipErrorOutOfMemory ? CUGP_ERR_NOMEM : CUGP_ERR_DEVICE, "cugp_group_create", e);
    }
    gr->ctx.bt.tab = gr->dtab;
    gr->ctx.bt.count = k;
    *out = gr;
    return CUGP_OK;
}

void cugp_group_destroy(cugp_group* gr)
{
    if (!gr) return;
    if (!gr->experts.empty()) {
        (void)hipSetDevice(gr->experts[0]->device);
        (void)hipStreamSynchronize(gr->experts[0]->stream);
    }
    for (hipGraphExec_t x : gr->gexec)
        if (x) (void)hipGraphExecDestroy(x);
    if (gr->dtab) (void)hipFree(gr->dtab);
    if (gr->ctx.tickets) (void)hipFree(gr->ctx.tickets);
    if (gr->ctx.dout) (void)hipFree(gr->ctx.dout);
    if (gr->ctx.hout) (void)hipHostFree(gr->ctx.hout);
    delete gr;
}

// Enqueue one evaluation of all experts of the group on the lead expert's stream(s); results stay on the device
// (ctx.dout, [k][8]) and travel to the pinned ctx.hout behind it.  cugp_group_fetch waits and reads them.
int cugp_group_enqueue(cugp_group* gr, int want_grad)
{
    if (!gr) return CUGP_ERR_INVALID;
    cugp_gp* lead = gr->experts[0];
    const int k = (int)gr->experts.size(), nt = lead->nt;
    int rc;
    if (gr->pending) return fail(CUGP_ERR_BUSY, "cugp_group_enqueue: an evaluation is already in flight");
    TuneScope ts(lead);                                       // the group runs on the lead expert's tuning
    if (nt > lead->tune[TUNE_GROUP_MAX_TILES]) return CUGP_ERR_INVALID;
    // the batched step kernel puts its workgroups on gridDim.y (65535 at most): larger experts go one by one
    if ((long long)nt * (nt + 1) / 2 + 16 > 65535) return CUGP_ERR_INVALID;
    for (cugp_gp* e : gr->experts) {
        if (!e->have_data || e->prof != 0 || pipe_block(e, want_grad != 0) != 0) return CUGP_ERR_INVALID;   // (experts of a BCM have their own overlap off)
        if (e->hp[0] != lead->hp[0] || e->hp[1] != lead->hp[1] || e->hp[2] != lead->hp[2]) return CUGP_ERR_INVALID;
    }
    if ((rc = use_device(lead))) return rc;
    for (cugp_gp* e : gr->experts) {
        if ((rc = fetch_eval(e))) return rc;
        const bool had = e->dA && e->dT && e->dU && (!want_grad || e->dKinv);
        if ((rc = ensure_factor_bufs(e))) return rc;
        if (want_grad && (rc = ensure_inverse_bufs(e))) return rc;
        if (!had) gr->tab_valid = false;
    }
    if (const int pe = prepare_kernels())
        return fail(CUGP_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)", (hipError_t)pe);
    if (!gr->tab_valid) {
        std::vector<ExpertPtrs> tab(k);
        for (int i = 0; i < k; i++) {
            cugp_gp* e = gr->experts[i];
            tab[i] = ExpertPtrs{e->dA, e->dT, e->dU, e->dKinv, e->d16, e->d64, e->dlogdet, e->dy, e->dz, e->dalpha,
                                e->dw, e->dpart, gr->ctx.dout + (size_t)i * 8, e->dX,
                                gr->ctx.tickets + (size_t)i * ticket_count(nt), e->n};
        }
        HIPCHK(hipStreamSynchronize(lead->stream));           // a captured graph may still be reading the old table
        HIPCHK(hipMemcpy(gr->dtab, tab.data(), tab.size() * sizeof(ExpertPtrs), hipMemcpyHostToDevice));
        gr->tab_valid = true;
    }
    for (cugp_gp* e : gr->experts) e->factor_valid = e->inverse_valid = false;
    *lead->hhs = scalars(lead);
    gr->ctx.overlap = lead->tune[TUNE_GROUP_OVERLAP] != 0;
    {   // one batched k_trtri_block launch holds count x G workgroups: G = the experts' smallest reserved share
        int cap = TRTRI_BLOCK_MAXWG;
        for (cugp_gp* e : gr->experts) { const int sh = barrier_share(e); if (sh < cap) cap = sh; }
        gr->ctx.gcap = gr->ctx.overlap && cap >= 16 ? cap / 2 : cap;
    }
    lead->grp = &gr->ctx;
    const int gi = want_grad ? 1 : 0;
    // (with the hand-over the sequence spans several streams: enqueued launch by launch, not replayed)
    if (lead->tune[TUNE_GRAPHS] != 0 && nt <= GRAPH_MAX_TILES && pipe_block(lead, want_grad != 0) == 0 &&
        panel_width(lead) == 1) {
        if (!gr->gexec[gi] || gr->gepoch[gi] != lead->cfg_epoch) {
            if (gr->gexec[gi]) (void)hipGraphExecDestroy(gr->gexec[gi]);
            gr->gexec[gi] = nullptr;
            hipGraph_t graph_obj = nullptr;
            hipError_t e = hipStreamBeginCapture(lead->stream, hipStreamCaptureModeThreadLocal);
            if (e != hipSuccess) { lead->grp = nullptr; return fail(CUGP_ERR_DEVICE, "hipStreamBeginCapture", e); }
            rc = record_eval(lead, want_grad != 0, lead->dhs);
            e = hipStreamEndCapture(lead->stream, &graph_obj);   // always leave capture mode
            if (rc == CUGP_OK && e == hipSuccess) e = hipGraphInstantiate(&gr->gexec[gi], graph_obj, nullptr, nullptr, 0);
            if (graph_obj) (void)hipGraphDestroy(graph_obj);
            if (rc || e != hipSuccess) {
                gr->gexec[gi] = nullptr;
                lead->grp = nullptr;
                return rc ? rc : fail(CUGP_ERR_DEVICE, "group graph capture", e);
            }
            gr->gepoch[gi] = lead->cfg_epoch;
        }
        rc = CUGP_OK;
        const hipError_t e = hipGraphLaunch(gr->gexec[gi], lead->stream);
        if (e != hipSuccess) rc = fail(CUGP_ERR_DEVICE, "hipGraphLaunch", e);
    } else {
        rc = record_eval(lead, want_grad != 0, lead->dhs);
    }
    lead->grp = nullptr;
    if (rc) return rc;
    gr->pending = true;
    gr->pending_grad = want_grad != 0;
    return CUGP_OK;
}

int cugp_group_fetch(cugp_group* gr, double* ll, double* g)
{
    if (!gr || !ll) return CUGP_ERR_INVALID;
    if (!gr->pending) return fail(CUGP_ERR_INVALID, "cugp_group_fetch: nothing enqueued");
    cugp_gp* lead = gr->experts[0];
    const int k = (int)gr->experts.size();
    int rc;
    if ((rc = use_device(lead))) return rc;
    HIPCHK(hipStreamSynchronize(lead->stream));
    gr->pending = false;
    bool timed_out = false;
    for (int i = 0; i < k; i++)
        if (gr->ctx.hout[(size_t)i * 8 + 6] != 0.0) { gr->ctx.hout[(size_t)i * 8 + 6] = 0.0; timed_out = true; }
    if (timed_out) {                                          // (see fetch_eval)
        for (cugp_gp* e : gr->experts) {
            e->factor_valid = e->inverse_valid = false;
            e->last_ll = e->last_quad = e->last_logdet = NAN;
            e->last_g[0] = e->last_g[1] = e->last_g[2] = NAN;
        }
        return fail(CUGP_ERR_DEVICE, "a stage barrier of k_trtri_block ran out of polls (the group's evaluation was abandoned)");
    }
    for (int i = 0; i < k; i++) {
        cugp_gp* e = gr->experts[i];
        const double* h = gr->ctx.hout + (size_t)i * 8;
        e->last_ll = h[0];
        e->last_quad = h[4];
        e->last_logdet = h[5];
        e->factor_valid = true;
        ll[i] = h[0];
        if (gr->pending_grad) {
            for (int j = 0; j < 3; j++) {
                e->last_g[j] = h[1 + j];
                if (g) g[3 * i + j] = h[1 + j];
            }
            e->inverse_valid = true;
        }
    }
    return CUGP_OK;
}

int cugp_internal_fail(int code, const char* what) { return fail(code, what); }

// rows [count][4] <- the first four of every 8-double result row of src, on `stream` (one 2D copy: a group's
// results packed for a collective)
int cugp_pack_result_rows(double* dst, const double* src, int count, void* stream)
{
    if (count <= 0) return CUGP_OK;
    HIPCHK(hipMemcpy2DAsync(dst, 4 * sizeof(double), src, 8 * sizeof(double), 4 * sizeof(double), (size_t)count,
                            hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CUGP_OK;
}

int cugp_copy_device_row(double* dst, const double* src, void* stream)
{
    HIPCHK(hipMemcpyAsync(dst, src, 4 * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CUGP_OK;
}

int cugp_copy_result_row(cugp_gp* g, double* dst)
{
    if (!g || !dst || !g->pending) return CUGP_ERR_INVALID;
    int rc;
    if ((rc = use_device(g))) return rc;
    return cugp_copy_device_row(dst, g->dout, g->stream);
}

int cugp_group_eval(cugp_group* gr, int want_grad, double* ll, double* g)
{
    if (!gr || !ll) return CUGP_ERR_INVALID;
    const int rc = cugp_group_enqueue(gr, want_grad);
    return rc ? rc : cugp_group_fetch(gr, ll, g);
}

// device side of the results of the evaluation in flight: [k][8] doubles, row i = {LL, g0, g1, g2, ...} of expert i,
// valid once everything enqueued on *stream so far has run
int cugp_group_device_results(cugp_group* gr, const double** dout, void** stream)
{
    if (!gr || !dout || !stream) return CUGP_ERR_INVALID;
    *dout = gr->ctx.dout;
    *stream = (void*)gr->experts[0]->stream;
    return CUGP_OK;
}

}  // extern "C"
