// bcm.cpp -- product-of-experts ("BCM") over the experts resident on the GPUs of one process.
//
// Mirrors class BCM (distributed_gp/BCM.h:2-27, BCM.cpp): K experts, each a full GP on its own rows,
// the objective is the plain sum of the experts' log-likelihoods / gradients (BCM.cpp:153-198) and
// the prediction is a product of experts with no prior-precision correction (BCM.cpp:45-62).
// The experts are padded to a common size and evaluated as a GROUP: one sequence of launches in which
// blockIdx.y selects the expert (group.h) -- 16 experts driven from 16 streams were bounded by the
// command processor's dispatch rate, not by the CUs.  When a group evaluation is not possible (experts
// with different hyper-parameters, profiling on) every expert is enqueued on its own stream before the
// first result is fetched.  Sums are taken in expert order on the host either way, so the result does
// not depend on completion order.
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/cugp.h"
#include <hip/hip_runtime.h>

#include "group.h"

// the experts of one device: evaluated as one group of shared launches when their shapes allow it
struct DeviceSet {
    int device = 0;
    std::vector<int> idx;            // global expert indices, ascending
    cugp_group* group = nullptr;     // null: a single expert, or shapes differ
    bool grouped_now = false;        // the evaluation in flight was enqueued as a group
};

struct cugp_bcm {
    std::vector<DeviceSet> sets;     // one per entry of the device list (the same device may be listed twice)
    std::vector<cugp_gp*> experts;   // global order k = 0..K-1
    std::vector<int> rows;
    int d = 0;
    double hp[3] = {0, 0, 0};
    double* pred_host = nullptr;     // pinned: [expert][mean nt | variance nt] of the prediction in flight
    size_t pred_cap = 0;             // ... in doubles
};

extern "C" {

// Experts k = 0..K-1 over the devices of one process: expert k lives on devices[k mod ndev] -- the reference's
// placement of chunk i on worker i mod W (cuda_scalingdist/cg_solver.cpp:93) -- and one host thread drives all
// of them: every device's experts are enqueued (one group of shared launches per device) before the first
// result is read, sums are taken in expert order on the host, so the numbers are those of the single-device
// BCM bit for bit whatever the device list (distributed_gp/BCM.cpp:153-198).
int cugp_bcm_create_multi(int ndev, const int* devices, int nexperts, const int* rows, int d, cugp_bcm** out)
{
    if (!out || ndev <= 0 || !devices || nexperts <= 0 || !rows || d <= 0) return CUGP_ERR_INVALID;
    cugp_bcm* b = new (std::nothrow) cugp_bcm;
    if (!b) return CUGP_ERR_NOMEM;
    b->d = d;
    // Common padded size (identity padding) so that the experts can share launches -- unless their row counts
    // differ by more than a tile or ~6 %: then padding the small ones would cost more than it gains.
    int nmax = 0, nmin = rows[0];
    for (int k = 0; k < nexperts; k++) {
        if (rows[k] <= 0) { delete b; return CUGP_ERR_INVALID; }
        nmax = rows[k] > nmax ? rows[k] : nmax;
        nmin = rows[k] < nmin ? rows[k] : nmin;
    }
    const int tmax = (nmax + 127) / 128, tmin = (nmin + 127) / 128;
    const int pad_to = (tmax - tmin <= (tmin / 16 > 1 ? tmin / 16 : 1)) ? nmax : 0;
    const int nsets = ndev < nexperts ? ndev : nexperts;
    b->sets.resize(nsets);
    for (int s = 0; s < nsets; s++) b->sets[s].device = devices[s];
    for (int k = 0; k < nexperts; k++) {
        DeviceSet& ds = b->sets[k % nsets];
        cugp_gp* g = nullptr;
        int rc = cugp_create_padded(rows[k], d, ds.device, pad_to, &g);
        if (rc) { cugp_bcm_destroy(b); return rc; }
        // several experts on one device already fill each other's idle time; the extra streams only cost launches
        if (nexperts > nsets) cugp_set_overlap(g, 0);
        b->experts.push_back(g);
        b->rows.push_back(rows[k]);
        ds.idx.push_back(k);
    }
    for (DeviceSet& ds : b->sets) {
        // A set of ONE expert beside larger sets (5 experts over 3 devices: 2 + 2 + 1) is a group of one: it then takes
        // its inverse in the same hand-over blocks as the other sets' groups.  Since round 5 an accumulate-form tile
        // product adds its C tile in the epilogue, so the bits of K^-1 depend on where the block boundaries are; with
        // its overlap off (below: nexperts > nsets) the lone expert would take the whole-matrix form instead.
        if (ds.idx.size() < 2 && !(nexperts > nsets)) continue;
        std::vector<cugp_gp*> mine;
        for (int k : ds.idx) mine.push_back(b->experts[k]);
        if (cugp_group_create(mine.data(), (int)mine.size(), &ds.group) != CUGP_OK) ds.group = nullptr;
    }
    *out = b;
    return CUGP_OK;
}

int cugp_bcm_create(int nexperts, const int* rows, int d, int device, cugp_bcm** out)
{
    return cugp_bcm_create_multi(1, &device, nexperts, rows, d, out);
}

// BCM.cpp:85-110 -- expert k gets rows [k*floor(N/K), ...), the last one also the remainder
int cugp_bcm_create_split_multi(const double* X, const double* y, int N, int D, int K, int ndev, const int* devices,
                                cugp_bcm** out)
{
    if (!X || !y || N <= 0 || D <= 0 || K <= 0 || K > N) return CUGP_ERR_INVALID;
    std::vector<int> rows(K), off(K);
    const int part = N / K;
    int start = 0;
    for (int k = 0; k < K; k++) {
        off[k] = start;
        rows[k] = (k == K - 1) ? (N - start) : part;
        start += part;
    }
    int rc = cugp_bcm_create_multi(ndev, devices, K, rows.data(), D, out);
    if (rc) return rc;
    for (int k = 0; k < K; k++) {
        rc = cugp_bcm_set_expert_data(*out, k, X + (size_t)off[k] * D, y + off[k]);
        if (rc) { cugp_bcm_destroy(*out); *out = nullptr; return rc; }
    }
    return CUGP_OK;
}

int cugp_bcm_create_split(const double* X, const double* y, int N, int D, int K, int device, cugp_bcm** out)
{
    return cugp_bcm_create_split_multi(X, y, N, D, K, 1, &device, out);
}

int cugp_bcm_destroy(cugp_bcm* b)
{
    if (!b) return CUGP_OK;
    for (DeviceSet& ds : b->sets) cugp_group_destroy(ds.group);
    for (cugp_gp* g : b->experts) cugp_destroy(g);
    if (b->pred_host) (void)hipHostFree(b->pred_host);
    delete b;
    return CUGP_OK;
}

int cugp_bcm_num_experts(const cugp_bcm* b, int* k)
{
    if (!b || !k) return CUGP_ERR_INVALID;
    *k = (int)b->experts.size();
    return CUGP_OK;
}

int cugp_bcm_expert(cugp_bcm* b, int k, cugp_gp** gp)
{
    if (!b || !gp || k < 0 || k >= (int)b->experts.size()) return CUGP_ERR_INVALID;
    *gp = b->experts[k];
    return CUGP_OK;
}

int cugp_bcm_set_expert_data(cugp_bcm* b, int k, const double* X, const double* y)
{
    if (!b || k < 0 || k >= (int)b->experts.size()) return CUGP_ERR_INVALID;
    return cugp_set_data(b->experts[k], X, y);
}

int cugp_bcm_set_loghyper(cugp_bcm* b, const double hp[3])
{
    if (!b || !hp) return CUGP_ERR_INVALID;
    for (int i = 0; i < 3; i++) b->hp[i] = hp[i];
    for (cugp_gp* g : b->experts) {
        int rc = cugp_set_loghyper(g, b->hp);
        if (rc) return rc;
    }
    return CUGP_OK;
}

int cugp_bcm_get_loghyper(const cugp_bcm* b, double hp[3])
{
    if (!b || !hp) return CUGP_ERR_INVALID;
    for (int i = 0; i < 3; i++) hp[i] = b->hp[i];
    return CUGP_OK;
}

// after an error half-way through an evaluation: fetch whatever is still in flight (results ignored) so that no group
// or expert keeps its "pending" flag -- the next evaluation then starts from a clean state instead of finding the
// shared-launch path "already in flight" for ever
static void bcm_drain(cugp_bcm* b)
{
    std::vector<double> lk, gk3;
    for (DeviceSet& ds : b->sets) {
        if (ds.grouped_now && ds.group) {
            lk.assign(ds.idx.size(), 0.0);
            gk3.assign(3 * ds.idx.size(), 0.0);
            (void)cugp_group_fetch(ds.group, lk.data(), gk3.data());
        } else {
            double l, g3[3];
            for (int k : ds.idx) (void)cugp_loglik_grad_fetch(b->experts[k], &l, g3);
        }
        ds.grouped_now = false;
    }
}

// all experts of all devices in flight (a group of shared launches per device where possible, else one stream
// per expert) before anything is read back
static int bcm_enqueue_all(cugp_bcm* b)
{
    for (DeviceSet& ds : b->sets) ds.grouped_now = false;
    for (DeviceSet& ds : b->sets) {
        if (ds.group) {
            const int rc = cugp_group_enqueue(ds.group, 1);
            if (rc == CUGP_OK) { ds.grouped_now = true; continue; }
            if (rc != CUGP_ERR_INVALID) { bcm_drain(b); return rc; }   // INVALID: not possible as a group right now
        }
        for (int k : ds.idx) {
            const int rc = cugp_loglik_grad_enqueue(b->experts[k], 1);
            if (rc) { bcm_drain(b); return rc; }
        }
    }
    return CUGP_OK;
}

// rows[k] = {LL_k, g_k[0..2]} for every expert of this handle, k in global order (what a multi-process BCM
// all-reduces across ranks)
int cugp_bcm_loglik_grad_rows(cugp_bcm* b, double* rows)
{
    if (!b || !rows) return CUGP_ERR_INVALID;
    int rc = bcm_enqueue_all(b);
    if (rc) return rc;
    std::vector<double> lk, gk3;
    for (DeviceSet& ds : b->sets) {
        const size_t n = ds.idx.size();
        lk.assign(n, 0.0);
        gk3.assign(3 * n, 0.0);
        if (ds.grouped_now) {
            if ((rc = cugp_group_fetch(ds.group, lk.data(), gk3.data()))) { ds.grouped_now = false; bcm_drain(b); return rc; }
            ds.grouped_now = false;
        } else {
            for (size_t i = 0; i < n; i++)
                if ((rc = cugp_loglik_grad_fetch(b->experts[ds.idx[i]], &lk[i], &gk3[3 * i]))) { bcm_drain(b); return rc; }
        }
        for (size_t i = 0; i < n; i++) {
            const size_t k = (size_t)ds.idx[i];
            rows[4 * k] = lk[i];
            for (int j = 0; j < 3; j++) rows[4 * k + 1 + j] = gk3[3 * i + j];
        }
    }
    return CUGP_OK;
}

// The same payload left ON THE DEVICE for a collective that never touches the host (RCCL all-reduce over
// xGMI in cugp_amd/bcm.py): row slot[k] of dev_rows ([.][4] doubles, device memory of the handle's first device,
// e.g. a zeroed buffer with one row per expert of the WHOLE model) receives {LL_k, g_k} of local expert k.
// Returns when the rows are in place (the evaluation itself is the wait); single-device handles only.
int cugp_bcm_loglik_grad_rows_device(cugp_bcm* b, double* dev_rows, const int* slot)
{
    if (!b || !dev_rows || !slot) return CUGP_ERR_INVALID;
    if (b->sets.size() != 1) return CUGP_ERR_INVALID;
    {   // dev_rows must be device memory of the handle's device (a host pointer or another GPU's buffer would fault
        // inside the copy kernels, or silently land elsewhere)
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dev_rows) != hipSuccess || at.type != hipMemoryTypeDevice ||
            at.device != b->sets[0].device) {
            (void)hipGetLastError();
            return CUGP_ERR_INVALID;
        }
    }
    int rc = bcm_enqueue_all(b);
    if (rc) return rc;
    DeviceSet& ds = b->sets[0];
    const size_t n = ds.idx.size();
    // (every error return drains what was enqueued and clears grouped_now, like cugp_bcm_loglik_grad_rows: the next
    //  evaluation of the handle starts clean instead of failing once with CUGP_ERR_BUSY)
    if (ds.grouped_now) {
        const double* dout = nullptr;
        void* stream = nullptr;
        if ((rc = cugp_group_device_results(ds.group, &dout, &stream))) { bcm_drain(b); return rc; }   // (the group is still in flight: drained as a group)
        for (size_t i = 0; i < n; i++)
            if ((rc = cugp_copy_device_row(dev_rows + 4 * (size_t)slot[ds.idx[i]], dout + 8 * i, stream))) { bcm_drain(b); return rc; }
        std::vector<double> lk(n), gk3(3 * n);
        rc = cugp_group_fetch(ds.group, lk.data(), gk3.data());       // waits for the stream: rows are in place
        ds.grouped_now = false;
        if (rc) bcm_drain(b);
        return rc;
    }
    for (size_t i = 0; i < n; i++) {
        cugp_gp* e = b->experts[ds.idx[i]];
        if ((rc = cugp_copy_result_row(e, dev_rows + 4 * (size_t)slot[ds.idx[i]]))) { bcm_drain(b); return rc; }
    }
    for (size_t i = 0; i < n; i++) {
        double l, g3[3];
        if ((rc = cugp_loglik_grad_fetch(b->experts[ds.idx[i]], &l, g3))) { bcm_drain(b); return rc; }
    }
    return CUGP_OK;
}

// The two halves of cugp_bcm_loglik_grad_allgather (comm.cpp) on the BCM's side.  enqueue: all experts in flight, their
// rows {LL, g} packed into dsend[i][4] (local order) BEHIND the evaluation on ITS stream, which is returned -- whatever
// the caller enqueues there next (the collective, the copy to the host) needs no host wait in between.  finish: waits
// for that stream and closes the evaluation (results into the handles, status word checked).
int cugp_bcm_enqueue_rows_packed(cugp_bcm* b, double* dsend, void** stream)
{
    if (!b || !dsend || !stream) return CUGP_ERR_INVALID;
    if (b->sets.size() != 1) return CUGP_ERR_INVALID;
    int rc = bcm_enqueue_all(b);
    if (rc) return rc;
    DeviceSet& ds = b->sets[0];
    const size_t n = ds.idx.size();
    if (ds.grouped_now) {
        const double* dout = nullptr;
        if ((rc = cugp_group_device_results(ds.group, &dout, stream)) ||
            (rc = cugp_pack_result_rows(dsend, dout, (int)n, *stream))) { bcm_drain(b); return rc; }
        return CUGP_OK;
    }
    // experts on streams of their own: every row behind its expert's evaluation; all but the first are waited for
    // here, so that what follows on the first expert's stream finds every row in place
    for (size_t i = 0; i < n; i++)
        if ((rc = cugp_copy_result_row(b->experts[ds.idx[i]], dsend + 4 * i))) { bcm_drain(b); return rc; }
    for (size_t i = 1; i < n; i++) {
        double l, g3[3];
        if ((rc = cugp_loglik_grad_fetch(b->experts[ds.idx[i]], &l, g3))) { bcm_drain(b); return rc; }
    }
    *stream = cugp_get_stream(b->experts[ds.idx[0]]);
    return CUGP_OK;
}

int cugp_bcm_finish_rows(cugp_bcm* b)
{
    if (!b || b->sets.size() != 1) return CUGP_ERR_INVALID;
    DeviceSet& ds = b->sets[0];
    int rc;
    if (ds.grouped_now) {
        std::vector<double> lk(ds.idx.size()), gk3(3 * ds.idx.size());
        rc = cugp_group_fetch(ds.group, lk.data(), gk3.data());
        ds.grouped_now = false;
        if (rc) bcm_drain(b);
        return rc;
    }
    double l, g3[3];
    if ((rc = cugp_loglik_grad_fetch(b->experts[ds.idx[0]], &l, g3))) bcm_drain(b);
    return rc;
}

int cugp_bcm_loglik_grad(cugp_bcm* b, double* ll, double g[3], double* per_expert_ll)
{
    if (!b) return CUGP_ERR_INVALID;
    const size_t K = b->experts.size();
    std::vector<double> rows(4 * K);
    int rc = cugp_bcm_loglik_grad_rows(b, rows.data());
    if (rc) return rc;
    double sll = 0.0, sg[3] = {0, 0, 0};
    for (size_t k = 0; k < K; k++) {
        const double l = rows[4 * k];
        const double* gk = &rows[4 * k + 1];
        sll = sll + l;                                   // BCM.cpp:190-194
        for (int i = 0; i < 3; i++) sg[i] = (k == 0) ? gk[i] : sg[i] + gk[i];   // BCM.cpp:161-173
        if (per_expert_ll) per_expert_ll[k] = l;
    }
    if (ll) *ll = sll;
    if (g)
        for (int i = 0; i < 3; i++) g[i] = sg[i];
    return CUGP_OK;
}

int cugp_bcm_predict_partial(cugp_bcm* b, const double* Xt, int nt, double* sum_prec, double* sum_prec_mean)
{
    if (!b || !Xt || nt <= 0 || !sum_prec || !sum_prec_mean) return CUGP_ERR_INVALID;
    int rc;
    // Round 5.  (1) Experts whose inverse quantities are not valid for the current hyper-parameters (a prediction right
    // after set_BCM_log_hyperparam, BCM.cpp:64-83 after :123-130) are brought up to date by ONE evaluation of the whole
    // model -- the groups of shared launches -- instead of one by one inside cugp_predict (16 x 1500 rows: 18 ms -> one
    // 2 ms evaluation), and every prediction then sees the experts in the state a BCM evaluation leaves them in,
    // whatever came before.  (2) All experts' predictions are in flight before the first is read: each on its own
    // stream, results into one pinned buffer; the two product-of-experts sums are taken in expert order (BCM.cpp:45-62).
    bool stale = false;
    for (cugp_gp* e : b->experts) stale = stale || !cugp_has_inverse(e);
    if (stale) {
        double ll, g3[3];
        if ((rc = cugp_bcm_loglik_grad(b, &ll, g3, nullptr))) return rc;
    }
    const size_t K = b->experts.size(), need = K * 2 * (size_t)nt;
    if (need > b->pred_cap) {
        if (b->pred_host) (void)hipHostFree(b->pred_host);
        b->pred_host = nullptr;
        b->pred_cap = 0;
        if (hipHostMalloc((void**)&b->pred_host, need * sizeof(double), hipHostMallocDefault) != hipSuccess) return CUGP_ERR_NOMEM;
        b->pred_cap = need;
    }
    size_t enq = 0;
    rc = CUGP_OK;
    for (; enq < K && rc == CUGP_OK; enq++) rc = cugp_predict_enqueue(b->experts[enq], Xt, nt, b->pred_host + enq * 2 * nt);
    if (rc) enq--;                                          // (the failing one enqueued nothing that needs a fetch)
    for (size_t k = 0; k < enq; k++) {
        const int rf = cugp_predict_fetch(b->experts[k]);
        if (rf && rc == CUGP_OK) rc = rf;
    }
    if (rc) return rc;
    for (int i = 0; i < nt; i++) sum_prec[i] = sum_prec_mean[i] = 0.0;
    for (size_t k = 0; k < K; k++) {
        const double* m = b->pred_host + k * 2 * nt;
        const double* v = m + nt;
        for (int i = 0; i < nt; i++) {                   // BCM.cpp:51-55
            const double inv = 1.0 / v[i];
            sum_prec[i] += inv;
            sum_prec_mean[i] += inv * m[i];
        }
    }
    return CUGP_OK;
}

int cugp_poe_finish(const double* sum_prec, const double* sum_prec_mean, int nt, double* mean, double* var)
{
    if (!sum_prec || !sum_prec_mean || !mean || !var || nt <= 0) return CUGP_ERR_INVALID;
    for (int i = 0; i < nt; i++) {                       // BCM.cpp:56-60
        const double tv = 1.0 / sum_prec[i];
        var[i] = tv;
        mean[i] = tv * sum_prec_mean[i];
    }
    return CUGP_OK;
}

int cugp_bcm_predict(cugp_bcm* b, const double* Xt, int nt, double* mean, double* var)
{
    if (!b || nt <= 0) return CUGP_ERR_INVALID;
    std::vector<double> sp(nt), spm(nt);
    int rc = cugp_bcm_predict_partial(b, Xt, nt, sp.data(), spm.data());
    if (rc) return rc;
    return cugp_poe_finish(sp.data(), spm.data(), nt, mean, var);
}

namespace {
void bcm_objective(void* ctx, const double th[3], double* f, double g[3])
{
    cugp_bcm* b = (cugp_bcm*)ctx;
    double ll = NAN;
    cugp_bcm_set_loghyper(b, th);
    if (cugp_bcm_loglik_grad(b, &ll, g, nullptr) != CUGP_OK) { ll = NAN; g[0] = g[1] = g[2] = NAN; }
    *f = -1.0 * ll;
}
}  // namespace

int cugp_bcm_cg_solve(cugp_bcm* b, int budget, double* trace, int trace_cap, int* nevals)
{
    if (!b) return CUGP_ERR_INVALID;
    double th[3] = {b->hp[0], b->hp[1], b->hp[2]};
    int rc = cugp_cg_minimize(bcm_objective, b, th, budget, trace, trace_cap, nevals);
    if (rc) return rc;
    return cugp_bcm_set_loghyper(b, th);
}

}  // extern "C"
