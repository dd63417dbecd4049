// wgtimes.hip -- where the factorisation's chain kernels lose their time beside the inverse streams: QUEUE WAIT (their
// workgroups are dispatched late: no free slot) or SLOWED EXECUTION (dispatched at once, but each workgroup runs long on
// CUs it shares with tile products)?  Diagnostic build of the whole library with -DCUGP_WGTIMES: every workgroup of the
// chain and tile kernels stamps s_memrealtime (100 MHz) at its start and end (kernels.hip, WgTimer).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCUGP_WGTIMES -w -x hip tools/wgtimes.hip cugp_amd/csrc/cugp_capi.cpp \
//         cugp_amd/csrc/bcm.cpp cugp_amd/csrc/minimize.cpp -o tools/bin/wgtimes          (kernels.hip is included here)
//   tools/bin/wgtimes [n=8192] [detail=0] [key=value ...]     (tuning keys of kernels.h, e.g. 16=224)
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"
#include "../cugp_amd/csrc/kernels.hip"
#include "../include/cugp.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

using namespace cugp;

struct Rec { int kind, param; double t0, t1, cyc; };   // us, shader cycles

static double med(std::vector<double> v)
{
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

struct Launch { double first = 1e300, last_start = -1e300, end = -1e300; std::vector<double> dur; };

static void analyse(const std::vector<Rec>& r, int nt, bool detail, const char* title)
{
    std::map<int, Launch> trsm, diag, potf2, tile;
    double tmin = 1e300, tmax = -1e300;
    for (const Rec& x : r) {
        tmin = std::min(tmin, x.t0);
        tmax = std::max(tmax, x.t1);
        std::map<int, Launch>* m = x.kind == WGT_TRSM ? &trsm : x.kind == WGT_DIAGUPD ? &diag : x.kind == WGT_POTF2 ? &potf2
                                   : x.kind == WGT_STEPTILE ? &tile : nullptr;
        if (!m) continue;
        Launch& l = (*m)[x.param];
        l.first = std::min(l.first, x.t0);
        l.last_start = std::max(l.last_start, x.t0);
        l.end = std::max(l.end, x.t1);
        l.dur.push_back(x.t1 - x.t0);
    }
    printf("\n== %s: %zu workgroup records, span %.1f us\n", title, r.size(), tmax - tmin);
    printf("   per step kb: [gap] = previous chain kernel's last workgroup end -> this kernel's first workgroup start\n");
    printf("   trsm: start spread = last wg start - first wg start (queue wait inside the launch); wg = median / max workgroup duration\n");
    double s_gap1[2] = {0, 0}, s_trsm[2] = {0, 0}, s_gap2[2] = {0, 0}, s_diag[2] = {0, 0}, s_potf2[2] = {0, 0};
    double s_spread[2] = {0, 0}, s_wgmed[2] = {0, 0}, s_wgmax[2] = {0, 0}, s_dwg[2] = {0, 0};
    int cnt[2] = {0, 0};
    for (int kb = 0; kb + 1 < nt; kb++) {
        if (!trsm.count(kb) || !diag.count(kb) || !potf2.count(kb)) continue;
        const Launch &t = trsm[kb], &d = diag[kb], &p = potf2[kb];
        const double prev_end = kb > 0 && potf2.count(kb - 1) ? potf2[kb - 1].end : t.first;
        const double gap1 = t.first - prev_end, gap2 = d.first - t.end;
        double wgmax = 0;
        for (double v : t.dur) wgmax = std::max(wgmax, v);
        const int h = kb >= nt / 2;
        s_gap1[h] += gap1; s_trsm[h] += t.end - t.first; s_gap2[h] += gap2; s_diag[h] += (p.end - p.dur[0]) - d.first;
        s_potf2[h] += p.dur[0]; s_spread[h] += t.last_start - t.first; s_wgmed[h] += med(t.dur); s_wgmax[h] += wgmax;
        s_dwg[h] += med(d.dur);
        cnt[h]++;
        if (detail)
            printf("   kb %2d  [%6.1f] trsm %3zu wgs span %6.1f (start spread %6.1f, wg %5.1f / %5.1f)  [%6.1f] diag-update span %6.1f (wg med %5.1f)  potf2 %5.1f   step tiles %4zu: wg med %6.1f, last end +%.1f after potf2\n",
                   kb, gap1, t.dur.size(), t.end - t.first, t.last_start - t.first, med(t.dur), wgmax, gap2,
                   (p.end - p.dur[0]) - d.first, med(d.dur), p.dur[0], tile.count(kb) ? tile[kb].dur.size() : 0,
                   tile.count(kb) ? med(tile[kb].dur) : 0.0, tile.count(kb) ? tile[kb].end - p.end : 0.0);
    }
    for (int h = 0; h < 2; h++) {
        if (!cnt[h]) continue;
        const double tot = s_gap1[h] + s_trsm[h] + s_gap2[h] + s_diag[h] + s_potf2[h];
        printf("   %s half (%d steps): chain %.0f us = boundary into trsm %.0f + trsm %.0f (of it start spread %.0f; wg median avg %.1f, max avg %.1f) "
               "+ boundary into step %.0f + diag update until the last ticket %.0f (wg median avg %.1f) + potf2 %.0f   -> %.1f us per step\n",
               h ? "second" : "first", cnt[h], tot, s_gap1[h], s_trsm[h], s_spread[h], s_wgmed[h] / cnt[h], s_wgmax[h] / cnt[h],
               s_gap2[h], s_diag[h], s_dwg[h] / cnt[h], s_potf2[h], tot / cnt[h]);
    }
}

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int n = argc > 1 ? atoi(argv[1]) : 8192, d = 10;
    const bool detail = argc > 2 && atoi(argv[2]) != 0;
    for (int i = 3; i < argc; i++) {
        int k = 0, v = 0;
        if (sscanf(argv[i], "%d=%d", &k, &v) == 2) { cugp_set_tuning(k, v); printf("tuning %d = %d\n", k, v); }
    }
    std::mt19937_64 rng(15618);
    std::uniform_real_distribution<double> U(-10.0, 10.0);
    std::normal_distribution<double> G(0.0, 0.1);
    std::vector<double> X((size_t)n * d), y(n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < d; j++) X[(size_t)i * d + j] = U(rng);
        y[i] = sin(X[(size_t)i * d]) + G(rng);
    }
    cugp_gp* g = nullptr;
    if (cugp_create(n, d, 0, &g)) { printf("create: %s\n", cugp_last_error()); return 1; }
    cugp_set_data(g, X.data(), y.data());
    std::vector<unsigned long long> raw(4u * WGT_CAP);
    for (int overlap = 1; overlap >= 0; overlap--) {
        cugp_set_overlap(g, overlap);
        double ll, gr[3];
        for (int it = 0; it < 4; it++) {
            const double hp[3] = {log(3.0) + 1e-3 * it, 0.0, log(0.1)};
            cugp_set_loghyper(g, hp);
            if (it == 3) { hipDeviceSynchronize(); wgt_reset(); }
            if (cugp_loglik_grad(g, &ll, gr)) { printf("eval: %s\n", cugp_last_error()); return 1; }
        }
        hipDeviceSynchronize();
        const unsigned cnt = wgt_fetch(raw.data(), WGT_CAP);
        std::vector<Rec> r(cnt);
        for (unsigned i = 0; i < cnt; i++)
            r[i] = Rec{(int)(raw[4 * i] & 255), (int)(raw[4 * i] >> 8), raw[4 * i + 1] * 0.01, raw[4 * i + 2] * 0.01, (double)raw[4 * i + 3]};
        printf("ll %.10g\n", ll);
        analyse(r, (n + 127) / 128, detail, overlap ? "overlap ON (inverse blocks beside the factorisation)" : "overlap OFF");
        // tile kernels: how long ONE workgroup takes in each regime
        const char* names[] = {"trsm", "diag-update", "potf2", "step tile", "border", "lauum", "level", "trtri_diag", "wide"};
        for (int k = 3; k <= 8; k++) {
            std::vector<double> v, ghz;
            for (const Rec& x : r)
                if (x.kind == k) {
                    v.push_back(x.t1 - x.t0);
                    if (x.t1 - x.t0 >= 20.0) ghz.push_back(x.cyc / (x.t1 - x.t0) * 1e-3);      // s_memtime cycles / s_memrealtime us
                }
            if (!v.empty()) {
                std::sort(v.begin(), v.end());
                printf("   %-11s %6zu workgroups: duration p10 %.1f  median %.1f  p90 %.1f  max %.1f us", names[k], v.size(), v[v.size() / 10],
                       v[v.size() / 2], v[v.size() * 9 / 10], v.back());
                if (!ghz.empty()) {
                    std::sort(ghz.begin(), ghz.end());
                    printf("   shader clock while they ran: p10 %.2f  median %.2f  p90 %.2f GHz", ghz[ghz.size() / 10], ghz[ghz.size() / 2], ghz[ghz.size() * 9 / 10]);
                }
                printf("\n");
            }
        }
    }
    cugp_destroy(g);
    return 0;
}
