// antq_search.hip -- calibration entry points of libantq: antq_search_sse, antq_search_sse_multi, antq_search_pick, antq_calibrate
// (reference: search_mse AQ/quant_modules.py:287-326, search_adaptive_numeric_type :328-415; OQ:189-256).  gfx950 only.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "antq_host.h"
#include "antq_k_fakequant.h"
#include "antq_k_search.h"
#include "antq_k_hist.h"
#include "antq_k_sweep.h"
#include "antq_k_sortsearch.h"

#include <type_traits>

namespace antq {

constexpr unsigned kSearchMinVpr = 64;      // rows of at least this many vectors: x-domain search kernels, single-read type selection

// ---- the histogram path (antq_k_hist.h): 16-bit tensors with ONE scale and no pair rule --------------------------------
// Worth it once the direct kernels' n x (types x candidates) evaluations outweigh the fixed cost of the three launches (the
// 65 536 x types x candidates literal evaluations of the scoring kernel and the slabs: ~30 us).  Measured
// (profiles/r05_hist_search.log): 1 M elements three types 0.106 -> 0.035 ms, one type 0.046 -> 0.033 ms; 256 K elements
// 0.043 -> 0.034 / 0.031 -> 0.033 ms.  The rule looks at
// the element count ONLY: a tensor's sums must not depend on how many types are searched with it (the single-read type
// selection and one search per type form the very same sums: test_calibration_sums_are_bit_reproducible).  knob 14 = 0
// switches the path off, = 2 takes it for every eligible tensor (tests).
template <typename T>
static bool hist_eligible(size_t n, bool ovp, const void *x, int nflat)
{
    if constexpr (std::is_same<T, float>::value) return false;
    if (g_knob_hist == 0 || (ovp && g_knob_hist == 3) || n % 8 != 0 || n >= ((size_t)1 << 31) || reinterpret_cast<uintptr_t>(x) % 16 != 0) return false;
    if (g_knob_hist == 2) return true;
    // With the pair rule every candidate also walks the list of outlier-capable pairs (~1 % of the pairs of a 3-sigma-clipped
    // tensor) and the direct kernels are enqueued behind as gated no-ops: it pays from ~1.5e8 element x candidate evaluations
    // (25 M elements x 176: 1.63 -> 0.12 ms; 8 M x 176: 0.57 -> 0.10; 1 M x 176: 0.098 -> 0.057; profiles/r05_hist_search.log).
    // Those sums are equal to rounding across call forms anyway (the list depends on the codebooks searched together), so the
    // rule may look at the candidate count; without the pair rule it must not (see above).
    if (ovp) return n >= ((size_t)1 << 20) && (double)n * (double)nflat >= 1.5e8;
    return n >= ((size_t)1 << 19);
}
// OliVe's pair rule: a lower bound (in units of gmax) of the smallest |d| that quantises to an outlier, from the plan's
// threshold list (the thresholds between a normal value and an outlier, either sign).  false: the plan has no such list.
static bool hist_outlier_bound(const void *plan_host, float gmax, float &bound)
{
    const PlanHeader *h = static_cast<const PlanHeader *>(plan_host);
    if (h->kind != kPlanLut || !h->hdom || !(gmax > 0.0f)) return false;
    const HThr *tl = plan_tlist(plan_host);
    float tmin = INFINITY;
    for (uint32_t i = 0; i < h->h_nthr; i++) {
        const bool lo_out = (tl[i].flags & 1u) != 0u, hi_out = (tl[i].flags & 2u) != 0u;
        if (lo_out != hi_out) tmin = std::min(tmin, fabsf(tl[i].T));
    }
    if (!(tmin < INFINITY)) tmin = 3.0e38f;              // (a codebook without outliers: nothing ever qualifies)
    bound = tmin / gmax * 0.999f;
    return true;
}
// The histogram search.  pairs: with OliVe's pair rule; *run_if then receives the device flag the caller's direct launches
// take as their run condition (set iff the pair list overflowed; the histogram kernels then wrote nothing).
// xmax_out (antq_calibrate with the abs-max statistic on a tensor the histogram search takes): the statistic's accumulator
// (zeroed, stream-ordered) that the counting pass fills on the way -- the separate abs-max pass over the tensor is not run.
// Passed down EXPLICITLY from antq_calibrate, which makes the eligibility decision once (`HistXmax`): a search that does not
// take the histogram path with a pending accumulator returns ANTQ_ERR_LAUNCH before it enqueues anything.
struct HistXmax {
    float *out = nullptr;        // non-null: the histogram search of this call must fill it
    bool taken = false;
};

template <typename T>
static int launch_hist_search(const void *x, size_t n, const float *xmax, const float *ratios, int ncand, const HistTypes &ht,
                              double *sse, void *ws, bool pairs, float tmin_over_gmax, const int **run_if, hipStream_t st,
                              HistXmax *hx = nullptr)
{
    if constexpr (std::is_same<T, float>::value) {
        return ANTQ_ERR_UNSUPPORTED;
    } else {
        const size_t nv = n / 8;
        uint32_t G = (uint32_t)std::min<size_t>(std::max<size_t>(n / 65536, 16), (size_t)kHistMaxG);
        G = (G + 7u) & ~7u;                          // (k_hist16's workgroup -> (chunk stream, sign) map works in groups of 16)
        char *w = static_cast<char *>(ws);
        uint32_t *slabs = reinterpret_cast<uint32_t *>(w);
        uint32_t *count = reinterpret_cast<uint32_t *>(w + kHistCountOff);
        HistPairs hp;
        hp.xmax = xmax; hp.ratios = ratios; hp.ncand = ncand; hp.tmin_over_gmax = tmin_over_gmax;
        hp.seg = reinterpret_cast<uint32_t *>(w + kHistSegOff);
        hp.seg_count = reinterpret_cast<uint32_t *>(w + kHistSegCountOff);
        hp.flags = reinterpret_cast<int *>(w + kHistFlagsOff);
        hp.list = reinterpret_cast<uint32_t *>(w + kHistListOff);
        const unsigned nflat = (unsigned)(ht.ntypes * ncand);
        size_t lds = 0;
        for (int t = 0; t < ht.ntypes; t++) lds = std::max(lds, (size_t)ht.pa[t].tab_units * 16);
        if (pairs) {
            hipLaunchKernelGGL(k_hist_clear_flags, dim3(1), dim3(64), 0, st, hp.flags);
            hipLaunchKernelGGL((k_hist16<T, true>), dim3(2 * G), dim3(1024), 0, st, static_cast<const uint4 *>(x), nv, G, slabs, hp, nullptr);
            hipLaunchKernelGGL((k_hist_reduce<true>), dim3(256), dim3(256), 0, st, slabs, G, count, hp, G * 16u);
            hipLaunchKernelGGL((k_hist_score<T, true>), dim3(nflat), dim3(1024), lds, st, count, xmax, ratios, ncand, ht, sse, hp);
            *run_if = hp.flags;                  // non-zero iff a segment of the pair list overflowed
        } else {
            float *const xo = (hx && !hx->taken) ? hx->out : nullptr;
            if (hx && xo) hx->taken = true;
            if (xo)
                hipLaunchKernelGGL((k_hist16<T, false, true>), dim3(2 * G), dim3(1024), 0, st, static_cast<const uint4 *>(x), nv, G, slabs, hp, xo);
            else
                hipLaunchKernelGGL((k_hist16<T, false>), dim3(2 * G), dim3(1024), 0, st, static_cast<const uint4 *>(x), nv, G, slabs, hp, nullptr);
            hipLaunchKernelGGL((k_hist_reduce<false>), dim3(256), dim3(256), 0, st, slabs, G, count, hp, 0u);
            hipLaunchKernelGGL((k_hist_score<T, false>), dim3(nflat), dim3(1024), lds, st, count, xmax, ratios, ncand, ht, sse, hp);
        }
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
}

// ---- grid of a clip-search launch -------------------------------------------------------------------------------------
// A search workgroup (4 wavefronts) walks its share of the tensor once per candidate of its chunk of the candidate list
// (blockIdx.y splits the list); the kernels hold 3-6 workgroups per CU (84-160 registers).  Measured (tools/
// probe_search_split.py, profiles/r04_search_split.log):
//   * launches of many rounds of workgroups run best with ~10 candidates per chunk whatever the tensor size (4096^2 fp32:
//     504 us with the whole list in one chunk, 462 us with 8 chunks, 501 us with 19): short workgroups balance the CUs and
//     XCDs dynamically, shorter ones pay the per-task load / unpack / table-build overhead too often;
//   * launches of one to three rounds are ruled by the round count: a grid of 1.1 or 1.5 rounds costs two (768 x 3072 per
//     row: 77.6 us with 960 workgroups on 1024 slots, 98.7 us with 1152), so there the split is the one with the smallest
//     rounds x work per workgroup, with the kernel's real occupancy on this device.
struct SearchGrid {
    size_t blocks;
    int chunks, chunk;
};
template <typename K>
static int resident_workgroups(K kernel, size_t dyn_lds)
{
    // (cached per kernel and dynamic-LDS size: the search launches of a calibration pass repeat a handful of pairs)
    struct Ent { const void *k; size_t lds; int n; };
    static std::mutex mu;
    static std::vector<Ent> seen;
    const void *key = reinterpret_cast<const void *>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    for (const Ent &e : seen)
        if (e.k == key && e.lds == dyn_lds) return e.n;
    int dev = 0, ncu = 256, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 4;
    (void)hipGetLastError();
    const int n = std::max(1, ncu) * per_cu;
    seen.push_back(Ent{key, dyn_lds, n});
    return n;
}
// units: wavefront-sized pieces (per tensor: tasks; per row: rows, each `unit_work` tasks long); n: entries of the candidate
// list; resident: workgroups the device holds at once
static SearchGrid search_grid(size_t units, size_t unit_work, bool pt, int n, int resident)
{
    const size_t cap = pt ? 256 * 4 : 256 * 8;
    const int cmin = (n + kPtCand - 1) / kPtCand;
    const int c_many = std::max(cmin, (n + 9) / 10);                 // ~10 candidates per chunk
    const bool many_rounds = std::min((units + 3) / 4, cap) * (size_t)c_many >= 3 * (size_t)resident;
    SearchGrid best{0, 0, 0};
    double best_cost = 0.0;
    const int want = g_knob_schunks > 0 ? std::max(cmin, std::min(g_knob_schunks, n)) : (many_rounds ? std::min(c_many, n) : 0);   // knob 12 (A/B)
    const int c_lo = want > 0 ? want : cmin, c_hi = want > 0 ? want : std::max(cmin, std::min(n, 64));
    for (int c = c_lo; c <= c_hi; c++) {
        const int chunk = (n + c - 1) / c;
        const int ce = (n + chunk - 1) / chunk;
        if (ce != c && c != cmin && want == 0) continue;             // (the same split as a smaller c)
        size_t blocks = std::min((units + 3) / 4, cap);
        if (pt) {
            if (ce > kWsSlots) break;
            blocks = std::min(blocks, (size_t)(kWsSlots / ce));
        }
        const size_t upw = (units + blocks * 4 - 1) / (blocks * 4);               // units a wavefront walks
        const size_t rounds = (blocks * (size_t)ce + (size_t)resident - 1) / (size_t)resident;
        // per unit and candidate ~1, plus the load / unpack / magnitude pass of the unit (~3 candidates' worth)
        const double cost = (double)rounds * (double)upw * (double)unit_work * ((double)chunk + 3.0);
        if (best.chunks == 0 || cost < best_cost * 0.98) { best = SearchGrid{blocks, ce, chunk}; best_cost = cost; }
    }
    if (getenv("ANTQ_DEBUG_GRID"))
        fprintf(stderr, "search_grid: units %zu x %zu pt %d n %d resident %d -> blocks %zu chunks %d chunk %d\n", units, unit_work,
                (int)pt, n, resident, best.blocks, best.chunks, best.chunk);
    return best;
}

// ---- the threshold sweep (antq_k_sweep.h): per-row scales, every codebook with a threshold list ------------------------
// ANTQ_ERR_UNSUPPORTED: not this launch (the caller goes on to the direct kernels).  Where it is taken (knob 19 = 1, the
// default): rows of 2048 .. 65536 elements and codebooks of at most 20 thresholds.  Measured, 70 candidates x 3 ANT codebooks
// (profiles/r06_sweep_search.log): 4096 x 4096 fp32 1.29 -> 0.71 ms, 4096 x 16384 5.07 -> 2.27 ms; rows of 3072: 0.213 -> 0.201,
// rows of 768: no gain -- the per-row set-up (thresholds x candidates moved into the x domain, the scans along the candidate
// axis, ~17 KB of LDS per wavefront) is worth ~1500 elements' work; above 65536 elements the fixed-point sums could pass 62
// bits.  OliVe's 28-29 thresholds and 75 .. 250 clip range (half a dozen thresholds sweep across every element) run 0.6-0.8 x
// the direct kernels' speed: not taken.  knob 19 = 2 takes every eligible launch from 256 elements per row (tests).
static bool sweep_type_ok(const void *plan_host, float gmax)
{
    const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
    return ph->kind == kPlanLut && ph->hdom && ph->h_nthr > 0 && ph->h_nthr <= (g_knob_sweep == 2 ? (uint32_t)kSweepMaxThr : 20u) && gmax > 0.0f;
}
template <typename T, bool OVP>
static bool sweep_shape_ok(const void *x, size_t rows, size_t row_len, int ncand, int ntypes)
{
    constexpr int EPL = IO<T>::EPL;
    if (!g_knob_sweep || rows < 2 || ncand < 1 || ntypes < 1 || ntypes > kMaxTypes) return false;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || row_len % EPL != 0 || row_len < (g_knob_sweep == 2 ? 256u : 2048u) || row_len > 65536) return false;
    // (the pair rule: every pair that may hold an outlier under SOME candidate is evaluated literally -- ~1 % of the pairs of a
    //  3-sigma-clipped tensor, and that alone costs more than the direct kernels' whole pass: forced only, knob 19 = 2)
    if (OVP && g_knob_sweep != 2) return false;
    return !(OVP && (row_len & 1));                                  // (pairs would straddle rows)
}

template <typename T, bool OVP>
static int launch_sweep(const void *x, size_t rows, size_t row_len, const float *xmax, const float *ratios, int ncand, int ntypes,
                        const float *gmax, const void *const *plan_host, const void *const *plan_dev, double *sse, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    if (!sweep_shape_ok<T, OVP>(x, rows, row_len, ncand, ntypes)) return ANTQ_ERR_UNSUPPORTED;
    SweepType ty[kMaxTypes];
    memset(ty, 0, sizeof(ty));
    // candidate lists longer than the kernel's tables (128) go out in pieces: a candidate's sums are formed from exact integer
    // counts, so they do not depend on which other candidates share its launch (as long as no element changes between the
    // step-function and the literal class, which the smallest scale of the piece decides)
    const uint32_t cp_max = (uint32_t)std::min(ncand, kSweepMaxCand) + 1u;
    for (int t = 0; t < ntypes; t++) {
        const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host[t]);
        if (!sweep_type_ok(plan_host[t], gmax[t])) return ANTQ_ERR_UNSUPPORTED;
        const HThr *tl = plan_tlist(plan_host[t]);
        ty[t].tlist = reinterpret_cast<const uint4 *>(static_cast<const char *>(plan_dev[t]) + ph->tlist_off);
        ty[t].grid = reinterpret_cast<const float *>(plan_tab_ptr(plan_dev[t]));
        ty[t].n_thr = ph->h_nthr;
        ty[t].m = ph->m;
        ty[t].gmax = gmax[t];
        const float flim = ph->fastlim * 0.99999f;
        ty[t].lim = flim < ph->xlim ? flim : ph->xlim;
        ty[t].kout_pos = ty[t].kout_neg = -1;
        for (uint32_t k = 0; k < ph->h_nthr; k++) {
            const bool lo_out = (tl[k].flags & 1u) != 0u, hi_out = (tl[k].flags & 2u) != 0u;
            if (k > 0 && !(tl[k].T > tl[k - 1].T)) return ANTQ_ERR_UNSUPPORTED;                   // (ascending: what the kernel's searches assume)
            if (!lo_out && hi_out) { if (ty[t].kout_pos >= 0 || !(tl[k].T > 0.0f)) return ANTQ_ERR_UNSUPPORTED; ty[t].kout_pos = (int)k; }
            if (lo_out && !hi_out) { if (ty[t].kout_neg >= 0 || !(tl[k].T < 0.0f)) return ANTQ_ERR_UNSUPPORTED; ty[t].kout_neg = (int)k; }
        }
        if (sweep_lds_bytes(ph->h_nthr, cp_max) > 64 * 1024) return ANTQ_ERR_UNSUPPORTED;
    }
    const unsigned blocks = (unsigned)std::min<size_t>(rows, (size_t)1 << 20);
    for (int t = 0; t < ntypes; t++)             // one codebook per launch: [ncand][rows] doubles each
        for (int c0 = 0; c0 < ncand; c0 += kSweepMaxCand) {
            const int nc = std::min(kSweepMaxCand, ncand - c0);
            hipLaunchKernelGGL((k_search_sweep<T, OVP>), dim3(blocks), dim3(64), sweep_lds_bytes(ty[t].n_thr, (uint32_t)nc + 1u), st,
                               static_cast<const uint4 *>(x), (uint32_t)(row_len / EPL), rows, xmax, ratios + c0,
                               sse + ((size_t)t * (size_t)ncand + (size_t)c0) * rows, ty[t], (uint32_t)nc, (uint32_t)nc + 1u);
        }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T>
static bool sweep_pt_shape_ok(const void *x, size_t n, int ncand)
{
    constexpr int EPL = IO<T>::EPL;
    if (!g_knob_sweep || sizeof(T) != 4 || ncand < 1 || ncand > kSweepMaxCand) return false;
    return reinterpret_cast<uintptr_t>(x) % 16 == 0 && n % EPL == 0 && n >= (g_knob_sweep == 2 ? 16384u : (1u << 22)) && n < ((size_t)1 << 31);      // (below 16 workgroups' worth the launcher declines: G < 16)
}

// The sweep for a tensor with ONE scale (fp32: 16-bit tensors take the histogram search).  ws: the search workspace (slabs of
// the workgroups' integer tables, then their totals).  ANTQ_ERR_UNSUPPORTED: not this launch.
template <typename T, bool OVP>
static int launch_sweep_pt(const void *x, size_t n, const float *xmax, const float *ratios, int ncand, int ntypes, const float *gmax,
                           const void *const *plan_host, const void *const *plan_dev, double *sse, void *ws, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    if (!sweep_pt_shape_ok<T>(x, n, ncand) || ntypes < 1 || ntypes > kMaxTypes || !ws || (OVP && g_knob_sweep != 2)) return ANTQ_ERR_UNSUPPORTED;
    SweepType ty[kMaxTypes];
    memset(ty, 0, sizeof(ty));
    const uint32_t cp = (uint32_t)ncand + 1u;
    uint32_t cells_max = 0;
    for (int t = 0; t < ntypes; t++) {
        const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host[t]);
        if (!sweep_type_ok(plan_host[t], gmax[t])) return ANTQ_ERR_UNSUPPORTED;
        const HThr *tl = plan_tlist(plan_host[t]);
        ty[t].tlist = reinterpret_cast<const uint4 *>(static_cast<const char *>(plan_dev[t]) + ph->tlist_off);
        ty[t].grid = reinterpret_cast<const float *>(plan_tab_ptr(plan_dev[t]));
        ty[t].n_thr = ph->h_nthr;
        ty[t].m = ph->m;
        ty[t].gmax = gmax[t];
        const float flim = ph->fastlim * 0.99999f;
        ty[t].lim = flim < ph->xlim ? flim : ph->xlim;
        ty[t].kout_pos = ty[t].kout_neg = -1;
        for (uint32_t k = 0; k < ph->h_nthr; k++) {
            const bool lo_out = (tl[k].flags & 1u) != 0u, hi_out = (tl[k].flags & 2u) != 0u;
            if (k > 0 && !(tl[k].T > tl[k - 1].T)) return ANTQ_ERR_UNSUPPORTED;
            if (!lo_out && hi_out) { if (ty[t].kout_pos >= 0 || !(tl[k].T > 0.0f)) return ANTQ_ERR_UNSUPPORTED; ty[t].kout_pos = (int)k; }
            if (lo_out && !hi_out) { if (ty[t].kout_neg >= 0 || !(tl[k].T < 0.0f)) return ANTQ_ERR_UNSUPPORTED; ty[t].kout_neg = (int)k; }
        }
        if (sweep_lds_bytes(ph->h_nthr, cp) > 64 * 1024) return ANTQ_ERR_UNSUPPORTED;
        cells_max = std::max(cells_max, sweep_slab_cells(ph->h_nthr, cp));
    }
    const size_t nv = n / EPL;
    const size_t ws_bytes = antq_search_workspace_bytes();
    size_t G = std::min<size_t>((nv + 255) / 256, 2048);                        // >= 4 rounds of 64 vectors per workgroup
    const uint32_t per_group = 64;
    G = std::min(G, ws_bytes / ((size_t)cells_max * 8) - (2048 / per_group + 2));
    if (G < 16) return ANTQ_ERR_UNSUPPORTED;
    const uint32_t ngroups = (uint32_t)((G + per_group - 1) / per_group);
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    const int fbits = std::min(38, 53 - lg);                                    // n * 2^(fbits + 8) <= 2^61
    long long *slabs = static_cast<long long *>(ws);
    for (int t = 0; t < ntypes; t++) {
        const uint32_t ncells = sweep_slab_cells(ty[t].n_thr, cp), nint = 2u * ty[t].n_thr * cp + 132u;
        long long *part = slabs + G * (size_t)ncells, *tot = part + (size_t)ngroups * ncells;
        const size_t lds = sweep_lds_bytes(ty[t].n_thr, cp);
        hipLaunchKernelGGL((k_search_sweep<T, OVP, true>), dim3((unsigned)G), dim3(64), lds, st, static_cast<const uint4 *>(x), nv, (size_t)1,
                           xmax, ratios, sse + (size_t)t * ncand, ty[t], (uint32_t)ncand, cp, slabs, fbits);
        hipLaunchKernelGGL(k_sweep_pt_total, dim3((ncells + 255) / 256, ngroups), dim3(256), 0, st, slabs, (uint32_t)G, per_group, ncells, nint, part);
        hipLaunchKernelGGL(k_sweep_pt_total, dim3((ncells + 255) / 256, 1), dim3(256), 0, st, part, ngroups, ngroups, ncells, nint, tot);
        hipLaunchKernelGGL(k_sweep_pt_finish, dim3(1), dim3(64), lds, st, tot, xmax, ratios, sse + (size_t)t * ncand, ty[t], (uint32_t)ncand, cp, fbits);
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}


// ---- the sorted-row search (antq_k_sortsearch.h): every codebook and candidate of a launch on ONE sort of the row -------
// ANTQ_ERR_UNSUPPORTED: not this launch (the caller goes on to the sweep / the direct kernels).  knob 20: 0 off, 1 the default
// rule, 2 every eligible launch (tests).
constexpr uint32_t kSortLdsMax = 64 * 1024;
static bool sort_type_ok(const void *plan_host, float gmax)
{
    const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
    return ph->kind == kPlanLut && ph->hdom && ph->h_nthr > 0 && ph->h_nthr <= 64u && gmax > 0.0f;
}
static bool sort_fill_types(SortTypes &stt, int ntypes, const float *gmax, const void *const *plan_host, const void *const *plan_dev)
{
    memset(&stt, 0, sizeof(stt));
    if (ntypes < 1 || ntypes > kMaxTypes) return false;
    stt.ntypes = ntypes;
    uint32_t nmax = 0;
    for (int t = 0; t < ntypes; t++) {
        if (!sort_type_ok(plan_host[t], gmax[t])) return false;
        const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host[t]);
        const HThr *tl = plan_tlist(plan_host[t]);
        SweepType &ty = stt.ty[t];
        ty.tlist = reinterpret_cast<const uint4 *>(static_cast<const char *>(plan_dev[t]) + ph->tlist_off);
        ty.grid = reinterpret_cast<const float *>(plan_tab_ptr(plan_dev[t]));
        ty.n_thr = ph->h_nthr;
        ty.m = ph->m;
        ty.gmax = gmax[t];
        const float flim = ph->fastlim * 0.99999f;
        ty.lim = flim < ph->xlim ? flim : ph->xlim;
        ty.kout_pos = ty.kout_neg = -1;
        stt.nneg[t] = 0;
        for (uint32_t k = 0; k < ph->h_nthr; k++) {
            const bool lo_out = (tl[k].flags & 1u) != 0u, hi_out = (tl[k].flags & 2u) != 0u;
            if (k > 0 && !(tl[k].T > tl[k - 1].T)) return false;                     // (ascending: what the kernel's searches assume)
            if (tl[k].T < 0.0f) stt.nneg[t] = k + 1u;
            if (!lo_out && hi_out) { if (ty.kout_pos >= 0 || !(tl[k].T > 0.0f)) return false; ty.kout_pos = (int)k; }
            if (lo_out && !hi_out) { if (ty.kout_neg >= 0 || !(tl[k].T < 0.0f)) return false; ty.kout_neg = (int)k; }
        }
        nmax = std::max(nmax, ph->h_nthr);
    }
    stt.nthr_pad = (nmax + (uint32_t)kSortKS - 1u) / (uint32_t)kSortKS * (uint32_t)kSortKS;
    return true;
}
// candidates per launch so that the workgroup's tables fit (the sums of a candidate do not depend on its company)
static int sort_piece(const SortTypes &stt, int ncand, bool ovp)
{
    // ... and a thread's items (partial terms in registers: kSortNI / kSortNC / kSortNL per thread) cover the launch
    const uint32_t nkg = stt.nthr_pad / (uint32_t)kSortKS;
    auto fits = [&](int nc) {
        const uint32_t ntc = (uint32_t)(stt.ntypes * nc);
        return sort_lds(ntc, stt.nthr_pad, stt.ntypes, ovp).total <= kSortLdsMax && ntc * nkg <= (uint32_t)(kSortNI * kSortNT) &&
               ntc <= (uint32_t)(kSortNL * kSortNT) && (!ovp || 4u * ntc <= (uint32_t)(kSortNC * kSortNT));
    };
    int nc = ncand;
    while (nc > 0 && !fits(nc)) nc = nc > 1 ? (nc + 1) / 2 : 0;
    return nc;
}
template <typename T, bool OVP>
static bool sort_shape_ok(const void *x, size_t rows, size_t row_len, int ncand, int ntypes)
{
    constexpr int EPL = IO<T>::EPL;
    if (!g_knob_sort || rows < 2 || ncand < 1 || ntypes < 1 || ntypes > kMaxTypes) return false;
    // Where it pays (profiles/r06_sort_scan.log): ANT codebooks from rows of 128 elements -- up to 1024 elements one row per
    // wavefront (2.5 x the direct kernels at 128 and 256, 3.7 x at 512, 3.9 x at 768, 4.7 x at 1024), above that one row per
    // workgroup in 4096-element chunks (3.1 x at 1152, 5.9 x at 4096); OliVe's pair rule from 256 (2.1 x; 3.1 x at 512, 3.2 x at
    // 768, 3.9 x at 1024 one row per wavefront; 2.5 x at 1152, 5.4 x at 4096 one row per workgroup)
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || row_len % EPL != 0 || row_len < (g_knob_sort == 2 ? 128u : (OVP ? 256u : 128u))) return false;
    return !(OVP && (row_len & 1));                                  // (pairs would straddle rows)
}
template <typename T, bool OVP>
static int launch_sorted(const void *x, size_t rows, size_t row_len, const float *xmax, const float *ratios, int ncand, int ntypes,
                         const float *gmax, const void *const *plan_host, const void *const *plan_dev, double *sse, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    if (!sort_shape_ok<T, OVP>(x, rows, row_len, ncand, ntypes)) return ANTQ_ERR_UNSUPPORTED;
    SortTypes stt;
    if (!sort_fill_types(stt, ntypes, gmax, plan_host, plan_dev)) return ANTQ_ERR_UNSUPPORTED;
    {
        // rows of at most 1024 elements: one row per wavefront (k_search_sorted_short; knob 21 = 0: the 4096-key kernel, A/B)
        if (row_len <= (size_t)kSortKSh && g_knob_sort_short) {
            int piece = ncand;
            auto lds_of = [&](int nc) { return (((uint32_t)ntypes * kSortTy * 4u + 15u) & ~15u) + 4u * sort_short_wave_bytes((uint32_t)(ntypes * nc)); };
            while (piece > 1 && (lds_of(piece) > kSortLdsMax || ntypes * piece > 512)) piece = (piece + 1) / 2;
            const unsigned blocks = (unsigned)std::min<size_t>((rows + 3) / 4, (size_t)1 << 20);
            for (int c0 = 0; c0 < ncand; c0 += piece) {
                const int nc = std::min(piece, ncand - c0);
                hipLaunchKernelGGL((k_search_sorted_short<T, OVP>), dim3(blocks), dim3(256), lds_of(nc), st, static_cast<const uint4 *>(x),
                                   (uint32_t)(row_len / EPL), rows, xmax, ratios + c0, sse + (size_t)c0 * rows, stt, (uint32_t)nc, (uint32_t)ncand);
            }
            return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
        }
    }
    const int piece = sort_piece(stt, ncand, OVP);
    if (piece < 1) return ANTQ_ERR_UNSUPPORTED;
    const unsigned blocks = (unsigned)std::min<size_t>(rows, (size_t)1 << 20);
    for (int c0 = 0; c0 < ncand; c0 += piece) {
        const int nc = std::min(piece, ncand - c0);
        const SortLds L = sort_lds((uint32_t)(ntypes * nc), stt.nthr_pad, ntypes, OVP);
        hipLaunchKernelGGL((k_search_sorted<T, OVP, false>), dim3(blocks), dim3(kSortNT), L.total, st, static_cast<const uint4 *>(x),
                           row_len / EPL, rows, xmax, ratios + c0, sse + (size_t)c0 * rows, stt, (uint32_t)nc, (uint32_t)ncand, nullptr);
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T>
static bool sort_pt_shape_ok(const void *x, size_t n, int ncand)
{
    constexpr int EPL = IO<T>::EPL;
    if (!g_knob_sort || sizeof(T) != 4 || ncand < 1) return false;     // (16-bit tensors with one scale: the histogram search)
    return reinterpret_cast<uintptr_t>(x) % 16 == 0 && n % EPL == 0 && n >= (g_knob_sort == 2 ? 4096u : (1u << 20)) && n < ((size_t)1 << 40);
}
// A tensor with ONE scale.  ws: the search workspace (the workgroups' slabs of partial terms, then their totals).
template <typename T, bool OVP>
static int launch_sorted_pt(const void *x, size_t n, const float *xmax, const float *ratios, int ncand, int ntypes, const float *gmax,
                            const void *const *plan_host, const void *const *plan_dev, double *sse, void *ws, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    if (!sort_pt_shape_ok<T>(x, n, ncand) || !ws || (OVP && (n & 1))) return ANTQ_ERR_UNSUPPORTED;
    SortTypes stt;
    if (!sort_fill_types(stt, ntypes, gmax, plan_host, plan_dev)) return ANTQ_ERR_UNSUPPORTED;
    const int piece = sort_piece(stt, ncand, OVP);
    if (piece < 1) return ANTQ_ERR_UNSUPPORTED;
    const size_t nv = n / EPL, nchunks = (n + kSortK - 1) / kSortK;
    const uint32_t nkg = stt.nthr_pad / (uint32_t)kSortKS, per_group = 32;
    const size_t ws_cells = antq_search_workspace_bytes() / 8;
    double *slabs = static_cast<double *>(ws);
    for (int c0 = 0; c0 < ncand; c0 += piece) {
        const int nc = std::min(piece, ncand - c0);
        const uint32_t ntc = (uint32_t)(ntypes * nc), ncell = ntc * nkg + ntc + (OVP ? 4u * ntc : 0u) + 1u;
        // a workgroup sets its tables up once (the thresholds of every candidate moved into the x domain: ~ a tenth of a
        // chunk's work): three chunks and more per workgroup, three workgroups per CU
        size_t G = std::min<size_t>((nchunks + 2) / 3, 768);
        G = std::min(G, ws_cells / ncell - (768 / per_group + 2));
        if (G < 1 || ws_cells / ncell < 768 / per_group + 3) return ANTQ_ERR_UNSUPPORTED;
        const uint32_t ngroups = (uint32_t)((G + per_group - 1) / per_group);
        double *part = slabs + G * (size_t)ncell, *tot = part + (size_t)ngroups * ncell;
        const SortLds L = sort_lds(ntc, stt.nthr_pad, ntypes, OVP);
        hipLaunchKernelGGL((k_search_sorted<T, OVP, true>), dim3((unsigned)G), dim3(kSortNT), L.total, st, static_cast<const uint4 *>(x), nv,
                           (size_t)1, xmax, ratios + c0, sse + c0, stt, (uint32_t)nc, (uint32_t)ncand, slabs);
        hipLaunchKernelGGL(k_sort_pt_total, dim3((ncell + 255) / 256, ngroups), dim3(256), 0, st, slabs, (uint32_t)G, per_group, ncell, part);
        hipLaunchKernelGGL(k_sort_pt_total, dim3((ncell + 255) / 256, 1), dim3(256), 0, st, part, ngroups, ngroups, ncell, tot);
        hipLaunchKernelGGL(k_sort_pt_finish, dim3((ntc + 255) / 256), dim3(256), 0, st, tot, ntc, nkg, OVP ? 1 : 0, (uint32_t)nc, (uint32_t)ncand, sse + c0);
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T, bool OVP>
static int launch_search(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                         const float *ratios, int ncand, float gmax, const PlanArgs &pa, const void *plan_host,
                         const void *plan_dev, double *sse, double *ws, hipStream_t st, HistXmax *hx = nullptr)
{
    constexpr int EPL = IO<T>::EPL;
    const size_t lds = (size_t)pa.tab_units * 16;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || (per_row ? row_len : rows * row_len) % EPL != 0) {
        size_t strips = per_row ? rows : (rows * row_len + 16383) / 16384;
        size_t blocks = (strips + 3) / 4;
        const size_t ychunks = ((size_t)ncand + kPtCand - 1) / kPtCand;
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (!per_row) {
            if (ychunks > (size_t)kWsSlots) return ANTQ_ERR_UNSUPPORTED;
            blocks = std::min(blocks, (size_t)kWsSlots / ychunks);
        }
        hipLaunchKernelGGL((k_search_sse_scalar<T, OVP>), dim3((unsigned)blocks, (unsigned)ychunks), dim3(256), lds, st, x, rows,
                           row_len, xmax, per_row, ratios, ncand, gmax, sse, ws, pa, plan_tab_ptr(plan_dev));
        if (!per_row)
            hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)ncand), dim3(256), 0, st, ws, (uint32_t)blocks, kPtCand, sse);
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
    if (per_row && rows > 1) {                   // per-row scales: the sorted-row search, then the threshold sweep, where they apply
        const void *ph1[1] = {plan_host}, *pd1[1] = {plan_dev};
        int rc = launch_sorted<T, OVP>(x, rows, row_len, xmax, ratios, ncand, 1, &gmax, ph1, pd1, sse, st);
        if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        rc = launch_sweep<T, OVP>(x, rows, row_len, xmax, ratios, ncand, 1, &gmax, ph1, pd1, sse, st);
        if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
    }
    if (!per_row) { row_len = rows * row_len; rows = 1; }
    const int *run_if = nullptr;                 // device flag: run the direct kernels only if it is set (pair-list overflow)
    float bound = 0.0f;
    if (rows == 1 && hist_eligible<T>(row_len, OVP, x, ncand) && (!OVP || hist_outlier_bound(plan_host, gmax, bound))) {
        HistTypes ht;
        memset(&ht, 0, sizeof(ht));
        ht.ntypes = 1;
        ht.pa[0] = pa;
        ht.plan_tab[0] = plan_tab_ptr(plan_dev);
        ht.gmax[0] = gmax;
        const int rc = launch_hist_search<T>(x, row_len, xmax, ratios, ncand, ht, sse, ws, OVP, bound, &run_if, st, hx);
        if (rc != ANTQ_OK || !OVP) return rc;
    }
    if (hx && hx->out && !hx->taken) return ANTQ_ERR_LAUNCH;     // (the caller counted on the histogram pass for its statistic)
    if (rows == 1 && !run_if) {                  // one scale, no histogram search in front: the sorted search / the sweep over many workgroups
        const void *ph1[1] = {plan_host}, *pd1[1] = {plan_dev};
        int rc = launch_sorted_pt<T, OVP>(x, row_len, xmax, ratios, ncand, 1, &gmax, ph1, pd1, sse, ws, st);
        if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        rc = launch_sweep_pt<T, OVP>(x, row_len, xmax, ratios, ncand, 1, &gmax, ph1, pd1, sse, ws, st);
        if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
    }
    const size_t vpr = row_len / EPL;
    if (vpr > 0xffffffffull) return ANTQ_ERR_UNSUPPORTED;
    // vectors per lane and task: the per-candidate work of a task that does not depend on its size (the row table for this
    // scale, ~60 instructions incl. the exact f64 threshold moves; the scale's division; the 64-lane sum of the squared
    // errors) is shared by twice the elements with 8 (tools/probe_search.py); short rows keep 4 (fewer idle lanes)
    int U = (g_knob_u == 4 || g_knob_u == 8) ? g_knob_u : ((vpr >= 512 && EPL == 4) ? 8 : 4);   // (16-bit: 4 measured faster)
    // Short rows (round 5): a row of 96 vectors (768 bf16 elements: 60 of BERT-base's 72 Linear weights) in a 4-vector task
    // leaves 62 % of the lane slots idle; rows of at most 128 / 64 vectors take 2 / 1 vectors per lane (per-row searches
    // only; knob 0 = 4 restores the 4-vector tasks for an A/B)
    const bool pt = rows == 1;
    if (!pt && vpr < kRowKernelMinVpr && g_knob_u == 0) U = vpr < kSearchMinVpr ? 1 : 2;
    const size_t tpr = (vpr + 64 * U - 1) / (64 * U);
    const size_t total = rows * tpr;
    if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
    // per tensor: tasks over all wavefronts; per row: one wavefront per row (it walks the row's tasks in order)
    const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
    // (the per-candidate x-domain table from 64 vectors per row on -- the fake-quant row kernels want 128: their table is
    //  built once per task, this one once per task AND candidate, but it replaces ~8 instructions per element and candidate)
    const bool xd = (g_knob_x != 0) && pa.kind == kPlanLut && ph->xdom && vpr >= kSearchMinVpr && (pt || U != 1);
    const XArgs xa = xargs_from_plan(plan_host, pa);
    const uint4 *xv = static_cast<const uint4 *>(x);
    SearchGrid sg{0, 0, 0};
#define ANTQ_LAUNCH_S(PT_, XD_, U_)                                                                                \
    do {                                                                                                           \
        const int resident_ = resident_workgroups(k_search_sse<T, OVP, U_, PT_, XD_>, (XD_) ? 0 : lds);            \
        sg = search_grid(PT_ ? total : rows, PT_ ? 1 : tpr, PT_, ncand, resident_);                                \
        if (sg.chunks == 0) return ANTQ_ERR_UNSUPPORTED;                                                           \
        hipLaunchKernelGGL((k_search_sse<T, OVP, U_, PT_, XD_>), dim3((unsigned)sg.blocks, (unsigned)sg.chunks), dim3(256), \
                           (XD_) ? 0 : lds, st, xv, (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, rows, xmax, per_row,  \
                           ratios, ncand, gmax, sse, ws, pa, plan_tab_ptr(plan_dev), sg.chunk, xa, run_if);         \
    } while (0)
#define ANTQ_LAUNCH_SU(PT_, XD_) do { if (U == 8) ANTQ_LAUNCH_S(PT_, XD_, 8); else ANTQ_LAUNCH_S(PT_, XD_, 4); } while (0)
    if (pt) { if (xd) ANTQ_LAUNCH_SU(true, true); else ANTQ_LAUNCH_SU(true, false); }
    else if (U == 2) { if (xd) ANTQ_LAUNCH_S(false, true, 2); else ANTQ_LAUNCH_S(false, false, 2); }
    else if (U == 1) ANTQ_LAUNCH_S(false, false, 1);
    else    { if (xd) ANTQ_LAUNCH_SU(false, true); else ANTQ_LAUNCH_SU(false, false); }
#undef ANTQ_LAUNCH_SU
#undef ANTQ_LAUNCH_S
    if (pt) hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)ncand), dim3(256), 0, st, ws, (uint32_t)sg.blocks, sg.chunk, sse, run_if);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

// all candidate types of a type selection on one read of the tensor; every plan must have the x-domain path
template <typename T, bool OVP>
static int launch_search_multi(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                               const float *ratios, int ncand, int ntypes, const float *gmax, const void *const *plan_host,
                               const void *const *plan_dev, double *sse, double *ws, hipStream_t st, HistXmax *hx = nullptr)
{
    constexpr int EPL = IO<T>::EPL;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || (per_row ? row_len : rows * row_len) % EPL != 0) return ANTQ_ERR_UNSUPPORTED;
    if (per_row && rows > 1) {                   // per-row scales: the sorted-row search, then the threshold sweep, where they apply
        int rc = launch_sorted<T, OVP>(x, rows, row_len, xmax, ratios, ncand, ntypes, gmax, plan_host, plan_dev, sse, st);
        if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        // (a codebook's sums must not depend on its company: one search per type if some of them would take the sorted search)
        if (sort_shape_ok<T, OVP>(x, rows, row_len, ncand, 1))
            for (int t = 0; t < ntypes; t++)
                if (sort_type_ok(plan_host[t], gmax[t])) return ANTQ_ERR_UNSUPPORTED;
        rc = launch_sweep<T, OVP>(x, rows, row_len, xmax, ratios, ncand, ntypes, gmax, plan_host, plan_dev, sse, st);
        if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        // a codebook's sums must not depend on which other codebooks are searched with it: if one search per type would send
        // SOME of these types through the sweep, the caller has to issue one search per type (it does on this return code)
        if (sweep_shape_ok<T, OVP>(x, rows, row_len, ncand, 1))
            for (int t = 0; t < ntypes; t++)
                if (sweep_type_ok(plan_host[t], gmax[t])) return ANTQ_ERR_UNSUPPORTED;
    }
    if (!per_row) { row_len = rows * row_len; rows = 1; }
    const int *run_if = nullptr;
    if (rows == 1 && hist_eligible<T>(row_len, OVP, x, ntypes * ncand)) {
        HistTypes ht;
        memset(&ht, 0, sizeof(ht));
        ht.ntypes = ntypes;
        float bound = 3.0e38f;
        bool ok = true;
        for (int t = 0; t < ntypes; t++) {
            PlanArgs pa;
            if (!plan_args_from_host(plan_host[t], pa)) return ANTQ_ERR_PLAN;
            ht.pa[t] = pa;
            ht.plan_tab[t] = plan_tab_ptr(plan_dev[t]);
            ht.gmax[t] = gmax[t];
            float b = 0.0f;
            if (OVP) { ok = ok && hist_outlier_bound(plan_host[t], gmax[t], b); bound = std::min(bound, b); }
        }
        if (ok) {
            const int rc = launch_hist_search<T>(x, row_len, xmax, ratios, ncand, ht, sse, ws, OVP, bound, &run_if, st, hx);
            if (rc != ANTQ_OK || !OVP) return rc;
        }
    }
    if (hx && hx->out && !hx->taken) return ANTQ_ERR_LAUNCH;     // (see launch_search)
    if (rows == 1 && !run_if) {
        int rc = launch_sorted_pt<T, OVP>(x, row_len, xmax, ratios, ncand, ntypes, gmax, plan_host, plan_dev, sse, ws, st);
        if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        if (sort_pt_shape_ok<T>(x, row_len, ncand) && !(OVP && (row_len & 1)))
            for (int t = 0; t < ntypes; t++)
                if (sort_type_ok(plan_host[t], gmax[t])) return ANTQ_ERR_UNSUPPORTED;
        rc = launch_sweep_pt<T, OVP>(x, row_len, xmax, ratios, ncand, ntypes, gmax, plan_host, plan_dev, sse, ws, st);
        if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        // (as for rows: a codebook's sums must not depend on its company -- one search per type if some types would sweep)
        if (sweep_pt_shape_ok<T>(x, row_len, ncand) && !(OVP && g_knob_sweep != 2))
            for (int t = 0; t < ntypes; t++)
                if (sweep_type_ok(plan_host[t], gmax[t])) return ANTQ_ERR_UNSUPPORTED;
    }
    const size_t vpr = row_len / EPL;
    if (vpr < kSearchMinVpr || vpr > 0xffffffffull) return ANTQ_ERR_UNSUPPORTED;
    MultiArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.ntypes = ntypes;
    for (int t = 0; t < ntypes; t++) {
        PlanArgs pa;
        if (!plan_args_from_host(plan_host[t], pa)) return ANTQ_ERR_PLAN;
        const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host[t]);
        if (!(g_knob_x != 0 && pa.kind == kPlanLut && ph->xdom)) return ANTQ_ERR_UNSUPPORTED;
        ma.xa[t] = xargs_from_plan(plan_host[t], pa);
        const uint4 *tab = plan_tab_ptr(plan_dev[t]);
        ma.entries[t] = tab + (pa.m_pad >> 2);
        ma.grid[t] = reinterpret_cast<const float *>(tab);
        ma.gmax[t] = gmax[t];
    }
    // (as in launch_search; 16-bit data stays at 4: with 8 the kernel needs 169 registers -- 2 waves per SIMD instead of 3)
    int U = (g_knob_u == 4 || g_knob_u == 8) ? g_knob_u : ((vpr >= 512 && EPL == 4) ? 8 : 4);
    const bool pt = rows == 1;
    if (!pt && vpr < kRowKernelMinVpr && g_knob_u == 0) U = 2;          // (as in launch_search: the same sums as one search per type)
    const size_t tpr = (vpr + 64 * U - 1) / (64 * U);
    const size_t total = rows * tpr;
    if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
    const int nflat = ntypes * ncand;
    const uint4 *xv = static_cast<const uint4 *>(x);
    SearchGrid sg{0, 0, 0};
#define ANTQ_LAUNCH_M(PT_, U_)                                                                                            \
    do {                                                                                                                  \
        const int resident_ = resident_workgroups(k_search_sse_multi<T, OVP, U_, PT_>, 0);                                \
        sg = search_grid(PT_ ? total : rows, PT_ ? 1 : tpr, PT_, nflat, resident_);                                       \
        if (sg.chunks == 0) return ANTQ_ERR_UNSUPPORTED;                                                                  \
        hipLaunchKernelGGL((k_search_sse_multi<T, OVP, U_, PT_>), dim3((unsigned)sg.blocks, (unsigned)sg.chunks), dim3(256), 0, st, \
                           xv, (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, rows, xmax, per_row, ratios, ncand, sse, ws, ma, sg.chunk, run_if); \
    } while (0)
    if (pt) {
        if (U == 8) ANTQ_LAUNCH_M(true, 8); else ANTQ_LAUNCH_M(true, 4);
        hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)nflat), dim3(256), 0, st, ws, (uint32_t)sg.blocks, sg.chunk, sse, run_if);
    } else {
        if (U == 8) ANTQ_LAUNCH_M(false, 8); else if (U == 2) ANTQ_LAUNCH_M(false, 2); else ANTQ_LAUNCH_M(false, 4);
    }
#undef ANTQ_LAUNCH_M
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

}  // namespace antq

using namespace antq;

static int search_sse_multi_impl(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                                 const float *ratios, int ncand, int ntypes, const float *gmax_host,
                                 const void *const *plan_host, const void *const *plan_dev, unsigned flags, int dtype,
                                 double *sse, void *workspace, void *stream, HistXmax *hx)
{
    if (rows == 0 || row_len == 0 || ncand == 0 || ntypes == 0) return ANTQ_OK;
    if (!x || !xmax || !ratios || !gmax_host || !plan_host || !plan_dev || !sse || ncand < 0 || ntypes < 0) return ANTQ_ERR_ARG;
    if ((!per_row || rows == 1) && !workspace) return ANTQ_ERR_ARG;
    double *ws = static_cast<double *>(workspace);
    if (ntypes > kMaxTypes) return ANTQ_ERR_UNSUPPORTED;
    for (int t = 0; t < ntypes; t++)
        if (!plan_host[t] || !plan_dev[t]) return ANTQ_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    const int pr = per_row ? 1 : 0;
#define ANTQ_SM(TT)                                                                                                     \
    (ovp ? launch_search_multi<TT, true>(x, rows, row_len, xmax, pr, ratios, ncand, ntypes, gmax_host, plan_host, plan_dev, sse, ws, st, hx)   \
         : launch_search_multi<TT, false>(x, rows, row_len, xmax, pr, ratios, ncand, ntypes, gmax_host, plan_host, plan_dev, sse, ws, st, hx))
    switch (dtype) {
    case ANTQ_F32: return ANTQ_SM(float);
    case ANTQ_BF16: return ANTQ_SM(bf16_tag);
    case ANTQ_F16: return ANTQ_SM(f16_tag);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
#undef ANTQ_SM
}

extern "C" int antq_search_sse_multi(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                                     const float *ratios, int ncand, int ntypes, const float *gmax_host,
                                     const void *const *plan_host, const void *const *plan_dev, unsigned flags, int dtype,
                                     double *sse, void *workspace, void *stream)
{
    return search_sse_multi_impl(x, rows, row_len, xmax, per_row, ratios, ncand, ntypes, gmax_host, plan_host, plan_dev, flags, dtype, sse,
                                 workspace, stream, nullptr);
}

static int search_sse_impl(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                           const float *ratios, int ncand, float gmax, const void *plan_host, const void *plan_dev,
                           unsigned flags, int dtype, double *sse, void *workspace, void *stream, HistXmax *hx)
{
    if (rows == 0 || row_len == 0 || ncand == 0) return ANTQ_OK;
    if (!x || !xmax || !ratios || !plan_host || !plan_dev || !sse || ncand < 0) return ANTQ_ERR_ARG;
    if ((!per_row || rows == 1) && !workspace) return ANTQ_ERR_ARG;
    double *ws = static_cast<double *>(workspace);
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    const int pr = per_row ? 1 : 0;
    switch (dtype) {
    case ANTQ_F32:
        return ovp ? launch_search<float, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st, hx)
                   : launch_search<float, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st, hx);
    case ANTQ_BF16:
        return ovp ? launch_search<bf16_tag, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st, hx)
                   : launch_search<bf16_tag, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st, hx);
    case ANTQ_F16:
        return ovp ? launch_search<f16_tag, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st, hx)
                   : launch_search<f16_tag, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st, hx);
    default:
        return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_search_sse(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                               const float *ratios, int ncand, float gmax, const void *plan_host, const void *plan_dev,
                               unsigned flags, int dtype, double *sse, void *workspace, void *stream)
{
    return search_sse_impl(x, rows, row_len, xmax, per_row, ratios, ncand, gmax, plan_host, plan_dev, flags, dtype, sse, workspace, stream, nullptr);
}

// (the direct kernels' workgroup partials, or -- never at the same time -- the histogram path's slabs and counts)
extern "C" size_t antq_search_workspace_bytes(void) { return std::max((size_t)kWsSlots * kPtCand * sizeof(double), kHistWorkspaceBytes); }

extern "C" int antq_search_pick(const double *sse, const float *xmax, const float *ratios, int ncand, size_t na,
                                size_t row_len, float *best_score, float *best_alpha, void *stream)
{
    if (na == 0) return ANTQ_OK;
    if (!sse || !xmax || !ratios || !best_score || !best_alpha || ncand < 0 || row_len == 0) return ANTQ_ERR_ARG;
    const size_t blocks = (na + 3) / 4;              // one wavefront per row
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(antq::k_search_pick, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), sse, xmax,
                       ratios, ncand, na, (double)row_len, best_score, best_alpha);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}



// ------------------------------------------------------------------------------------------------
// antq_calibrate: one call = x_max, every candidate type's clip search, the per-row choice and the type choice
// ------------------------------------------------------------------------------------------------
namespace {
inline size_t up256(size_t v) { return (v + 255u) & ~(size_t)255u; }
struct CalibLayout {
    size_t search, ratios, sse, score, sums, total;
};
inline CalibLayout calib_layout(size_t na, int ncand, int ntypes)
{
    CalibLayout L;
    L.search = 0;
    L.ratios = up256(antq_search_workspace_bytes());
    L.sse = L.ratios + up256((size_t)std::max(ncand, 1) * sizeof(float));
    L.score = L.sse + up256((size_t)ntypes * (size_t)std::max(ncand, 1) * na * sizeof(double));
    L.sums = L.score + up256((size_t)ntypes * na * sizeof(float));
    L.total = L.sums + up256(2 * na * sizeof(double));
    return L;
}
inline int calib_ncand(int lb, int ub, int step) { return (step > 0 && ub > lb) ? (ub - lb + step - 1) / step : 0; }
}  // namespace

extern "C" size_t antq_calibrate_workspace_bytes(size_t rows, int alpha_per_row, int lb, int ub, int step, int ntypes)
{
    if (ntypes < 1 || step < 1) return 0;
    return calib_layout(alpha_per_row ? rows : 1, calib_ncand(lb, ub, step), ntypes).total;
}

extern "C" int antq_calibrate(const void *x, size_t rows, size_t row_len, int alpha_per_row, int dtype, int xmax_mode,
                              float *xmax, int lb, int ub, int step, int ntypes, const float *gmax_host,
                              const void *const *plan_host, const void *const *plan_dev, unsigned flags, float *alpha,
                              float *score, int32_t *type, void *workspace, size_t workspace_bytes, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !xmax || !gmax_host || !plan_host || !plan_dev || !alpha || !score || !type || !workspace || ntypes < 1 || step < 1)
        return ANTQ_ERR_ARG;
    if (dtype != ANTQ_F32 && dtype != ANTQ_BF16 && dtype != ANTQ_F16) return ANTQ_ERR_UNSUPPORTED;
    for (int t = 0; t < ntypes; t++)
        if (!plan_host[t] || !plan_dev[t]) return ANTQ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return ANTQ_ERR_ALIGN;
    const size_t na = alpha_per_row ? rows : 1;
    const size_t n_per = alpha_per_row ? row_len : rows * row_len;
    const int ncand = calib_ncand(lb, ub, step);
    const CalibLayout L = calib_layout(na, ncand, ntypes);
    if (workspace_bytes < L.total) return ANTQ_ERR_ARG;
    char *w = static_cast<char *>(workspace);
    void *ws_search = w + L.search;
    float *ratios = reinterpret_cast<float *>(w + L.ratios);
    double *sse = reinterpret_cast<double *>(w + L.sse);
    float *best = reinterpret_cast<float *>(w + L.score);
    double *sums = reinterpret_cast<double *>(w + L.sums);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int pr = alpha_per_row ? 1 : 0;
    int rc = ANTQ_OK;
    // 1. the clip statistic: abs-max (ANT, AQ:289 / :308), mean +- 3 sigma (OliVe, OQ:193-197 / :213-218), or the caller's
    //    (the candidate ratios do not depend on the data: their kernel goes first and also zeroes the accumulator of a
    //     whole-tensor abs-max)
    //    A 16-bit tensor with ONE scale that the histogram search will take gets its abs-max from the counting pass.
    bool xmax_in_hist = false;
    if (na == 1 && xmax_mode == ANTQ_XMAX_ABSMAX && !(flags & ANTQ_FLAG_OVP) && ncand > 0 && g_knob_hist_xmax)
        xmax_in_hist = dtype == ANTQ_BF16 ? hist_eligible<bf16_tag>(n_per, false, x, 0)
                     : dtype == ANTQ_F16  ? hist_eligible<f16_tag>(n_per, false, x, 0) : false;
    HistXmax hx;                                   // (decided here, once; handed to the search explicitly)
    float *const zero = (xmax_mode == ANTQ_XMAX_ABSMAX && ncand > 0 && (!pr || xmax_in_hist)) ? xmax : nullptr;
    if (ncand > 0)
        hipLaunchKernelGGL(k_calib_ratios, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, ratios, lb, step, ncand, zero);
    if (xmax_in_hist) hx.out = xmax;
    else if (xmax_mode == ANTQ_XMAX_ABSMAX)
        rc = zero ? antq_absmax_into(x, xmax, rows * row_len, dtype, stream) : antq_absmax(x, xmax, rows, row_len, pr, dtype, stream);
    else if (xmax_mode == ANTQ_XMAX_3SIGMA) {
        rc = antq_moments(x, rows, row_len, pr, dtype, sums, ws_search, stream);
        if (rc == ANTQ_OK) rc = antq_xmax_3sigma(sums, na, n_per, dtype, xmax, stream);
    } else if (xmax_mode != ANTQ_XMAX_GIVEN) return ANTQ_ERR_ARG;
    if (rc != ANTQ_OK) return rc;
    const unsigned nab = (unsigned)std::min<size_t>((na + 255) / 256, 0x7fffffffu);
    if ((na + 255) / 256 > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    if (ncand == 0) {
        hipLaunchKernelGGL(k_calib_none, dim3(nab), dim3(256), 0, st, xmax, na, ntypes, best, alpha);
    } else {
        // 2. sum of squared errors of every (type, candidate) per row: kMaxTypes types per read of the tensor where the
        //    single-read kernel applies, one read per type otherwise
        const size_t per_type = (size_t)ncand * na;
        for (int b = 0; b < ntypes; b += kMaxTypes) {
            const int nt = std::min(kMaxTypes, ntypes - b);
            rc = nt > 1 ? search_sse_multi_impl(x, rows, row_len, xmax, pr, ratios, ncand, nt, gmax_host + b, plan_host + b,
                                                plan_dev + b, flags, dtype, sse + (size_t)b * per_type, ws_search, stream, &hx)
                        : ANTQ_ERR_UNSUPPORTED;
            if (rc == ANTQ_ERR_UNSUPPORTED) {
                for (int t = b; t < b + nt; t++) {
                    rc = search_sse_impl(x, rows, row_len, xmax, pr, ratios, ncand, gmax_host[t], plan_host[t], plan_dev[t], flags,
                                         dtype, sse + (size_t)t * per_type, ws_search, stream, &hx);
                    if (rc != ANTQ_OK) return rc;
                }
            }
            if (rc != ANTQ_OK) return rc;
        }
        if (hx.out && !hx.taken) return ANTQ_ERR_LAUNCH;  // (cannot happen: every search entry refuses before enqueueing anything)
        if (na == 1) {
            // 3 + 4 for a tensor with one scale: picks, scores and the type pick in one launch
            hipLaunchKernelGGL(k_calib_pick_one_scale, dim3(1), dim3(256), 0, st, sse, xmax, ratios, ncand, (double)n_per, ntypes, best,
                               alpha, score, type);
            return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
        }
        // 3. per row: the first strict minimum (AQ:299-306), every type in one launch (blockIdx.y)
        const size_t pblocks = (na + 3) / 4;
        if (pblocks > 0x7fffffffull || ntypes > 65535) return ANTQ_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_search_pick, dim3((unsigned)pblocks, (unsigned)ntypes), dim3(256), 0, st, sse, xmax, ratios, ncand, na,
                           (double)n_per, best, alpha, per_type);
    }
    // 4. per tensor: the type with the smallest sum of best MSEs (AQ:326, :413-415)
    hipLaunchKernelGGL(k_calib_type_score_pick, dim3(1), dim3(256), 0, st, best, na, ntypes, score, type);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// antq_calibrate_install: the state and the output of a calibrating call whose type pick stays on the device
// ------------------------------------------------------------------------------------------------
namespace antq {
struct SelPlans {
    PlanArgs pa[kMaxTypes];
    const uint4 *tab[kMaxTypes];
    float gmax[kMaxTypes];
    int ntypes;
};

static __global__ void __launch_bounds__(256)
k_calib_install(const int32_t *__restrict__ type, int ntypes, const float *__restrict__ alpha, const float *__restrict__ score,
                const float *__restrict__ grids, int grid_len, float *__restrict__ grid_out, const float *__restrict__ outl,
                int outl_len, float *__restrict__ outl_out, float *__restrict__ alpha_out, float *__restrict__ mse_out)
{
    int t = type[0];
    t = t < 0 ? 0 : (t >= ntypes ? ntypes - 1 : t);
    for (int i = (int)threadIdx.x; i < grid_len; i += 256) grid_out[i] = grids[(size_t)t * grid_len + i];
    if (outl && outl_out)
        for (int i = (int)threadIdx.x; i < outl_len; i += 256) outl_out[i] = outl[(size_t)t * outl_len + i];
    if (threadIdx.x == 0) {
        alpha_out[0] = alpha[t];
        if (mse_out) mse_out[0] = score[t];
    }
}

// _forward of a tensor with ONE scale through the codebook the device picked: the d-domain element path (table where the
// quotient lies inside it, literal scan elsewhere: quant_vec), the plan chosen by *type at run time.  Not a throughput
// kernel -- it runs once per quantiser, in its calibrating forward -- but one pass instead of ntypes passes and a gather.
template <typename T, bool OVP>
__global__ void __launch_bounds__(256)
k_fq_select(const uint4 *__restrict__ x, uint4 *__restrict__ out, size_t nv, const int32_t *__restrict__ type,
            const float *__restrict__ alpha, SelPlans sp)
{
    constexpr int EPL = IO<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    int t = type[0];
    t = t < 0 ? 0 : (t >= sp.ntypes ? sp.ntypes - 1 : t);
    PlanArgs pa = sp.pa[0];
    const uint4 *tab = sp.tab[0];
    float gmax = sp.gmax[0];
#pragma unroll
    for (int k = 1; k < kMaxTypes; k++)
        if (t == k) { pa = sp.pa[k]; tab = sp.tab[k]; gmax = sp.gmax[k]; }
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = tab[threadIdx.x];
    const PlanLds L = stage_plan(pa, tab, smem, tab0);
    __syncthreads();
    const Scale sc = make_scale(alpha[t], gmax);
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256u) {
        const uint4 v = ld_stream(x + i);
        float xf[EPL], of[EPL];
        int j[EPL];
        IO<T>::unpack(v, xf);
        quant_vec<EPL, OVP, false>(pa, L, sc, xf, of, j);
        st_stream(out + i, IO<T>::pack(of));
    }
}

template <typename T>
static int launch_install_forward(const void *x, void *out, size_t n, unsigned flags, const int32_t *type, const float *alpha,
                                  const SelPlans &sp, size_t lds, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const size_t nv = n / EPL;
    const unsigned blocks = (unsigned)std::min<size_t>((nv + 255) / 256, 256 * 16);
    if (flags & ANTQ_FLAG_OVP)
        hipLaunchKernelGGL((k_fq_select<T, true>), dim3(blocks), dim3(256), lds, st, static_cast<const uint4 *>(x), static_cast<uint4 *>(out), nv, type, alpha, sp);
    else
        hipLaunchKernelGGL((k_fq_select<T, false>), dim3(blocks), dim3(256), lds, st, static_cast<const uint4 *>(x), static_cast<uint4 *>(out), nv, type, alpha, sp);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq

extern "C" int antq_calibrate_install(const void *x, void *out, size_t n, int dtype, int ntypes, const float *gmax_host,
                                      const void *const *plan_host, const void *const *plan_dev, unsigned flags,
                                      const int32_t *type, const float *alpha, const float *score, const float *grids,
                                      int grid_len, float *grid_out, const float *outl, int outl_len, float *outl_out,
                                      float *alpha_out, float *mse_out, void *stream)
{
    if (!x || !out || !gmax_host || !plan_host || !plan_dev || !type || !alpha || !score || !grids || !grid_out || !alpha_out ||
        ntypes < 1 || grid_len < 1 || n == 0 || (flags & ~ANTQ_FLAG_OVP))
        return ANTQ_ERR_ARG;
    if (ntypes > kMaxTypes) return ANTQ_ERR_UNSUPPORTED;
    const int epl = dtype == ANTQ_F32 ? 4 : 8;
    if (dtype != ANTQ_F32 && dtype != ANTQ_BF16 && dtype != ANTQ_F16) return ANTQ_ERR_UNSUPPORTED;
    if (n % (size_t)epl != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return ANTQ_ERR_UNSUPPORTED;
    SelPlans sp;
    memset(&sp, 0, sizeof(sp));
    sp.ntypes = ntypes;
    size_t lds = 0;
    for (int t = 0; t < kMaxTypes; t++) {
        const int u = t < ntypes ? t : 0;
        if (!plan_host[u] || !plan_dev[u]) return ANTQ_ERR_ARG;
        if (!plan_args_from_host(plan_host[u], sp.pa[t])) return ANTQ_ERR_PLAN;
        sp.tab[t] = plan_tab_ptr(plan_dev[u]);
        sp.gmax[t] = gmax_host[u];
        lds = std::max(lds, (size_t)sp.pa[t].tab_units * 16);
    }
    if (lds > 64 * 1024) return ANTQ_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_calib_install, dim3(1), dim3(256), 0, st, type, ntypes, alpha, score, grids, grid_len, grid_out, outl, outl_len,
                       outl_out, alpha_out, mse_out);
    switch (dtype) {
    case ANTQ_F32: return launch_install_forward<float>(x, out, n, flags, type, alpha, sp, lds, st);
    case ANTQ_BF16: return launch_install_forward<bf16_tag>(x, out, n, flags, type, alpha, sp, lds, st);
    default: return launch_install_forward<f16_tag>(x, out, n, flags, type, alpha, sp, lds, st);
    }
}

extern "C" size_t antq_calibrate_batch_workspace_bytes(const antq_calib_job *jobs, int n)
{
    if (!jobs || n < 0) return 0;
    size_t need = 256;
    for (int i = 0; i < n; i++) {
        const antq_calib_job &J = jobs[i];
        const size_t b = antq_calibrate_workspace_bytes(J.rows, J.alpha_per_row, J.lb, J.ub, J.step, J.ntypes);
        if (b == 0) return 0;
        need = std::max(need, b);
    }
    return need;
}

extern "C" int antq_calibrate_batch(const antq_calib_job *jobs, int n, int dtype, unsigned flags, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    if (n == 0) return ANTQ_OK;
    if (!jobs || n < 0) return ANTQ_ERR_ARG;
    for (int i = 0; i < n; i++) {
        const antq_calib_job &J = jobs[i];
        const int rc = antq_calibrate(J.x_dev, J.rows, J.row_len, J.alpha_per_row, dtype, J.xmax_mode, J.xmax_dev, J.lb, J.ub,
                                      J.step, J.ntypes, J.gmax_host, J.plan_host, J.plan_dev, flags, J.alpha_dev, J.score_dev,
                                      J.type_dev, workspace, workspace_bytes, stream);
        if (rc != ANTQ_OK) return rc;
    }
    return ANTQ_OK;
}

namespace antq {
int prefetch_unit_search()        // antq_prefetch_kernels (antq_kernels.hip): load this unit's code object now
{
    hipFuncAttributes at;
    return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_search_pick)) == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq
