// antq_search.hip -- calibration entry points of libantq: antq_search_sse, antq_search_sse_multi, antq_search_pick
// (reference: search_mse AQ/quant_modules.py:287-326, search_adaptive_numeric_type :328-415; OQ:189-256).  gfx950 only.
#include "antq_host.h"
#include "antq_k_fakequant.h"
#include "antq_k_search.h"

namespace antq {

template <typename T, bool OVP>
static int launch_search(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                         const float *ratios, int ncand, float gmax, const PlanArgs &pa, const void *plan_host,
                         const void *plan_dev, double *sse, double *ws, hipStream_t st)
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
    if (!per_row) { row_len = rows * row_len; rows = 1; }
    const size_t vpr = row_len / EPL;
    if (vpr > 0xffffffffull) return ANTQ_ERR_UNSUPPORTED;
    constexpr int U = 4;
    const size_t tpr = (vpr + 64 * U - 1) / (64 * U);
    const size_t total = rows * tpr;
    if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
    const bool pt = rows == 1;
    // per tensor: tasks over all wavefronts; per row: one wavefront per row (it walks the row's tasks in order)
    size_t blocks = ((pt ? total : rows) + 3) / 4;
    const size_t cap = pt ? 256 * 4 : 256 * 8;
    if (blocks > cap) blocks = cap;
    // enough wavefronts to fill 256 CUs x 8 waves/SIMD: split the candidates when there are few rows
    int chunks = (int)std::min<size_t>((size_t)ncand, std::max<size_t>(1, (size_t)2048 / blocks));
    chunks = std::max(chunks, (ncand + kPtCand - 1) / kPtCand);
    const int cand_chunk = (ncand + chunks - 1) / chunks;
    chunks = (ncand + cand_chunk - 1) / cand_chunk;
    if (pt) {
        if (chunks > kWsSlots) return ANTQ_ERR_UNSUPPORTED;
        blocks = std::min(blocks, (size_t)(kWsSlots / chunks));
    }
    const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
    const bool xd = (g_knob_x != 0) && pa.kind == kPlanLut && ph->xdom && vpr >= kRowKernelMinVpr;
    const XArgs xa = xargs_from_plan(plan_host, pa);
    const dim3 gdim((unsigned)blocks, (unsigned)chunks), bdim(256);
    const uint4 *xv = static_cast<const uint4 *>(x);
#define ANTQ_LAUNCH_S(PT_, XD_)                                                                                    \
    hipLaunchKernelGGL((k_search_sse<T, OVP, U, PT_, XD_>), gdim, bdim, (XD_) ? 0 : lds, st, xv, (uint32_t)total,    \
                       (uint32_t)vpr, (uint32_t)tpr, rows, xmax, per_row, ratios, ncand, gmax, sse, ws, pa,          \
                       plan_tab_ptr(plan_dev), cand_chunk, xa)
    if (pt) { if (xd) ANTQ_LAUNCH_S(true, true); else ANTQ_LAUNCH_S(true, false); }
    else    { if (xd) ANTQ_LAUNCH_S(false, true); else ANTQ_LAUNCH_S(false, false); }
#undef ANTQ_LAUNCH_S
    if (pt) hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)ncand), dim3(256), 0, st, ws, (uint32_t)blocks, cand_chunk, sse);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

// all candidate types of a type selection on one read of the tensor; every plan must have the x-domain path
template <typename T, bool OVP>
static int launch_search_multi(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                               const float *ratios, int ncand, int ntypes, const float *gmax, const void *const *plan_host,
                               const void *const *plan_dev, double *sse, double *ws, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || (per_row ? row_len : rows * row_len) % EPL != 0) return ANTQ_ERR_UNSUPPORTED;
    if (!per_row) { row_len = rows * row_len; rows = 1; }
    const size_t vpr = row_len / EPL;
    if (vpr < kRowKernelMinVpr || vpr > 0xffffffffull) return ANTQ_ERR_UNSUPPORTED;
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
    constexpr int U = 4;
    const size_t tpr = (vpr + 64 * U - 1) / (64 * U);
    const size_t total = rows * tpr;
    if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
    const bool pt = rows == 1;
    size_t blocks = ((pt ? total : rows) + 3) / 4;
    const size_t cap = pt ? 256 * 4 : 256 * 8;
    if (blocks > cap) blocks = cap;
    // enough wavefronts to fill 256 CUs x 8 waves/SIMD: split the flattened (type, ratio) list when there are few rows
    const int nflat = ntypes * ncand;
    int chunks = (int)std::min<size_t>((size_t)nflat, std::max<size_t>(1, (size_t)2048 / blocks));
    chunks = std::max(chunks, (nflat + kPtCand - 1) / kPtCand);
    const int flat_chunk = (nflat + chunks - 1) / chunks;
    chunks = (nflat + flat_chunk - 1) / flat_chunk;
    if (pt) {
        if (chunks > kWsSlots) return ANTQ_ERR_UNSUPPORTED;
        blocks = std::min(blocks, (size_t)(kWsSlots / chunks));
    }
    const dim3 gdim((unsigned)blocks, (unsigned)chunks), bdim(256);
    const uint4 *xv = static_cast<const uint4 *>(x);
    if (pt) {
        hipLaunchKernelGGL((k_search_sse_multi<T, OVP, U, true>), gdim, bdim, 0, st, xv, (uint32_t)total, (uint32_t)vpr,
                           (uint32_t)tpr, rows, xmax, per_row, ratios, ncand, sse, ws, ma, flat_chunk);
        hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)nflat), dim3(256), 0, st, ws, (uint32_t)blocks, flat_chunk, sse);
    } else
        hipLaunchKernelGGL((k_search_sse_multi<T, OVP, U, false>), gdim, bdim, 0, st, xv, (uint32_t)total, (uint32_t)vpr,
                           (uint32_t)tpr, rows, xmax, per_row, ratios, ncand, sse, ws, ma, flat_chunk);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

}  // namespace antq

using namespace antq;

extern "C" int antq_search_sse_multi(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                                     const float *ratios, int ncand, int ntypes, const float *gmax_host,
                                     const void *const *plan_host, const void *const *plan_dev, unsigned flags, int dtype,
                                     double *sse, void *workspace, void *stream)
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
    (ovp ? launch_search_multi<TT, true>(x, rows, row_len, xmax, pr, ratios, ncand, ntypes, gmax_host, plan_host, plan_dev, sse, ws, st)   \
         : launch_search_multi<TT, false>(x, rows, row_len, xmax, pr, ratios, ncand, ntypes, gmax_host, plan_host, plan_dev, sse, ws, st))
    switch (dtype) {
    case ANTQ_F32: return ANTQ_SM(float);
    case ANTQ_BF16: return ANTQ_SM(bf16_tag);
    case ANTQ_F16: return ANTQ_SM(f16_tag);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
#undef ANTQ_SM
}

extern "C" int antq_search_sse(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                               const float *ratios, int ncand, float gmax, const void *plan_host, const void *plan_dev,
                               unsigned flags, int dtype, double *sse, void *workspace, void *stream)
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
        return ovp ? launch_search<float, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st)
                   : launch_search<float, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st);
    case ANTQ_BF16:
        return ovp ? launch_search<bf16_tag, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st)
                   : launch_search<bf16_tag, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st);
    case ANTQ_F16:
        return ovp ? launch_search<f16_tag, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st)
                   : launch_search<f16_tag, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_host, plan_dev, sse, ws, st);
    default:
        return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" size_t antq_search_workspace_bytes(void) { return (size_t)kWsSlots * kPtCand * sizeof(double); }

extern "C" int antq_search_pick(const double *sse, const float *xmax, const float *ratios, int ncand, size_t na,
                                size_t row_len, float *best_score, float *best_alpha, void *stream)
{
    if (na == 0) return ANTQ_OK;
    if (!sse || !xmax || !ratios || !best_score || !best_alpha || ncand < 0 || row_len == 0) return ANTQ_ERR_ARG;
    const size_t blocks = (na + 255) / 256;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(antq::k_search_pick, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), sse, xmax,
                       ratios, ncand, na, (double)row_len, best_score, best_alpha);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

