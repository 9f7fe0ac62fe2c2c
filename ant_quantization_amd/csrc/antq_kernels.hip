// antq_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ANT / OliVe
// fake-quant hot path + their C-ABI launchers (include/antq.h).
//
// Reference semantics being reproduced (bit-exact):
//   nearest-value scan   ant_quantization/quant/quant_kernel.cu:20-38
//   Quantizer._forward   ant_quantization/antquant/quant_modules.py:535-551
//   OliVe _forward + outlier-victim pairs  olive_quantization/antquant/quant_modules.py:294-330
//   AsymmetricQuantFunction  ant_quantization/antquant/quant_affine.py:95-115
//
// Design: the op is element-wise and HBM-bound, so the kernels are
// shaped by bytes, not flops: 16 B per lane per access (global_load_dwordx4), a
// wavefront owns a contiguous 1-4 KiB run of ONE quant group (row) so the scale is
// wave-uniform (SGPRs), the grid's decision table sits in LDS (one ds_read_b128 +
// one compare per element instead of the reference's M-step scan), the division
// x/scale is an exact 5-FMA sequence on a per-row reciprocal, and everything the
// reference does in 7-17 separate PyTorch kernels (div, scan, OVP mask ops, STE
// add, rescale) happens in registers between one load and one store.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (no fast-math: every
// float op below must round exactly as written).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/antq.h"
#include "antq_internal.h"

#include "antq_device.h"
#include "antq_k_fakequant.h"
#include "antq_k_nearest.h"
#include "antq_k_aux.h"
#include "antq_k_search.h"

namespace antq {

// tuning knobs (dev / bench only; see antq_debug_set).  THREAD-LOCAL: they change the dispatch of the calling thread's
// later calls only, so a probe that forgets to reset them cannot change which kernel another thread's calls run, and the
// library keeps no process-global mutable state.
static thread_local int g_knob_u = 0;        // force U of the uniform kernel (0 = heuristic)
static thread_local int g_knob_encwg = 2048;  // persistent workgroups of the 4-bit encoder (256 CUs x 8)
static thread_local int g_knob_x = 1;        // 0 disables the x-domain row kernel (A/B measurements)
static thread_local int g_knob_nearest_fast = 1;   // 0: antq_nearest always runs the literal scan
static thread_local int g_knob_lane_rows = 1;   // 0: rows of a power of two of vectors through the per-row table kernels (A/B)
static thread_local int g_knob_a = 1;        // 0 disables the approximate-quotient element path (quant_vec_a): exact division

// ------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------
static bool plan_args_from_host(const void *plan_host, PlanArgs &pa)
{
    const PlanHeader *h = static_cast<const PlanHeader *>(plan_host);
    if (h->magic != kPlanMagic || h->version != kPlanVersion) return false;
    if (h->m < 1 || h->m > ANTQ_MAX_GRID || h->m_pad != ((h->m + 3) & ~3u)) return false;
    pa.kind = h->kind;
    pa.m = h->m;
    pa.m_pad = h->m_pad;
    pa.shift = h->shift;
    pa.kmin = h->kmin;
    pa.kmax = h->kmax;
    pa.keymask = h->keymask;
    pa.nbneg = h->nbneg;
    pa.fastlim = h->fastlim;
    pa.n_entries = (h->kind == kPlanLut) ? h->n_entries : 0;
    pa.tab_units = pa.n_entries + (pa.m_pad >> 2);
    pa.linear = h->linear;
    pa.lin_scale = h->lin_scale;
    pa.lin_bias = h->lin_bias;
    pa.adom = (h->kind == kPlanLut && g_knob_a != 0) ? h->adom : 0u;
    pa.xlim = h->xlim;
    pa.atab_slots = h->atab_slots;
    return true;
}

static inline const uint4 *plan_tab_ptr(const void *plan_dev)
{
    return reinterpret_cast<const uint4 *>(static_cast<const char *>(plan_dev) + sizeof(PlanHeader));
}

// dynamic LDS of a kernel that stages the plan's table (stage_plan) or, for plans with adom, its converted image (stage_atab)
static inline size_t lds_table(const PlanArgs &pa, bool idx)
{
    const size_t plain = (size_t)pa.tab_units * 16;
    return pa.adom ? std::max(plain, (size_t)atab_units(pa.atab_slots, pa.m_pad, idx) * 16) : plain;
}

template <typename T, bool OVP, bool IDX, bool DYN>
static int launch_uniform(const void *x, void *out, int16_t *idx, size_t rows, size_t vpr, const float *alpha,
                          int per_row, float gmax, float ratio, float *alpha_out, const PlanArgs &pa,
                          const void *plan_host, const void *plan_dev, size_t lds, hipStream_t st)
{
    const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
    const uint4 *tab = plan_tab_ptr(plan_dev);
    const uint4 *xv = static_cast<const uint4 *>(x);
    uint4 *ov = static_cast<uint4 *>(out);
    // (dynamic rows of <= 256 vectors -- fp32: <= 128 -- run faster through the exact per-element decision of the d-domain
    //  kernel than through a per-row table that can only be built once the row's abs-max is known: see antq_batch_build)
    const bool small_dyn = DYN && pa.adom && vpr <= (IO<T>::EPL == 4 ? 128u : 256u) && g_knob_u != 1;
    const bool use_x = (g_knob_x != 0) && pa.kind == kPlanLut && ph->xdom && vpr >= kRowKernelMinVpr && (!DYN || vpr <= 8192) &&
                       !small_dyn;
    if (use_x) {
        // x-domain row kernel: 4 or 8 KiB of one row per wavefront (the per-row table is rebuilt per task)
        // 4 KiB of the row per wavefront measured best at steady clocks (79 % of 8 TB/s on 1 GiB); 2 or 3 KiB when that
        // leaves fewer idle lanes (rows of 128 vectors: 2; 144 / 288 / 576: 3)
        int U = (int)row_task_u((uint32_t)std::min<size_t>(vpr, 0x7fffffffu));
        if (DYN) U = vpr <= 128 ? 2 : vpr <= 192 ? 3 : (vpr <= 256 || (vpr > 512 && vpr <= 1024) || (vpr > 2048 && vpr <= 4096)) ? 4 : 8;
        if (g_knob_u) U = DYN ? U : g_knob_u;
        const bool wpr4 = DYN && vpr > 512 && vpr <= 2048;   // one row per workgroup: 4 wavefronts x U x 64 vectors
        const bool wpr16 = DYN && vpr > 2048;                // one row per 1024-thread workgroup: 16 wavefronts
        const size_t tpr = wpr16 ? 16 : wpr4 ? 4 : (vpr + (size_t)64 * U - 1) / ((size_t)64 * U);
        const size_t total = rows * tpr;
        if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
        XArgs xa;
        xa.m = pa.m; xa.shift = pa.shift; xa.kmin = pa.kmin; xa.kmax = pa.kmax; xa.keymask = pa.keymask;
        xa.nbneg = pa.nbneg; xa.n_entries = pa.n_entries; xa.xlim = ph->xlim; xa.vout = ph->vout;
        xa.linear = pa.linear; xa.lin_scale = pa.lin_scale; xa.lin_bias = pa.lin_bias;
        const uint4 *entries = tab + (pa.m_pad >> 2);
        const float *grid = reinterpret_cast<const float *>(tab);
        const dim3 grid_dim((unsigned)((total + 3) / 4)), block(256);
#define ANTQ_LAUNCH_X(UU)                                                                                           \
    hipLaunchKernelGGL((k_fq_xrow<T, OVP, IDX, UU, DYN>), grid_dim, block, 0, st, xv, ov, idx, (uint32_t)total,     \
                       (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out, xa, entries, grid)
        if (wpr16) {
            const dim3 g16((unsigned)rows), b16(1024);
            if (U == 8)
                hipLaunchKernelGGL((k_fq_xrow<T, OVP, IDX, 8, DYN, DYN ? 16 : 1>), g16, b16, 0, st, xv, ov, idx,
                                   (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out,
                                   xa, entries, grid);
            else
                hipLaunchKernelGGL((k_fq_xrow<T, OVP, IDX, 4, DYN, DYN ? 16 : 1>), g16, b16, 0, st, xv, ov, idx,
                                   (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out,
                                   xa, entries, grid);
            return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
        }
        if (wpr4) {
            if (U == 8)
                hipLaunchKernelGGL((k_fq_xrow<T, OVP, IDX, 8, DYN, DYN ? 4 : 1>), grid_dim, block, 0, st, xv, ov, idx,
                                   (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out,
                                   xa, entries, grid);
            else
                hipLaunchKernelGGL((k_fq_xrow<T, OVP, IDX, 4, DYN, DYN ? 4 : 1>), grid_dim, block, 0, st, xv, ov, idx,
                                   (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out,
                                   xa, entries, grid);
            return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
        }
        switch (U) {
        case 8: ANTQ_LAUNCH_X(8); break;
        case 4: ANTQ_LAUNCH_X(4); break;
        case 3: ANTQ_LAUNCH_X(3); break;
        case 2: ANTQ_LAUNCH_X(2); break;
        default: ANTQ_LAUNCH_X(1); break;
        }
#undef ANTQ_LAUNCH_X
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
    // U: 1 .. 4 KiB of one row per task, keeping lane utilisation high at the row tail
    int U = 4;
    if (DYN) {
        U = vpr <= 64 ? 1 : vpr <= 128 ? 2 : vpr <= 256 ? 4 : 8;
        if (vpr > 512) return ANTQ_ERR_UNSUPPORTED;  // caller falls back to absmax + static
    } else {
        double best = -1.0;
        for (int cand : {4, 2, 1}) {
            const size_t span = (size_t)64 * cand;
            const double util = (double)vpr / (double)(((vpr + span - 1) / span) * span);
            if (util > best + 0.05) { best = util; U = cand; }
        }
        if (g_knob_u) U = g_knob_u;
    }
    const size_t tpr = (vpr + (size_t)64 * U - 1) / ((size_t)64 * U);
    const size_t total = rows * tpr;
    if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
    size_t blocks = (total + 3) / 4;
    const bool loop = !DYN && lds > 3072;   // (int-8: 255 buckets = 5 KiB of table per 16 KiB of data)
    if (loop) {
        // staging a big table per 16 KiB of data would dominate: persistent workgroups instead
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>(8, (size_t)(144 * 1024) / lds));
        blocks = std::min(blocks, (size_t)256 * per_cu);
        const dim3 grid_l((unsigned)blocks), block_l(256);
        hipLaunchKernelGGL((k_fq_uniform<T, OVP, IDX, 4, false, true>), grid_l, block_l, lds, st, xv, ov, idx,
                           (uint32_t)((rows * ((vpr + 255) / 256))), (uint32_t)vpr, (uint32_t)((vpr + 255) / 256), alpha,
                           per_row, gmax, ratio, alpha_out, pa, tab);
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)blocks), block(256);
#define ANTQ_LAUNCH_U(UU)                                                                                          \
    hipLaunchKernelGGL((k_fq_uniform<T, OVP, IDX, UU, DYN>), grid, block, lds, st, xv, ov, idx, (uint32_t)total,  \
                       (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out, pa, tab)
    switch (U) {
    case 8: ANTQ_LAUNCH_U(8); break;
    case 4: ANTQ_LAUNCH_U(4); break;
    case 2: ANTQ_LAUNCH_U(2); break;
    default: ANTQ_LAUNCH_U(1); break;
    }
#undef ANTQ_LAUNCH_U
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T, bool OVP, bool IDX>
static int launch_fq(const void *x, void *out, int16_t *idx, size_t rows, size_t row_len,
                     const float *alpha, int per_row, float gmax, const PlanArgs &pa,
                     const void *plan_host, const void *plan_dev, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const size_t n = rows * row_len;
    const size_t lds = lds_table(pa, IDX);
    const uint4 *tab = plan_tab_ptr(plan_dev);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                         (!idx || reinterpret_cast<uintptr_t>(idx) % 16 == 0);
    if (!per_row) { rows = 1; row_len = n; }

    if (aligned && row_len % EPL == 0) {
        const size_t vpr = row_len / EPL;
        // Rows of a power of two of vectors (4096, 8192, ... elements) with an exact-decision plan: the lane kernel (alpha
        // index = a shift) instead of a table per row -- since the instruction diet of the element path it is ahead at
        // every tensor size: 33.5 MB bf16 59.3 -> 62.9 %, fp32 69.4 -> 74.1 %; 134 MB 74.0 -> 77.1 / 79.7 -> 81.9 %
        // (tools/probe_lane_rows.py; knob 5 = 0 restores the row kernel).  Other row lengths pay ~8 instructions per vector for
        // the row index (f64 reciprocal + fix-up) and still gain: fp32 1-3.5 points (4608 / 11008 / 28672 wide: 70.7 -> 71.8,
        // 68.6 -> 72.1, 69.0 -> 71.9 %), bf16 0-3.5 on three boxes (57.3 -> 60.0, 58.2 -> 59.9, 58.9 -> 60.4 % on the last;
        // 768-wide rows: equal) -- tools/probe_lane_rows_np2.py
        const bool lane_rows = pa.adom && g_knob_lane_rows != 0;
        if (vpr >= kRowKernelMinVpr && !lane_rows) {
            if (vpr > 0xffffffffull) return ANTQ_ERR_UNSUPPORTED;
            return launch_uniform<T, OVP, IDX, false>(x, out, idx, rows, vpr, alpha, per_row, gmax, 1.0f, nullptr, pa,
                                                      plan_host, plan_dev, lds, st);
        } else {
            const size_t n_vec = n / EPL;
            int vshift = -1;
            if ((vpr & (vpr - 1)) == 0) { vshift = 0; while (((size_t)1 << vshift) < vpr) vshift++; }
            constexpr int U = 2;
            const size_t blocks = (n_vec + 256 * U - 1) / (256 * U);
            if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
            if (pa.adom)
                hipLaunchKernelGGL((k_fq_lane<T, OVP, IDX, U, false, true>), dim3((unsigned)blocks), dim3(256), lds, st,
                                   static_cast<const uint4 *>(x), static_cast<uint4 *>(out), idx, n_vec, (uint32_t)vpr,
                                   vshift, alpha, per_row, gmax, 1.0f, (float *)nullptr, pa, tab);
            else
                hipLaunchKernelGGL((k_fq_lane<T, OVP, IDX, U, false, false>), dim3((unsigned)blocks), dim3(256), lds, st,
                                   static_cast<const uint4 *>(x), static_cast<uint4 *>(out), idx, n_vec, (uint32_t)vpr,
                                   vshift, alpha, per_row, gmax, 1.0f, (float *)nullptr, pa, tab);
        }
    } else if (aligned && !per_row && n >= (size_t)64 * EPL) {
        // per-tensor scale with a ragged tail: vector body + element tail
        const size_t n_body = (n / EPL) * EPL;
        int rc = launch_fq<T, OVP, IDX>(x, out, idx, 1, n_body, alpha, 0, gmax, pa, plan_host, plan_dev, st);
        if (rc != ANTQ_OK) return rc;
        const size_t n_tail = n - n_body;
        const size_t pairs = (n_tail + 1) / 2;
        hipLaunchKernelGGL((k_fq_scalar<T, OVP, IDX>), dim3((unsigned)((pairs + 255) / 256)), dim3(256), lds, st, x, out,
                           idx, n_body, n_tail, n, n, alpha, 0, gmax, pa, tab);
    } else {
        const size_t pairs = (n + 1) / 2;
        const size_t blocks = (pairs + 255) / 256;
        if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((k_fq_scalar<T, OVP, IDX>), dim3((unsigned)blocks), dim3(256), lds, st, x, out, idx,
                           (size_t)0, n, n, row_len, alpha, per_row, gmax, pa, tab);
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T>
static int launch_fq_flags(const void *x, void *out, int16_t *idx, size_t rows, size_t row_len,
                           const float *alpha, int per_row, float gmax, const PlanArgs &pa,
                           const void *plan_host, const void *plan_dev, unsigned flags, hipStream_t st)
{
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    if (ovp) {
        if (idx) return launch_fq<T, true, true>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, st);
        return launch_fq<T, true, false>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, st);
    }
    if (idx) return launch_fq<T, false, true>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, st);
    return launch_fq<T, false, false>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, st);
}

}  // namespace antq

// ======================================================================================
// C ABI
// ======================================================================================
using namespace antq;

extern "C" int antq_abi_version(void) { return ANTQ_ABI_VERSION; }

extern "C" const char *antq_strerror(int code)
{
    switch (code) {
    case ANTQ_OK: return "ok";
    case ANTQ_ERR_ARG: return "invalid argument";
    case ANTQ_ERR_UNSUPPORTED: return "unsupported dtype / size for this entry point";
    case ANTQ_ERR_PLAN: return "malformed or undersized plan blob";
    case ANTQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case ANTQ_ERR_ALIGN: return "pointer not aligned to the element size";
    default: return "unknown antq error";
    }
}

extern "C" int antq_nearest(const void *x, void *z, int16_t *idx, size_t n, const void *grid, int m, int dtype,
                            void *stream)
{
    if (n == 0) return ANTQ_OK;
    if (!x || !z || !grid || m < 1) return ANTQ_ERR_ARG;
    if (m > ANTQ_MAX_GRID) return ANTQ_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ANTQ_F32 || dtype == ANTQ_F64) {
        const size_t blocks = (n + 1023) / 1024;
        if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        if (dtype == ANTQ_F32) {
            if (m <= 256 && g_knob_nearest_fast)
                hipLaunchKernelGGL((k_nearest_fast<float>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, st,
                                   static_cast<const float *>(x), static_cast<float *>(z), idx, n,
                                   static_cast<const float *>(grid), m);
            else
                hipLaunchKernelGGL((k_nearest<float>), dim3((unsigned)blocks), dim3(256), 0, st,
                                   static_cast<const float *>(x), static_cast<float *>(z), idx, n,
                                   static_cast<const float *>(grid), m);
        } else {
            if (m <= 256 && g_knob_nearest_fast)
                hipLaunchKernelGGL((k_nearest_fast<double>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, st,
                                   static_cast<const double *>(x), static_cast<double *>(z), idx, n,
                                   static_cast<const double *>(grid), m);
            else
                hipLaunchKernelGGL((k_nearest<double>), dim3((unsigned)blocks), dim3(256), 0, st,
                                   static_cast<const double *>(x), static_cast<double *>(z), idx, n,
                                   static_cast<const double *>(grid), m);
        }
    } else if (dtype == ANTQ_BF16 || dtype == ANTQ_F16) {
        const size_t blocks = (n + 255) / 256;
        if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        if (dtype == ANTQ_BF16)
            hipLaunchKernelGGL((k_nearest16<bf16_tag>), dim3((unsigned)blocks), dim3(256), 0, st, x, z, idx, n,
                               static_cast<const float *>(grid), m);
        else
            hipLaunchKernelGGL((k_nearest16<f16_tag>), dim3((unsigned)blocks), dim3(256), 0, st, x, z, idx, n,
                               static_cast<const float *>(grid), m);
    } else {
        return ANTQ_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

// quant_cuda.quant for a grid the caller has a plan for (quantisers' static codebooks): table lookup instead of the scan.
// gcheck != nullptr: the hinted form (see k_nearest_plan).
static int launch_nearest_plan(const void *x, void *z, int16_t *idx, size_t n, const void *plan_host, const void *plan_dev,
                               const float *gcheck, int m_check, int *stale, int dtype, void *stream)
{
    if (n == 0) return ANTQ_OK;
    if (!x || !z || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    if (gcheck && (uint32_t)m_check != pa.m) return ANTQ_ERR_ARG;
    const int epl = dtype == ANTQ_F32 ? 4 : (dtype == ANTQ_BF16 || dtype == ANTQ_F16) ? 8 : 0;
    if (!epl) return ANTQ_ERR_UNSUPPORTED;
    if (n % epl || reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(z) % 16 ||
        (idx && reinterpret_cast<uintptr_t>(idx) % 16))
        return ANTQ_ERR_UNSUPPORTED;                     // ragged / unaligned: use antq_nearest (the literal scan)
    const size_t n_vec = n / epl;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)pa.tab_units * 16;
    const bool u2 = !gcheck && lds <= 2048;
    const size_t per_wg = u2 ? 512 : 1024;
    const size_t blocks = (n_vec + per_wg - 1) / per_wg;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    const uint4 *xv = static_cast<const uint4 *>(x);
    uint4 *zv = static_cast<uint4 *>(z);
    const uint4 *tab = plan_tab_ptr(plan_dev);
#define ANTQ_LAUNCH_N2(TT, II, HH, UU)                                                                             \
    hipLaunchKernelGGL((k_nearest_plan<TT, II, HH, UU>), dim3((unsigned)blocks), dim3(256), lds, st, xv, zv, idx, n_vec, pa, \
                       tab, gcheck, stale)
#define ANTQ_LAUNCH_N(TT)                                                                                          \
    do {                                                                                                           \
        if (gcheck) { if (idx) ANTQ_LAUNCH_N2(TT, true, true, 4); else ANTQ_LAUNCH_N2(TT, false, true, 4); }       \
        else if (u2) { if (idx) ANTQ_LAUNCH_N2(TT, true, false, 2); else ANTQ_LAUNCH_N2(TT, false, false, 2); }    \
        else { if (idx) ANTQ_LAUNCH_N2(TT, true, false, 4); else ANTQ_LAUNCH_N2(TT, false, false, 4); }            \
    } while (0)
    switch (dtype) {
    case ANTQ_F32: ANTQ_LAUNCH_N(float); break;
    case ANTQ_BF16: ANTQ_LAUNCH_N(bf16_tag); break;
    default: ANTQ_LAUNCH_N(f16_tag); break;
    }
#undef ANTQ_LAUNCH_N
#undef ANTQ_LAUNCH_N2
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

extern "C" int antq_nearest_plan(const void *x, void *z, int16_t *idx, size_t n, const void *plan_host, const void *plan_dev,
                                 int dtype, void *stream)
{
    return launch_nearest_plan(x, z, idx, n, plan_host, plan_dev, nullptr, 0, nullptr, dtype, stream);
}

extern "C" int antq_nearest_hinted(const void *x, void *z, int16_t *idx, size_t n, const float *grid_dev, int m,
                                   const void *plan_host, const void *plan_dev, int *stale, int dtype, void *stream)
{
    if (!grid_dev || m < 1) return ANTQ_ERR_ARG;
    return launch_nearest_plan(x, z, idx, n, plan_host, plan_dev, grid_dev, m, stale, dtype, stream);
}

extern "C" int antq_fakequant(const void *x, void *out, int16_t *idx, size_t rows, size_t row_len,
                              const float *alpha, int alpha_per_row, float gmax, const void *plan_host,
                              const void *plan_dev, unsigned flags, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !alpha || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_row = alpha_per_row ? 1 : 0;
    switch (dtype) {
    case ANTQ_F32:
        if (reinterpret_cast<uintptr_t>(x) % 4 || reinterpret_cast<uintptr_t>(out) % 4) return ANTQ_ERR_ALIGN;
        return launch_fq_flags<float>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, flags, st);
    case ANTQ_BF16:
        if (reinterpret_cast<uintptr_t>(x) % 2 || reinterpret_cast<uintptr_t>(out) % 2) return ANTQ_ERR_ALIGN;
        return launch_fq_flags<bf16_tag>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, flags, st);
    case ANTQ_F16:
        if (reinterpret_cast<uintptr_t>(x) % 2 || reinterpret_cast<uintptr_t>(out) % 2) return ANTQ_ERR_ALIGN;
        return launch_fq_flags<f16_tag>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, flags, st);
    default:
        return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_affine(const float *x, float *out, int32_t *q, size_t rows, size_t row_len, int k,
                           const float *xmin, const float *xmax, int per_row, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !xmin || !xmax || k < 1 || k > 24) return ANTQ_ERR_ARG;
    const size_t n = rows * row_len;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = (per_row ? row_len : n) % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(out) % 16 == 0 && (!q || reinterpret_cast<uintptr_t>(q) % 16 == 0);
    if (vec) {
        const size_t n_vec = n / 4, vpr = (per_row ? row_len : n) / 4;
        const size_t blocks = (n_vec + 256 * kAffineU - 1) / (256 * kAffineU);
        if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_affine_vec, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const uint4 *>(x),
                           reinterpret_cast<uint4 *>(out), reinterpret_cast<int4 *>(q), n_vec, vpr, k, xmin, xmax,
                           per_row ? 1 : 0);
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
    const size_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_affine, dim3((unsigned)blocks), dim3(256), 0, st, x, out, q, n,
                       row_len, k, xmin, xmax, per_row ? 1 : 0);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

extern "C" int antq_copy(const void *src, void *dst, size_t bytes, void *stream)
{
    if (bytes == 0) return ANTQ_OK;
    if (!src || !dst || bytes % 16 || reinterpret_cast<uintptr_t>(src) % 16 || reinterpret_cast<uintptr_t>(dst) % 16)
        return ANTQ_ERR_ARG;
    const size_t n_vec = bytes / 16;
    const size_t blocks = (n_vec + 1023) / 1024;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_copy, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n_vec);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

extern "C" int antq_debug_set(int key, int value)
{
    if (key == 0) g_knob_u = value;
    else if (key == 1) g_knob_encwg = value > 0 ? value : 2048;
    else if (key == 2) g_knob_x = value;
    else if (key == 3) g_knob_nearest_fast = value;
    else if (key == 4) g_knob_a = value;
    else if (key == 5) g_knob_lane_rows = value;
    else return ANTQ_ERR_ARG;
    return ANTQ_OK;
}

namespace antq {

template <typename T>
static int launch_absmax(const void *x, float *amax, size_t rows, size_t row_len, int per_row, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const bool al = reinterpret_cast<uintptr_t>(x) % 16 == 0;
    const int vec_ok = per_row ? (al && row_len % EPL == 0) : al;
    if (per_row && vec_ok) {
        const size_t vpr = row_len / EPL;
        if (vpr <= 64 && (vpr & (vpr - 1)) == 0) {       // small groups: several per wavefront
            int vshift = 0;
            while (((size_t)1 << vshift) < vpr) vshift++;
            const size_t n_vec = rows * vpr, gblocks = (n_vec + 1023) / 1024;
            if (gblocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
            hipLaunchKernelGGL((k_absmax_groups<T>), dim3((unsigned)gblocks), dim3(256), 0, st, static_cast<const uint4 *>(x), amax,
                               n_vec, (uint32_t)vpr, vshift);
            return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
        }
    }
    size_t waves = per_row ? rows : (rows * row_len + 64 * EPL * 4 - 1) / (64 * EPL * 4);
    size_t blocks = (waves + 3) / 4;
    // (per tensor: one workgroup per CU -- every workgroup ends with an atomicMax on ONE address, and a thousand of them arriving
    //  together cost more than the read: 23.6 -> 16.4 us for 67 MB with 256 instead of 1024 workgroups)
    if (blocks > (per_row ? 4096u : 256u)) blocks = per_row ? 4096 : 256;
    if (blocks < 1) blocks = 1;
    // the whole-tensor maximum is an atomicMax of workgroup maxima into a zero (stream-ordered, capturable)
    if (!per_row) hipLaunchKernelGGL(k_zero_f32, dim3(1), dim3(64), 0, st, amax);
    hipLaunchKernelGGL((k_absmax<T>), dim3((unsigned)blocks), dim3(256), 0, st, x, amax, rows, row_len, per_row, vec_ok);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T, bool OVP, bool IDX>
static int launch_dynamic(const void *x, void *out, int16_t *idx, float *alpha_out, size_t rows, size_t row_len,
                          float ratio, float gmax, const PlanArgs &pa, const void *plan_host, const void *plan_dev, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const size_t lds = lds_table(pa, IDX);
    const uint4 *tab = plan_tab_ptr(plan_dev);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                         (!idx || reinterpret_cast<uintptr_t>(idx) % 16 == 0);
    if (aligned && row_len % EPL == 0) {
        const size_t vpr = row_len / EPL;
        const bool pow2 = (vpr & (vpr - 1)) == 0;
        if ((vpr <= 64 && pow2) || (pa.adom && EPL == 8 && vpr == 128 && g_knob_u != 1)) {
            // several groups per wavefront (or one: 64 vectors): butterfly max over vpr adjacent lanes; 16-bit rows of
            // 128 vectors: the 2 wavefronts of a group exchange their maxima through LDS
            int vshift = 0;
            while (((size_t)1 << vshift) < vpr) vshift++;
            const size_t n_vec = rows * vpr;
            constexpr int U = 2;
            const size_t blocks = (n_vec + 256 * U - 1) / (256 * U);
            if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
            if (pa.adom)
                hipLaunchKernelGGL((k_fq_lane<T, OVP, IDX, U, true, true>), dim3((unsigned)blocks), dim3(256), lds, st,
                                   static_cast<const uint4 *>(x), static_cast<uint4 *>(out), idx, n_vec, (uint32_t)vpr,
                                   vshift, (const float *)nullptr, 1, gmax, ratio, alpha_out, pa, tab);
            else
                hipLaunchKernelGGL((k_fq_lane<T, OVP, IDX, U, true, false>), dim3((unsigned)blocks), dim3(256), lds, st,
                                   static_cast<const uint4 *>(x), static_cast<uint4 *>(out), idx, n_vec, (uint32_t)vpr,
                                   vshift, (const float *)nullptr, 1, gmax, ratio, alpha_out, pa, tab);
            return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
        }
        if (vpr <= 8192) {
            // one quant group (row) per wavefront (<= 512 vectors) or per workgroup (<= 2048: 4 wavefronts, <= 8192:
            // 16): the row lives in registers, single HBM read.  Plans without the x-domain table only have the
            // wavefront variant; longer rows fall through to the two-pass scheme.
            int rc = launch_uniform<T, OVP, IDX, true>(x, out, idx, rows, vpr, nullptr, 1, gmax, ratio, alpha_out, pa,
                                                       plan_host, plan_dev, lds, st);
            if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        }
    }
    // long or ragged rows: abs-max pass (read) + static pass (read + write)
    if (!alpha_out) return ANTQ_ERR_ARG;
    int rc = launch_absmax<T>(x, alpha_out, rows, row_len, 1, st);
    if (rc != ANTQ_OK) return rc;
    hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, alpha_out, rows, ratio);
    return launch_fq<T, OVP, IDX>(x, out, idx, rows, row_len, alpha_out, 1, gmax, pa, plan_host, plan_dev, st);
}

template <typename T>
static int launch_dynamic_flags(const void *x, void *out, int16_t *idx, float *alpha_out, size_t rows, size_t row_len,
                                float ratio, float gmax, const PlanArgs &pa, const void *plan_host, const void *plan_dev, unsigned flags,
                                hipStream_t st)
{
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    if (ovp) {
        if (idx) return launch_dynamic<T, true, true>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, st);
        return launch_dynamic<T, true, false>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, st);
    }
    if (idx) return launch_dynamic<T, false, true>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, st);
    return launch_dynamic<T, false, false>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, st);
}

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
    XArgs xa;
    xa.m = pa.m; xa.shift = pa.shift; xa.kmin = pa.kmin; xa.kmax = pa.kmax; xa.keymask = pa.keymask;
    xa.nbneg = pa.nbneg; xa.n_entries = pa.n_entries; xa.xlim = ph->xlim; xa.vout = ph->vout;
        xa.linear = pa.linear; xa.lin_scale = pa.lin_scale; xa.lin_bias = pa.lin_bias;
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
        XArgs &xa = ma.xa[t];
        xa.m = pa.m; xa.shift = pa.shift; xa.kmin = pa.kmin; xa.kmax = pa.kmax; xa.keymask = pa.keymask;
        xa.nbneg = pa.nbneg; xa.n_entries = pa.n_entries; xa.xlim = ph->xlim; xa.vout = ph->vout;
        xa.linear = pa.linear; xa.lin_scale = pa.lin_scale; xa.lin_bias = pa.lin_bias;
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

extern "C" int antq_fakequant_dynamic(const void *x, void *out, int16_t *idx, float *alpha_out, size_t rows,
                                      size_t row_len, float ratio, float gmax, const void *plan_host,
                                      const void *plan_dev, unsigned flags, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_dynamic_flags<float>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, flags, st);
    case ANTQ_BF16: return launch_dynamic_flags<bf16_tag>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, flags, st);
    case ANTQ_F16: return launch_dynamic_flags<f16_tag>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, flags, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

namespace antq {
template <typename T>
static int launch_alpha_grad(const void *x, const void *out, const void *gout, double *gsum, double *ws, size_t rows,
                             size_t row_len, int per_row, hipStream_t st)
{
    static_assert(kPartialStride == kPtCand, "k_sum_partials reads partials kPtCand doubles apart");
    constexpr int EPL = IO<T>::EPL;
    const bool al = (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gout)) % 16 == 0;
    const int vec_ok = per_row ? (al && row_len % EPL == 0) : al;
    size_t waves = per_row ? rows : (rows * row_len + 64 * EPL * 2 - 1) / (64 * EPL * 2);
    size_t blocks = (waves + 3) / 4;
    const size_t cap = per_row ? 4096 : 1024;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_alpha_grad<T>), dim3((unsigned)blocks), dim3(256), 0, st, x, out, gout, gsum, ws, rows, row_len,
                       per_row, vec_ok);
    if (!per_row) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, ws, (uint32_t)blocks, 1, gsum);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq

extern "C" int antq_alpha_grad(const void *x, const void *out, const void *gout, size_t rows, size_t row_len, int per_row,
                               double *gsum, void *workspace, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !gout || !gsum || (!per_row && !workspace)) return ANTQ_ERR_ARG;
    double *ws = static_cast<double *>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_alpha_grad<float>(x, out, gout, gsum, ws, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_BF16: return launch_alpha_grad<bf16_tag>(x, out, gout, gsum, ws, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_F16: return launch_alpha_grad<f16_tag>(x, out, gout, gsum, ws, rows, row_len, per_row ? 1 : 0, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_absmax(const void *x, float *amax, size_t rows, size_t row_len, int per_row, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !amax) return ANTQ_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_absmax<float>(x, amax, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_BF16: return launch_absmax<bf16_tag>(x, amax, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_F16: return launch_absmax<f16_tag>(x, amax, rows, row_len, per_row ? 1 : 0, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
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

// ======================================================================================
// Batched launch (antq_batch_build / antq_fakequant_batch)
// ======================================================================================
#include "antq_k_batch.h"

extern "C" size_t antq_batch_capacity(const antq_job *jobs, int n, int dtype)
{
    const int epl = epl_of(dtype);
    if (!jobs || n < 1 || !epl) return 0;
    size_t blocks = 0;
    // (the dynamic variant gives rows of 257..1024 vectors a workgroup each: never more than max(static, rows))
    // (x-domain rows may be cut into tasks of 2 or 3 vectors per lane instead of 4: at most twice the blocks)
    for (int i = 0; i < n; i++) blocks += std::max(2 * job_blocks(jobs[i], epl, nullptr) + 1, jobs[i].rows);
    return sizeof(BatchHeader) + sizeof(BatchDesc) * (size_t)n + 4 * blocks;
}

extern "C" int antq_batch_build(const antq_job *jobs, int n, int dtype, unsigned flags, void *blob, size_t cap)
{
    const int epl = epl_of(dtype);
    if (!jobs || !blob || n < 1 || n > 65535) return ANTQ_ERR_ARG;
    if (!epl) return ANTQ_ERR_UNSUPPORTED;
    const bool dyn = (flags & ANTQ_FLAG_DYNAMIC) != 0;
    char *p = static_cast<char *>(blob);
    BatchHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = kBatchMagic; h.n = (uint32_t)n; h.dtype = (uint32_t)dtype; h.flags = flags;
    h.map_offset = (uint32_t)(sizeof(BatchHeader) + sizeof(BatchDesc) * (size_t)n);
    if (cap < h.map_offset) return ANTQ_ERR_PLAN;
    BatchDesc *descs = reinterpret_cast<BatchDesc *>(p + sizeof(BatchHeader));
    uint32_t *map = reinterpret_cast<uint32_t *>(p + h.map_offset);
    std::vector<uint8_t> fam((size_t)n);
    std::vector<size_t> nblk((size_t)n);
    size_t fam_blocks[kBatchFamilies] = {0, 0, 0, 0, 0}, lds = 0;
    bool any_da = false;
    for (int i = 0; i < n; i++) {
        const antq_job &J = jobs[i];
        if (!J.x_dev || !J.out_dev || (!J.alpha_dev && !dyn) || !J.plan_host || !J.plan_dev) return ANTQ_ERR_ARG;
        const uintptr_t esz = (dtype == ANTQ_F32) ? 4 : 2;
        if (reinterpret_cast<uintptr_t>(J.x_dev) % esz || reinterpret_cast<uintptr_t>(J.out_dev) % esz) return ANTQ_ERR_ALIGN;
        BatchDesc d;
        memset(&d, 0, sizeof(d));
        size_t blocks = job_blocks(J, epl, &d);
        if (blocks == 0) return ANTQ_ERR_UNSUPPORTED;
        if (!plan_args_from_host(J.plan_host, d.pa)) return ANTQ_ERR_PLAN;
        const PlanHeader *ph = static_cast<const PlanHeader *>(J.plan_host);
        const bool xdom = g_knob_x && d.pa.kind == kPlanLut && ph->xdom;
        d.vout = ph->vout;
        d.ratio = 1.0f;
        int f;
        d.u = (uint32_t)kBatchU;
        if (!dyn) {
            if (d.kind == 0 && d.pa.adom && (dtype == ANTQ_F32 || g_knob_lane_rows == 2) && g_knob_lane_rows != 0) {
                // fp32 long rows as lane jobs (alpha index = a shift, or the f64-reciprocal quotient): 16 x 4096^2 78.9 -> 81.0 %,
                // with OliVe's pairs 79.4 -> 81.0 %, BERT-base's 768 / 3072-wide rows 79.0 -> 80.8 %, ResNet-50 75.2 -> 76.0 %
                // against the per-row table kernel.  16-bit rows stay on the table kernel: equal without the pair rule (80.3 vs
                // 80.8, 81.1 vs 81.2 %), 0.6-1.3 points ahead with it, 0.7 ahead on BERT's shapes (tools/probe_batch_lane.py;
                // knob 5 = 0 restores the table kernel for fp32 too, 2 makes every long row a lane job)
                d.kind = 1; d.total_tasks = 0; d.tpr = 1; d.vshift = -1;
                if ((d.vpr & (d.vpr - 1u)) == 0u) { d.vshift = 0; while ((1u << d.vshift) < d.vpr) d.vshift++; }
                blocks = (size_t)((d.n_vec + 256u * kBatchU - 1u) / (256u * kBatchU));
            }
            if (d.kind == 0 && xdom) {
                // x-domain rows: the task size that leaves the fewest idle lanes for this row length
                d.kind = 2;
                d.u = row_task_u(d.vpr);
                d.tpr = (d.vpr + 64u * d.u - 1u) / (64u * d.u);
                const size_t total = (J.alpha_per_row ? J.rows : (size_t)1) * (size_t)d.tpr;   // per tensor: ONE row
                if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
                d.total_tasks = (uint32_t)total;
                blocks = (total + 3) / 4;
            }
            // (groups of 16 / 32 / 64 vectors: a per-group x-domain table was round 1's answer for bf16 group-128 ... 512; the
            //  lane kernel with the exact per-element decision matches it for 16-bit data -- 75.2-76.7 vs 76.5-77.5 % -- and
            //  beats it for fp32 with 2-vector tasks -- 79.7-80.4 vs 77-79 %: profiles/r02_lane_task_ab.log -- so it is gone)
            f = d.kind == 2 ? 0 : (d.kind == 3 ? -1 : (d.pa.adom ? 1 : 2));
        } else {
            // alpha computed in the kernel: the group / row has to live in the registers of a few lanes, one wavefront
            // or one workgroup
            if (!J.alpha_per_row || d.kind == 3 || J.rows > 0x3ffffff0ull) return ANTQ_ERR_UNSUPPORTED;
            if (d.kind == 0 && d.pa.adom && dtype != ANTQ_F32 && d.vpr == 128u && g_knob_u != 1) {
                // 16-bit rows of 128 vectors as lane jobs whose groups span 2 wavefronts of a workgroup (LDS exchange of
                // the wavefront maxima): 4 vectors in flight per lane instead of a wavefront per row: 71 -> 75 %; at 256
                // vectors the wavefront-per-row kernel stays ahead (79 vs 75 %)
                d.kind = 1; d.total_tasks = 0; d.tpr = 1; d.vshift = 7;
                blocks = (size_t)((d.n_vec + 256u * kBatchU - 1u) / (256u * kBatchU));
            }
            if (d.kind == 1) {
                if (d.vshift < 0 || d.vpr > 256u) return ANTQ_ERR_UNSUPPORTED;  // butterfly over a power-of-two group
                f = d.pa.adom ? 1 : 2;       // (per-group tables with the abs-max in front measured slower: 66 vs 71 %)
            } else if (xdom && !(d.pa.adom && d.vpr <= (dtype == ANTQ_F32 ? 128u : 256u) && g_knob_u != 1)) {
                // (rows of <= 256 vectors: bf16 / f16 measured faster through the exact per-element decision below -- 70 vs
                //  61 % at 128 vectors, 79 vs 72 % at 256 -- fp32 only at 128; profiles/r02_group_sweep.log)
                if (d.vpr > 8192u) return ANTQ_ERR_UNSUPPORTED;
                // one wavefront per row up to 512 vectors (4 or 8 per lane), one workgroup per row beyond: 4 wavefronts up
                // to 2048 vectors, 16 (a 1024-thread workgroup) up to 8192
                f = 3;
                if (d.vpr <= 256u) {
                    d.kind = 4; d.tpr = 1; d.total_tasks = (uint32_t)J.rows; blocks = (J.rows + 3) / 4;
                    d.u = d.vpr <= 128u ? 2u : d.vpr <= 192u ? 3u : 4u;
                }
                else if (d.vpr <= 512u && dtype == ANTQ_F32 && g_knob_u != 8) {
                    // fp32 rows of 257..512 vectors over the 4 wavefronts of a workgroup (80 vs 78 %); 16-bit rows of that many
                    // vectors are twice the elements and stay in one wavefront (74 vs 63 %)
                    d.kind = 12; d.tpr = 4; d.total_tasks = (uint32_t)(J.rows * 4); blocks = J.rows;
                }
                else if (d.vpr <= 512u) { d.kind = 6; d.tpr = 1; d.total_tasks = (uint32_t)J.rows; blocks = (J.rows + 3) / 4; }
                else if (d.vpr <= 2048u) { d.kind = d.vpr <= 1024u ? 5 : 7; d.tpr = 4; d.total_tasks = (uint32_t)(J.rows * 4); blocks = J.rows; }
                else { d.kind = d.vpr <= 4096u ? 9 : 10; d.tpr = 16; d.total_tasks = (uint32_t)(J.rows * 16); blocks = J.rows; f = 4; }
            } else {
                if (d.vpr > 64u * kBatchU) return ANTQ_ERR_UNSUPPORTED;         // the row in one wavefront's registers
                d.tpr = 1; d.total_tasks = (uint32_t)J.rows; blocks = (J.rows + 3) / 4;
                f = d.pa.adom ? 1 : 2;
            }
        }
        any_da = any_da || f == 1;
        d.x = static_cast<const uint4 *>(J.x_dev);
        d.out = static_cast<uint4 *>(J.out_dev);
        d.alpha = J.alpha_dev;
        d.plan_tab = plan_tab_ptr(J.plan_dev);
        d.per_row = J.alpha_per_row ? 1 : 0;
        d.gmax = J.gmax;
        descs[i] = d;
        fam[(size_t)i] = (uint8_t)(f < 0 ? 255 : f);
        nblk[(size_t)i] = blocks;
        if (f == 1 || f == 2 || f < 0) lds = std::max(lds, lds_table(d.pa, false));
    }
    // element-granular jobs (exact arithmetic, no table path) ride along with whichever d-domain launch exists
    for (int i = 0; i < n; i++)
        if (fam[(size_t)i] == 255) fam[(size_t)i] = any_da ? 1 : 2;
    {
        // Lane jobs (kind 1, adom) take 2 instead of 4 vectors per lane when the batch is only a few rounds of workgroups
        // (256 CUs x 8 workgroups = 2048 per round): smaller workgroups shorten the ramp and the tail of a short pass
        // (ResNet-50 group-16, 3 rounds: 72 -> ~75 %).  Knob 0: 2 forces it, 4 forbids it (A/B).
        size_t all_blocks = 0;
        for (int i = 0; i < n; i++) all_blocks += nblk[(size_t)i];
        // fp32 lane jobs always: 79.7-80.4 % with 2 against 75.7-76.3 % with 4 vectors per lane on 16 x 4096^2
        const bool small = g_knob_u == 2 || (g_knob_u != 4 && (all_blocks < 4u * 2048u || dtype == ANTQ_F32));
        for (int i = 0; i < n && small; i++) {
            BatchDesc &d = descs[i];
            if (d.kind == 1 && d.pa.adom && !(dyn && d.vpr > 64u)) {    // (groups of 2 wavefronts keep 4 vectors per lane)
                d.u = 2u;
                nblk[(size_t)i] = (size_t)((d.n_vec + 511u) / 512u);
            }
        }
    }
    if (!dyn) {
        // jobs of more than one static family: ONE launch of the all-in-one kernel instead of a launch per family
        bool seen[kBatchFamilies] = {false, false, false, false, false};
        int nf = 0;
        for (int i = 0; i < n; i++)
            if (!seen[fam[(size_t)i]]) { seen[fam[(size_t)i]] = true; nf++; }
        if (nf > 1) {
            h.pad = 1u;
            for (int i = 0; i < n; i++) fam[(size_t)i] = 0;
        }
    }
    size_t total_blocks = 0;
    for (int i = 0; i < n; i++) {
        descs[i].first_block = (uint32_t)fam_blocks[fam[(size_t)i]];
        fam_blocks[fam[(size_t)i]] += nblk[(size_t)i];
        total_blocks += nblk[(size_t)i];
    }
    if (total_blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    if (cap < h.map_offset + 4 * total_blocks) return ANTQ_ERR_PLAN;
    size_t off[kBatchFamilies], acc = 0;
    for (int f = 0; f < kBatchFamilies; f++) { off[f] = acc; acc += fam_blocks[f]; h.fam_blocks[f] = (uint32_t)fam_blocks[f]; }
    for (int i = 0; i < n; i++) {
        uint32_t *m = map + off[fam[(size_t)i]] + descs[i].first_block;
        for (size_t b = 0; b < nblk[(size_t)i]; b++) m[b] = (uint32_t)i;
    }
    h.total_blocks = (uint32_t)total_blocks;
    h.lds_bytes = (uint32_t)lds;
    h.bytes = (uint32_t)(h.map_offset + 4 * total_blocks);
    memcpy(p, &h, sizeof(h));
    return (int)h.bytes;
}

extern "C" int antq_fakequant_batch(const void *batch_host, const void *batch_dev, void *stream)
{
    if (!batch_host || !batch_dev) return ANTQ_ERR_ARG;
    const BatchHeader *h = static_cast<const BatchHeader *>(batch_host);
    if (h->magic != kBatchMagic) return ANTQ_ERR_PLAN;
    if (h->total_blocks == 0) return ANTQ_OK;
    const char *pd = static_cast<const char *>(batch_dev);
    const BatchDesc *descs = reinterpret_cast<const BatchDesc *>(pd + sizeof(BatchHeader));
    const uint32_t *map = reinterpret_cast<const uint32_t *>(pd + h->map_offset);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 block(256);
    const bool ovp = (h->flags & ANTQ_FLAG_OVP) != 0;
    const bool dyn = (h->flags & ANTQ_FLAG_DYNAMIC) != 0;
    const uint32_t *fmap[kBatchFamilies];
    {
        const uint32_t *m = map;
        for (int f = 0; f < kBatchFamilies; f++) { fmap[f] = m; m += h->fam_blocks[f]; }
    }
#define ANTQ_LAUNCH_D(TT, OO, AA)                                                                                   \
    do {                                                                                                            \
        const int f_ = (AA) ? 1 : 2;                                                                                \
        if (dyn) hipLaunchKernelGGL((k_fq_batch_d<TT, OO, AA, true>), dim3(h->fam_blocks[f_]), block, h->lds_bytes, st, descs, fmap[f_]);  \
        else hipLaunchKernelGGL((k_fq_batch_d<TT, OO, AA, false>), dim3(h->fam_blocks[f_]), block, h->lds_bytes, st, descs, fmap[f_]);     \
    } while (0)
#define ANTQ_LAUNCH_B(TT)                                                                                         \
    do {                                                                                                          \
        if (h->pad) {      /* mixed static batch: the all-in-one kernel */                                       \
            if (ovp) hipLaunchKernelGGL((k_fq_batch_all<TT, true>), dim3(h->fam_blocks[0]), block, h->lds_bytes, st, descs, fmap[0]);  \
            else hipLaunchKernelGGL((k_fq_batch_all<TT, false>), dim3(h->fam_blocks[0]), block, h->lds_bytes, st, descs, fmap[0]);     \
            break;                                                                                                \
        }                                                                                                         \
        if (h->fam_blocks[0]) {                                                                                   \
            if (ovp) hipLaunchKernelGGL((k_fq_batch<TT, true>), dim3(h->fam_blocks[0]), block, 0, st, descs, fmap[0]);   \
            else hipLaunchKernelGGL((k_fq_batch<TT, false>), dim3(h->fam_blocks[0]), block, 0, st, descs, fmap[0]);      \
        }                                                                                                         \
        if (h->fam_blocks[1]) { if (ovp) ANTQ_LAUNCH_D(TT, true, true); else ANTQ_LAUNCH_D(TT, false, true); }    \
        if (h->fam_blocks[2]) { if (ovp) ANTQ_LAUNCH_D(TT, true, false); else ANTQ_LAUNCH_D(TT, false, false); }  \
        if (h->fam_blocks[3]) {                                                                                   \
            if (ovp) hipLaunchKernelGGL((k_fq_batch_dyn<TT, true>), dim3(h->fam_blocks[3]), block, 0, st, descs, fmap[3]);   \
            else hipLaunchKernelGGL((k_fq_batch_dyn<TT, false>), dim3(h->fam_blocks[3]), block, 0, st, descs, fmap[3]);      \
        }                                                                                                         \
        if (h->fam_blocks[4]) {                                                                                   \
            if (ovp) hipLaunchKernelGGL((k_fq_batch_dyn16<TT, true>), dim3(h->fam_blocks[4]), dim3(1024), 0, st, descs, fmap[4]);   \
            else hipLaunchKernelGGL((k_fq_batch_dyn16<TT, false>), dim3(h->fam_blocks[4]), dim3(1024), 0, st, descs, fmap[4]);      \
        }                                                                                                         \
    } while (0)
    switch (h->dtype) {
    case ANTQ_F32: ANTQ_LAUNCH_B(float); break;
    case ANTQ_BF16: ANTQ_LAUNCH_B(bf16_tag); break;
    case ANTQ_F16: ANTQ_LAUNCH_B(f16_tag); break;
    default: return ANTQ_ERR_UNSUPPORTED;
    }
#undef ANTQ_LAUNCH_B
#undef ANTQ_LAUNCH_D
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

// ======================================================================================
// Packed 4-bit codec (antq_encode4 / antq_decode4)
// ======================================================================================
#include "antq_k_codec.h"

extern "C" int antq_encode4(const void *x, uint8_t *codes, size_t rows, size_t row_len, const float *alpha, int per_row,
                            float gmax, const void *plan_host, const void *plan_dev, int n_normal, unsigned flags,
                            int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !codes || !alpha || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(codes) % 4) return ANTQ_ERR_ALIGN;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_codec<float>(true, x, codes, nullptr, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    case ANTQ_BF16: return launch_codec<bf16_tag>(true, x, codes, nullptr, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    case ANTQ_F16: return launch_codec<f16_tag>(true, x, codes, nullptr, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_decode4(const uint8_t *codes, void *out, size_t rows, size_t row_len, const float *alpha, int per_row,
                            float gmax, const void *plan_host, const void *plan_dev, int n_normal, unsigned flags,
                            int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!codes || !out || !alpha || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(codes) % 4) return ANTQ_ERR_ALIGN;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_codec<float>(false, nullptr, out, codes, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    case ANTQ_BF16: return launch_codec<bf16_tag>(false, nullptr, out, codes, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    case ANTQ_F16: return launch_codec<f16_tag>(false, nullptr, out, codes, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}



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
