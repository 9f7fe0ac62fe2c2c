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
// This translation unit: antq_fakequant / antq_fakequant_dynamic (one tensor per launch).
#include "antq_host.h"
#include "antq_k_fakequant.h"
#include "antq_k_hrow.h"
#include "antq_k_aux.h"

#include <hip/hip_ext.h>
#include <type_traits>

namespace antq {

// ANTQ_FLAG_UNORDERED of the call being dispatched (set by the entry point, read by the launch helpers of this file)
static thread_local bool t_unordered = false;

// One launch.  `unordered`: the dispatch packet goes out without the barrier bit (hipExtAnyOrderLaunch), so the kernel may
// start while the launches queued before it on the same stream are still draining -- the caller has promised that it
// does not depend on them (weights at rest).  Later ordinary launches still wait for it.
template <typename... KArgs, typename... Args>
static inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args)
{
    if (t_unordered) hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, static_cast<KArgs>(args)...);
    else hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, static_cast<KArgs>(args)...);
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
        XArgs xa = xargs_from_plan(plan_host, pa);
        xa.inv_gmax = 1.0 / (double)gmax;
        const uint4 *entries = tab + (pa.m_pad >> 2);
        const float *grid = reinterpret_cast<const float *>(tab);
        const dim3 grid_dim((unsigned)((total + 3) / 4)), block(256);
#define ANTQ_LAUNCH_X(UU)                                                                                           \
    launch_k(k_fq_xrow<T, OVP, IDX, UU, DYN>, grid_dim, block, 0, st, xv, ov, idx, (uint32_t)total,                 \
             (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out, xa, entries, grid)
        // static rows, no index output: ONE wavefront per workgroup when the launch is unordered (or knob 6 = 1) -- the
        // tables are wave-private, there is no workgroup barrier, and wavefronts that start and retire one by one keep
        // the memory system busy across the launch boundary (tools/exp_lane.hip: 73.7 -> 74.7 % unordered, U = 4)
        if constexpr (!DYN && !IDX) {
            const int wpb = g_knob_waves ? g_knob_waves : (t_unordered ? 1 : 4);
            if (wpb == 1 && U >= 2 && U <= 4) {
                const dim3 g1((unsigned)total), b1(64);
                if (total > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
#define ANTQ_LAUNCH_X1(UU)                                                                                          \
    launch_k(k_fq_xrow<T, OVP, false, UU, false, 1, 1>, g1, b1, 0, st, xv, ov, idx, (uint32_t)total,                \
             (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out, xa, entries, grid)
                if (U == 4) ANTQ_LAUNCH_X1(4); else if (U == 3) ANTQ_LAUNCH_X1(3); else ANTQ_LAUNCH_X1(2);
#undef ANTQ_LAUNCH_X1
                return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
            }
        }
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

// 16-bit rows of >= 128 vectors in their own 16-bit domain (antq_k_hrow.h): one wavefront per workgroup, up to 4 KiB of
// one row per wavefront.  Returns ANTQ_ERR_UNSUPPORTED when the plan / dtype / shape has no such path.
template <typename T, bool OVP>
static int launch_hrow(const void *x, void *out, size_t rows, size_t vpr, const float *alpha, int per_row, float gmax,
                       const void *plan_host, const void *plan_dev, hipStream_t st)
{
    if constexpr (std::is_same<T, float>::value) {
        return ANTQ_ERR_UNSUPPORTED;
    } else {
        HArgs ha;
        if (g_knob_h == 0 || vpr < kRowKernelMinVpr || vpr > 0xffffffffull || !hargs_from_plan(plan_host, IO<T>::DTYPE, gmax, ha))
            return ANTQ_ERR_UNSUPPORTED;
        const PlanHeader *php = static_cast<const PlanHeader *>(plan_host);
        int U = g_knob_h == 2 ? (int)row_task_u((uint32_t)vpr) : (int)hrow_static_u((uint32_t)vpr, g_knob_x != 0 && php->xdom != 0u);
        if (U == 0) return ANTQ_ERR_UNSUPPORTED;          // (short / awkward rows: the fp32-domain row table, see hrow_static_u)
        // An ORDERED launch of 1024 ... 4096 wavefronts when a wavefront takes 8 vectors per lane -- one 4096 x 4096 bf16 tensor:
        // 4096 wavefronts, each a whole row, its table built once -- starts and drains as one front: 13.3 -> 12.9 us (63.3 ->
        // 65.1 %), 1024 x 4096: 5.03 -> 4.84 us.  Fewer wavefronts than that leave CUs idle (256 rows: 3.6 -> 4.2 us), more
        // (8192 x 4096: 23.3 -> 23.7 us) and every unordered launch (77 vs 74.4 %) do better with 4
        // (tools/probe_per_tensor.py, profiles/r04_per_tensor_shapes.log)
        if (U == 4 && !t_unordered && vpr % 512u == 0u && rows * (vpr / 512u) >= 1024u && rows * (vpr / 512u) <= 4096u) U = 8;
        if ((g_knob_u >= 2 && g_knob_u <= 4) || g_knob_u == 8) U = g_knob_u;      // knob 0 (A/B)
        const int W = (g_knob_waves == 4 || g_knob_waves == 1) ? g_knob_waves : 1;
        if (W == 4 && U == 8) U = 4;               // (the 4-wavefront form exists for 2 / 3 / 4 vectors per lane: the task size below must be the launched kernel's)
        const size_t tpr = (vpr + (size_t)64 * U - 1) / ((size_t)64 * U);
        const size_t total = rows * tpr;
        if (total > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        // The occupancy cap (kHRowLdsPad: 24 workgroups per CU) pays once a launch is many rounds of workgroups; one
        // 33.5 MB tensor is a single round and wants every slot (tools/probe_per_tensor.py)
        const bool big = total >= (size_t)4 * 8192;
        const unsigned pad = g_knob_hlds >= 0 ? (unsigned)g_knob_hlds : (big ? kHRowLdsPad : 0u);
        const dim3 g((unsigned)((total + W - 1) / W)), b(64 * W);
        const uint4 *tl = plan_tlist_dev(plan_host, plan_dev);
        const float *grid = reinterpret_cast<const float *>(plan_tab_ptr(plan_dev));
#define ANTQ_LAUNCH_H(UU, WW)                                                                                      \
    launch_k(k_fq_hrow<T, OVP, UU, WW>, g, b, (WW) == 1 ? pad : 0u, st, static_cast<const uint4 *>(x), static_cast<uint4 *>(out), (uint32_t)total,  \
             (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ha, tl, grid)
        if (W == 4) { if (U == 4) ANTQ_LAUNCH_H(4, 4); else if (U == 3) ANTQ_LAUNCH_H(3, 4); else ANTQ_LAUNCH_H(2, 4); }
        else { if (U == 8) ANTQ_LAUNCH_H(8, 1); else if (U == 4) ANTQ_LAUNCH_H(4, 1); else if (U == 3) ANTQ_LAUNCH_H(3, 1); else ANTQ_LAUNCH_H(2, 1); }
#undef ANTQ_LAUNCH_H
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
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
        // Unordered launches overlap their neighbours, i.e. run in something like the batched kernels' steady state, where
        // the per-row table kernel's leaner element loop wins (tools/exp_lane.hip: 74.7 % against 71.1 %): they take it
        // whenever the plan has the x-domain form (knob 5 = 2 keeps the lane kernel)
        if constexpr (!IDX) {
            const int rc = launch_hrow<T, OVP>(x, out, rows, vpr, alpha, per_row, gmax, plan_host, plan_dev, st);
            if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        }
        const PlanHeader *ph_ = static_cast<const PlanHeader *>(plan_host);
        const bool rows_unordered = t_unordered && !IDX && g_knob_lane_rows == 1 && g_knob_x != 0 && pa.kind == kPlanLut &&
                                    ph_->xdom && vpr >= kRowKernelMinVpr;
        const bool lane_rows = pa.adom && g_knob_lane_rows != 0 && !rows_unordered;
        if (vpr >= kRowKernelMinVpr && !lane_rows) {
            if (vpr > 0xffffffffull) return ANTQ_ERR_UNSUPPORTED;
            return launch_uniform<T, OVP, IDX, false>(x, out, idx, rows, vpr, alpha, per_row, gmax, 1.0f, nullptr, pa,
                                                      plan_host, plan_dev, lds, st);
        } else {
            const size_t n_vec = n / EPL;
            int vshift = -1;
            if ((vpr & (vpr - 1)) == 0) { vshift = 0; while (((size_t)1 << vshift) < vpr) vshift++; }
            // Launch shape of the exact-decision lane kernel (tools/exp_lane.hip, same-box A/B on 16 x 4096^2 bf16): an
            // ordinary launch 4 wavefronts per workgroup, 2 vectors per lane (62.5 %; one-wavefront workgroups re-stage the
            // table four times as often: 59-61 %); an unordered one 1 wavefront per workgroup, 4 vectors per lane (71.1 %
            // against 69.7 %).  Knobs 6 / 7 force wavefronts per workgroup / vectors per lane (A/B).
            int W = IDX ? 4 : (g_knob_waves ? g_knob_waves : (t_unordered ? 1 : 4));
            int U = IDX ? 2 : (g_knob_lane_u ? g_knob_lane_u : (t_unordered ? 4 : 2));
            if (!pa.adom) { W = 4; U = 2; }
            if (W != 1 && W != 4) W = 4;
            if (U != 1 && U != 2 && U != 4) U = 2;
            const size_t per_wg = (size_t)64 * W * U;
            const size_t blocks = (n_vec + per_wg - 1) / per_wg;
            if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
            const uint4 *xv = static_cast<const uint4 *>(x);
            uint4 *ov = static_cast<uint4 *>(out);
#define ANTQ_LANE(UU, WW, AA)                                                                                          \
    launch_k(k_fq_lane<T, OVP, IDX, UU, false, AA, WW>, dim3((unsigned)blocks), dim3(64 * WW), lds, st, xv, ov, idx, n_vec,  \
             (uint32_t)vpr, vshift, alpha, per_row, gmax, 1.0f, (float *)nullptr, pa, tab)
            if (!pa.adom) ANTQ_LANE(2, 4, false);
            else if constexpr (IDX) ANTQ_LANE(2, 4, true);
            else if (W == 1) { if (U == 1) ANTQ_LANE(1, 1, true); else if (U == 2) ANTQ_LANE(2, 1, true); else ANTQ_LANE(4, 1, true); }
            else             { if (U == 1) ANTQ_LANE(1, 4, true); else if (U == 2) ANTQ_LANE(2, 4, true); else ANTQ_LANE(4, 4, true); }
#undef ANTQ_LANE
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
        if constexpr (!IDX && !std::is_same<T, float>::value) {
            // 16-bit rows of 128 .. 8192 vectors in their own domain (antq_k_hrow.h): the row in 1 / 4 / 16 wavefronts
            HArgs ha;
            if (g_knob_h != 0 && vpr >= kRowKernelMinVpr && vpr <= 8192 && rows <= 0x7fffffffull &&
                hargs_from_plan(plan_host, IO<T>::DTYPE, gmax, ha)) {
                const HDynShape sh = hrow_dyn_shape((uint32_t)vpr);
                const unsigned pad = g_knob_hlds >= 0 ? (unsigned)g_knob_hlds : (rows >= 4 * 8192 ? hrow_dyn_lds_pad(sh) : 0u);
                const uint4 *tl = plan_tlist_dev(plan_host, plan_dev);
                const float *grid = reinterpret_cast<const float *>(tab);
                const dim3 g((unsigned)rows), b(64u * sh.wpr);
#define ANTQ_HD(WW) hipLaunchKernelGGL((k_fq_hrow_dyn<T, OVP, WW>), g, b, (WW) == 1 ? pad : 0u, st, static_cast<const uint4 *>(x), \
                                       static_cast<uint4 *>(out), (uint32_t)rows, (uint32_t)vpr, (uint32_t)sh.vpt, ratio, alpha_out, gmax, ha, tl, grid)
                if (sh.wpr == 1) ANTQ_HD(1); else if (sh.wpr == 4) ANTQ_HD(4); else ANTQ_HD(16);
#undef ANTQ_HD
                return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
            }
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
    int rc = antq_absmax(x, alpha_out, rows, row_len, 1, IO<T>::DTYPE, st);     // (antq_kernels.hip)
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

}  // namespace antq

using namespace antq;

extern "C" int antq_fakequant(const void *x, void *out, int16_t *idx, size_t rows, size_t row_len,
                              const float *alpha, int alpha_per_row, float gmax, const void *plan_host,
                              const void *plan_dev, unsigned flags, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !alpha || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    if ((flags & ANTQ_FLAG_UNORDERED) && idx) return ANTQ_ERR_ARG;      // (the bindings refuse the pair as well)
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_row = alpha_per_row ? 1 : 0;
    struct Unordered {       // scoped: the flag never outlives the call
        explicit Unordered(bool on) { t_unordered = on; }
        ~Unordered() { t_unordered = false; }
    } scope((flags & ANTQ_FLAG_UNORDERED) != 0);
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
int prefetch_unit_fq()        // antq_prefetch_kernels (antq_kernels.hip): load this unit's code object now
{
    hipFuncAttributes at;
    return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_fq_hrow<bf16_tag, false, 4, 1>)) == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq
