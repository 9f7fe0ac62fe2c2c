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
// This translation unit: nearest-value operator, affine quantiser, copy, abs-max, alpha gradient, packed 4-bit codec, knobs.
// (antq_fq.hip: one tensor per launch; antq_batch.hip: batched launch; antq_search.hip: calibration.)
#include <type_traits>

#include "antq_host.h"
#include "antq_k_fakequant.h"
#include "antq_k_nearest.h"
#include "antq_k_aux.h"
#include "antq_k_reduce.h"
#include "antq_k_search.h"   // k_sum_partials: the fixed-order sum of workgroup partials (alpha gradient per tensor)

namespace antq {

thread_local int g_knob_u = 0;
thread_local int g_knob_encwg = 2048;
thread_local int g_knob_x = 1;
thread_local int g_knob_nearest_fast = 1;
thread_local int g_knob_lane_rows = 1;
thread_local int g_knob_a = 1;
thread_local int g_knob_waves = 0;
thread_local int g_knob_lane_u = 0;
thread_local int g_knob_rot = 0;
thread_local int g_knob_h = 1;
thread_local int g_knob_hlds = -1;
thread_local int g_knob_dlds = -1;
thread_local int g_knob_schunks = 0;
thread_local int g_knob_exp = 0;
thread_local int g_knob_hist = 1;
thread_local int g_knob_hist_xmax = 1;
thread_local int g_knob_rows_stream = 1;
thread_local int g_knob_tk_group = 0;     // A/B: workgroups per ticket group of the one-launch reductions (0 = default)
thread_local int g_knob_tk_blocks = 0;
thread_local int g_knob_sort_short = 1;
thread_local int g_knob_sort = 1;         // clip searches from the sorted row (antq_k_sortsearch.h): 0 off, 1 default rule, 2 every eligible launch
thread_local int g_knob_sweep = 1;        // per-row clip searches through the threshold-sweep kernel (antq_k_sweep.h): 0 = the direct kernels (A/B, tests)    // A/B: workgroups of the one-launch reductions (0 = default)

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

// HIP loads a translation unit's code object when one of its kernels is first launched (or asked about): ~50 ms for the
// four units of this library, which would otherwise land inside the first calibrating forward.  hipFuncGetAttributes on
// one kernel per unit does the loading now, on the calling thread's current device; nothing is launched.
namespace antq {
int prefetch_unit_fq();        // antq_fq.hip
int prefetch_unit_batch();     // antq_batch.hip
int prefetch_unit_search();    // antq_search.hip
}
extern "C" int antq_prefetch_kernels(void)
{
    hipFuncAttributes at;
    int rc = hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_copy)) == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    if (antq::prefetch_unit_fq() != ANTQ_OK) rc = ANTQ_ERR_LAUNCH;
    if (antq::prefetch_unit_batch() != ANTQ_OK) rc = ANTQ_ERR_LAUNCH;
    if (antq::prefetch_unit_search() != ANTQ_OK) rc = ANTQ_ERR_LAUNCH;
    (void)hipGetLastError();
    return rc;
}

extern "C" int antq_debug_set(int key, int value)
{
    if (key == 0) g_knob_u = value;
    else if (key == 1) g_knob_encwg = value > 0 ? value : 2048;
    else if (key == 2) g_knob_x = value;
    else if (key == 3) g_knob_nearest_fast = value;
    else if (key == 4) g_knob_a = value;
    else if (key == 5) g_knob_lane_rows = value;
    else if (key == 6) g_knob_waves = value;
    else if (key == 7) g_knob_lane_u = value;
    else if (key == 8) g_knob_rot = value;
    else if (key == 9) g_knob_h = value;
    else if (key == 10) g_knob_hlds = value;
    else if (key == 11) g_knob_dlds = value;
    else if (key == 12) g_knob_schunks = value;
    else if (key == 13) g_knob_exp = value;
    else if (key == 14) g_knob_hist = value;
    else if (key == 15) g_knob_hist_xmax = value;
    else if (key == 16) g_knob_rows_stream = value;
    else if (key == 17) g_knob_tk_group = value;
    else if (key == 18) g_knob_tk_blocks = value;
    else if (key == 19) g_knob_sweep = value;
    else if (key == 20) g_knob_sort = value;
    else if (key == 21) g_knob_sort_short = value;
    else return ANTQ_ERR_ARG;
    return ANTQ_OK;
}

namespace antq {

template <typename T>
static int launch_absmax(const void *x, float *amax, size_t rows, size_t row_len, int per_row, hipStream_t st, bool zero = true)
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
    if (per_row && vec_ok && g_knob_rows_stream && rows <= 0x7fffffffull) {
        // rows of 128 / 256 / 512 / 1024 vectors: one single-wavefront workgroup per row, the whole row in flight (streaming)
        const size_t vpr = row_len / EPL;
        const dim3 g((unsigned)(rows < 262144u ? rows : 262144u)), b(64);
        const uint4 *xv = static_cast<const uint4 *>(x);
        bool done = true;
        if (vpr == 128) hipLaunchKernelGGL((k_absmax_rows<T, 2>), g, b, 0, st, xv, amax, (uint32_t)rows);
        else if (vpr == 256) hipLaunchKernelGGL((k_absmax_rows<T, 4>), g, b, 0, st, xv, amax, (uint32_t)rows);
        else if (vpr == 512) hipLaunchKernelGGL((k_absmax_rows<T, 8>), g, b, 0, st, xv, amax, (uint32_t)rows);
        else if (vpr == 1024) hipLaunchKernelGGL((k_absmax_rows<T, 16>), g, b, 0, st, xv, amax, (uint32_t)rows);
        else done = false;
        if (done) return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
    size_t waves = per_row ? rows : (rows * row_len + 64 * EPL * 4 - 1) / (64 * EPL * 4);
    size_t blocks = (waves + 3) / 4;
    // (per tensor: one workgroup per CU.  More measured slower with and without the closing atomics -- 16384^2 bf16: 83.7 us
    //  with 256 workgroups, 90 / 109 / 113 / 117 us with 512 / 1024 / 2048 / 4096 -- the block-strided walk wants few, long streams; a chunked walk with
    //  1024 workgroups reads 33 MB in the per-row kernel's 7.5 us but then spends 15 us on its 1024 atomics to ONE address, which
    //  all arrive together: 23 us against this shape's 9.3)
    // (round 5: the chunked walk with SIXTEEN wavefronts per workgroup -- 4096 contiguous streams, 256 atomics -- is slower still:
    //  10.9 us for the 33.5 MB bf16 tensor, 1024-thread workgroups start slowly; profiles/r05_absmax_experiments.log, .patch)
    if (blocks > (per_row ? 4096u : 256u)) blocks = per_row ? 4096 : 256;
    if (blocks < 1) blocks = 1;
    // the whole-tensor maximum is an atomicMax of workgroup maxima into a zero (stream-ordered, capturable)
    if (!per_row && zero) hipLaunchKernelGGL(k_zero_f32, dim3(1), dim3(64), 0, st, amax);
    hipLaunchKernelGGL((k_absmax<T>), dim3((unsigned)blocks), dim3(256), 0, st, x, amax, rows, row_len, per_row, vec_ok);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq

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

namespace antq {
template <typename T>
static int launch_moments(const void *x, double *sums, double *ws, size_t rows, size_t row_len, int per_row, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const bool al = reinterpret_cast<uintptr_t>(x) % 16 == 0;
    const int vec_ok = per_row ? (al && row_len % EPL == 0) : al;
    size_t waves = per_row ? rows : (rows * row_len + 64 * EPL * 4 - 1) / (64 * EPL * 4);
    size_t blocks = (waves + 3) / 4;
    const size_t cap = per_row ? 4096 : 1024;            // (per tensor: <= kWsSlots partials)
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_moments<T>), dim3((unsigned)blocks), dim3(256), 0, st, x, sums, ws, rows, row_len, per_row, vec_ok);
    if (!per_row) hipLaunchKernelGGL(k_sum_partials, dim3(2), dim3(256), 0, st, ws, (uint32_t)blocks, 2, sums);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq

extern "C" int antq_moments(const void *x, size_t rows, size_t row_len, int per_row, int dtype, double *sums, void *workspace,
                            void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !sums || (!per_row && !workspace)) return ANTQ_ERR_ARG;
    double *ws = static_cast<double *>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_moments<float>(x, sums, ws, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_BF16: return launch_moments<bf16_tag>(x, sums, ws, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_F16: return launch_moments<f16_tag>(x, sums, ws, rows, row_len, per_row ? 1 : 0, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_xmax_3sigma(const double *sums, size_t na, size_t n_per, int dtype, float *xmax, void *stream)
{
    if (na == 0) return ANTQ_OK;
    if (!sums || !xmax || n_per == 0) return ANTQ_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((na + 255) / 256)), b(256);
    switch (dtype) {
    case ANTQ_F32: hipLaunchKernelGGL((k_xmax_3sigma<float>), g, b, 0, st, sums, na, (double)n_per, xmax); break;
    case ANTQ_BF16: hipLaunchKernelGGL((k_xmax_3sigma<bf16_tag>), g, b, 0, st, sums, na, (double)n_per, xmax); break;
    case ANTQ_F16: hipLaunchKernelGGL((k_xmax_3sigma<f16_tag>), g, b, 0, st, sums, na, (double)n_per, xmax); break;
    default: return ANTQ_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
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

extern "C" int antq_absmax_into(const void *x, float *amax, size_t n, int dtype, void *stream)
{
    if (n == 0) return ANTQ_OK;
    if (!x || !amax) return ANTQ_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_absmax<float>(x, amax, 1, n, 0, st, false);
    case ANTQ_BF16: return launch_absmax<bf16_tag>(x, amax, 1, n, 0, st, false);
    case ANTQ_F16: return launch_absmax<f16_tag>(x, amax, 1, n, 0, st, false);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

// ======================================================================================
// Whole-tensor reductions in one launch (antq_k_reduce.h, ABI 7)
// ======================================================================================
namespace antq {
template <typename T>
static int launch_absmax_t(const void *x, float *amax, size_t n, void *ws, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const int vec_ok = reinterpret_cast<uintptr_t>(x) % 16 == 0;
    size_t blocks = (n + 64 * EPL * 4 * 4 - 1) / (64 * EPL * 4 * 4);
    const size_t cap = g_knob_tk_blocks > 0 ? (size_t)g_knob_tk_blocks : 256;
    if (blocks > cap) blocks = cap;              // one workgroup per CU: few, long streams (see launch_absmax)
    if (blocks < 1) blocks = 1;
    const uint32_t group = g_knob_tk_group > 0 ? (uint32_t)g_knob_tk_group : 16u;
    hipLaunchKernelGGL((k_absmax_t<T>), dim3((unsigned)blocks), dim3(256), 0, st, x, amax, n, vec_ok, ws, group);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
template <typename T>
static int launch_alpha_grad_t(const void *x, const void *out, const void *gout, double *gsum, size_t n, void *ws, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const int vec_ok = (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gout)) % 16 == 0;
    size_t blocks = (n + 64 * EPL * 2 * 4 - 1) / (64 * EPL * 2 * 4);       // a chunk of 128 vectors per wavefront
    // (16 x 4096^2, profiles/r06_aux_kernels.log: bf16 128 / 256 / 512 / 1024 workgroups 41 / 64.2 / 65.6 / 65.7 %, fp32 49 /
    //  77.0 / 75.0 / 74.2 %; workgroups per ticket group 8 .. 64: no difference, one group for all: 56 / 43 %)
    const size_t cap = g_knob_tk_blocks > 0 ? (size_t)g_knob_tk_blocks : (sizeof(T) == 4 ? 256 : 512);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const uint32_t group = g_knob_tk_group > 0 ? (uint32_t)g_knob_tk_group : 32u;
    hipLaunchKernelGGL((k_alpha_grad_t<T>), dim3((unsigned)blocks), dim3(256), 0, st, x, out, gout, gsum, n, vec_ok, ws, group);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq

extern "C" int antq_absmax_t(const void *x, float *amax, size_t n, int dtype, void *reduce_ws, void *stream)
{
    if (!x || !amax || !reduce_ws || n == 0) return ANTQ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(reduce_ws) % 16) return ANTQ_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_absmax_t<float>(x, amax, n, reduce_ws, st);
    case ANTQ_BF16: return launch_absmax_t<bf16_tag>(x, amax, n, reduce_ws, st);
    case ANTQ_F16: return launch_absmax_t<f16_tag>(x, amax, n, reduce_ws, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_alpha_grad_t(const void *x, const void *out, const void *gout, size_t n, double *gsum, int dtype,
                                 void *reduce_ws, void *stream)
{
    if (!x || !out || !gout || !gsum || !reduce_ws || n == 0) return ANTQ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(reduce_ws) % 16) return ANTQ_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_alpha_grad_t<float>(x, out, gout, gsum, n, reduce_ws, st);
    case ANTQ_BF16: return launch_alpha_grad_t<bf16_tag>(x, out, gout, gsum, n, reduce_ws, st);
    case ANTQ_F16: return launch_alpha_grad_t<f16_tag>(x, out, gout, gsum, n, reduce_ws, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

// ======================================================================================
// Fused Quantizer._forward of a float64 tensor (antq_k_f64.h)
// ======================================================================================
#include "antq_k_f64.h"

extern "C" int antq_fakequant_f64(const double *x, double *out, size_t rows, size_t row_len, const double *alpha,
                                  int alpha_per_row, double gmax, const void *plan_host, const void *plan_dev, unsigned flags,
                                  void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !alpha || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    if (flags & ~ANTQ_FLAG_OVP) return ANTQ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(x) % 8 || reinterpret_cast<uintptr_t>(out) % 8 || reinterpret_cast<uintptr_t>(alpha) % 8) return ANTQ_ERR_ALIGN;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    const size_t n = rows * row_len, npairs = (n + 1) / 2;
    size_t blocks = (npairs + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const size_t lds = lds_table(pa, false);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (flags & ANTQ_FLAG_OVP)
        hipLaunchKernelGGL((antq::k_fq_f64<true>), dim3((unsigned)blocks), dim3(256), lds, st, x, out, n, row_len, alpha,
                           alpha_per_row ? 1 : 0, gmax, pa, plan_tab_ptr(plan_dev));
    else
        hipLaunchKernelGGL((antq::k_fq_f64<false>), dim3((unsigned)blocks), dim3(256), lds, st, x, out, n, row_len, alpha,
                           alpha_per_row ? 1 : 0, gmax, pa, plan_tab_ptr(plan_dev));
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


