// antq_k_search.h -- clip search (search_mse): SSE of every candidate on one read, and the selection step
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_SEARCH_H
#define ANTQ_K_SEARCH_H

#include "antq_device.h"
#include "antq_k_fakequant.h"

namespace antq {

// ------------------------------------------------------------------------------------
// Clip search (search_mse, AQ:287-326): for every candidate ratio the squared error of the
// fake-quantised row against the row itself, WITHOUT writing the quantised tensor: x is
// read once into registers and all `ncand` candidates are evaluated on it.
//   sse[c, r] += sum_col fl32( fl32|out - x| ^ 2 )      (fp32 terms, fp64 accumulation)
// Tasks of U*64 vectors.  Every sum is formed in ONE fixed order (no floating-point atomics, so a score never depends
// on which workgroup finished first and near-tied candidates resolve the same way on every run and rank):
//   per row    : one wavefront owns a row and walks its tasks in order; its per-candidate sums live in LDS
//   per tensor : every workgroup writes its per-candidate partial to the caller's workspace and k_sum_partials adds
//                the partials of a candidate in a fixed tree
// Nothing has to be zeroed by the caller.
// ------------------------------------------------------------------------------------
constexpr int kPtCand = 128;     // candidates per workgroup (LDS accumulators); blockIdx.y splits longer lists
constexpr int kWsSlots = 8192;   // workgroup partials the per-tensor workspace holds (kWsSlots * kPtCand doubles = 8 MiB)

// per-wavefront accumulators: lane 0 of the owning wavefront adds, its lanes read back (LDS is in order inside a wavefront)
struct WaveAcc {
    double (*w)[kPtCand];
    __device__ __forceinline__ void zero(uint32_t wv, uint32_t lane) const
    {
        for (int c = (int)lane; c < kPtCand; c += 64) w[wv][c] = 0.0;
    }
};

// out[f] = sum over the n_wg workgroup partials of candidate f, in a fixed order (strided per thread, then a tree)
static __global__ void __launch_bounds__(256)
k_sum_partials(const double *__restrict__ ws, uint32_t n_wg, int chunk, double *__restrict__ out, const int *__restrict__ run_if = nullptr)
{
    if (run_if && !*run_if) return;             // (behind a histogram search with a pair list: only when that list overflowed)
    const int f = (int)blockIdx.x;
    const int y = f / chunk, c = f - y * chunk;
    const double *p = ws + ((size_t)y * n_wg) * kPtCand + c;
    double s = 0.0;
    for (uint32_t w = threadIdx.x; w < n_wg; w += 256u) s += p[(size_t)w * kPtCand];
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 128u; off >= 1u; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[f] = sh[0];
}

// XD: grids with an x-domain plan (PlanHeader::xdom: every ANT / OliVe 4-bit codebook).  For every candidate the
// wavefront rebuilds its row table for THAT scale (closed-form thresholds, ~14 ops per lane) and the element loop is
// the one of k_fq_xrow -- no division, no straight-through arithmetic, no multiply: ~12 instead of ~20 VALU ops per
// candidate evaluation.
// Sums over the 64 lanes of FOUR per-lane doubles at once (the squared-error sums of four consecutive candidates): two
// transposing exchanges (lanes 32 apart, then 16 apart: each lane gives away the half it does not keep) leave row r of 16
// lanes with candidate r's partial sums, which four DPP steps (no LDS crossbar) add up -- 3 crossbar exchanges of a double
// per FOUR candidates instead of 6 per candidate (the reduction was 7-10 % of the calibration kernels' time).  Returns, in
// every lane of row r (lanes 16 r .. 16 r + 15), the total of value r.  One fixed order: bit-reproducible.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum4(double a0, double a1, double a2, double a3, uint32_t lane)
{
    const bool up = (lane & 32u) != 0u;              // lanes 32..63 keep values 2, 3 and give away 0, 1
    double k0 = up ? a2 : a0, k1 = up ? a3 : a1;
    const double s0 = up ? a0 : a2, s1 = up ? a1 : a3;
    k0 += __shfl_xor(s0, 32, 64);
    k1 += __shfl_xor(s1, 32, 64);
    const bool odd = (lane & 16u) != 0u;             // rows 1, 3 keep the second value of their pair
    double k = odd ? k1 : k0;
    const double s = odd ? k0 : k1;
    k += __shfl_xor(s, 16, 64);
    k += dpp_f64<0xB1>(k);                           // quad_perm [1,0,3,2]
    k += dpp_f64<0x4E>(k);                           // quad_perm [2,3,0,1]
    k += dpp_f64<0x141>(k);                          // row_half_mirror
    k += dpp_f64<0x140>(k);                          // row_mirror
    return k;
}

template <typename T, bool OVP, int U, bool PT, bool XD>
__global__ void __launch_bounds__(256)
k_search_sse(const uint4 *__restrict__ x, uint32_t total_tasks, uint32_t vpr, uint32_t tpr, size_t rows,
             const float *__restrict__ xmax, int per_row, const float *__restrict__ ratios, int ncand, float gmax,
             double *__restrict__ sse, double *__restrict__ ws, PlanArgs pa, const uint4 *__restrict__ plan_tab,
             int cand_chunk, XArgs xa, const int *__restrict__ run_if = nullptr)
{
    constexpr int EPL = IO<T>::EPL;
    if (run_if && !*run_if) return;             // (see k_sum_partials)
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    __shared__ __attribute__((aligned(16))) uint4 wtab_all[XD ? 4 : 1][XD ? 256 : 1];
    const uint32_t lane = threadIdx.x & 63u;
    // small tensors do not have enough rows to fill the chip: blockIdx.y splits the candidate list
    const int c_begin = (int)blockIdx.y * cand_chunk;
    const int c_end = min(ncand, c_begin + cand_chunk);
    PlanLds L;
    L.lut = nullptr;
    L.grid = nullptr;
    uint4 ent = make_uint4(f2u(__builtin_inff()), 0u, 0u, 0u), ent2 = ent;
    uint4 *wtab = wtab_all[XD ? (threadIdx.x >> 6) : 0];
    const float *grid_g = reinterpret_cast<const float *>(plan_tab);
    const uint4 *entries = plan_tab + (pa.m_pad >> 2);
    if (XD) {
        if (lane < xa.n_entries) ent = entries[lane];
        if (xa.n_entries > 64u && lane + 64u < xa.n_entries) ent2 = entries[lane + 64u];
    } else {
        uint4 tab0 = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
        L = stage_plan(pa, plan_tab, smem, tab0);
    }
    // Per-candidate sums of this wavefront (PT: over all its tasks; rows: over the tasks of the row it owns).
    __shared__ double wacc[4][kPtCand];
    const uint32_t wv = threadIdx.x >> 6;
    const WaveAcc W{wacc};
    if (PT) W.zero(wv, lane);
    __syncthreads();
    const size_t na = per_row ? rows : 1;
    // PT: tasks grid-strided over all wavefronts.  rows: `units` = rows, a wavefront walks the tpr tasks of its row.
    const uint32_t units = PT ? total_tasks : (uint32_t)rows;
    for (uint32_t unit = blockIdx.x * 4u + wv; unit < units; unit += gridDim.x * 4u) {
        const uint32_t row = PT ? 0u : unit;
        const uint32_t g_end = PT ? 1u : tpr;
        if (!PT && tpr != 1) W.zero(wv, lane);
        for (uint32_t gi = 0; gi < g_end; gi++) {
            const uint32_t task = PT ? unit : row * tpr + gi;
            const uint32_t g = PT ? unit : gi;
            uint4 v[U];
            float xm;
            task_load<T, U>(x, xmax, per_row, task, vpr, tpr, lane, false, v, xm);
            const uint32_t v0 = g * (64u * U) + lane;
            // the vectors' magnitude maxima, once for all candidates: per candidate ONE comparison per vector then says
            // whether the per-element domain checks of the table path can be skipped (see quant_vec_x)
            uint32_t vamax[U];
#pragma unroll
            for (int u = 0; u < U; u++) vamax[u] = IO<T>::amax_acc(0u, v[u]);
            double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;      // the last four candidates' per-lane sums (q3 the newest)
            int filled = 0;
            for (int c = c_begin; c < c_end; c++) {
                const float a = xm * ratios[c];  // AQ:300  new_alpha = base_alpha * fl32(i*0.01)
                const Scale sc = make_scale(a, gmax);
                bool rowfast = false;
                if (XD) rowfast = build_row_table(xa, ent, ent2, sc, wtab, lane);
                const uint32_t lkey = IO<T>::lim_key(fminf(xa.flim, xa.xlim) * sc.s * 0.999f);
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (v0 + 64u * u < vpr) {
                        float xf[EPL], of[EPL];
                        int j[EPL];
                        IO<T>::unpack(v[u], xf);
                        if (XD) quant_vec_x<EPL, OVP, false>(xa, wtab, grid_g, sc, rowfast, rowfast && IO<T>::all_below(vamax[u], lkey), xf, of, j);
                        else quant_vec<EPL, OVP, false>(pa, L, sc, xf, of, j);
                        float part = 0.0f;
#pragma unroll
                        for (int e = 0; e < EPL; e++) {
                            const float df = fabsf(of[e] - xf[e]);  // AQ:282 (q - x).abs().pow(2)
                            part += df * df;
                        }
                        acc += (double)part;
                    }
                }
                q0 = q1; q1 = q2; q2 = q3; q3 = acc;
                if (++filled == 4 || c == c_end - 1) {
                    // four (or the last few) candidates summed over the lanes together: row r of lanes holds slot r
                    const double tot = wave_sum4(q0, q1, q2, q3, lane);
                    const int slot = (int)(lane >> 4), cc = c - 3 + slot;
                    if ((lane & 15u) == 0u && slot >= 4 - filled) {
                        if (!PT && tpr == 1) sse[(size_t)cc * na + row] = tot;
                        else wacc[wv][cc - c_begin] += tot;
                    }
                    q0 = q1 = q2 = q3 = 0.0;
                    filled = 0;
                }
            }
        }
        if (!PT && tpr != 1)
            for (int c = (int)lane; c < c_end - c_begin; c += 64) sse[(size_t)(c_begin + c) * na + row] = wacc[wv][c];
    }
    if (PT) {
        // this workgroup's partial, in a fixed order; k_sum_partials adds the workgroups
        __syncthreads();
        double *part = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kPtCand;
        for (int c = (int)threadIdx.x; c < c_end - c_begin; c += 256)
            part[c] = (wacc[0][c] + wacc[1][c]) + (wacc[2][c] + wacc[3][c]);
    }
}

// ------------------------------------------------------------------------------------
// The whole type selection on ONE read of the tensor (search_adaptive_numeric_type, AQ:328-415 / OQ:235-256: the
// reference runs search_mse once per candidate type, 75-88 full passes each).  The task's vectors stay in registers
// while the wavefront walks the flattened (type, clip ratio) list: for every entry it rebuilds its x-domain row table
// for THAT codebook at THAT scale and accumulates the squared error -- the loop of k_search_sse<XD>, with the static
// bucket entries of all (<= kMaxTypes) codebooks parked in LDS.  sse layout: [type][candidate][row].
// ------------------------------------------------------------------------------------
constexpr int kMaxTypes = 4;
struct MultiArgs {
    XArgs xa[kMaxTypes];
    const uint4 *entries[kMaxTypes];
    const float *grid[kMaxTypes];
    float gmax[kMaxTypes];
    int ntypes;
};

// (16-bit data, 4-vector tasks: held to 128 registers = 4 waves per SIMD; the four-candidate lane sum's eight extra
//  registers would otherwise cost the kernel a wave per SIMD and 3.5 %)
template <typename T, bool OVP, int U, bool PT>
__global__ void __launch_bounds__(256, (IO<T>::EPL == 8 && U == 4) ? 4 : 1)
k_search_sse_multi(const uint4 *__restrict__ x, uint32_t total_tasks, uint32_t vpr, uint32_t tpr, size_t rows,
                   const float *__restrict__ xmax, int per_row, const float *__restrict__ ratios, int ncand,
                   double *__restrict__ sse, double *__restrict__ ws, MultiArgs ma, int flat_chunk, const int *__restrict__ run_if = nullptr)
{
    constexpr int EPL = IO<T>::EPL;
    if (run_if && !*run_if) return;             // (see k_sum_partials)
    __shared__ __attribute__((aligned(16))) uint4 wtab_all[4][256];
    __shared__ __attribute__((aligned(16))) uint4 s_ent[kMaxTypes][128];
    __shared__ double wacc[4][kPtCand];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const WaveAcc W{wacc};
    const int nflat = ma.ntypes * ncand;
    const int f_begin = (int)blockIdx.y * flat_chunk;            // flat_chunk <= kPtCand
    const int f_end = min(nflat, f_begin + flat_chunk);
    const uint4 inf = make_uint4(f2u(__builtin_inff()), 0u, 0u, 0u);
    for (uint32_t i = threadIdx.x; i < (uint32_t)ma.ntypes * 128u; i += 256u) {
        const uint32_t t = i >> 7, b = i & 127u;
        s_ent[t][b] = b < ma.xa[t].n_entries ? ma.entries[t][b] : inf;
    }
    if (PT) W.zero(wv, lane);
    __syncthreads();
    uint4 *wtab = wtab_all[wv];
    const size_t na = per_row ? rows : 1;
    const uint32_t units = PT ? total_tasks : (uint32_t)rows;    // rows: a wavefront owns a row and walks its tasks in order
    for (uint32_t unit = blockIdx.x * 4u + wv; unit < units; unit += gridDim.x * 4u) {
        const uint32_t row = PT ? 0u : unit;
        const uint32_t g_end = PT ? 1u : tpr;
        if (!PT && tpr != 1) W.zero(wv, lane);
        for (uint32_t gi = 0; gi < g_end; gi++) {
            const uint32_t task = PT ? unit : row * tpr + gi;
            const uint32_t g = PT ? unit : gi;
            uint4 v[U];
            float xm;
            task_load<T, U>(x, xmax, per_row, task, vpr, tpr, lane, false, v, xm);
            const uint32_t v0 = g * (64u * U) + lane;
            uint32_t vamax[U];             // (as in k_search_sse: one domain comparison per vector and candidate)
#pragma unroll
            for (int u = 0; u < U; u++) vamax[u] = IO<T>::amax_acc(0u, v[u]);
            // the flattened list [f_begin, f_end) type by type: a type's bucket entries and plan fields are fetched once
            int f = f_begin;
            for (int t = f_begin / ncand; f < f_end; t++) {
            const XArgs xa = ma.xa[t];
            const uint4 ent = s_ent[t][lane], ent2 = s_ent[t][lane + 64u];
            const float gmax_t = ma.gmax[t];
            const float *grid_t = ma.grid[t];
            const int c_first = f - t * ncand, c_last = min(ncand, f_end - t * ncand);
            double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;      // (as in k_search_sse: four candidates per lane reduction)
            int filled = 0;
            for (int c = c_first; c < c_last; c++, f++) {
                const float a = xm * ratios[c];  // AQ:300  new_alpha = base_alpha * fl32(i*0.01)
                const Scale sc = make_scale(a, gmax_t);
                const bool rowfast = build_row_table(xa, ent, ent2, sc, wtab, lane);
                const uint32_t lkey = IO<T>::lim_key(fminf(xa.flim, xa.xlim) * sc.s * 0.999f);
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (v0 + 64u * u < vpr) {
                        float xf[EPL], of[EPL];
                        int j[EPL];
                        IO<T>::unpack(v[u], xf);
                        quant_vec_x<EPL, OVP, false>(xa, wtab, grid_t, sc, rowfast, rowfast && IO<T>::all_below(vamax[u], lkey), xf, of, j);
                        float part = 0.0f;
#pragma unroll
                        for (int e = 0; e < EPL; e++) {
                            const float df = fabsf(of[e] - xf[e]);  // AQ:282 (q - x).abs().pow(2)
                            part += df * df;
                        }
                        acc += (double)part;
                    }
                }
                q0 = q1; q1 = q2; q2 = q3; q3 = acc;
                if (++filled == 4 || c == c_last - 1) {
                    const double tot = wave_sum4(q0, q1, q2, q3, lane);
                    const int slot = (int)(lane >> 4), back = 3 - slot;          // candidate c - back, flat index f - back
                    if ((lane & 15u) == 0u && slot >= 4 - filled) {
                        if (!PT && tpr == 1) sse[((size_t)t * ncand + (c - back)) * na + row] = tot;
                        else wacc[wv][f - back - f_begin] += tot;
                    }
                    q0 = q1 = q2 = q3 = 0.0;
                    filled = 0;
                }
            }
            }
        }
        if (!PT && tpr != 1)
            for (int f = (int)lane; f < f_end - f_begin; f += 64) sse[(size_t)(f_begin + f) * na + row] = wacc[wv][f];
    }
    if (PT) {
        __syncthreads();
        double *part = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kPtCand;
        for (int c = (int)threadIdx.x; c < f_end - f_begin; c += 256)
            part[c] = (wacc[0][c] + wacc[1][c]) + (wacc[2][c] + wacc[3][c]);
    }
}

// Element-granular clip search for ragged rows (row_len % EPL != 0, e.g. 3x3x3 conv rows) or
// unaligned buffers: one wavefront per row (per strip of 16 Ki elements for a per-tensor
// scale), exact slow-path arithmetic (true division + literal scan).  With OVP the partner
// element (i ^ 1, or element 0 for the last element of an odd-sized tensor) is quantised
// with ITS row's candidate alpha, as the reference does when it quantises the whole tensor.
template <typename T, bool OVP>
__global__ void __launch_bounds__(256)
k_search_sse_scalar(const void *__restrict__ x, size_t rows, size_t row_len, const float *__restrict__ xmax,
                    int per_row, const float *__restrict__ ratios, int ncand, float gmax, double *__restrict__ sse,
                    double *__restrict__ ws, PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    __shared__ double wacc[4][kPtCand];                // per tensor: this wavefront's sums over its strips
    const int c_begin = (int)blockIdx.y * kPtCand, c_end = min(ncand, c_begin + kPtCand);
    if (!per_row)
        for (int c = (int)(threadIdx.x & 63u); c < kPtCand; c += 64) wacc[threadIdx.x >> 6][c] = 0.0;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const size_t n = rows * row_len;
    const size_t strip = per_row ? row_len : (size_t)16384;
    const size_t nstrips = per_row ? rows : (n + strip - 1) / strip;
    const size_t na = per_row ? rows : 1;
    for (size_t st = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6); st < nstrips; st += (size_t)gridDim.x * 4u) {
        const size_t b = st * strip;
        const size_t e = per_row ? b + row_len : (b + strip < n ? b + strip : n);
        for (int c = c_begin; c < c_end; c++) {
            const float r = ratios[c];
            double acc = 0.0;
            for (size_t i = b + lane; i < e; i += 64) {
                const float xv = IO<T>::load1(x, i);
                const float s0 = (xmax[per_row ? i / row_len : 0] * r) / gmax;
                const float d = xv / s0;
                int j;
                float q = scan_lds(d, L.grid, (int)pa.m, j);
                if (OVP) {
                    size_t ip = i ^ (size_t)1;
                    if (ip >= n) ip = 0;                       // odd numel: torch.roll wrap-around
                    const bool has_partner = (ip != i);
                    if (has_partner) {
                        const float s1 = (xmax[per_row ? ip / row_len : 0] * r) / gmax;
                        int jp;
                        const float qp = scan_lds(IO<T>::load1(x, ip) / s1, L.grid, (int)pa.m, jp);
                        const bool me = fabsf(q) > 32.0f, mp = fabsf(qp) > 32.0f;
                        bool victim;
                        if (i & 1) victim = mp;                 // odd element: victim iff its even partner is an outlier
                        else if ((i ^ 1) < n) victim = mp && !me;  // even element with a real odd partner
                        else victim = mp;                       // last element of an odd-sized tensor
                        q = q * (victim ? 0.0f : 1.0f);
                    }
                }
                const float t = (q - d) + d;
                const float df = fabsf(t * s0 - xv);
                acc += (double)(df * df);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) {
                if (per_row) sse[(size_t)c * na + st] = acc;
                else wacc[threadIdx.x >> 6][c - c_begin] += acc;
            }
        }
    }
    if (!per_row) {
        __syncthreads();
        double *part = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kPtCand;
        for (int c = (int)threadIdx.x; c < c_end - c_begin; c += 256)
            part[c] = (wacc[0][c] + wacc[1][c]) + (wacc[2][c] + wacc[3][c]);
    }
}

// search_mse's selection loop (AQ:299-306: best = 1e10, candidates in ascending order, strict '<' keeps the earliest best):
// = the smallest (score, index) pair among the candidates whose score is below 1e10.  One wavefront per row, the candidates
// over its lanes (a lane walks c = lane, lane + 64, ... in ascending order), then a lexicographic minimum over the lanes.
// (One thread per row walked 76 dependent f64 divisions: 18-26 us for a per-tensor quantiser, 150 times per BERT-base
//  calibration pass.)
// blockIdx.y: the candidate type (antq_calibrate picks for all its types in one launch: sse / best_score / best_alpha of type t
// start t * type_stride_sse / t * na further on; the single-type entry point launches with gridDim.y = 1)
static __global__ void __launch_bounds__(256)
k_search_pick(const double *__restrict__ sse, const float *__restrict__ xmax, const float *__restrict__ ratios,
              int ncand, size_t na, double row_len, float *__restrict__ best_score, float *__restrict__ best_alpha,
              size_t type_stride_sse = 0)
{
    const size_t r = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= na) return;
    sse += (size_t)blockIdx.y * type_stride_sse;
    best_score += (size_t)blockIdx.y * na;
    best_alpha += (size_t)blockIdx.y * na;
    const int lane = (int)(threadIdx.x & 63u);
    float best = 1e10f;
    int bc = 0x7fffffff;
    for (int c = lane; c < ncand; c += 64) {
        const float score = (float)(sse[(size_t)c * na + r] / row_len);
        if (score < best) { best = score; bc = c; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oc = __shfl_xor(bc, off, 64);
        if (ob < best || (ob == best && oc < bc)) { best = ob; bc = oc; }
    }
    if (lane == 0) {
        const float xm = xmax[r];
        best_score[r] = best;
        best_alpha[r] = bc == 0x7fffffff ? xm : xm * ratios[bc];
    }
}

// k_search_pick for every type + the type's score + the type pick in ONE launch when the tensor has ONE scale (na == 1: every
// activation quantiser): wavefront w picks for types w, w + 4, ...; the score of a type with one row IS its row's best MSE
// ((float)(double)best); then the first smallest score, NaN last.  Same comparisons, same results as the two kernels.
static __global__ void __launch_bounds__(256)
k_calib_pick_one_scale(const double *__restrict__ sse, const float *__restrict__ xmax, const float *__restrict__ ratios, int ncand,
                       double row_len, int ntypes, float *__restrict__ best_score, float *__restrict__ best_alpha,
                       float *__restrict__ score, int32_t *__restrict__ type)
{
    __shared__ float sc[64];
    const int lane = (int)(threadIdx.x & 63u);
    for (int t = (int)(threadIdx.x >> 6); t < ntypes; t += 4) {
        const double *p = sse + (size_t)t * (size_t)ncand;
        float best = 1e10f;
        int bc = 0x7fffffff;
        for (int c = lane; c < ncand; c += 64) {
            const float s = (float)(p[c] / row_len);
            if (s < best) { best = s; bc = c; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oc = __shfl_xor(bc, off, 64);
            if (ob < best || (ob == best && oc < bc)) { best = ob; bc = oc; }
        }
        if (lane == 0) {
            const float xm = xmax[0];
            best_score[t] = best;
            best_alpha[t] = bc == 0x7fffffff ? xm : xm * ratios[bc];
            score[t] = best;
            if (t < 64) sc[t] = best;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int bt = 0;
        bool have = false;
        for (int t = 0; t < ntypes; t++) {
            const float v = t < 64 ? sc[t] : score[t];
            if (v != v) continue;
            const float b = bt < 64 ? sc[bt] : score[bt];
            if (!have || v < b) { bt = t; have = true; }
        }
        type[0] = bt;
    }
}

// antq_calibrate's small steps.  The candidate ratios: fl32(i * 0.01), i * 0.01 evaluated in double as Python does (AQ:296).
static __global__ void __launch_bounds__(256)
k_calib_ratios(float *__restrict__ ratios, int lb, int step, int ncand, float *__restrict__ zero = nullptr)
{
    const int c = (int)(blockIdx.x * 256u + threadIdx.x);
    if (zero && c == 0) *zero = 0.0f;                // (the accumulator of a whole-tensor abs-max that follows: one launch less)
    if (c < ncand) ratios[c] = (float)((double)(lb + c * step) * 0.01);
}

// no candidate at all (range(lb, ub, step) empty): the reference's loop body never runs -- best = 1e10, alpha = x_max
static __global__ void __launch_bounds__(256)
k_calib_none(const float *__restrict__ xmax, size_t na, int ntypes, float *__restrict__ best_score, float *__restrict__ alpha)
{
    const size_t r = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= na) return;
    for (int t = 0; t < ntypes; t++) { best_score[(size_t)t * na + r] = 1e10f; alpha[(size_t)t * na + r] = xmax[r]; }
}

// score of a type = the sum of its rows' best MSE (search_mse returns best_score.sum(), AQ:326): double accumulators, one fixed
// order (thread-strided partials, then a tree over the 256 threads); then the type with the smallest score, the first one on
// ties, NaN last (np.argsort(mse)[0], AQ:413-415).  One workgroup walks the types one after the other (round 5: one launch
// instead of two; the same additions in the same order as before).
static __global__ void __launch_bounds__(256)
k_calib_type_score_pick(const float *__restrict__ best_score, size_t na, int ntypes, float *__restrict__ score, int32_t *__restrict__ type)
{
    __shared__ double part[256];
    __shared__ float sc[64];
    for (int t = 0; t < ntypes; t++) {
        const float *p = best_score + (size_t)t * na;
        double s = 0.0;
        // (eight loads in flight, the additions in the same order: one load latency per row block made this 16 us for 4096 rows)
#pragma unroll 8
        for (size_t r = threadIdx.x; r < na; r += 256u) s += (double)p[r];
        part[threadIdx.x] = s;
        __syncthreads();
        for (uint32_t w = 128u; w > 0u; w >>= 1) {
            if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float v = (float)part[0];
            score[t] = v;
            if (t < 64) sc[t] = v;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int best = 0;
        bool have = false;
        for (int t = 0; t < ntypes; t++) {
            const float v = t < 64 ? sc[t] : score[t];
            if (v != v) continue;
            const float b = best < 64 ? sc[best] : score[best];
            if (!have || v < b) { best = t; have = true; }
        }
        type[0] = best;
    }
}

}  // namespace antq

#endif  // ANTQ_K_SEARCH_H
