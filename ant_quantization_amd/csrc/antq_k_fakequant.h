// antq_k_fakequant.h -- fused Quantizer._forward kernels: K1a rows (d-domain table), K1x rows (per-row x-domain table), K1b lane groups, K1c element-granular
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_FAKEQUANT_H
#define ANTQ_K_FAKEQUANT_H

#include "antq_device.h"
#include "antq_k_approx.h"

namespace antq {

// ------------------------------------------------------------------------------------
// K1a  wave-uniform scale.  A task = up to U*64 consecutive 16-byte vectors of ONE row
// (quant group) = one wavefront; the row's alpha is a scalar load and scale / reciprocal
// are wave-uniform.  All U loads of the task are issued before anything else; with 6-8
// resident wavefronts per SIMD that keeps > 100 KiB per CU in flight, which is what hides
// HBM latency (a persistent ping-pong variant measured slower: it doubles the registers).
// Rows need row_len % EPL == 0; lanes past the row end are masked.
//   vpr = vectors per row, tpr = tasks per row = ceil(vpr / (64*U)).
// DYN: alpha is not read but computed: alpha = max|row| * ratio (requires tpr == 1, the
// whole row sits in this wave's registers; one HBM read of x in total).
// Launch: 256 threads (4 wavefronts); grid = ceil(total_tasks / 4).
// ------------------------------------------------------------------------------------
template <typename T, int U>
__device__ __forceinline__ void task_load(const uint4 *__restrict__ x, const float *__restrict__ alpha, int per_row,
                                          uint32_t task, uint32_t vpr, uint32_t tpr, uint32_t lane, bool dyn,
                                          uint4 (&v)[U], float &a)
{
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint32_t v0 = g * (64u * U) + lane;
    const uint4 *p = x + (size_t)row * vpr;
    // The scale FIRST: loads return in order, so whoever needs only the scale (the per-row table build) waits for it with
    // the U data loads still in flight behind it instead of for everything.
    a = 1.0f;
    if (!dyn) a = ld_global(alpha + (per_row ? row : 0));
    // Unconditional loads (lanes past the row end re-read the row's last vector and are
    // masked at the store): no exec-mask branches between the loads, so all U of them are
    // in flight together.
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ld_stream(p + min(v0 + 64u * u, vpr - 1u));
}

// max over aligned groups of g = 1, 2, 4, ... 64 adjacent lanes (g wave-uniform): DPP lane exchanges up to 16 lanes
// (one VALU instruction per step: quad_perm, row_half_mirror, row_mirror), ds_bpermute beyond.
__device__ __forceinline__ uint32_t group_max_u32(uint32_t m, uint32_t g)
{
    if (g >= 2) m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    if (g >= 4) m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    if (g >= 8) m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, false));   // row_half_mirror
    if (g >= 16) m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, false));  // row_mirror
    if (g >= 32) m = max(m, (uint32_t)__shfl_xor((int)m, 16, 64));
    if (g >= 64) m = max(m, (uint32_t)__shfl_xor((int)m, 32, 64));
    return m;
}
// The same for N independent values, step by step across all of them (the N exchanges of a step are in flight together)
template <int N>
__device__ __forceinline__ void group_max_multi(uint32_t (&m)[N], uint32_t g)
{
#define ANTQ_GM_STEP(COND, EXPR)                                   \
    if (COND) {                                                    \
        uint32_t o[N];                                             \
        _Pragma("unroll") for (int i = 0; i < N; i++) o[i] = (uint32_t)(EXPR);   \
        _Pragma("unroll") for (int i = 0; i < N; i++) m[i] = max(m[i], o[i]);    \
    }
    ANTQ_GM_STEP(g >= 2, __builtin_amdgcn_update_dpp(0, (int)m[i], 0xB1, 0xf, 0xf, false))
    ANTQ_GM_STEP(g >= 4, __builtin_amdgcn_update_dpp(0, (int)m[i], 0x4E, 0xf, 0xf, false))
    ANTQ_GM_STEP(g >= 8, __builtin_amdgcn_update_dpp(0, (int)m[i], 0x141, 0xf, 0xf, false))
    ANTQ_GM_STEP(g >= 16, __builtin_amdgcn_update_dpp(0, (int)m[i], 0x140, 0xf, 0xf, false))
    ANTQ_GM_STEP(g >= 32, __shfl_xor((int)m[i], 16, 64))
    ANTQ_GM_STEP(g >= 64, __shfl_xor((int)m[i], 32, 64))
#undef ANTQ_GM_STEP
}
// wave-wide max of a non-negative float (bit patterns order like integers); NaN propagates
// as in torch.max because a NaN's magnitude bits exceed every finite value's.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t m) { return group_max_u32(m, 64u); }

// ADM: 1 / 0 = the approximate-quotient element path is / is not compiled in; -1 = both, chosen per launch by pa.adom
template <typename T, bool OVP, bool IDX, int U, bool DYN, int ADM = -1>
__device__ __forceinline__ void task_run(uint4 *__restrict__ out, int16_t *__restrict__ idx,
                                         float *__restrict__ alpha_out, float ratio,
                                         uint32_t task, uint32_t vpr, uint32_t tpr, uint32_t lane, float gmax,
                                         const PlanArgs &pa, const PlanLds &L, const ATab &A, const uint4 (&v)[U], float a)
{
    constexpr int EPL = IO<T>::EPL;
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint32_t v0 = g * (64u * U) + lane;
    const size_t base = (size_t)row * vpr + v0;
    if (DYN) {
        // alpha = fl32(max|x| * ratio): AQ/quant_modules.py:474 (x_max) and :300 (x_max * ratio)
        uint32_t m = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t mu = IO<T>::amax_acc(0u, v[u]);
            if (v0 + 64u * u < vpr) m = IO<T>::amax_acc(m, v[u]);  // lanes past the row end hold a duplicate
            (void)mu;
        }
        m = IO<T>::amax_bits(m);
        m = wave_max_u32(m);
        a = u2f(m) * ratio;
        if (alpha_out && lane == 0) st_global(alpha_out + row, a);
    }
    // plans with `adom` (every ANT / OliVe codebook whose table is too big for a per-row copy: int-8, flint-5..8, ...):
    // approximate quotient + margin test instead of the exact division per element (wave-uniform choice)
    const bool ad = ADM < 0 ? (pa.adom != 0u) : (ADM != 0);
    ScaleA sa;
    Scale sc;
    if (ad) sa = make_scale_a(a, gmax, 1.0 / (double)gmax); else sc = make_scale(a, gmax);
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (v0 + 64u * u < vpr) {
            float xf[EPL], of[EPL];
            int j[EPL];
            IO<T>::unpack(v[u], xf);
            if (ad) quant_vec_a<EPL, OVP, IDX>(pa, A, sa, xf, of, j);
            else quant_vec<EPL, OVP, IDX>(pa, L, sc, xf, of, j);
            st_stream(out + base + 64u * u, IO<T>::pack(of));
            if (IDX) store_idx<EPL>(idx, base + 64u * u, j);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep one vector's working set live at a time
    }
}

template <typename T, bool OVP, bool IDX, int U, bool DYN, bool LOOP = false>
__global__ void __launch_bounds__(256)
k_fq_uniform(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
             uint32_t total_tasks, uint32_t vpr, uint32_t tpr,
             const float *__restrict__ alpha, int per_row, float gmax, float ratio,
             float *__restrict__ alpha_out, PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t stride = gridDim.x * 4u;   // one-shot launch: stride >= total_tasks, the loop runs once

    // table fetch is issued FIRST (L2 hit) so that its wait (vmcnt is in-order) does not
    // also wait for the HBM loads of the task, which are issued right behind it
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (pa.adom) tab0 = atab_prefetch<IDX>(pa, plan_tab);
    else if (threadIdx.x < pa.tab_units) tab0 = ld_global(plan_tab + threadIdx.x);

    uint4 v[U];
    float a;
    bool active = task < total_tasks;
    task_load<T, U>(x, alpha, per_row, active ? task : total_tasks - 1u, vpr, tpr, lane, DYN, v, a);

    PlanLds L;
    ATab A;
    if (pa.adom) A = stage_atab<IDX>(pa, plan_tab, smem, tab0);
    else L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    // Big tables (8-bit grids: up to 48 KiB) are staged once per workgroup and amortised over a
    // grid-stride loop of tasks; small tables use a one-shot grid (loop runs once).
    if (!LOOP) {
        if (active) task_run<T, OVP, IDX, U, DYN>(out, idx, alpha_out, ratio, task, vpr, tpr, lane, gmax, pa, L, A, v, a);
        return;
    }
    while (active) {
        task_run<T, OVP, IDX, U, DYN>(out, idx, alpha_out, ratio, task, vpr, tpr, lane, gmax, pa, L, A, v, a);
        task += stride;
        active = task < total_tasks;
        if (active) task_load<T, U>(x, alpha, per_row, task, vpr, tpr, lane, DYN, v, a);
    }
}

// ------------------------------------------------------------------------------------
// K1x  x-domain row kernel: the fast path for rows of >= kRowKernelMinVpr (128) vectors: the headline shape and every
// Linear / conv row of 512 fp32 or 1024 bf16 elements and up; the closed-form threshold keeps the per-task table cheap.
//
// K1a spends most of its VALU time on per-element work that only depends on the ROW:
// dividing by the row's scale, and mapping the quotient back (straight-through add,
// multiply by the scale).  Here each wavefront first rebuilds the grid's bucket table for
// ITS row -- lane b owns bucket b:
//     U_b   = min { x : fl(x / s) >= T_b }       (threshold moved into the x domain, exact: x_threshold)
//     O_lo  = fl(v_lo * s),  O_hi = fl(v_hi * s)  (= the reference's output when the straight-through step
//                                                  (q-d)+d is exact in every region: PlanHeader::xdom,
//                                                  checked per region in antq_plan.cpp)
// into a wave-private LDS table of up to 256 sign-interleaved 16-byte slots (no workgroup
// barrier), then per element does
//     bucket from x * rcp(s)  (approximate quotient: only picks the bucket; thresholds keep
//                              2^-20 clear of bucket edges, so a 2-ulp error cannot matter)
//     out = (x >= U_b) ? O_hi : O_lo
// i.e. 1 mul + 5 integer ops + 1 LDS read + compare/select: ~10 VALU ops per element instead
// of ~18.  Lanes whose |x * rcp(s)| >= xlim (clipped far beyond the grid, Inf, NaN) and rows
// with an odd scale take the exact reference sequence (true division, literal scan).
// ------------------------------------------------------------------------------------
struct XArgs {
    uint32_t m;
    uint32_t shift;
    uint32_t kmin;
    uint32_t kmax;
    uint32_t keymask;
    uint32_t nbneg;
    uint32_t n_entries;
    float xlim;
    float vout;
    uint32_t linear;     // PlanHeader::linear: slot = trunc(clamp(fma(x * rcp(s), lin_scale, lin_bias), 0, kmax)), no sign slots
    float lin_scale;
    float lin_bias;
    float flim;          // |x * rcp(s)| below this: the table's DECISION is right (its whole domain, PlanHeader::fastlim)
    float vmin, vmax;    // the grid's extreme values: what an element clipped beyond xlim quantises to (by sign)
    double inv_gmax;     // 1.0 / (double)gmax of THIS launch (set by the launcher, 0: unknown): scale without a division
};

// The scale of a row task.  s = fl32(alpha / gmax) (AQ:536: a true division) as fl32((double)alpha * (1 / (double)gmax)):
// the double product is within 2^-52 of the quotient and a quotient of two floats is never that close to a float
// rounding boundary without lying on the same side of it (make_scale_a, antq_k_approx.h); scales outside the table
// path's range (denormal, zero, negative, Inf, NaN) are divided literally.  The reciprocal only steers the bucket choice
// (thresholds keep 2^-20 clear of bucket edges): v_rcp_f32.  ~6 instructions instead of two IEEE divisions (~24).
__device__ __forceinline__ Scale row_scale(float alpha, float gmax, double inv_gmax)
{
    Scale sc;
    sc.s = (float)((double)alpha * inv_gmax);
    sc.ok = (inv_gmax != 0.0) && (sc.s >= kScaleLo) && (sc.s <= kScaleHi);
    if (!sc.ok) {
        sc.s = alpha / gmax;
        const float as = fabsf(sc.s);
        sc.ok = (as >= kScaleLo) && (as <= kScaleHi);
    }
    sc.rs = __builtin_amdgcn_rcpf(sc.s);
    return sc;
}

// host: XArgs of a plan (every launcher of an x-domain kernel goes through this)
static inline XArgs xargs_from_plan(const void *plan_host, const PlanArgs &pa)
{
    const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
    XArgs xa;
    xa.m = pa.m; xa.shift = pa.shift; xa.kmin = pa.kmin; xa.kmax = pa.kmax; xa.keymask = pa.keymask;
    xa.nbneg = pa.nbneg; xa.n_entries = pa.n_entries; xa.xlim = ph->xlim; xa.vout = ph->vout;
    xa.linear = pa.linear; xa.lin_scale = pa.lin_scale; xa.lin_bias = pa.lin_bias;
    xa.flim = pa.fastlim * 0.99999f;                        // (the approximate quotient is within 2^-22 of fl(x / s))
    xa.inv_gmax = 0.0;
    const float *g = plan_grid(plan_host);
    xa.vmin = xa.vmax = g[0];
    for (uint32_t i = 1; i < pa.m; i++) { xa.vmin = g[i] < xa.vmin ? g[i] : xa.vmin; xa.vmax = g[i] > xa.vmax ? g[i] : xa.vmax; }
    return xa;
}

__device__ __forceinline__ float f_up(float c)   // next float towards +inf (c != 0)
{
    const uint32_t u = f2u(c);
    return u2f((int32_t)u >= 0 ? u + 1u : u - 1u);
}
__device__ __forceinline__ float f_dn(float c)   // next float towards -inf (c != 0)
{
    const uint32_t u = f2u(c);
    return u2f((int32_t)u >= 0 ? u - 1u : u + 1u);
}

// U = min { x : RN(x / s) >= T }, s > 0, T finite and non-zero -- in closed form.  With P = pred(T) and the rounding
// boundary M = (P + T) / 2:  RN(y) >= T  <=>  y > M, or y == M and T's mantissa is even (ties-to-even).  So
// U is the smallest float above (or at, on an even tie) the real number M * s, and M * s is EXACT in double
// (25 x 24 significant bits).  Replaces a 7-division search (55 VALU ops per lane and task) by ~14 ops.
__device__ __forceinline__ float x_threshold(float T, float s, float rs, bool &ok)
{
    (void)rs;
    ok = true;
    const double M = 0.5 * ((double)f_dn(T) + (double)T);
    const double prod = M * (double)s;                 // exact
    const float xf = (float)prod;                      // round to nearest even
    const double back = (double)xf;
    const bool t_even = (f2u(T) & 1u) == 0u;
    const bool take = (back > prod) || (back == prod && t_even);
    return take ? xf : f_up(xf);
}

// Slot of the per-row table for EPL approximate quotients dt = x * rcp(s) (all |dt| < xlim).
//   slot = 2 * clamp(key) + sign: positive and negative buckets interleaved, so that the sign costs one v_alignbit and the
//   clamp one v_med3 (an unsigned grid keeps a negative key: it clamps to kmin, slot 1); linear-key plans: the bucket number.
// (slots of a float-bits table are 2 * kmin too high: the caller folds that into the table's base address)
template <int EPL>
__device__ __forceinline__ void x_slots(const XArgs &xa, const float (&dt)[EPL], uint32_t (&slots)[EPL])
{
    const uint32_t sh = xa.shift, wd = 31u - xa.shift;
    const bool mag = xa.keymask != 0xffffffffu;     // key = magnitude bits [shift, 31) (v_bfe_u32); unsigned grid: arithmetic shift
    const int32_t lo = (int32_t)xa.kmin, hi = (int32_t)xa.kmax;
    const bool lin = xa.linear != 0u;               // wave-uniform
    const float khi = (float)xa.kmax;
    if (lin) {
#pragma unroll
        for (int e = 0; e < EPL; e++)
            slots[e] = (uint32_t)__builtin_amdgcn_fmed3f(__builtin_fmaf(dt[e], xa.lin_scale, xa.lin_bias), 0.0f, khi);
    } else if (mag) {                                 // (a branch per loop, not a select per element)
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const uint32_t u = f2u(dt[e]);
            const int32_t t = (int32_t)__builtin_amdgcn_ubfe(u, sh, wd);
            int32_t ck;
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(ck) : "v"(t), "v"(lo), "v"(hi));
            slots[e] = __builtin_amdgcn_alignbit((uint32_t)ck, u, 31);
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int32_t u = (int32_t)f2u(dt[e]);
            const int32_t t = u >> sh;
            int32_t ck;
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(ck) : "v"(t), "v"(lo), "v"(hi));
            slots[e] = __builtin_amdgcn_alignbit((uint32_t)ck, (uint32_t)u, 31);
        }
    }
}

template <int EPL, bool OVP, bool IDX, bool PRE = true>
__device__ __forceinline__ void quant_vec_x(const XArgs &xa, const uint4 *wtab, const float *__restrict__ grid,
                                            const Scale &sc, bool rowfast, bool pre, const float (&x)[EPL],
                                            float (&o)[EPL], int (&j)[EPL])
{
    // pre: the caller already knows (from the raw words, IO<T>::all_below) that every |x| of this vector lies inside both
    // the table's domain and the straight-through-exact range for this row's scale -- the per-element checks below are
    // then skipped (2.5 of 10.75 instructions per bf16 element; they run for a wavefront only when one of its lanes fails).
    // fast: the table decides (|d| inside its domain).  Elements clipped beyond xlim (twice the outermost values: OliVe's
    // planted outliers at a 3-sigma alpha, activations far above a calibrated clip) keep the table's decision -- they
    // quantise to the grid's extreme value of their sign -- and only redo the straight-through arithmetic with the true
    // quotient (round 2 sent the whole lane through the literal scan: C3 / C4 lost 2-4 points to one lane in a thousand).
    bool fast = rowfast;
    float dt[EPL];
    float dmax = 0.0f;
    if constexpr (PRE) {
#pragma unroll
        for (int e = 0; e < EPL; e++) dt[e] = x[e] * sc.rs;
        if (!pre) {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                fast = fast && (fabsf(dt[e]) < xa.flim);     // false for NaN / Inf / beyond the table's domain
                dmax = __builtin_fmaxf(dmax, fabsf(dt[e]));
            }
        }
    } else {
        // (PRE = false: the callers that never pass pre -- the pair kernel, compiled for 64 registers -- keep the loop they
        //  were tuned with: checks fused with the products)
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            dt[e] = x[e] * sc.rs;
            fast = fast && (fabsf(dt[e]) < xa.flim);
            dmax = __builtin_fmaxf(dmax, fabsf(dt[e]));
        }
    }
    if (fast) {
        const bool lin = xa.linear != 0u;               // wave-uniform
        const char *t0 = reinterpret_cast<const char *>(wtab) - (lin ? 0 : ((int32_t)xa.kmin << 5));
        const float othr = xa.vout * sc.s;
        uint32_t slots[EPL];
        x_slots<EPL>(xa, dt, slots);
        // {U, O_lo, O_hi, idx pair}: one ds_read_b128 each, NB of them in flight before the first use (all 8 would cost
        // the one-launch-per-tensor kernel its 8th wave per SIMD: 72 registers)
        constexpr int NB = EPL < 4 ? EPL : 4;
#pragma unroll
        for (int b = 0; b < EPL; b += NB) {
            AEnt ents[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) ents[k].v = *reinterpret_cast<const u32x4_t *>(t0 + (slots[b + k] << 4));
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int e = b + k;
                AEnt ent = ents[k];
                ent.pin();
                const bool c = x[e] >= u2f(ent.v.x);
                o[e] = c ? u2f(ent.v.z) : u2f(ent.v.y);
                if (IDX) j[e] = (int)((c ? (ent.v.w >> 16) : ent.v.w) & kIdxMask);
            }
        }
        if (dmax >= xa.xlim) {
            // (rare) far-clipped elements: q is the extreme grid value of the element's sign; (q - d) + d is not q out here
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                if (fabsf(dt[e]) >= xa.xlim) {
                    const float d = x[e] / sc.s;
                    const float q = (d > 0.0f ? xa.vmax : xa.vmin) + 0.0f;
                    const float t = (q - d) + d;
                    o[e] = t * sc.s;
                }
            }
        }
        if (OVP) {
            // |q| > 32 (OQ:314) read off the output: |o| >= fl(vout * s), vout = the smallest magnitude above 32 in the grid
            // (PlanHeader::vout) -- also for a far-clipped element, whose output is within a few ulps of an outlier's.  The
            // two flags of a pair are formed right where they are used (eight live masks cost the pair kernel a wave per SIMD).
#pragma unroll
            for (int p = 0; p < EPL / 2; p++) {
                const bool me = fabsf(o[2 * p]) >= othr, mo = fabsf(o[2 * p + 1]) >= othr;
                const bool ve = mo && !me;
                o[2 * p] = ve ? 0.0f : o[2 * p];          // ((q*0 - d) + d) * s == +0 for s > 0
                o[2 * p + 1] = me ? 0.0f : o[2 * p + 1];
                if (IDX) {
                    if (ve) j[2 * p] = ANTQ_IDX_VICTIM;
                    if (me) j[2 * p + 1] = ANTQ_IDX_VICTIM;
                }
            }
        }
    } else {
        // exact reference sequence for this lane's EPL elements
        float d[EPL], q[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            d[e] = x[e] / sc.s;
            int jj;
            q[e] = scan_lds(d[e], grid, (int)xa.m, jj);
            if (IDX) j[e] = jj;
        }
        if (OVP) {
#pragma unroll
            for (int p = 0; p < EPL / 2; p++) {
                const bool me = fabsf(q[2 * p]) > 32.0f;
                const bool mo = fabsf(q[2 * p + 1]) > 32.0f;
                const bool ve = mo && !me;
                q[2 * p] = q[2 * p] * (ve ? 0.0f : 1.0f);
                q[2 * p + 1] = q[2 * p + 1] * (me ? 0.0f : 1.0f);
                if (IDX) {
                    if (ve) j[2 * p] = ANTQ_IDX_VICTIM;
                    if (me) j[2 * p + 1] = ANTQ_IDX_VICTIM;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const float t = (q[e] - d[e]) + d[e];
            o[e] = t * sc.s;
        }
    }
}

// The wave-private table of one (row, scale): lane b moves the threshold of its bucket(s) into the x domain and
// pre-multiplies the two outputs.  `ent` / `ent2` are the lane's static plan entries (bucket b and b + 64).  Returns
// whether the table path may be used for this scale (Scale::ok and s > 0).  Ends with the wave's LDS writes landed.
__device__ __forceinline__ bool build_row_table(const XArgs &xa, const uint4 &ent, const uint4 &ent2,
                                                const Scale &sc, uint4 *wtab, uint32_t lane)
{
    bool rowfast = sc.ok && (sc.s > 0.0f);
    // entry i of a float-bits table is positive bucket i (slot 2i) or negative bucket i - nb (slot 2(i - nb) + 1);
    // a linear table has one bucket per threshold and no sign slots (slot i)
    const uint32_t nbp = xa.n_entries - xa.nbneg;
    const bool lin = xa.linear != 0u;
    auto put = [&](uint32_t i, const uint4 &e) {
        bool ok = true;
        float Ux = u2f(e.x);
        if (rowfast && i < xa.n_entries && Ux < __builtin_inff()) Ux = x_threshold(Ux, sc.s, sc.rs, ok);
        // (v + 0) * s: a -0.0 grid entry must come out as +0.0, like the reference's (q - d) + d
        const uint4 w = make_uint4(f2u(Ux), f2u((u2f(e.y) + 0.0f) * sc.s), f2u((u2f(e.z) + 0.0f) * sc.s), e.w);
        if (i < xa.n_entries) wtab[lin ? i : (i < nbp ? 2u * i : 2u * (i - nbp) + 1u)] = w;
        return w;
    };
    const uint4 w0 = put(lane, ent);
    if (!lin && xa.nbneg == 0u && lane == 0u) wtab[1] = w0;     // unsigned grid: every negative x lands in slot 1
    if (xa.n_entries > 64u) put(lane + 64u, ent2);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
    return rowfast;
}

// Body of the x-domain row kernel for one wavefront task (shared by k_fq_xrow and k_fq_batch).
template <typename T, bool OVP, bool IDX, int U, bool DYN, int WPR>
__device__ __forceinline__ void xrow_task(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
                                          uint32_t task, uint32_t vpr, uint32_t tpr,
                                          const float *__restrict__ alpha, int per_row, float gmax, float ratio,
                                          float *__restrict__ alpha_out, const XArgs &xa,
                                          const uint4 *__restrict__ entries, const float *__restrict__ grid,
                                          uint4 *wtab, uint32_t lane, uint32_t wv)
{
    constexpr int EPL = IO<T>::EPL;
    // static bucket entries of this lane (L2 hits), issued ahead of the HBM loads; tables of
    // 65..128 buckets (e.g. unsigned int-4) give every lane a second entry
    // (unconditional, clamped addresses: a load under an exec mask makes the compiler wait for it before the mask is
    //  restored -- i.e. BEFORE the HBM loads below are even issued, a serial L2 round trip per wavefront)
    const uint4 none = make_uint4(f2u(__builtin_inff()), 0u, 0u, 0u);
    const bool two = xa.n_entries > 64u;
    uint4 ent = ld_global(entries + min(lane, xa.n_entries - 1u)), ent2 = none;
    if (two) ent2 = ld_global(entries + min(lane + 64u, xa.n_entries - 1u));

    uint4 v[U];
    float a;
    task_load<T, U>(x, alpha, per_row, task, vpr, tpr, lane, DYN, v, a);
    __builtin_amdgcn_sched_barrier(0);                 // nothing that consumes a load is scheduled above this line
    if (lane >= xa.n_entries) ent = none;
    if (lane + 64u >= xa.n_entries) ent2 = none;

    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint32_t v0 = g * (64u * U) + lane;
    const size_t base = (size_t)row * vpr + v0;
    if (DYN) {
        uint32_t m = 0;
#pragma unroll
        for (int u = 0; u < U; u++)
            if (v0 + 64u * u < vpr) m = IO<T>::amax_acc(m, v[u]);
        m = wave_max_u32(IO<T>::amax_bits(m));
        if (WPR > 1) {
            // the row spans the WPR (4 or 16) wavefronts of this workgroup (tpr == WPR): combine their maxima
            __shared__ uint32_t wmax[WPR];
            if (lane == 0) wmax[wv] = m;
            __syncthreads();
            m = wmax[0];
#pragma unroll
            for (int w = 1; w < WPR; w++) m = max(m, wmax[w]);
        }
        a = u2f(m) * ratio;
        if (alpha_out && lane == 0 && (WPR == 1 || wv == 0)) st_global(alpha_out + row, a);
    }
    const Scale sc = row_scale(a, gmax, xa.inv_gmax);

    // per-row table: thresholds into the x domain, outputs pre-multiplied by the scale
    const bool rowfast = build_row_table(xa, ent, ent2, sc, wtab, lane);
    // |x| below this => |fl(x * rs)| < min(flim, xlim) with room for the reciprocal's and the product's rounding
    const uint32_t lkey = IO<T>::lim_key(fminf(xa.flim, xa.xlim) * sc.s * 0.999f);

#pragma unroll
    for (int u = 0; u < U; u++) {
        if (v0 + 64u * u < vpr) {
            float xf[EPL], of[EPL];
            int j[EPL];
            // (not in the pair kernel: it runs at 64 registers for 8 waves per SIMD, the extra live values spill, and whole
            //  models lost up to 12 points with 4-vector tasks -- OPT-6.7B 78 -> 66 % -- for +0.4 with 2-vector tasks)
            const bool pre = OVP ? false : (rowfast && IO<T>::all_below(IO<T>::amax_acc(0u, v[u]), lkey));
            IO<T>::unpack(v[u], xf);
            quant_vec_x<EPL, OVP, IDX, !OVP>(xa, wtab, grid, sc, rowfast, pre, xf, of, j);
            st_stream(out + base + 64u * u, IO<T>::pack(of));
            if (IDX) store_idx<EPL>(idx, base + 64u * u, j);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// WPR = 16 (DYN only): rows of up to 64 * U * 16 = 4096 / 8192 vectors (C4's 28 672-wide rows: 3584 bf16 / 7168 fp32
// vectors) held in the registers of ONE 1024-thread workgroup -- abs-max and quantisation on a single HBM read.
template <typename T, bool OVP, bool IDX, int U, bool DYN, int WPR = 1, int WPB = 4>
__global__ void __launch_bounds__(WPR > 4 ? 64 * WPR : 64 * WPB)
k_fq_xrow(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
          uint32_t total_tasks, uint32_t vpr, uint32_t tpr,
          const float *__restrict__ alpha, int per_row, float gmax, float ratio,
          float *__restrict__ alpha_out, XArgs xa, const uint4 *__restrict__ entries,
          const float *__restrict__ grid)
{
    constexpr uint32_t WAVES = WPR > 4 ? WPR : WPB;      // wavefronts per workgroup (WPR > 1: a row spans them, WPB = 4)
    __shared__ __attribute__((aligned(16))) uint4 wtab_all[WAVES][256];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t task = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wv);
    if (task >= total_tasks) return;   // no workgroup barrier in this kernel (WPR > 1: whole workgroups exit)
    xrow_task<T, OVP, IDX, U, DYN, WPR>(x, out, idx, task, vpr, tpr, alpha, per_row, gmax, ratio, alpha_out, xa, entries,
                                        grid, wtab_all[wv], lane, wv);
}

// ------------------------------------------------------------------------------------
// K1b  per-lane scale: small rows / small groups (vpr < 64: several quant groups share a
// wavefront, e.g. group-16 = 2 bf16 lanes or 4 fp32 lanes per group).  Each lane gathers
// its own alpha and builds its own scale.  vshift >= 0 when vpr is a power of two.
// ------------------------------------------------------------------------------------
// AD: the plan allows the approximate-quotient element path (quant_vec_a): no exact division and no straight-through
// arithmetic in the element loop -- what keeps 16-element groups of bf16 (2 lanes per group) from being VALU-bound.
// DYN: alpha = max|group| * ratio from a butterfly over the group's lanes (vpr a power of two <= 256; beyond 64 lanes the
// wavefronts of the workgroup exchange their maxima through LDS).
// Body shared by k_fq_lane and the batched d-domain kernel (k_fq_batch_d).
// TPB: threads per workgroup (the U vectors of a lane are TPB vectors apart).  64 = one wavefront per workgroup: the
// wavefronts of a launch then start and retire one by one instead of four at a time, which measured +1.5 ... +5 points on a
// plain copy of 1 GiB (tools/stream_shapes.hip) and more on one launch per 33.5 MB tensor.
template <typename T, bool OVP, bool IDX, int U, bool DYN, bool AD, bool XW = false, int TPB = 256>
__device__ __forceinline__ void lane_task(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
                                          size_t n_vec, uint32_t vpr, int vshift, const float *__restrict__ alpha, int per_row,
                                          float gmax, float ratio, float *__restrict__ alpha_out, const PlanArgs &pa,
                                          const uint4 *__restrict__ plan_tab, uint4 *smem, size_t first)
{
    constexpr int EPL = IO<T>::EPL;
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (AD) tab0 = atab_prefetch<IDX>(pa, plan_tab);
    else if (threadIdx.x < pa.tab_units) tab0 = ld_global(plan_tab + threadIdx.x);
    uint4 v[U];
    float a[U];
    const double inv_vpr = 1.0 / (double)vpr;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * (size_t)TPB;
        v[u] = make_uint4(0, 0, 0, 0);
        a[u] = 1.0f;
        if (vi < n_vec) {
            v[u] = ld_stream(x + vi);
            if (!DYN) {
                size_t row = 0;
                if (per_row)   // a shift; else the f64-reciprocal quotient (exact below 2^32 vectors); else a 64-bit division
                    row = (vshift >= 0) ? (vi >> vshift)
                                        : (n_vec <= 0xffffffffull ? (size_t)oct_row((uint32_t)vi, vpr, inv_vpr) : vi / vpr);
                a[u] = ld_global(alpha + row);
            }
        }
    }
    // DYN: group = vpr (a power of two) adjacent lanes.  Butterfly max inside the wavefront, the U vectors of a lane (U
    // different groups) step by step together.  XW (vpr = 128): a group spans 2 wavefronts of the workgroup (its 256
    // threads hold 256 consecutive vectors per u), which exchange their maxima through LDS across the barrier the table
    // staging needs anyway.  Lanes past n_vec hold zeros and belong to no real group (n_vec % vpr == 0).
    __shared__ uint32_t s_gmax[(DYN && XW) ? U : 1][4];
    uint32_t m[U];
    if constexpr (DYN && XW) {
#pragma unroll
        for (int u = 0; u < U; u++) m[u] = IO<T>::amax_bits(IO<T>::amax_acc(0u, v[u]));
        group_max_multi<U>(m, 64u);
        if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
            for (int u = 0; u < U; u++) s_gmax[u][threadIdx.x >> 6] = m[u];
        }
    }
    PlanLds L;
    ATab A;
    if (AD) A = stage_atab<IDX>(pa, plan_tab, smem, tab0);
    else L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    const double inv_gmax = 1.0 / (double)gmax;
    if (DYN) {
        if constexpr (XW) {
            const uint32_t w = threadIdx.x >> 6;
#pragma unroll
            for (int u = 0; u < U; u++) m[u] = max(s_gmax[u][w & 2u], s_gmax[u][w | 1u]);
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) m[u] = IO<T>::amax_bits(IO<T>::amax_acc(0u, v[u]));
            if (EPL == 8) group_max_multi<U>(m, vpr);     // (fp32, 2 vectors per lane: one after the other measured 0.5 points better)
            else {
#pragma unroll
                for (int u = 0; u < U; u++) m[u] = group_max_u32(m[u], vpr);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t vi = first + (size_t)u * (size_t)TPB;
            a[u] = u2f(m[u]) * ratio;             // AQ:474 (x_max), :300 (x_max * ratio)
            if (alpha_out && vi < n_vec && (vi & (vpr - 1)) == 0) st_global(alpha_out + (vi >> vshift), a[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * (size_t)TPB;
        float xf[EPL];
        IO<T>::unpack(v[u], xf);
        if (vi < n_vec) {
            float of[EPL];
            int j[EPL];
            if (AD) {
                const ScaleA sc = make_scale_a(a[u], gmax, inv_gmax);
                quant_vec_a<EPL, OVP, IDX>(pa, A, sc, xf, of, j);
            } else {
                const Scale sc = make_scale(a[u], gmax);
                quant_vec<EPL, OVP, IDX>(pa, L, sc, xf, of, j);
            }
            st_stream(out + vi, IO<T>::pack(of));
            if (IDX) store_idx<EPL>(idx, vi, j);
        }
    }
}

template <typename T, bool OVP, bool IDX, int U, bool DYN, bool AD, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES)
k_fq_lane(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
          size_t n_vec, uint32_t vpr, int vshift,
          const float *__restrict__ alpha, int per_row, float gmax, float ratio,
          float *__restrict__ alpha_out, PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    constexpr uint32_t TPB = 64u * WAVES;
    if (WAVES == 4 && DYN && AD && vpr > 64u)            // 16-bit rows of 128 vectors: groups of 2 wavefronts
        lane_task<T, OVP, IDX, U, DYN, AD, true>(x, out, idx, n_vec, vpr, vshift, alpha, per_row, gmax, ratio, alpha_out, pa,
                                                 plan_tab, smem, ((size_t)blockIdx.x * U) * 256u + threadIdx.x);
    else
        lane_task<T, OVP, IDX, U, DYN, AD, false, (int)TPB>(x, out, idx, n_vec, vpr, vshift, alpha, per_row, gmax, ratio, alpha_out,
                                                            pa, plan_tab, smem, ((size_t)blockIdx.x * U) * TPB + threadIdx.x);
}

// ------------------------------------------------------------------------------------
// K1c  element-granular fallback: any row_len (e.g. conv1's K = 147), any alignment,
// and the < EPL tail of a per-tensor launch.  One thread per PAIR (2p, 2p+1) of the flat
// tensor so the OliVe victim rule stays inside a thread; with an odd element count the
// last element's "partner" is element 0 (torch.roll wrap-around, OQ:315-318).
//   elements [e0, e0 + n_here) of a tensor with n_total elements, e0 even.
// ------------------------------------------------------------------------------------
// pair p of the range: elements e0 + 2p, e0 + 2p + 1 (shared by k_fq_scalar and the batched kernel's ragged jobs)
template <typename T, bool OVP, bool IDX>
__device__ __forceinline__ void scalar_pair(const void *__restrict__ x, void *__restrict__ out, int16_t *__restrict__ idx,
                                            size_t p, size_t e0, size_t n_here, size_t n_total, size_t row_len,
                                            const float *__restrict__ alpha, int per_row, float gmax,
                                            const PlanArgs &pa, const PlanLds &L)
{
    const size_t i0 = e0 + 2 * p;
    if (2 * p >= n_here) return;
    const bool has_odd = (2 * p + 1 < n_here);
    const size_t i1 = has_odd ? i0 + 1 : 0;  // wrap partner (only read when !has_odd && OVP)
    const bool need1 = has_odd || (OVP && i0 + 1 == n_total);

    float xs[2] = {IO<T>::load1(x, i0), need1 ? IO<T>::load1(x, i1) : 0.0f};
    float d[2], q[2];
    float s[2];
    int j[2] = {ANTQ_IDX_NONE, ANTQ_IDX_NONE};
    for (int e = 0; e < 2; e++) {
        const size_t ii = e ? i1 : i0;
        const float a = alpha[per_row ? (ii / row_len) : 0];
        s[e] = a / gmax;
        d[e] = xs[e] / s[e];
        int jj;
        q[e] = scan_lds(d[e], L.grid, (int)pa.m, jj);
        j[e] = jj;
    }
    if (OVP && need1) {
        const bool me = fabsf(q[0]) > 32.0f;
        const bool mo = fabsf(q[1]) > 32.0f;
        if (has_odd) {
            const bool ve = mo && !me;
            q[0] = q[0] * (ve ? 0.0f : 1.0f);
            q[1] = q[1] * (me ? 0.0f : 1.0f);
            if (ve) j[0] = ANTQ_IDX_VICTIM;
            if (me) j[1] = ANTQ_IDX_VICTIM;
        } else {
            // odd numel: the last (even-indexed) element is zeroed iff element 0 is an outlier
            q[0] = q[0] * (mo ? 0.0f : 1.0f);
            if (mo) j[0] = ANTQ_IDX_VICTIM;
        }
    }
    {
        float t = (q[0] - d[0]) + d[0];
        IO<T>::store1(out, i0, t * s[0]);
        if (IDX) idx[i0] = (int16_t)j[0];
    }
    if (has_odd) {
        float t = (q[1] - d[1]) + d[1];
        IO<T>::store1(out, i1, t * s[1]);
        if (IDX) idx[i1] = (int16_t)j[1];
    }
}

template <typename T, bool OVP, bool IDX>
__global__ void __launch_bounds__(256)
k_fq_scalar(const void *__restrict__ x, void *__restrict__ out, int16_t *__restrict__ idx,
            size_t e0, size_t n_here, size_t n_total, size_t row_len,
            const float *__restrict__ alpha, int per_row, float gmax,
            PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    scalar_pair<T, OVP, IDX>(x, out, idx, (size_t)blockIdx.x * 256u + threadIdx.x, e0, n_here, n_total, row_len, alpha,
                             per_row, gmax, pa, L);
}

}  // namespace antq

#endif  // ANTQ_K_FAKEQUANT_H
