// antq_k_approx.h -- the element path for small groups and big tables: bucket from an approximate quotient, EXACT decision
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_APPROX_H
#define ANTQ_K_APPROX_H

#include "antq_device.h"

namespace antq {

// ------------------------------------------------------------------------------------
// Why.  The d-domain core (quant_vec) spends most of its VALU time on per-element work whose only purpose is to be
// exact: the 5-FMA division d = fl(x / s), the straight-through add / subtract and the multiply back.  For rows of
// >= 128 vectors the x-domain kernels hoist all of that into a per-row table; for 16-element groups (2 bf16 lanes per
// scale) there is nothing to hoist it into, and the kernel is VALU-bound (~21 ops per bf16 element, 61 % of HBM).
//
// What.  Plans with `adom` (every ANT / OliVe codebook) let the decision be made on x itself, per element, exactly:
//     fl(x / s) >= T   <=>   x / s > M,  or  x / s == M and T has an even mantissa  (M = the rounding boundary below T:
//                            the midpoint of pred(T) and T; round-to-nearest-even)
//                      <=>   x - M' * s >= 0   with  M' = M (T even)  or  the double just above M (T odd)
// M has 25 significant bits, s 24: M * s is EXACT in double, and ONE v_fma_f64 gives the correctly rounded x - M' * s,
// whose sign is exact (for odd T the perturbation of M' is below the smallest possible gap between x and M * s and only
// turns an exact tie negative).  Exact ties are not rare: with bf16 data and alpha = the group's abs-max, d = 10 * x /
// alpha hits the codebook's mid-points (7.5, 3.75, ...) for about one element in 200 -- a margin test with an exact
// redo inside the margin spent most of its time in the redo.
//     dt = x * rcp(s)                 only picks the bucket (within 2^-22 of fl(x / s); the plan builder duplicates a
//                                     threshold within 2^-20 of a bucket edge into the neighbouring bucket)
//     {M', v_lo, v_hi} = atab[bucket] one ds_read_b128 (the table is part of the plan blob; every workgroup copies it to LDS)
//     q   = fma64(-M', s, x) >= 0 ? v_hi : v_lo
//     out = fl(q * s)                 == ((q - d) + d) * s: the straight-through step is exact in every region of an
//                                     `adom` plan (Sterbenz); -0.0 codebook entries are stored as +0.0, as (q - d) + d gives
// ~13 VALU ops per element (3 of them f64, full rate on CDNA4) and no data-dependent branch.  Elements with
// |dt| >= xlim (clipped beyond twice the outermost value), NaN / Inf and groups whose scale is not in [2^-40, 2^40]
// take the literal reference sequence (true division, scan, straight-through arithmetic), per lane, rarely.
// ------------------------------------------------------------------------------------
struct ScaleA {
    float s;
    float rs;     // v_rcp_f32(s): only steers the bucket choice
    double sd;    // (double)s
    bool ok;      // s in [2^-40, 2^40] (positive, finite): the table path may be used
};
__device__ __forceinline__ ScaleA make_scale_a(float alpha, float gmax)
{
    ScaleA sc;
    sc.s = alpha / gmax;                                // exactly as the reference divides it (AQ:536)
    sc.ok = (sc.s >= kScaleLo) && (sc.s <= kScaleHi);
    sc.rs = __builtin_amdgcn_rcpf(sc.s);
    sc.sd = (double)sc.s;
    return sc;
}

// The a-table is part of the plan blob (built once per grid on the host, antq_plan.cpp): behind the plan's entries,
//     atab[slots] x {double M'; float v_lo; float v_hi}     slot = bucket (linear key), or 2 * bucket + sign
//     aidx[slots] x packed index pair (as LutEntry::idx)     read only when an index output is wanted
// LDS image, staged per workgroup by a plain copy: [atab] [grid: m_pad floats] [IDX only: aidx]
struct ATab {
    const uint4 *tab;
    const float *grid;
    const uint32_t *idx;
};
__host__ __device__ inline uint32_t atab_units(uint32_t slots, uint32_t m_pad, bool idx)
{
    return slots + (m_pad >> 2) + (idx ? ((slots + 3u) >> 2) : 0u);      // 16-byte units to stage
}

// first = the caller's early fetch of atab_src(threadIdx.x) (issued ahead of its HBM loads, like stage_plan's)
__device__ __forceinline__ uint4 atab_src(const PlanArgs &pa, const uint4 *__restrict__ plan_tab, uint32_t i)
{
    // plan_tab: [grid m_pad/4 units][entries n_entries units][atab slots units][aidx]
    const uint32_t gu = pa.m_pad >> 2, s = pa.atab_slots;
    const uint4 *at = plan_tab + gu + pa.n_entries;
    return i < s ? at[i] : (i < s + gu ? plan_tab[i - s] : at[i - gu]);
}
template <bool IDX>
__device__ __forceinline__ ATab stage_atab(const PlanArgs &pa, const uint4 *__restrict__ plan_tab, uint4 *smem, uint4 first)
{
    const uint32_t units = atab_units(pa.atab_slots, pa.m_pad, IDX);
    if (threadIdx.x < units) smem[threadIdx.x] = first;
    for (uint32_t i = threadIdx.x + blockDim.x; i < units; i += blockDim.x) smem[i] = atab_src(pa, plan_tab, i);
    ATab A;
    A.tab = smem;
    A.grid = reinterpret_cast<const float *>(smem + pa.atab_slots);
    A.idx = reinterpret_cast<const uint32_t *>(smem + pa.atab_slots + (pa.m_pad >> 2));
    return A;
}
template <bool IDX>
__device__ __forceinline__ uint4 atab_prefetch(const PlanArgs &pa, const uint4 *__restrict__ plan_tab)
{
    uint4 first = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < atab_units(pa.atab_slots, pa.m_pad, IDX)) first = atab_src(pa, plan_tab, threadIdx.x);
    return first;
}

template <int EPL, bool OVP, bool IDX>
__device__ __forceinline__ void quant_vec_a(const PlanArgs &pa, const ATab &A, const ScaleA &sc, const float (&x)[EPL],
                                            float (&o)[EPL], int (&j)[EPL])
{
    float dt[EPL], q[EPL];
    bool fast = sc.ok;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        dt[e] = x[e] * sc.rs;
        fast = fast && (fabsf(dt[e]) < pa.xlim);  // false for NaN / Inf / far beyond the grid
    }
    if (fast) {
        uint32_t slot[EPL];
        if (pa.linear) {
            const float khi = (float)pa.kmax;
#pragma unroll
            for (int e = 0; e < EPL; e++)
                slot[e] = (uint32_t)__builtin_amdgcn_fmed3f(__builtin_fmaf(dt[e], pa.lin_scale, pa.lin_bias), 0.0f, khi);
        } else {
            // slot = 2 * (clamp(key) - kmin) + sign: one v_med3 and one v_alignbit (an unsigned grid keeps a negative key:
            // it clamps to kmin, slot 1)
            const int32_t sh = (int32_t)pa.shift, km = (int32_t)pa.keymask;
            const int32_t lo = (int32_t)pa.kmin, hi = (int32_t)pa.kmax;
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int32_t u = (int32_t)f2u(dt[e]);
                const int32_t t = (u >> sh) & km;
                int32_t ck;
                asm("v_med3_i32 %0, %1, %2, %3" : "=v"(ck) : "v"(t), "v"(lo), "v"(hi));
                slot[e] = __builtin_amdgcn_alignbit((uint32_t)ck, (uint32_t)u, 31);    // 2 * kmin too high: folded into tab0
            }
        }
        const uint4 *tab0 = pa.linear ? A.tab : A.tab - 2u * pa.kmin;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            uint4 ent = tab0[slot[e]];
            asm volatile("" : "+v"(ent.x), "+v"(ent.y), "+v"(ent.z), "+v"(ent.w));   // one ds_read_b128, not two b64 halves
            const double Mp = __longlong_as_double((long long)(((unsigned long long)ent.y << 32) | ent.x));
            const bool c = __builtin_fma(-Mp, sc.sd, (double)x[e]) >= 0.0;
            q[e] = c ? u2f(ent.w) : u2f(ent.z);
            if (IDX) {
                const uint32_t w = (pa.linear ? A.idx : A.idx - 2u * pa.kmin)[slot[e]];
                j[e] = (int)((c ? (w >> 16) : w) & kIdxMask);
            }
        }
        if (OVP) {
#pragma unroll
            for (int p = 0; p < EPL / 2; p++) {
                const bool me = fabsf(q[2 * p]) > 32.0f, mo = fabsf(q[2 * p + 1]) > 32.0f;   // OQ:314
                const bool ve = mo && !me;
                q[2 * p] = ve ? 0.0f : q[2 * p];          // ((q*0 - d) + d) * s == +0 for s > 0
                q[2 * p + 1] = me ? 0.0f : q[2 * p + 1];
                if (IDX) {
                    if (ve) j[2 * p] = ANTQ_IDX_VICTIM;
                    if (me) j[2 * p + 1] = ANTQ_IDX_VICTIM;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; e++) o[e] = q[e] * sc.s;
    } else {
        // exact reference sequence for this lane's EPL elements
        float d[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            d[e] = x[e] / sc.s;
            int jj;
            q[e] = scan_lds(d[e], A.grid, (int)pa.m, jj);
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

}  // namespace antq

#endif  // ANTQ_K_APPROX_H
