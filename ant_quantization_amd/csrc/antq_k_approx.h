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
// scale) there is nothing to hoist it into, and the kernel is VALU-bound (~21 ops per bf16 element, 61-65 % of HBM; with the
// path below: 78-80 %).
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
// ~11 VALU ops per element (3 of them f64, full rate on CDNA4) and no data-dependent branch.  Elements clipped beyond
// twice the outermost value (|dt| >= xlim) keep the table's q and redo only the arithmetic with the true quotient;
// NaN / Inf / |d| beyond the table's domain and groups whose scale is not in [2^-40, 2^40] take the literal reference
// sequence (true division, scan, straight-through arithmetic) -- per lane, rarely.
// ------------------------------------------------------------------------------------
struct ScaleA {
    float s;
    float rs;     // v_rcp_f32(s): only steers the bucket choice
    double sd;    // (double)s
    bool ok;      // s in [2^-40, 2^40] (positive, finite): the table path may be used
};
// inv_gmax = 1.0 / (double)gmax.  s = fl32(alpha / gmax) (AQ:536) without the division: the double product is within
// 2^-52 of the quotient, and a quotient of two floats is never that close to a float rounding boundary without being on
// the same side of it (boundary x gmax has <= 49 significant bits, alpha 24: their distance is >= 2^-49 relative), so
// rounding the product to float gives the correctly rounded quotient.  Outside the table path's scale range (denormal,
// zero, negative, Inf, NaN) the division is done literally.
__device__ __forceinline__ ScaleA make_scale_a(float alpha, float gmax, double inv_gmax)
{
    ScaleA sc;
    sc.s = (float)((double)alpha * inv_gmax);
    sc.ok = (sc.s >= kScaleLo) && (sc.s <= kScaleHi);
    if (!sc.ok) sc.s = alpha / gmax;
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
// One a-table entry as the element loop reads it: ONE ds_read_b128 into four consecutive registers (the barrier keeps
// the compiler from splitting it into two b64 halves), M' = the register pair .xy, no copies.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
struct AEnt {
    u32x4_t v;
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ double Mp() const { return __builtin_bit_cast(double, (u32x2_t)v.xy); }
    __device__ __forceinline__ uint32_t lo() const { return v.z; }     // bits of v_lo / v_hi (the encoder: their codes)
    __device__ __forceinline__ uint32_t hi() const { return v.w; }
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
    return ld_global(i < s ? at + i : (i < s + gu ? plan_tab + (i - s) : at + (i - gu)));
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

// max |v[e]| over a lane's EPL values (v_max3 with |.| modifiers; max drops NaNs) and whether any of them is a NaN
// (v_cmp_u on pairs): 10 instructions for 8 values instead of a compare per value and limit.
template <int EPL>
__device__ __forceinline__ float absmax_nan(const float (&v)[EPL], bool &nan)
{
    float m = __builtin_fmaxf(fabsf(v[0]), fabsf(v[1]));
    nan = __builtin_isunordered(v[0], v[1]);
#pragma unroll
    for (int e = 2; e < EPL; e += 2) {
        m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v[e])), fabsf(v[e + 1]));
        nan = nan || __builtin_isunordered(v[e], v[e + 1]);
    }
    return m;
}

// Slot of the a-table for EPL approximate quotients (all |dt| inside the table's domain).
template <int EPL>
__device__ __forceinline__ void a_slots(const PlanArgs &pa, const float (&dt)[EPL], uint32_t (&slot)[EPL])
{
    if (pa.linear) {
        const float khi = (float)pa.kmax;
#pragma unroll
        for (int e = 0; e < EPL; e++)
            slot[e] = (uint32_t)__builtin_amdgcn_fmed3f(__builtin_fmaf(dt[e], pa.lin_scale, pa.lin_bias), 0.0f, khi);
    } else {
        // slot = 2 * (clamp(key) - kmin) + sign: one v_med3 and one v_alignbit (an unsigned grid keeps a negative key:
        // it clamps to kmin, slot 1)
        // key: the magnitude bits [shift, 31) (v_bfe_u32), or for an unsigned grid the arithmetic shift (keymask all ones)
        const uint32_t sh = pa.shift, wd = 31u - pa.shift;
        const bool mag = pa.keymask != 0xffffffffu;
        const int32_t lo = (int32_t)pa.kmin, hi = (int32_t)pa.kmax;
        if (mag) {                                        // (a branch per loop, not a select per element)
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const uint32_t u = f2u(dt[e]);
                const int32_t t = (int32_t)__builtin_amdgcn_ubfe(u, sh, wd);
                int32_t ck;
                asm("v_med3_i32 %0, %1, %2, %3" : "=v"(ck) : "v"(t), "v"(lo), "v"(hi));
                slot[e] = __builtin_amdgcn_alignbit((uint32_t)ck, u, 31);    // 2 * kmin too high: folded into tab0
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int32_t u = (int32_t)f2u(dt[e]);
                const int32_t t = u >> sh;
                int32_t ck;
                asm("v_med3_i32 %0, %1, %2, %3" : "=v"(ck) : "v"(t), "v"(lo), "v"(hi));
                slot[e] = __builtin_amdgcn_alignbit((uint32_t)ck, (uint32_t)u, 31);
            }
        }
    }
}

template <int EPL, bool OVP, bool IDX>
__device__ __forceinline__ void quant_vec_a(const PlanArgs &pa, const ATab &A, const ScaleA &sc, const float (&x)[EPL],
                                            float (&o)[EPL], int (&j)[EPL])
{
    float dt[EPL], q[EPL];
    // fast: the table decides (|d| inside the table's domain); ste: the straight-through step is exact as well (|d| < xlim).
    // Elements clipped beyond xlim but inside the table's domain keep the table's q and only redo the arithmetic.
    const float flim = pa.fastlim * 0.99999f;             // (the approximate quotient is within 2^-22 of d)
#pragma unroll
    for (int e = 0; e < EPL; e++) dt[e] = x[e] * sc.rs;
    bool nan;
    const float dmax = absmax_nan<EPL>(dt, nan);
    const bool fast = sc.ok && !nan && dmax < flim;       // false for NaN / Inf / huge
    const bool ste = dmax < pa.xlim;
    if (fast) {
        uint32_t slot[EPL];
        a_slots<EPL>(pa, dt, slot);
        const AEnt *tab0 = reinterpret_cast<const AEnt *>(pa.linear ? A.tab : A.tab - 2u * pa.kmin);
        AEnt ents[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) ents[e] = tab0[slot[e]];     // all reads in flight before the first use
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            AEnt ent = ents[e];
            ent.pin();
            const bool c = __builtin_fma(-ent.Mp(), sc.sd, (double)x[e]) >= 0.0;
            q[e] = c ? u2f(ent.hi()) : u2f(ent.lo());
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
        if (ste) {
#pragma unroll
            for (int e = 0; e < EPL; e++) o[e] = q[e] * sc.s;
        } else {
            // clipped far beyond the grid (|d| >= twice the outermost value): q is right, (q - d) + d is not q -- literally
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const float d = x[e] / sc.s;
                const float t = (q[e] - d) + d;
                o[e] = t * sc.s;
            }
        }
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
