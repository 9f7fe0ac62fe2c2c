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
//     {M', v_lo, v_hi} = atab[bucket] one ds_read_b128 (the table is rebuilt from the plan's entries by every workgroup)
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

// LDS image: [slots x {double M'; float v_lo; float v_hi}] [grid: m_pad floats] [IDX only: slots x packed index pair]
struct ATab {
    const char *tab;
    const float *grid;
    const uint32_t *idx;
};
__host__ __device__ inline uint32_t atab_slots(uint32_t n_entries, uint32_t nbneg, uint32_t linear)
{
    return linear ? n_entries : 2u * (n_entries - nbneg);       // float-bits key: positive / negative buckets interleaved
}
__host__ __device__ inline size_t atab_bytes(uint32_t n_entries, uint32_t nbneg, uint32_t linear, uint32_t m_pad, bool idx)
{
    const size_t slots = atab_slots(n_entries, nbneg, linear);
    return slots * 16u + (size_t)m_pad * 4u + (idx ? slots * 4u : 0u);
}

__device__ __forceinline__ double atab_boundary(float T)
{
    if (!(T < __builtin_inff())) return (double)__builtin_inff();       // bucket without a threshold: never >=
    const uint32_t u = f2u(T);
    const float P = u2f((int32_t)u >= 0 ? u - 1u : u + 1u);             // pred(T): next float towards -inf (T != 0)
    const double M = 0.5 * ((double)P + (double)T);
    if ((u & 1u) == 0u) return M;                                       // even T: a tie rounds up to T
    const long long b = __double_as_longlong(M);
    return __longlong_as_double(M > 0.0 ? b + 1 : b - 1);               // odd T: strictly above M (M != 0)
}

template <bool IDX>
__device__ __forceinline__ ATab stage_atab(const PlanArgs &pa, const uint4 *__restrict__ plan_tab, uint4 *smem)
{
    const uint32_t slots = atab_slots(pa.n_entries, pa.nbneg, pa.linear);
    const uint4 *entries = plan_tab + (pa.m_pad >> 2);
    uint4 *tab = smem;
    float *grid = reinterpret_cast<float *>(smem + slots);
    uint32_t *idx = reinterpret_cast<uint32_t *>(grid + pa.m_pad);
    const uint32_t nbp = pa.n_entries - pa.nbneg;
    const bool lin = pa.linear != 0u;
    const bool one_sided = !lin && pa.nbneg == 0u;      // unsigned grid: every negative x belongs to the lowest region
    auto conv = [](const uint4 &e) {
        const double Mp = atab_boundary(u2f(e.x));
        const unsigned long long mb = (unsigned long long)__double_as_longlong(Mp);
        return make_uint4((uint32_t)mb, (uint32_t)(mb >> 32), f2u(u2f(e.y) + 0.0f), f2u(u2f(e.z) + 0.0f));
    };
    uint4 e0 = make_uint4(0, 0, 0, 0);
    if (one_sided) e0 = entries[0];
    for (uint32_t i = threadIdx.x; i < pa.n_entries; i += blockDim.x) {
        const uint4 e = entries[i];
        const uint32_t slot = lin ? i : (i < nbp ? 2u * i : 2u * (i - nbp) + 1u);
        tab[slot] = conv(e);
        if (IDX) idx[slot] = e.w;
        if (one_sided) {
            tab[2u * i + 1u] = conv(e0);
            if (IDX) idx[2u * i + 1u] = e0.w;
        }
    }
    const float *g = reinterpret_cast<const float *>(plan_tab);
    for (uint32_t i = threadIdx.x; i < pa.m_pad; i += blockDim.x) grid[i] = g[i];
    ATab A;
    A.tab = reinterpret_cast<const char *>(tab);
    A.grid = grid;
    A.idx = idx;
    return A;
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
                slot[e] = __builtin_amdgcn_alignbit((uint32_t)(ck - lo), (uint32_t)u, 31);
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const uint4 ent = *reinterpret_cast<const uint4 *>(A.tab + (slot[e] << 4));
            const double Mp = __longlong_as_double((long long)(((unsigned long long)ent.y << 32) | ent.x));
            const bool c = __builtin_fma(-Mp, sc.sd, (double)x[e]) >= 0.0;
            q[e] = c ? u2f(ent.w) : u2f(ent.z);
            if (IDX) {
                const uint32_t w = A.idx[slot[e]];
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
