// antq_k_hrow.h -- K1h: rows of 16-bit elements (bf16 / f16) quantised in the tensor's OWN 16-bit domain
// Part of libantq's device translation units (antq_batch.hip, antq_fq.hip include it); gfx950 only.
//
// Quantizer._forward (AQ/quant_modules.py:535-551, OQ:294-330) of one row is a monotone step function of x once the row's
// scale s = fl32(alpha / gmax) is fixed: out(x) = fl16(fl32(v_r * s)) for x in region r, the regions being separated by
// the x-domain thresholds U_i = min { x : RN(x / s) >= T_i } (x_threshold, antq_k_fakequant.h; T_i = the reference scan's
// decision thresholds, antq_plan.cpp).  For a 16-bit tensor x only takes 16-bit values, so "x >= U_i" is an INTEGER
// compare of the element's bit pattern against the first 16-bit pattern at or above U_i -- no conversion to fp32, no
// multiply by 1/s, no rounding of the result: the element never leaves its 16 bits.
//
// Per row, one wavefront builds a wave-private LDS table of 8-byte slots keyed by (sign, exponent, top mantissa bits) of
// the PATTERN (at most one threshold per slot):
//     word 0 = threshold pattern << 16 (sign bit set in the negative slots; 0xffffffff: no threshold in this slot)
//     word 1 = output pattern below the threshold | output pattern at / above it << 16
// and then spends per element:  v_bfe (key) + v_med3 (clamp) + v_alignbit (sign) + v_lshl_add (address) + ds_read_b64 +
// v_cmp + v_cndmask  -- 6.5 VALU instructions on average against the 8.25 + unpack / pack of the fp32-domain row table
// (K1x), and ONE table build per row instead of one per 2 KiB task.  The table ends in a SENTINEL slot: patterns at or
// beyond the row's limit (far-clipped elements where (q - d) + d is no longer q, Inf, NaN) read the impossible output
// 0xffff, which a packed max over the vector's four output words finds; such a lane redoes its eight elements with the
// literal reference sequence.  Rows whose scale is outside the table path's range do so for every vector.
#ifndef ANTQ_K_HROW_H
#define ANTQ_K_HROW_H

#include "antq_k_fakequant.h"

namespace antq {

constexpr uint32_t kHNoThr = 0xffffffffu;
constexpr uint32_t kHSentinel = 0xffffu;
// (HThr, kHSlots, kHMaxThr: antq_internal.h -- the plan blob carries the threshold list)

struct HArgs {        // by value: SGPRs
    uint32_t n_thr;   // thresholds, 1 .. kHMaxThr
    uint32_t n_neg;   // of which negative (lanes [0, n_neg))
    uint32_t hshift;  // key = magnitude pattern >> hshift
    uint32_t m;       // grid entries (literal scan of the exact path)
    float lim;        // |x * rcp(s)| below this: the step function is the whole story (table's domain and (q - d) + d == q)
    float flim;       // |x * rcp(s)| below this (>= lim): the step function still DECIDES (an element beyond `lim` quantises
                      // to the extreme grid value of its sign) but the straight-through arithmetic has to be done literally
    float vmin, vmax; // the grid's extreme values
    float vout;       // smallest grid magnitude above 32 (+inf: none)
    double inv_gmax;
};

// 16-bit pattern helpers.  up(U): the smallest magnitude pattern whose value is >= U; down(A): the largest whose value
// is <= A (U, A > 0 and finite); out(o): the pattern tensor.to(dtype) gives the fp32 value o.
template <typename T> struct H16;
template <> struct H16<bf16_tag> {
    static constexpr uint32_t MANT = 7;
    static constexpr uint32_t INF = 0x7f80u;
    __device__ __forceinline__ static uint32_t up(float U) { return (f2u(U) + 0xffffu) >> 16; }
    __device__ __forceinline__ static uint32_t down(float A) { return f2u(A) >> 16; }
    __device__ __forceinline__ static uint32_t out(float o) { return IO<bf16_tag>::pk(o, 0.0f) & 0xffffu; }
    __device__ __forceinline__ static float val(uint32_t p) { return u2f(p << 16); }
};
template <> struct H16<f16_tag> {
    static constexpr uint32_t MANT = 10;
    static constexpr uint32_t INF = 0x7c00u;
    __device__ __forceinline__ static uint32_t up(float U)
    {
        uint32_t h = IO<f16_tag>::f2h(U);                       // nearest; beyond 65520: Inf (0x7c00)
        if (IO<f16_tag>::h2f(h) < U) h++;                       // (65504 < U < 65520: 0x7bff -> 0x7c00)
        return h;
    }
    __device__ __forceinline__ static uint32_t down(float A)
    {
        uint32_t h = IO<f16_tag>::f2h(A);
        if (IO<f16_tag>::h2f(h) > A) h--;                       // (Inf -> 65504)
        return h;
    }
    __device__ __forceinline__ static uint32_t out(float o) { return IO<f16_tag>::f2h(o); }
    __device__ __forceinline__ static float val(uint32_t p) { return IO<f16_tag>::h2f(p); }
};

struct HRow {
    uint32_t kmin, klim;   // key clamp range of this row (wave-uniform)
    bool fast;             // the table may be used
};

// Build the row's table.  `thr` = this lane's HThr (lanes >= n_thr: anything).  tab: the wavefront's kHSlots * 2 slots.
// othr_inf (OVP): the row has no table when the outliers' outputs overflow the 16-bit format (pattern(vout * s) >= Inf).
template <typename T>
__device__ __forceinline__ HRow hrow_build(const HArgs &ha, const uint4 &thr, const Scale &sc, uint2 *tab, uint32_t lane)
{
    HRow R;
    const float s = sc.s;
    bool ok = sc.ok && (s > 0.0f);
    const bool mine = lane < ha.n_thr;
    const bool neg = lane < ha.n_neg;
    // patterns at or above lim16 leave the table path: |x| < value(lim16) implies |x * rcp(s)| < lim with room for the
    // reciprocal's and the product's rounding (same margin as the fp32-domain kernel's lim_key)
    const float limx = ha.lim * s * 0.999f;
    const uint32_t lim16 = ok ? H16<T>::down(fminf(limx, 3.0e38f)) : 0u;
    const uint32_t klim = lim16 >> ha.hshift;
    bool tok;
    const float Tt = mine ? u2f(thr.x) : 1.0f;
    const float U = x_threshold(Tt, ok ? s : 1.0f, sc.rs, tok);
    uint32_t t16 = neg ? H16<T>::down(-U) + 1u : H16<T>::up(U);
    const uint32_t key = mine ? (t16 >> ha.hshift) : 0u;
    const uint32_t o_lo = H16<T>::out((u2f(thr.y) + 0.0f) * s), o_hi = H16<T>::out((u2f(thr.z) + 0.0f) * s);
    const uint32_t first = neg ? o_hi : o_lo, second = neg ? o_lo : o_hi;      // by rising MAGNITUDE
    // the neighbouring threshold in the direction of larger magnitudes (lane - 1 on the negative side, lane + 1 on the
    // positive one); the last of a side runs up to the sentinel slot
    const uint32_t k_prev = (uint32_t)__shfl((int)key, (int)((lane + 63u) & 63u), 64);
    const uint32_t k_next = (uint32_t)__shfl((int)key, (int)((lane + 1u) & 63u), 64);
    const bool last_of_side = neg ? (lane == 0u) : (lane + 1u == ha.n_thr);
    const bool first_of_side = neg ? (lane + 1u == ha.n_neg) : (lane == ha.n_neg);
    const uint32_t nxt = last_of_side ? klim : (neg ? k_prev : k_next);
    // lowest key: the smaller of the two sides' innermost thresholds
    const uint32_t kpos = ha.n_neg < ha.n_thr ? (uint32_t)__builtin_amdgcn_readlane((int)key, (int)ha.n_neg) : 0xffffffffu;
    const uint32_t kneg = ha.n_neg > 0u ? (uint32_t)__builtin_amdgcn_readlane((int)key, (int)(ha.n_neg - 1u)) : 0xffffffffu;
    const uint32_t kmin = min(kpos, kneg);
    // every threshold in a slot of its own, below the sentinel slot; the whole range inside the table
    const bool lane_ok = !mine || (key < nxt);
    ok = ok && (klim >= kmin) && (klim - kmin < kHSlots) && (__ballot(lane_ok) == ~0ull);
    R.kmin = kmin; R.klim = klim; R.fast = ok;
    if (!ok) return R;
    auto slot = [&](uint32_t k, bool ng) -> uint2 & { return tab[((k - kmin) << 1) + (ng ? 1u : 0u)]; };
    const uint32_t sbit = neg ? 0x80000000u : 0u;
    if (mine) {
        slot(key, neg) = make_uint2(sbit | (t16 << 16), first | (second << 16));
        // (plain rolled loops, a handful of iterations: the loop vectoriser turned them into a kilobyte of unrolled
        //  ds_write_b32 bodies with scalar epilogues)
#pragma clang loop vectorize(disable) unroll(disable)
        for (uint32_t k = key + 1u; k < nxt; k++) slot(k, neg) = make_uint2(kHNoThr, second | (second << 16));
        if (first_of_side) {
#pragma clang loop vectorize(disable) unroll(disable)
            for (uint32_t k = kmin; k < key; k++) slot(k, neg) = make_uint2(kHNoThr, first | (first << 16));
        }
        if (last_of_side) slot(klim, neg) = make_uint2(sbit | (lim16 << 16), second | (kHSentinel << 16));
    }
    // a side without thresholds (unsigned grid: every negative x is in the lowest region; all-negative grid: the mirror
    // image): one region up to the sentinel
    if (ha.n_neg == 0u || ha.n_neg == ha.n_thr) {
        const bool ng = ha.n_neg == 0u;
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)first, ng ? 0 : (int)(ha.n_thr - 1u));
#pragma clang loop vectorize(disable) unroll(disable)
        for (uint32_t k = kmin + lane; k < klim; k += 64u) slot(k, ng) = make_uint2(kHNoThr, o | (o << 16));
        if (lane == 0u) slot(klim, ng) = make_uint2((ng ? 0x80000000u : 0u) | (lim16 << 16), o | (kHSentinel << 16));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
    return R;
}

// The two elements of one 32-bit word: their two output patterns, packed.  13 VALU instructions:
//   high element  v_bfe_u32 (key) . v_med3_i32 (clamp) . v_alignbit_b32 (slot = 2 key + sign) . v_lshl_add_u32 (address)
//   low element   v_lshlrev_b32 16 . the same four
//   ds_read_b64 x 2;  v_cmp_ge_u32 + v_cndmask_b32_sdwa per element, the second one writing the high half of the result
// (the compiler's own version of the selects -- shifts, masks, an or -- came to 19).
__device__ __forceinline__ uint32_t hrow_pair(uint32_t w, uint32_t tbase, uint32_t vkmin, uint32_t vklim, uint32_t kpos,
                                               uint32_t vkwid)
{
    uint32_t th, tl, wl, o;
    uint2 eh, el;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(th) : "v"(w), "s"(kpos), "v"(vkwid));
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(th) : "v"(th), "v"(vkmin), "v"(vklim));
    wl = w << 16;
    th = __builtin_amdgcn_alignbit(th, w, 31);
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(tl) : "v"(wl), "s"(kpos), "v"(vkwid));
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(tl) : "v"(tl), "v"(vkmin), "v"(vklim));
    tl = __builtin_amdgcn_alignbit(tl, wl, 31);
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t rh = *(const __attribute__((address_space(3))) u32x2_t *)(uintptr_t)((th << 3) + tbase);
    const u32x2_t rl = *(const __attribute__((address_space(3))) u32x2_t *)(uintptr_t)((tl << 3) + tbase);
    eh = make_uint2(rh.x, rh.y);
    el = make_uint2(rl.x, rl.y);
    // low half: o = (wl >= el.x) ? el.y >> 16 : el.y & 0xffff;  high half likewise from eh, kept in place
    asm("v_cmp_ge_u32 vcc, %1, %2\n\t"
        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
        : "=&v"(o) : "v"(wl), "v"(el.x), "v"(el.y) : "vcc");
    asm("v_cmp_ge_u32 vcc, %1, %2\n\t"
        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1"
        : "+v"(o) : "v"(w), "v"(eh.x), "v"(eh.y) : "vcc");
    return o;
}

// word i (wave-uniform i) of a vector, and its replacement, without a register array (dynamic indexing of one would spill)
__device__ __forceinline__ uint32_t vec_word(const uint4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ void vec_set(uint4 &v, int i, uint32_t w)
{
    if (i == 0) v.x = w; else if (i == 1) v.y = w; else if (i == 2) v.z = w; else v.w = w;
}

// The exact reference sequence for one lane's vector (rare): d = x / s, literal scan, pair rule, (q - d) + d, * s.  One
// pair at a time in a rolled loop: this path must not cost the common one registers.
template <typename T, bool OVP>
__device__ __forceinline__ uint4 hrow_exact(const uint4 &v, float s, const float *__restrict__ grid, uint32_t m)
{
    uint4 o = v;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
        const uint32_t w = vec_word(v, i);
        const float x0 = H16<T>::val(w & 0xffffu), x1 = H16<T>::val(w >> 16);
        const float d0 = x0 / s, d1 = x1 / s;
        float q0 = 0.0f, q1 = 0.0f, m0 = 102400.0f, m1 = 102400.0f;
#pragma unroll 1
        for (uint32_t k = 0; k < m; k++) {
            const float g = ld_global(grid + k);
            const float s0 = fabsf(d0 - g), s1 = fabsf(d1 - g);
            if (s0 <= m0) { m0 = s0; q0 = g; }
            if (s1 <= m1) { m1 = s1; q1 = g; }
        }
        if (OVP) {
            const bool me = fabsf(q0) > 32.0f, mo = fabsf(q1) > 32.0f;
            q0 = q0 * ((mo && !me) ? 0.0f : 1.0f);
            q1 = q1 * (me ? 0.0f : 1.0f);
        }
        const float t0 = (q0 - d0) + d0, t1 = (q1 - d1) + d1;
        vec_set(o, i, H16<T>::out(t0 * s) | (H16<T>::out(t1 * s) << 16));
    }
    return o;
}

// OliVe's pair rule on the packed outputs of one word (pairs (2p, 2p + 1) = its two halves, OQ:311-320): the odd element
// is a victim when its even partner is an outlier; the even one when its odd partner is an outlier and it is not one
// itself.  An output is an outlier's iff its magnitude pattern is >= othr = pattern of fl16(fl32(vout * s)).
__device__ __forceinline__ uint32_t hrow_victims(uint32_t o, uint32_t othr)
{
    const uint32_t mg = o & 0x7fff7fffu;
    const bool me = (mg & 0xffffu) >= othr, mo = (mg >> 16) >= othr;
    return o & (((mo && !me) ? 0u : 0xffffu) | (me ? 0u : 0xffff0000u));
}

struct HFar { float rs, flim, vmin, vmax; };     // what the far-clipped arithmetic needs

// A vector that holds an element in the sentinel slot, formed again (see hrow_task).
template <typename T, bool OVP>
__device__ __forceinline__ uint4 hrow_vec_far(const uint4 &v, uint32_t tbase, uint32_t vkmin, uint32_t vklim, uint32_t kpos,
                                              uint32_t vkwid, uint32_t othr, float s, const HFar &far,
                                              const float *__restrict__ grid, uint32_t m)
{
    uint4 o = v;
    bool decided = true;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
        const uint32_t w = vec_word(v, i);
        uint32_t ow = hrow_pair(w, tbase, vkmin, vklim, kpos, vkwid);
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            if (((ow >> (16 * hf)) & 0xffffu) == kHSentinel) {
                const float xe = H16<T>::val((w >> (16 * hf)) & 0xffffu);
                decided = decided && (fabsf(xe * far.rs) < far.flim);          // false for NaN / Inf
                const float d = xe / s;
                const float q = (d > 0.0f ? far.vmax : far.vmin) + 0.0f;
                const float t = (q - d) + d;
                const uint32_t pat = H16<T>::out(t * s);
                ow = hf ? ((ow & 0xffffu) | (pat << 16)) : ((ow & 0xffff0000u) | pat);
            }
        }
        if (OVP) ow = hrow_victims(ow, othr);
        vec_set(o, i, ow);
    }
    if (!decided) return hrow_exact<T, OVP>(v, s, grid, m);
    return o;
}

// One task: VPT 16-byte vectors per lane of ONE row, through the row's table.  Outputs are stored as they are formed;
// a packed max over every output word of the lane finds the sentinel (an element at or beyond the row's limit: far-clipped,
// Inf, NaN) at the end, and only then does the lane redo the vectors that hold such an element with the literal sequence
// and store them again (same lane, same address: the second store lands after the first).
template <typename T, bool OVP, int VPT>
__device__ __forceinline__ void hrow_task(const uint4 (&v)[VPT], uint4 *__restrict__ out, uint32_t v0, uint32_t vpr,
                                          const HRow &R, uint32_t tab_addr, uint32_t hshift, uint32_t othr, float s,
                                          const HFar &far, const float *__restrict__ grid, uint32_t m, uint32_t nv = VPT)
{
    // nv (wave-uniform, <= VPT): vectors per lane actually in use -- the dynamic kernels are compiled once for VPT = 8 and
    // walk rows of any length with it (steps beyond nv are skipped by a scalar branch)
    const uint32_t kpos = 16u + hshift;
    uint32_t vkmin = R.kmin, vklim = R.klim, vkwid = 15u - hshift;
    asm volatile("" : "+v"(vkmin), "+v"(vklim), "+v"(vkwid));          // (VGPR copies: gfx9 takes one SGPR per VOP3)
    const uint32_t tbase = tab_addr - (R.kmin << 4);
    if (R.fast) {
        u16x2_t acc = as_u16x2(0u);
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            if ((uint32_t)u < nv && v0 + 64u * u < vpr) {
                const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    o[i] = hrow_pair(w[i], tbase, vkmin, vklim, kpos, vkwid);
                    acc = __builtin_elementwise_max(acc, as_u16x2(o[i]));
                    if (OVP) o[i] = hrow_victims(o[i], othr);
                }
                st_stream(out + 64u * u, make_uint4(o[0], o[1], o[2], o[3]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint32_t na = ~as_u32(acc);
        if (__builtin_expect((na & 0xffffu) != 0u && (na >> 16) != 0u, 1)) return;
    }
    // rare: this lane holds an element beyond the row's limit (or the row has no table at all).  Vectors with such an
    // element are formed again: a finite element inside the table's decision domain keeps the table's decision -- the
    // extreme value of its sign -- and redoes only the straight-through arithmetic with the true quotient (OliVe's planted
    // outliers at a 3-sigma alpha, activations far above a calibrated clip); anything else (Inf, NaN, beyond flim) sends
    // the vector through the literal sequence.
    const uint32_t slot0 = R.klim << hshift;          // first pattern of the sentinel slot (a superset of "beyond the limit")
    // (a rolled loop with the vector picked by wave-uniform selects: ONE copy of the far-clipped and the literal code in the
    //  kernel instead of VPT of each)
#pragma unroll 1
    for (uint32_t u = 0; u < nv; u++) {
        uint4 cur = v[0];
#pragma unroll
        for (int k = 1; k < VPT; k++)
            if (u == (uint32_t)k) cur = v[k];
        if (v0 + 64u * u < vpr) {
            const uint32_t top = IO<T>::amax_acc(0u, cur);
            if (!R.fast) st_stream(out + 64u * u, hrow_exact<T, OVP>(cur, s, grid, m));
            else if (max(top & 0xffffu, top >> 16) >= slot0)
                st_stream(out + 64u * u, hrow_vec_far<T, OVP>(cur, tbase, vkmin, vklim, kpos, vkwid, othr, s, far, grid, m));
        }
    }
}

// Dynamic LDS on top of the 2 KiB table so that 24 one-wavefront workgroups fit a CU (160 KiB): with 4 KiB of reads per
// wavefront that is 96 KiB in flight per CU.  More in flight measured SLOWER (32 workgroups: 78.6 % against 81.4 % of
// 8 TB/s on 32 x 4096^2 bf16), fewer starve the CU whenever the clocks dip (16: 82.8 % steady but 76 % right after an
// idle period); tools/exp_hrow.hip, profiles/r04_exp_hrow_*.log.
constexpr unsigned kHRowLdsPad = 6656u - kHSlots * 16u;

// What follows the loads of a wavefront task: the row's scale, its table, the elements.
template <typename T, bool OVP, int VPT>
__device__ __forceinline__ void hrow_wave_finish(const uint4 (&v)[VPT], uint4 *__restrict__ out, uint32_t v0, uint32_t vpr,
                                                 float a, float gmax, const HArgs &ha, const uint4 &thr,
                                                 const float *__restrict__ grid, uint2 *tab, uint32_t lane, uint32_t nv = VPT)
{
    const Scale sc = row_scale(a, gmax, ha.inv_gmax);
    HRow R = hrow_build<T>(ha, thr, sc, tab, lane);
    const uint32_t othr = OVP ? H16<T>::out(ha.vout * sc.s) & 0x7fffu : 0u;
    if (OVP && othr >= H16<T>::INF && ha.vout < __builtin_inff()) R.fast = false;     // the outliers' outputs overflow
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)tab;
    const HFar far = {sc.rs, ha.flim, ha.vmin, ha.vmax};
    hrow_task<T, OVP, VPT>(v, out, v0, vpr, R, tab_addr, ha.hshift, othr, sc.s, far, grid, ha.m, nv);
}

// One wavefront task of the row kernels: up to 64 * VPT vectors of ONE row (quant group); shared by the batched launch
// (k_fq_hbatch) and the one-tensor launch (k_fq_hrow).  The threshold list entry and the scale are requested first, then
// all VPT vectors; the table is built while they are in flight.
template <typename T, bool OVP, int VPT>
__device__ __forceinline__ void hrow_wave_task(const uint4 *__restrict__ x, uint4 *__restrict__ out, uint32_t task, uint32_t vpr,
                                               uint32_t tpr, const float *__restrict__ alpha, int per_row, float gmax,
                                               const HArgs &ha, const uint4 *__restrict__ tlist, const float *__restrict__ grid,
                                               uint2 *tab, uint32_t lane)
{
    // (VPT is a template parameter on purpose: with a run-time vector count the loads sit under scalar branches, the
    //  compiler no longer knows how many of them follow the threshold-list load, and waits for ALL of them -- vmcnt(0) --
    //  before the table build instead of building the table while the data is in flight)
    constexpr uint32_t nv = VPT;
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    uint4 thr = ld_global(tlist + min(lane, ha.n_thr - 1u));
    const float a = ld_global(alpha + (per_row ? row : 0));
    const uint32_t v0 = g * (64u * nv) + lane;
    const uint4 *p = x + (size_t)row * vpr;
    uint4 v[VPT];
#pragma unroll
    for (int u = 0; u < VPT; u++) v[u] = ld_stream(p + min(v0 + 64u * u, vpr - 1u));
    __builtin_amdgcn_sched_barrier(0);                 // nothing that consumes a load is scheduled above this line
    // (all four registers of `thr` stay claimed until here: with the unused fourth one free the allocator handed it to an
    //  address computation BETWEEN the data loads, which then had to wait -- vmcnt(0) -- for the first vector to land before
    //  the second was even requested)
    asm volatile("" : "+v"(thr.x), "+v"(thr.y), "+v"(thr.z), "+v"(thr.w));
    hrow_wave_finish<T, OVP, VPT>(v, out + (size_t)row * vpr + v0, v0, vpr, a, gmax, ha, thr, grid, tab, lane, nv);
}

// The same with the scale computed from the row (ANTQ_FLAG_DYNAMIC): alpha = fl32(max |row| * ratio) (AQ:473-477, :300) from
// the 16-bit magnitude patterns of the registers that hold the row -- one HBM read.  The row lives in the WPR (1, 4 or 16)
// wavefronts of the workgroup, nv <= 8 vectors per lane each (vpr <= 64 * nv * WPR; nv wave-uniform: ONE instantiation walks
// every row length, the unused vector steps cost a scalar branch each); task = row * WPR + wavefront.
constexpr int kHDynV = 8;
template <typename T, bool OVP, int WPR>
__device__ __forceinline__ void hrow_wave_task_dyn(const uint4 *__restrict__ x, uint4 *__restrict__ out, uint32_t task, uint32_t vpr,
                                                   uint32_t nv, float ratio, float *__restrict__ alpha_out, float gmax,
                                                   const HArgs &ha, const uint4 *__restrict__ tlist, const float *__restrict__ grid,
                                                   uint2 *tab, uint32_t lane, uint32_t wv)
{
    const uint32_t row = WPR == 1 ? task : task / WPR, g = WPR == 1 ? 0u : task - row * WPR;
    uint4 thr = ld_global(tlist + min(lane, ha.n_thr - 1u));
    const uint32_t v0 = g * (64u * nv) + lane;
    const uint4 *p = x + (size_t)row * vpr;
    uint4 v[kHDynV];
#pragma unroll
    for (int u = 0; u < kHDynV; u++) {
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)u < nv) v[u] = ld_stream(p + min(v0 + 64u * u, vpr - 1u));
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(thr.x), "+v"(thr.y), "+v"(thr.z), "+v"(thr.w));      // (see hrow_wave_task)
    uint32_t m = 0;
#pragma unroll
    for (int u = 0; u < kHDynV; u++)
        if ((uint32_t)u < nv && v0 + 64u * u < vpr) m = IO<T>::amax_acc(m, v[u]);       // (lanes past the row end hold a duplicate: masked)
    m = wave_max_u32(IO<T>::amax_bits(m));
    if (WPR > 1) {
        __shared__ uint32_t wmax[WPR];
        if (lane == 0) wmax[wv] = m;
        __syncthreads();
        m = wmax[0];
#pragma unroll
        for (int w = 1; w < WPR; w++) m = max(m, wmax[w]);
    }
    const float a = u2f(m) * ratio;
    if (alpha_out && lane == 0 && (WPR == 1 || wv == 0)) st_global(alpha_out + row, a);
    hrow_wave_finish<T, OVP, kHDynV>(v, out + (size_t)row * vpr + v0, v0, vpr, a, gmax, ha, thr, grid, tab, lane, nv);
}

// One tensor per launch, ANTQ_FLAG_DYNAMIC: a row per wavefront (WPR = 1, one wavefront per workgroup) or per workgroup.
template <typename T, bool OVP, int WPR>
__global__ void __launch_bounds__(64 * WPR)
k_fq_hrow_dyn(const uint4 *__restrict__ x, uint4 *__restrict__ out, uint32_t rows, uint32_t vpr, uint32_t nv, float ratio,
              float *__restrict__ alpha_out, float gmax, HArgs ha, const uint4 *__restrict__ tlist, const float *__restrict__ grid)
{
    __shared__ __attribute__((aligned(16))) uint2 tab[WPR][kHSlots * 2];
    const uint32_t wv = threadIdx.x >> 6;
    if (blockIdx.x >= rows) return;                     // (whole workgroups: the barrier inside is never split)
    const uint32_t task = __builtin_amdgcn_readfirstlane(blockIdx.x * WPR + wv);
    hrow_wave_task_dyn<T, OVP, WPR>(x, out, task, vpr, nv, ratio, alpha_out, gmax, ha, tlist, grid, tab[wv], threadIdx.x & 63u, wv);
}

// Vectors per lane and task (4, 3 or 2) of a STATIC row of vpr vectors, or 0 when the row is better left to the fp32-domain
// row table (have_xdom): K1h pays ~150 instructions per task for its table -- short rows (<= 144 vectors: 75 against 82 % of
// 8 TB/s) and rows that only split into tasks with many idle lanes (288 vectors: 75 against 77 %) lose that against the
// 14-instruction closed form of the older kernel; everything else is level or ahead, and far ahead whenever the clocks are
// not settled or the pair rule is on (tools/probe_hrow_rows.py, profiles/r04_hrow_rows*.log).
static inline uint32_t hrow_static_u(uint32_t vpr, bool have_xdom)
{
    uint32_t best_u = 0;
    double best = -1.0;
    for (uint32_t u = 4; u >= 2; u--) {
        const uint32_t span = 64u * u, tasks = (vpr + span - 1u) / span;
        const double util = (double)vpr / (double)(tasks * span);
        if (util > best + 0.02) { best = util; best_u = u; }          // (near ties: the larger task)
    }
    // (2-vector tasks: the table build is half the task -- 11008-wide rows, 1376 vectors, 74 against 81 %)
    if (have_xdom && (vpr < 192u || best < 0.93 || best_u < 3u)) return 0u;
    return best_u;
}

// Shape of a dynamic row job: wavefronts per row and vectors per lane for a row of vpr vectors (128 .. 8192)
struct HDynShape { int wpr, vpt; };      // vpt: vectors per lane in use (the `nv` of the kernels)
static inline HDynShape hrow_dyn_shape(uint32_t vpr)
{
    const int wpr = vpr <= 512u ? 1 : (vpr <= 2048u ? 4 : 16);
    return {wpr, (int)((vpr + 64u * wpr - 1u) / (64u * wpr))};
}
// dynamic LDS per workgroup (occupancy): about 96 KiB of reads in flight per CU (see kHRowLdsPad) -- wavefronts that hold
// 4 KiB are capped at 24 per CU, 6 KiB at 16, 8 KiB at 12; shorter ones are not capped
static inline unsigned hrow_dyn_lds_pad(const HDynShape &sh)
{
    if (sh.wpr != 1 || sh.vpt < 4) return 0u;
    const unsigned per_cu = 96u / (unsigned)sh.vpt;                    // workgroups per CU
    const unsigned lds = (160u * 1024u / per_cu) & ~511u;
    return lds > kHSlots * 16u ? lds - kHSlots * 16u : 0u;
}

// One tensor per launch: WAVES wavefronts per workgroup (wave-private tables, no barrier), one task each.
template <typename T, bool OVP, int VPT, int WAVES = 1>
__global__ void __launch_bounds__(64 * WAVES)
k_fq_hrow(const uint4 *__restrict__ x, uint4 *__restrict__ out, uint32_t total_tasks, uint32_t vpr, uint32_t tpr,
          const float *__restrict__ alpha, int per_row, float gmax, HArgs ha, const uint4 *__restrict__ tlist,
          const float *__restrict__ grid)
{
    __shared__ __attribute__((aligned(16))) uint2 tab[WAVES][kHSlots * 2];
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t task = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wv);
    if (task >= total_tasks) return;
    hrow_wave_task<T, OVP, VPT>(x, out, task, vpr, tpr, alpha, per_row, gmax, ha, tlist, grid, tab[wv], threadIdx.x & 63u);
}

// host: the by-value arguments of the row kernels for one plan / dtype / gmax; false when the plan does not allow the path
static inline bool hargs_from_plan(const void *plan_host, int dtype, float gmax, HArgs &ha)
{
    const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
    const int t = dtype == ANTQ_BF16 ? 0 : dtype == ANTQ_F16 ? 1 : -1;
    if (t < 0 || ph->kind != kPlanLut || !(ph->hdom & (1u << t))) return false;
    ha.n_thr = ph->h_nthr; ha.n_neg = ph->h_nneg;
    ha.hshift = (ph->hshift >> (8 * t)) & 0xffu;
    ha.m = ph->m;
    ha.flim = ph->fastlim * 0.99999f;                  // (the approximate quotient is within 2^-22 of fl(x / s))
    ha.lim = ha.flim < ph->xlim ? ha.flim : ph->xlim;
    const float *g = plan_grid(plan_host);
    ha.vmin = ha.vmax = g[0];
    for (uint32_t i = 1; i < ph->m; i++) { ha.vmin = g[i] < ha.vmin ? g[i] : ha.vmin; ha.vmax = g[i] > ha.vmax ? g[i] : ha.vmax; }
    ha.vout = ph->vout;
    ha.inv_gmax = 1.0 / (double)gmax;
    return true;
}
static inline const uint4 *plan_tlist_dev(const void *plan_host, const void *plan_dev)
{
    return reinterpret_cast<const uint4 *>(static_cast<const char *>(plan_dev) + static_cast<const PlanHeader *>(plan_host)->tlist_off);
}


}  // namespace antq

#endif  // ANTQ_K_HROW_H
