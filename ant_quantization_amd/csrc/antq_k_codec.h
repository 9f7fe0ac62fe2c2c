// antq_k_codec.h -- packed 4-bit codec kernels and their launcher
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_CODEC_H
#define ANTQ_K_CODEC_H

#include "antq_device.h"
#include "antq_k_approx.h"
#include "antq_k_hrow.h"

namespace antq {

// The encoder only needs codes: its LDS table is the plan's a-table with the two values of every entry replaced, once per
// workgroup, by their nibbles (bit 8 = the pair rule's outlier test |v| > 32): {M' (double), code_lo, code_hi}.  One
// ds_read_b128, the exact decision (antq_k_approx.h) and one select per element; no second table read, no dequantised value.
constexpr uint32_t kOutlierBit = 0x100u;
template <bool OVP>
__device__ __forceinline__ uint32_t code_of(uint32_t j, float v, int n_normal)
{
    uint32_t c = (OVP && (int)j >= n_normal) ? j - (uint32_t)n_normal : j;
    c &= 15u;
    if (OVP && fabsf(v) > 32.0f) c |= kOutlierBit;                // OQ:314
    return c;
}
template <bool OVP>
__device__ __forceinline__ void atab_to_codes(const PlanArgs &pa, uint4 *smem, const ATab &A, int n_normal)
{
    for (uint32_t i = threadIdx.x; i < pa.atab_slots; i += blockDim.x) {
        uint4 e = smem[i];
        const uint32_t w = A.idx[i];
        e.z = code_of<OVP>(w & kIdxMask, u2f(e.z), n_normal);
        e.w = code_of<OVP>((w >> 16) & kIdxMask, u2f(e.w), n_normal);
        smem[i] = e;
    }
}
// 8 codes (4 pairs) -> one word of nibbles; OVP: the pair rule on the outlier bits first (OQ:313-320)
template <bool OVP>
__device__ __forceinline__ uint32_t pack_codes(uint32_t (&c)[8])
{
    uint32_t pr[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        uint32_t c0 = c[2 * p], c1 = c[2 * p + 1];
        if (OVP) {
            const bool me = c0 >= kOutlierBit, mo = c1 >= kOutlierBit;
            const bool ve = mo && !me;
            c0 = ve ? 15u : c0;                                  // the identifier 1111 marks the victim
            c1 = me ? 15u : c1;
        }
        pr[p] = (c1 << 4) | c0;                                  // byte 0 = the pair; the outlier bits land in byte 1
    }
    if (OVP) {
        const uint32_t w01 = __builtin_amdgcn_perm(pr[1], pr[0], 0x0c0c0400u);
        const uint32_t w23 = __builtin_amdgcn_perm(pr[3], pr[2], 0x0c0c0400u);
        return __builtin_amdgcn_perm(w23, w01, 0x05040100u);
    }
    return pr[0] | (pr[1] << 8) | (pr[2] << 16) | (pr[3] << 24);
}
// 8 consecutive elements of one row (the first at an index that is a multiple of 8) -> 8 nibbles
template <bool OVP>
__device__ __forceinline__ uint32_t encode_oct_a(const PlanArgs &pa, const uint4 *etab, const float *grid, const ScaleA &sc,
                                                 const float (&x)[8], int n_normal, int zero_code)
{
    float dt[8];
    uint32_t c[8];
    const float flim = pa.fastlim * 0.99999f;
#pragma unroll
    for (int e = 0; e < 8; e++) dt[e] = x[e] * sc.rs;
    bool nan;
    const float dmax = absmax_nan<8>(dt, nan);
    if (sc.ok && !nan && dmax < flim) {                 // false for NaN / Inf / beyond the table's domain
        uint32_t slot[8];
        a_slots<8>(pa, dt, slot);
        const AEnt *tab0 = reinterpret_cast<const AEnt *>(pa.linear ? etab : etab - 2u * pa.kmin);
        AEnt ents[8];
#pragma unroll
        for (int e = 0; e < 8; e++) ents[e] = tab0[slot[e]];      // all reads in flight before the first use
#pragma unroll
        for (int e = 0; e < 8; e++) {
            AEnt ent = ents[e];
            ent.pin();
            c[e] = (__builtin_fma(-ent.Mp(), sc.sd, (double)x[e]) >= 0.0) ? ent.hi() : ent.lo();
        }
    } else {
        // the literal sequence (true division, scan) for this lane's octet
#pragma unroll
        for (int e = 0; e < 8; e++) {
            int jj;
            const float q = scan_lds(x[e] / sc.s, grid, (int)pa.m, jj);
            c[e] = jj == ANTQ_IDX_NONE ? (uint32_t)zero_code : code_of<OVP>((uint32_t)jj, q, n_normal);
        }
    }
    return pack_codes<OVP>(c);
}

// One lane: 8 consecutive elements (4 pairs) of one row -> 4 bytes of codes; VEC: the octet is one (bf16 / f16) or two
// (fp32) 16-byte loads, U octets in flight per thread.
template <typename T> struct Oct {                     // 8 elements <- 16-byte vectors
    __device__ __forceinline__ static void load(const void *p, size_t o, float (&f)[8])
    {
        IO<T>::unpack(ld_stream(static_cast<const uint4 *>(p) + o), f);
    }
};
template <> struct Oct<float> {
    __device__ __forceinline__ static void load(const void *p, size_t o, float (&f)[8])
    {
        const uint4 *v = static_cast<const uint4 *>(p) + 2 * o;
        const uint4 a = ld_stream(v), b = ld_stream(v + 1);
        f[0] = u2f(a.x); f[1] = u2f(a.y); f[2] = u2f(a.z); f[3] = u2f(a.w);
        f[4] = u2f(b.x); f[5] = u2f(b.y); f[6] = u2f(b.z); f[7] = u2f(b.w);
    }
};

// Persistent workgroups (the plan's table is staged once), each looping over tasks of 256 * U octets.
template <typename T, bool OVP, bool VEC, int UE>
__global__ void __launch_bounds__(256)
k_encode4(const void *__restrict__ x, uint32_t *__restrict__ codes, size_t n_oct, size_t row_len, size_t n_tasks,
          const float *__restrict__ alpha, int per_row, float gmax, int n_normal, int zero_code,
          PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    constexpr int U = VEC ? UE : 1;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (pa.adom) tab0 = atab_prefetch<true>(pa, plan_tab);         // exact decision on x (antq_k_approx.h), with indices
    else if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    PlanLds L;
    ATab A;
    if (pa.adom) {
        A = stage_atab<true>(pa, plan_tab, smem, tab0);
        __syncthreads();
        atab_to_codes<OVP>(pa, smem, A, n_normal);
    } else L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    const size_t opr = row_len / 8;                    // row_len % 8 == 0: an octet (4 pairs) lies inside one row -> one scale
    const bool small = n_oct <= 0xffffffffull && opr <= 0xffffffffull;
    const bool pow2 = (opr & (opr - 1)) == 0;          // the usual case: the row index is a shift
    const int rsh = pow2 ? __builtin_ctzll(opr | (1ull << 63)) : 0;
    const double inv = 1.0 / (double)opr;
    const float a0 = per_row ? 1.0f : alpha[0];
    const double inv_gmax = 1.0 / (double)gmax;
    for (size_t task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const size_t first = (task * U) * 256u + threadIdx.x;   // octet index: elements [8o, 8o+8)
        float xf[U][8], a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t o = first + (size_t)u * 256u;
            a[u] = a0;
#pragma unroll
            for (int e = 0; e < 8; e++) xf[u][e] = 0.0f;
            if (o < n_oct) {
                if (per_row) a[u] = alpha[pow2 ? (o >> rsh) : small ? (size_t)oct_row((uint32_t)o, (uint32_t)opr, inv) : o / opr];
                if constexpr (VEC) Oct<T>::load(x, o, xf[u]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; e++) xf[u][e] = IO<T>::load1(x, o * 8 + e);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t o = first + (size_t)u * 256u;
            if (o >= n_oct) continue;
            if (pa.adom) {
                codes[o] = encode_oct_a<OVP>(pa, A.tab, A.grid, make_scale_a(a[u], gmax, inv_gmax), xf[u], n_normal, zero_code);
                continue;
            }
            float of[8];
            int j[8];
            const Scale sc = make_scale(a[u], gmax);
            quant_vec<8, OVP, true>(pa, L, sc, xf[u], of, j);
            uint32_t word = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int jj = j[e];
                uint32_t c;
                if (jj == ANTQ_IDX_VICTIM) c = 15u;
                else if (jj == ANTQ_IDX_NONE) c = (uint32_t)zero_code;
                else if (OVP && jj >= n_normal) c = (uint32_t)(jj - n_normal);
                else c = (uint32_t)jj;
                word |= (c & 15u) << (4 * e);
            }
            codes[o] = word;
        }
    }
}

// ------------------------------------------------------------------------------------
// Row-table encoder (round 3): rows of >= 128 vectors with an x-domain plan (every ANT / OliVe 4-bit codebook).  The exact-
// decision encoder above spends ~23 instructions per element (f64 decision, per-octet scale, table of 16-byte entries with
// 42 % bank conflicts) and is VALU-bound at 46-50 % of HBM for bf16.  Here a wavefront owns a task of ONE row, as in
// k_fq_xrow: lane b moves threshold T_b into the x domain once (x_threshold) and stores {U_b, code_lo | code_hi << 16} --
// an 8-byte slot, the codes already mapped (outlier -> index in the outlier list, bit 8 = the pair rule's outlier test) --
// into a wave-private table; per element: slot from x * rcp(s), one ds_read_b64, one compare, one SDWA select of the
// 16-bit half.  No division, no f64, no dequantised value.  One wavefront per workgroup, no barrier.
// fp32: a lane's vector is half an octet; the two halves of an octet sit in adjacent lanes and meet through one DPP move.
// ------------------------------------------------------------------------------------
// (x >= u) ? w >> 16 : w & 0xffff -- compare + one SDWA select that reads the 16-bit half directly (no shift, no mask)
__device__ __forceinline__ uint32_t select_half_ge(uint32_t w, float x, float u)
{
    uint32_t r;
    asm("v_cmp_ge_f32 vcc, %1, %2\n\t"
        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
        : "=v"(r) : "v"(x), "v"(u), "v"(w) : "vcc");
    return r;
}

template <bool OVP>
__device__ __forceinline__ uint32_t pack_codes4(uint32_t (&c)[4])          // 4 codes (2 pairs) -> 16 bits of nibbles
{
    uint32_t pr[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        uint32_t c0 = c[2 * p], c1 = c[2 * p + 1];
        if (OVP) {
            const bool me = c0 >= kOutlierBit, mo = c1 >= kOutlierBit;
            const bool ve = mo && !me;
            c0 = ve ? 15u : c0;
            c1 = me ? 15u : c1;
        }
        pr[p] = ((c1 & 15u) << 4) | (c0 & 15u);
    }
    return pr[0] | (pr[1] << 8);
}

template <typename T, bool OVP, int U>
__global__ void __launch_bounds__(64)
k_encode4_xrow(const uint4 *__restrict__ x, uint32_t *__restrict__ codes, uint32_t total_tasks, uint32_t vpr, uint32_t tpr,
               const float *__restrict__ alpha, int per_row, float gmax, XArgs xa, const uint4 *__restrict__ entries,
               const float *__restrict__ grid, int n_normal, int zero_code, float flim)
{
    constexpr int EPL = IO<T>::EPL;
    __shared__ __attribute__((aligned(16))) uint2 wtab[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t task = blockIdx.x;
    if (task >= total_tasks) return;
    const uint4 none = make_uint4(f2u(__builtin_inff()), 0u, 0u, 0u);
    const bool two = xa.n_entries > 64u;
    uint4 ent = ld_global(entries + min(lane, xa.n_entries - 1u)), ent2 = none;
    if (two) ent2 = ld_global(entries + min(lane + 64u, xa.n_entries - 1u));
    uint4 v[U];
    float a;
    task_load<T, U>(x, alpha, per_row, task, vpr, tpr, lane, false, v, a);
    __builtin_amdgcn_sched_barrier(0);
    if (lane >= xa.n_entries) ent = none;
    if (lane + 64u >= xa.n_entries) ent2 = none;
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint32_t v0 = g * (64u * U) + lane;
    const size_t base = (size_t)row * vpr + v0;
    const Scale sc = row_scale(a, gmax, xa.inv_gmax);
    const bool rowfast = sc.ok && (sc.s > 0.0f);
    {
        // the wave-private code table: thresholds into the x domain, the two indices of an entry as final codes
        const uint32_t nbp = xa.n_entries - xa.nbneg;
        const bool lin = xa.linear != 0u;
        auto put = [&](uint32_t i, const uint4 &e) {
            bool ok = true;
            float Ux = u2f(e.x);
            if (rowfast && i < xa.n_entries && Ux < __builtin_inff()) Ux = x_threshold(Ux, sc.s, sc.rs, ok);
            const uint32_t cl = code_of<OVP>(e.w & kIdxMask, u2f(e.y), n_normal), ch = code_of<OVP>((e.w >> 16) & kIdxMask, u2f(e.z), n_normal);
            const uint2 w = make_uint2(f2u(Ux), cl | (ch << 16));
            if (i < xa.n_entries) wtab[lin ? i : (i < nbp ? 2u * i : 2u * (i - nbp) + 1u)] = w;
            return w;
        };
        const uint2 w0 = put(lane, ent);
        if (!lin && xa.nbneg == 0u && lane == 0u) wtab[1] = w0;
        if (two) put(lane + 64u, ent2);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
    }
    const char *t0 = reinterpret_cast<const char *>(wtab) - (xa.linear ? 0 : ((int32_t)xa.kmin << 4));
#pragma unroll
    for (int u = 0; u < U; u++) {
        float xf[EPL];
        IO<T>::unpack(v[u], xf);
        uint32_t c[EPL];
        bool fast = rowfast;
        float dt[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            dt[e] = xf[e] * sc.rs;
            // (codes only need the table's DECISION, which holds over its whole domain -- not just below xlim, where the
            //  straight-through arithmetic of the dequantised value is exact too: clipped elements stay on this path)
            fast = fast && (fabsf(dt[e]) < flim);
        }
        if (fast) {
            uint32_t slots[EPL];
            x_slots<EPL>(xa, dt, slots);
            uint2 ents[EPL];
#pragma unroll
            for (int e = 0; e < EPL; e++) ents[e] = *reinterpret_cast<const uint2 *>(t0 + (slots[e] << 3));
#pragma unroll
            for (int e = 0; e < EPL; e++) c[e] = select_half_ge(ents[e].y, xf[e], u2f(ents[e].x));
        } else {
            // clipped far beyond the grid, NaN / Inf, or a scale outside the table path's range: the literal sequence
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                int jj;
                const float q = scan_lds(xf[e] / sc.s, grid, (int)xa.m, jj);
                c[e] = jj == ANTQ_IDX_NONE ? (uint32_t)zero_code : code_of<OVP>((uint32_t)jj, q, n_normal);
            }
        }
        const bool live = v0 + 64u * u < vpr;
        if constexpr (EPL == 8) {
            const uint32_t word = pack_codes<OVP>(c);
            if (live) __builtin_nontemporal_store(word, codes + base + 64u * u);
        } else {
            const uint32_t half = pack_codes4<OVP>(c);
            const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)half, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
            // vectors 2k / 2k + 1 of the tensor are octet k: the even lane stores both halves (vpr even: row_len % 8 == 0)
            if (live && (lane & 1u) == 0u) __builtin_nontemporal_store(half | (other << 16), codes + ((base + 64u * u) >> 1));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------
// 16-bit-domain row encoder (round 5): the K1h treatment (antq_k_hrow.h) for the codes.  A bf16 / f16 element never leaves
// its 16 bits: the wave-private slot table is keyed by the element's own (sign, exponent, top mantissa bits) and holds the
// threshold PATTERN in word 0 and, in word 1, two BYTES instead of two output patterns:
//     byte 0 = code of the region below the threshold (by magnitude), byte 1 = code at / above it
//     code byte = nibble (scan-order index; an outlier: its index in the outlier list) | 0x10 if |v| > 32 (OQ:314)
// Per element: v_bfe (key) + v_med3 (clamp) + v_alignbit (sign) + v_lshl_add (address) + ds_read_b64 + v_cmp +
// v_cndmask_sdwa that drops the chosen byte straight into byte i of an accumulator -- the low elements of a vector's four
// words collect in A, the high ones in B, and the vector's 32 bits of nibbles are A | B << 4 (one v_lshl_or); the pair rule is
// five more word-wide logic operations on the 0x10 flags of all four pairs at once.  No conversion to fp32, no multiply
// by 1 / s, no per-pair shifts and masks: 6.5 + ~1 VALU instructions per element against 16 in the fp32-domain row encoder.
// The sentinel slot's threshold is the pattern of the table's DECISION limit (flim), not of the narrower straight-through
// limit the fake-quant kernel needs: a far-clipped element keeps the extreme code of its sign on the fast path, and only
// Inf / NaN / magnitudes beyond ~2^20 grid units reach the sentinel byte 0x80 -- such a lane redoes its vector literally.
// ------------------------------------------------------------------------------------
constexpr uint32_t kHCodeSentinel = 0x80u;
constexpr uint32_t kHCodeOutlier = 0x10u;

template <bool OVP>
__device__ __forceinline__ uint32_t hcode_byte(uint32_t j, bool outlier, uint32_t n_normal)
{
    uint32_t c = (OVP && j >= n_normal) ? j - n_normal : j;
    c &= 15u;
    if (OVP && outlier) c |= kHCodeOutlier;
    return c;
}

// hrow_build (antq_k_hrow.h) with codes instead of output patterns; same slot geometry, same conditions for `fast`.
template <typename T, bool OVP>
__device__ __forceinline__ HRow hrow_build_codes(const HArgs &ha, const uint4 &thr, const Scale &sc, uint2 *tab, uint32_t lane,
                                                 uint32_t n_normal)
{
    HRow R;
    const float s = sc.s;
    bool ok = sc.ok && (s > 0.0f);
    const bool mine = lane < ha.n_thr;
    const bool neg = lane < ha.n_neg;
    const float limx = ha.lim * s * 0.999f;
    const uint32_t lim16 = ok ? H16<T>::down(fminf(limx, 3.0e38f)) : 0u;
    const uint32_t klim = lim16 >> ha.hshift;
    // the sentinel slot decides on the pattern of the DECISION limit (>= lim16: keys beyond klim clamp into this slot)
    const uint32_t flim16 = ok ? max(H16<T>::down(fminf(ha.flim * s * 0.999f, 3.0e38f)), lim16) : 0u;
    bool tok;
    const float Tt = mine ? u2f(thr.x) : 1.0f;
    const float U = x_threshold(Tt, ok ? s : 1.0f, sc.rs, tok);
    uint32_t t16 = neg ? H16<T>::down(-U) + 1u : H16<T>::up(U);
    const uint32_t key = mine ? (t16 >> ha.hshift) : 0u;
    const uint32_t o_lo = hcode_byte<OVP>((thr.w >> 8) & 0x3ffu, (thr.w & 1u) != 0u, n_normal);
    const uint32_t o_hi = hcode_byte<OVP>((thr.w >> 18) & 0x3ffu, (thr.w & 2u) != 0u, n_normal);
    const uint32_t first = neg ? o_hi : o_lo, second = neg ? o_lo : o_hi;      // by rising MAGNITUDE
    const uint32_t k_prev = (uint32_t)__shfl((int)key, (int)((lane + 63u) & 63u), 64);
    const uint32_t k_next = (uint32_t)__shfl((int)key, (int)((lane + 1u) & 63u), 64);
    const bool last_of_side = neg ? (lane == 0u) : (lane + 1u == ha.n_thr);
    const bool first_of_side = neg ? (lane + 1u == ha.n_neg) : (lane == ha.n_neg);
    const uint32_t nxt = last_of_side ? klim : (neg ? k_prev : k_next);
    const uint32_t kpos = ha.n_neg < ha.n_thr ? (uint32_t)__builtin_amdgcn_readlane((int)key, (int)ha.n_neg) : 0xffffffffu;
    const uint32_t kneg = ha.n_neg > 0u ? (uint32_t)__builtin_amdgcn_readlane((int)key, (int)(ha.n_neg - 1u)) : 0xffffffffu;
    const uint32_t kmin = min(kpos, kneg);
    const bool lane_ok = !mine || (key < nxt);
    ok = ok && (klim >= kmin) && (klim - kmin < kHSlots) && (__ballot(lane_ok) == ~0ull);
    R.kmin = kmin; R.klim = klim; R.fast = ok;
    if (!ok) return R;
    auto slot = [&](uint32_t k, bool ng) -> uint2 & { return tab[((k - kmin) << 1) + (ng ? 1u : 0u)]; };
    const uint32_t sbit = neg ? 0x80000000u : 0u;
    if (mine) {
        slot(key, neg) = make_uint2(sbit | (t16 << 16), first | (second << 8));
#pragma clang loop vectorize(disable) unroll(disable)
        for (uint32_t k = key + 1u; k < nxt; k++) slot(k, neg) = make_uint2(kHNoThr, second | (second << 8));
        if (first_of_side) {
#pragma clang loop vectorize(disable) unroll(disable)
            for (uint32_t k = kmin; k < key; k++) slot(k, neg) = make_uint2(kHNoThr, first | (first << 8));
        }
        if (last_of_side) slot(klim, neg) = make_uint2(sbit | (flim16 << 16), second | (kHCodeSentinel << 8));
    }
    if (ha.n_neg == 0u || ha.n_neg == ha.n_thr) {
        const bool ng = ha.n_neg == 0u;
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)first, ng ? 0 : (int)(ha.n_thr - 1u));
#pragma clang loop vectorize(disable) unroll(disable)
        for (uint32_t k = kmin + lane; k < klim; k += 64u) slot(k, ng) = make_uint2(kHNoThr, o | (o << 8));
        if (lane == 0u) slot(klim, ng) = make_uint2((ng ? 0x80000000u : 0u) | (flim16 << 16), o | (kHCodeSentinel << 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
    return R;
}

// The two elements of word I (0..3) of a vector: their code bytes into byte I of A (low element) and B (high element).
#define ANTQ_HCODE_SEL(DST, SRCW, ENT, BYTE)                                                                                   \
    asm("v_cmp_ge_u32 vcc, %1, %2\n\t"                                                                                         \
        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:" BYTE " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1"       \
        : "+v"(DST) : "v"(SRCW), "v"((ENT).x), "v"((ENT).y) : "vcc")
#define ANTQ_HCODE_SEL0(DST, SRCW, ENT)                                                                                        \
    asm("v_cmp_ge_u32 vcc, %1, %2\n\t"                                                                                         \
        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"              \
        : "=v"(DST) : "v"(SRCW), "v"((ENT).x), "v"((ENT).y) : "vcc")
template <int I>
__device__ __forceinline__ void hcode_pair(uint32_t w, uint32_t tbase, uint32_t vkmin, uint32_t vklim, uint32_t kpos, uint32_t vkwid,
                                           uint32_t &A, uint32_t &B)
{
    uint32_t th, tl, wl;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(th) : "v"(w), "s"(kpos), "v"(vkwid));
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(th) : "v"(th), "v"(vkmin), "v"(vklim));
    wl = w << 16;
    th = __builtin_amdgcn_alignbit(th, w, 31);
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(tl) : "v"(wl), "s"(kpos), "v"(vkwid));
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(tl) : "v"(tl), "v"(vkmin), "v"(vklim));
    tl = __builtin_amdgcn_alignbit(tl, wl, 31);
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t eh = *(const __attribute__((address_space(3))) u32x2_t *)(uintptr_t)((th << 3) + tbase);
    const u32x2_t el = *(const __attribute__((address_space(3))) u32x2_t *)(uintptr_t)((tl << 3) + tbase);
    if constexpr (I == 0) { ANTQ_HCODE_SEL0(A, wl, el); ANTQ_HCODE_SEL0(B, w, eh); }        // (the first byte pads the rest with zeros)
    if constexpr (I == 1) { ANTQ_HCODE_SEL(A, wl, el, "BYTE_1"); ANTQ_HCODE_SEL(B, w, eh, "BYTE_1"); }
    if constexpr (I == 2) { ANTQ_HCODE_SEL(A, wl, el, "BYTE_2"); ANTQ_HCODE_SEL(B, w, eh, "BYTE_2"); }
    if constexpr (I == 3) { ANTQ_HCODE_SEL(A, wl, el, "BYTE_3"); ANTQ_HCODE_SEL(B, w, eh, "BYTE_3"); }
}
#undef ANTQ_HCODE_SEL
#undef ANTQ_HCODE_SEL0

// A (low elements) / B (high elements) of the four pairs of a vector -> its 8 nibbles; OVP: the pair rule (OQ:313-320) on the
// 0x10 flags of all four pairs at once: the even element is a victim when its odd partner is an outlier and it is not one
// itself, the odd one when its even partner is an outlier; a victim's nibble becomes the identifier 15.
template <bool OVP>
__device__ __forceinline__ uint32_t hcode_pack(uint32_t A, uint32_t B)
{
    if (OVP) {
        const uint32_t me = A & 0x10101010u, mo = B & 0x10101010u;
        const uint32_t ve = mo & ~me;
        A = (A & 0x0f0f0f0fu) | (ve - (ve >> 4));        // 0x10 -> 0x0f in the bytes of the victims
        B = (B & 0x0f0f0f0fu) | (me - (me >> 4));
    }
    return A | (B << 4);
}

// The literal reference sequence for one vector (rare): d = x / s, scan, the code of the winner (no entry within 102400:
// the code of the grid's zero), pair rule.
template <typename T, bool OVP>
__device__ __forceinline__ uint32_t hcode_exact(const uint4 &v, float s, const float *__restrict__ grid, uint32_t m, int n_normal,
                                                int zero_code)
{
    uint32_t c[8];
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
        const uint32_t w = vec_word(v, i);
        const float d0 = H16<T>::val(w & 0xffffu) / s, d1 = H16<T>::val(w >> 16) / s;
        float q0 = 0.0f, q1 = 0.0f, m0 = 102400.0f, m1 = 102400.0f;
        int j0 = ANTQ_IDX_NONE, j1 = ANTQ_IDX_NONE;
#pragma unroll 1
        for (uint32_t k = 0; k < m; k++) {
            const float g = ld_global(grid + k);
            const float s0 = fabsf(d0 - g), s1 = fabsf(d1 - g);
            if (s0 <= m0) { m0 = s0; q0 = g; j0 = (int)k; }
            if (s1 <= m1) { m1 = s1; q1 = g; j1 = (int)k; }
        }
        const uint32_t c0 = j0 == ANTQ_IDX_NONE ? (uint32_t)zero_code : code_of<OVP>((uint32_t)j0, q0, n_normal);
        const uint32_t c1 = j1 == ANTQ_IDX_NONE ? (uint32_t)zero_code : code_of<OVP>((uint32_t)j1, q1, n_normal);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i == k) { c[2 * k] = c0; c[2 * k + 1] = c1; }
    }
    return pack_codes<OVP>(c);
}

template <typename T, bool OVP, int VPT, int MEM = 0>          // MEM (experiment): bit 0 plain loads, bit 1 plain stores
__global__ void __launch_bounds__(64)
k_encode4_hrow(const uint4 *__restrict__ x, uint32_t *__restrict__ codes, uint32_t total_tasks, uint32_t vpr, uint32_t tpr,
               const float *__restrict__ alpha, int per_row, float gmax, HArgs ha, const uint4 *__restrict__ tlist,
               const float *__restrict__ grid, int n_normal, int zero_code)
{
    __shared__ __attribute__((aligned(16))) uint2 tab[kHSlots * 2];
    const uint32_t lane = threadIdx.x;
    const uint32_t task = __builtin_amdgcn_readfirstlane(blockIdx.x);
    if (task >= total_tasks) return;
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    uint4 thr = ld_global(tlist + min(lane, ha.n_thr - 1u));
    const float a = ld_global(alpha + (per_row ? row : 0));
    const uint32_t v0 = g * (64u * VPT) + lane;
    const uint4 *p = x + (size_t)row * vpr;
    uint4 v[VPT];
#pragma unroll
    for (int u = 0; u < VPT; u++) v[u] = (MEM & 1) ? ld_global(p + min(v0 + 64u * u, vpr - 1u)) : ld_stream(p + min(v0 + 64u * u, vpr - 1u));
    __builtin_amdgcn_sched_barrier(0);                 // (see hrow_wave_task: the table is built while the loads are in flight)
    asm volatile("" : "+v"(thr.x), "+v"(thr.y), "+v"(thr.z), "+v"(thr.w));
    const Scale sc = row_scale(a, gmax, ha.inv_gmax);
    const HRow R = hrow_build_codes<T, OVP>(ha, thr, sc, tab, lane, (uint32_t)n_normal);
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)tab;
    const uint32_t kpos = 16u + ha.hshift;
    uint32_t vkmin = R.kmin, vklim = R.klim, vkwid = 15u - ha.hshift;
    asm volatile("" : "+v"(vkmin), "+v"(vklim), "+v"(vkwid));          // (VGPR copies: gfx9 takes one SGPR per VOP3)
    const uint32_t tbase = tab_addr - (R.kmin << 4);
    uint32_t *out = codes + (size_t)row * vpr + v0;                    // one word of nibbles per 16-byte vector
    uint32_t acc = 0u;
    if (R.fast) {
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            uint32_t A, B;
            hcode_pair<0>(v[u].x, tbase, vkmin, vklim, kpos, vkwid, A, B);
            hcode_pair<1>(v[u].y, tbase, vkmin, vklim, kpos, vkwid, A, B);
            hcode_pair<2>(v[u].z, tbase, vkmin, vklim, kpos, vkwid, A, B);
            hcode_pair<3>(v[u].w, tbase, vkmin, vklim, kpos, vkwid, A, B);
            acc |= A | B;
            if (v0 + 64u * u < vpr) {
                if (MEM & 2) out[64u * u] = hcode_pack<OVP>(A, B);
                else __builtin_nontemporal_store(hcode_pack<OVP>(A, B), out + 64u * u);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (__builtin_expect((acc & 0x80808080u) == 0u, 1)) return;
    }
    // rare: an element at or beyond the table's decision limit (Inf, NaN, absurd magnitudes), or a row without a table
    const uint32_t slot0 = R.klim << ha.hshift;
#pragma unroll 1
    for (uint32_t u = 0; u < (uint32_t)VPT; u++) {
        uint4 cur = v[0];
#pragma unroll
        for (int k = 1; k < VPT; k++)
            if (u == (uint32_t)k) cur = v[k];
        if (v0 + 64u * u < vpr) {
            const uint32_t top = IO<T>::amax_acc(0u, cur);
            if (!R.fast || max(top & 0xffffu, top >> 16) >= slot0)
                __builtin_nontemporal_store(hcode_exact<T, OVP>(cur, sc.s, grid, ha.m, n_normal, zero_code), out + 64u * u);
        }
    }
}

// Decoder.  VEC: one 16-byte vector of output per thread and access (EPL = 4 fp32 / 8 bf16 elements = 2 / 4 bytes of
// codes), 4 in flight: every store instruction of a wavefront covers 1 KiB contiguous.  Otherwise 8 elements per thread
// with element stores.
template <typename T, bool OVP, bool VEC>
__global__ void __launch_bounds__(256)
k_decode4(const uint32_t *__restrict__ codes, void *__restrict__ out, size_t n_oct, size_t row_len,
          const float *__restrict__ alpha, int per_row, float gmax, int n_normal,
          const float *__restrict__ grid, int m)
{
    constexpr int EPL = VEC ? IO<T>::EPL : 8;          // elements per thread and access
    constexpr int U = VEC ? 4 : 1;
    __shared__ float g[32];
    // (+ 0.0f: a codebook's -0 decodes to +0, which is what the reference's (q - d) + d makes of it for every finite d)
    if (threadIdx.x < 32) g[threadIdx.x] = ((int)threadIdx.x < m) ? grid[threadIdx.x] + 0.0f : 0.0f;
    const size_t n_units = n_oct * (8 / EPL);
    const size_t first = ((size_t)blockIdx.x * U) * 256u + threadIdx.x;
    uint32_t word[U];
    float a[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t t = first + (size_t)u * 256u;
        word[u] = 0u;
        a[u] = 1.0f;
        if (t < n_units) {
            word[u] = (EPL == 8) ? codes[t] : (uint32_t) reinterpret_cast<const uint16_t *>(codes)[t];
            a[u] = alpha[per_row ? (t * EPL / row_len) : 0];     // row_len % 8 == 0: a unit lies inside one row
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t t = first + (size_t)u * 256u;
        if (t >= n_units) continue;
        const float s = a[u] / gmax;                  // AQ:536 scale = alpha / max(grid)
        float of[EPL];
#pragma unroll
        for (int p = 0; p < EPL / 2; p++) {
            const uint32_t c0 = (word[u] >> (8 * p)) & 15u, c1 = (word[u] >> (8 * p + 4)) & 15u;
            float q0, q1;
            if (OVP) {
                // identifier 15 in one nibble: that element is the victim (0), its partner an outlier
                q0 = (c0 == 15u) ? 0.0f : ((c1 == 15u) ? g[n_normal + c0] : g[c0]);
                q1 = (c1 == 15u) ? 0.0f : ((c0 == 15u) ? g[n_normal + c1] : g[c1]);
            } else {
                q0 = g[c0];
                q1 = g[c1];
            }
            of[2 * p] = q0 * s;
            of[2 * p + 1] = q1 * s;
        }
        if constexpr (VEC) st_stream(static_cast<uint4 *>(out) + t, IO<T>::pack(of));
        else {
#pragma unroll
            for (int e = 0; e < EPL; e++) IO<T>::store1(out, t * EPL + e, of[e]);
        }
    }
}

template <typename T>
static int launch_codec(bool enc, const void *x, void *codes_or_out, const uint8_t *codes_in, size_t rows, size_t row_len,
                        const float *alpha, int per_row, float gmax, const PlanArgs &pa, const void *plan_host,
                        const void *plan_dev, int n_normal, bool ovp, hipStream_t st)
{
    const size_t n = rows * row_len;
    if (row_len % 8 != 0) return ANTQ_ERR_UNSUPPORTED;
    const int m = (int)pa.m;
    if (ovp) { if (n_normal < 1 || n_normal > 15 || m - n_normal > 15 || m - n_normal < 0) return ANTQ_ERR_UNSUPPORTED; }
    else if (m > 16) return ANTQ_ERR_UNSUPPORTED;
    const size_t n_oct = n / 8;
    const bool vec = reinterpret_cast<uintptr_t>(enc ? x : codes_or_out) % 16 == 0;     // 16-byte vector I/O
    constexpr int kEncU = 2;                           // octets per thread and task when encoding
    const size_t per_block = vec ? (enc ? 256 * kEncU : 1024 * IO<T>::EPL / 8) : 256;     // octets per workgroup (task)
    const size_t n_tasks = (n_oct + per_block - 1) / per_block;
    size_t blocks = n_tasks;
    if (enc && blocks > (size_t)g_knob_encwg) blocks = (size_t)g_knob_encwg;   // persistent workgroups
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    const float *grid_host = plan_grid(plan_host);
    int zero_code = 0;
    for (int i = 0; i < (ovp ? n_normal : m); i++) if (grid_host[i] == 0.0f) zero_code = i;
    const dim3 gd((unsigned)blocks), bd(256);
    if constexpr (!std::is_same<T, float>::value) {
        // bf16 / f16 rows of >= 128 vectors with a 16-bit-domain plan: the K1h encoder (knob 9 = 0: the fp32-domain row encoder)
        HArgs ha;
        const size_t rl = per_row ? row_len : n;
        const size_t vpr = rl / 8;
        if (enc && vec && g_knob_h != 0 && vpr >= kRowKernelMinVpr && vpr <= 0xffffffffull &&
            hargs_from_plan(plan_host, IO<T>::DTYPE, gmax, ha)) {
            // vectors per lane and task: the largest of 8 / 4 / 3 / 2 that keeps the lanes of a row's tasks busy (the table is
            // built once per task); knob 0 forces a size (A/B)
            uint32_t U = 2;
            double best = -1.0;
            for (uint32_t u : {8u, 4u, 3u, 2u}) {
                const size_t span = 64u * u, tasks = (vpr + span - 1) / span;
                const double util = (double)vpr / (double)(tasks * span);
                if (util > best + 0.02) { best = util; U = u; }
            }
            if ((g_knob_u >= 2 && g_knob_u <= 4) || g_knob_u == 8) U = (uint32_t)g_knob_u;
            const size_t tpr = (vpr + 64 * U - 1) / (64 * U), total = (per_row ? rows : 1) * tpr;
            if (total <= 0x7fffffffull) {
                const uint4 *tl = plan_tlist_dev(plan_host, plan_dev);
                const float *grid = reinterpret_cast<const float *>(plan_tab_ptr(plan_dev));
                const unsigned pad = g_knob_hlds >= 0 ? (unsigned)g_knob_hlds : 0u;
#define ANTQ_ENCH(O, U_) hipLaunchKernelGGL((k_encode4_hrow<T, O, U_>), dim3((unsigned)total), dim3(64), pad, st, static_cast<const uint4 *>(x), \
                                        static_cast<uint32_t *>(codes_or_out), (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, \
                                        gmax, ha, tl, grid, n_normal, zero_code)
#define ANTQ_ENCHM(O, U_, M_) hipLaunchKernelGGL((k_encode4_hrow<T, O, U_, M_>), dim3((unsigned)total), dim3(64), pad, st, static_cast<const uint4 *>(x), \
                                        static_cast<uint32_t *>(codes_or_out), (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, \
                                        gmax, ha, tl, grid, n_normal, zero_code)
                // whole 8 KiB tasks store their 256-byte code words through the cache (plain stores: 71.1 -> 76.5 % ANT, 70.4 ->
                // 75.4 % OliVe on 16384 x 8192 bf16; nontemporal loads stay -- plain loads cost 4 points); knob 13 = 1: nontemporal
                // stores (A/B).  profiles/r05_codec_h.log
                if (U == 8 && g_knob_exp != 1) { if (ovp) ANTQ_ENCHM(true, 8, 2); else ANTQ_ENCHM(false, 8, 2); }
                else
                if (U == 8) { if (ovp) ANTQ_ENCH(true, 8); else ANTQ_ENCH(false, 8); }
                else if (U == 4) { if (ovp) ANTQ_ENCH(true, 4); else ANTQ_ENCH(false, 4); }
                else if (U == 3) { if (ovp) ANTQ_ENCH(true, 3); else ANTQ_ENCH(false, 3); }
                else { if (ovp) ANTQ_ENCH(true, 2); else ANTQ_ENCH(false, 2); }
#undef ANTQ_ENCH
#undef ANTQ_ENCHM
                return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
            }
        }
    }
    if (enc && vec && g_knob_x != 0 && g_knob_lane_rows != 2) {
        // rows of >= 128 vectors with an x-domain plan: the row-table encoder (knob 2 = 0 or knob 5 = 2: the element encoder, A/B)
        const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
        const size_t rl = per_row ? row_len : n;
        const size_t vpr = rl / IO<T>::EPL;
        if (pa.kind == kPlanLut && ph->xdom && vpr >= kRowKernelMinVpr && vpr <= 0xffffffffull && rl % 8 == 0) {
            // 4 vectors per lane for rows of >= 256 vectors (the encoder is VALU-bound: the per-task table build is then shared
            // by twice the elements: +3 ... +5 points, tools/probe_codec_u.py; knob 0 = 2 / 4 / 8 forces a size for A/B)
            const uint32_t U = (g_knob_u == 2 || g_knob_u == 4 || g_knob_u == 8) ? (uint32_t)g_knob_u : (vpr >= 256 ? 4u : 2u);
            const size_t tpr = (vpr + 64 * U - 1) / (64 * U), total = (per_row ? rows : 1) * tpr;
            if (total <= 0x7fffffffull) {
                XArgs xa = xargs_from_plan(plan_host, pa);
                xa.inv_gmax = 1.0 / (double)gmax;
                const uint4 *tab = plan_tab_ptr(plan_dev);
#define ANTQ_ENCX(O, U_) hipLaunchKernelGGL((k_encode4_xrow<T, O, U_>), dim3((unsigned)total), dim3(64), 0, st, static_cast<const uint4 *>(x), \
                                        static_cast<uint32_t *>(codes_or_out), (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, \
                                        gmax, xa, tab + (pa.m_pad >> 2), reinterpret_cast<const float *>(tab), n_normal, zero_code,    \
                                        pa.fastlim * 0.99999f)
                if (U == 8) { if (ovp) ANTQ_ENCX(true, 8); else ANTQ_ENCX(false, 8); }
                else if (U == 4) { if (ovp) ANTQ_ENCX(true, 4); else ANTQ_ENCX(false, 4); }
                else        { if (ovp) ANTQ_ENCX(true, 2); else ANTQ_ENCX(false, 2); }
#undef ANTQ_ENCX
                return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
            }
        }
    }
    if (enc) {
        const size_t lds = lds_table(pa, true);
        uint32_t *codes = static_cast<uint32_t *>(codes_or_out);
#define ANTQ_ENC(O, V) hipLaunchKernelGGL((k_encode4<T, O, V, kEncU>), gd, bd, lds, st, x, codes, n_oct, row_len, n_tasks, alpha, per_row, gmax, \
                                          n_normal, zero_code, pa, plan_tab_ptr(plan_dev))
        if (ovp) { if (vec) ANTQ_ENC(true, true); else ANTQ_ENC(true, false); }
        else     { if (vec) ANTQ_ENC(false, true); else ANTQ_ENC(false, false); }
#undef ANTQ_ENC
    } else {
        const float *grid_dev = reinterpret_cast<const float *>(plan_tab_ptr(plan_dev));
        const uint32_t *codes = reinterpret_cast<const uint32_t *>(codes_in);
#define ANTQ_DEC(O, V) hipLaunchKernelGGL((k_decode4<T, O, V>), gd, bd, 0, st, codes, codes_or_out, n_oct, row_len, alpha, per_row, \
                                          gmax, n_normal, grid_dev, m)
        if (ovp) { if (vec) ANTQ_DEC(true, true); else ANTQ_DEC(true, false); }
        else     { if (vec) ANTQ_DEC(false, true); else ANTQ_DEC(false, false); }
#undef ANTQ_DEC
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

}  // namespace antq

#endif  // ANTQ_K_CODEC_H
