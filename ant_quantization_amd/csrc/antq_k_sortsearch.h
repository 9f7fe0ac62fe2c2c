// antq_k_sortsearch.h -- clip search from the SORTED row: every codebook and every candidate of a type selection on one sort
// Part of libantq's calibration translation unit (antq_search.hip includes it); gfx950 only.
//
// search_mse (AQ/quant_modules.py:287-326, OQ:189-233) scores C clip candidates per row, search_adaptive_numeric_type
// (AQ:328-415, OQ:235-256) does so for every candidate codebook.  For a fixed scale s the quantiser is a step function of x
// (antq_k_sweep.h), so a candidate's squared error is a closed form in N_k = #{x >= X_k} and S_k = sum{x : x >= X_k} over its
// x-domain thresholds X_k -- and both are ONE binary search away once the row is sorted and its prefix sums are known:
//     sum_x (O_J(x) - x)^2 = sum x^2 + n O_b^2 - 2 O_b S + sum_{k >= b} [ (O_{k+1}^2 - O_k^2) N_k - 2 (O_{k+1} - O_k) S_k ]
//                                                        - sum_{k <  b} [ (O_{k+1}^2 - O_k^2) M_k - 2 (O_{k+1} - O_k) R_k ]
// (M_k, R_k: count and sum of the elements BELOW X_k; b: the first non-negative threshold, so O_b is the value next to zero.
// Written around the cell that holds zero, every term counts only the elements beyond its threshold as seen from zero: with
// the sums anchored at the most negative value O_0, a row whose statistic is large next to its elements -- one huge element in
// a row of small ones -- cancels n O_0^2 against the rest and loses 1e-6 of the result in double).
// The threshold sweep (antq_k_sweep.h) pays per element and per threshold that sweeps across it (LDS atomics; OliVe's 28
// thresholds over a 75 .. 250 % clip range: half a dozen per element, slower than the direct kernels); here the elements pay
// one sort (a 4096-element bitonic network: 78 compare-exchange steps, ALL of them on registers) shared by every codebook of
// the launch, and the (codebook, candidate, threshold) triples pay 13 LDS probes each, whatever the clip range.
//
//   chunk    4096 elements of a row (a row is walked chunk by chunk; a tensor with ONE scale is spread over many workgroups)
//   keys     the float's bits mapped to an unsigned integer of the same order (-0 folded into +0); elements that are no
//            step-function elements (NaN / Inf / far-clipped: |x| >= lim * s_min) carry the sentinel 0xffffffff: they sort to
//            the end, count for nothing, and are evaluated literally (the reference sequence) for every candidate
//   sort     256 threads x 16 keys.  The index bits of the network that are REGISTER bits change with the layout: bits 0-3
//            (blocked), 4-7, 8-11 (6-9 inside a wavefront); a size-2^s merge walks its strides top down through at most three
//            layouts, moving between them through LDS (20 transposes in all, 14 of them inside a wavefront's own quarter of
//            the buffer without a workgroup barrier; address = i + (i >> 4): every layout nearly conflict-free).  The
//            first step of a merge (partner i ^ (2^s - 1)) is folded into the transposing READ (the upper half block is read
//            mirrored), so every comparator sorts ascending: v_min_u32 + v_max_u32 per pair.
//   sums     x in fixed point (2^-38 of the row statistic's binade, exact for every element within 2^15 of it: antq_k_sweep.h),
//            prefix sums as 64-bit integers at every second sorted position (every fourth with the pair rule: sort_psh); a
//            probe adds the one (three) elements in between
//   search   work item = (codebook, candidate, group of 4 thresholds): 4 interleaved binary searches, the closed form's
//            terms in double in ONE fixed order -- the sums of a (codebook, candidate) do not depend on which other codebooks
//            or candidates share the launch
//   pairs    OliVe's pair rule (OQ:311-320): every element stays in the sorted set with its plain step-function error; a
//            pair one of whose members can be an outlier under SOME candidate is also put on a list, and each (codebook,
//            candidate) adds the victim's correction v^2 - (O(v) - v)^2 for the pairs that hold an outlier under IT (exact:
//            a victim's output is 0 * s).  Pairs with a member that is no step-function element take the literal sequence.
//
// Rows whose candidate scales are unusable (an Inf / denormal statistic, a non-monotone ratio list) take the literal sequence
// for every element: slow (0.2 - 1 ms per such row), never wrong; a zero or NaN statistic gives NaN for every candidate, said
// directly.  Rows of at most 1024 elements under codebooks without the pair rule: k_search_sorted_short below (one row per
// wavefront).  A tensor with ONE scale: the PT form of the kernel + k_sort_pt_total / k_sort_pt_finish.
#ifndef ANTQ_K_SORTSEARCH_H
#define ANTQ_K_SORTSEARCH_H

#include <type_traits>

#include "antq_device.h"
#include "antq_k_fakequant.h"
#include "antq_k_search.h"
#include "antq_k_sweep.h"

namespace antq {

constexpr int kSortB = 12, kSortR = 4;                       // 2^12 keys per chunk, 2^4 per thread
constexpr int kSortK = 1 << kSortB, kSortEPT = 1 << kSortR, kSortNT = 1 << (kSortB - kSortR);
constexpr int kSortPad = kSortK + (kSortK >> kSortR);        // dwords of the key buffer (address = i + (i >> R))
constexpr int kSortKS = 4;                                   // thresholds per work item
constexpr uint32_t kSortSent = 0xffffffffu;
// A thread's partial terms live in REGISTERS across the chunks of a row (LDS holds what every thread reads; 10 - 15 KB of
// accumulators were the difference between two and three workgroups per CU): at most kSortNI (type, candidate, threshold
// group) items, kSortNC pair-correction items and kSortNL (type, candidate) literal sums per thread -- the launcher cuts the
// candidate list into pieces that fit.
constexpr int kSortNI = 6, kSortNC = 3, kSortNL = 2;
constexpr uint32_t kSortTy = 332;                            // dwords of a type's block in LDS: values [66], thresholds T [64], scalars [130..139]
                                                             // (n_thr, kout_pos, kout_neg, gmax, lim, m, nneg, even-mantissa masks [137..138]),
                                                             // rounding boundaries M_k as doubles [140..267], the codebook in scan order
                                                             // [268..331] (literal elements; codebooks of more than 64 values: from memory)
constexpr uint32_t kSortGridLds = 64;
// Prefix sums at every 2^PSH-th sorted position.  A probe converts the (up to 2^PSH - 1) keys between the stored position and
// its own back to fixed point -- 10 instructions each, a quarter of a look-up at every fourth -- so every SECOND position is
// kept (8 KB more LDS than every fourth).  With the pair rule that only fits next to three workgroups per CU because OliVe's
// launches do not keep their 20 KB table of x-domain thresholds (sort_inline): a (codebook, candidate, threshold) is moved
// into the x domain by the item that probes it -- once per chunk instead of once per row, which a row of several chunks
// pays back through the cheaper probes -- and the pair rule's own look-ups work in the grid domain (RN(x / s) >= T_k <=> x >= X_k).
__host__ __device__ constexpr int sort_psh(bool ovp) { (void)ovp; return 1; }
__host__ __device__ constexpr bool sort_inline(bool ovp) { return ovp; }

struct SortTypes {
    SweepType ty[kMaxTypes];
    uint32_t nneg[kMaxTypes];    // thresholds below zero (PlanHeader::h_nneg): the closed form is written around the cell that holds 0
    int ntypes;
    uint32_t nthr_pad;           // max n_thr over the types, rounded up to a multiple of kSortKS
};

// LDS of a workgroup: keys | prefix sums | thresholds | scales | tables
struct SortLds {
    uint32_t off_p4, off_x, off_s, off_v, off_misc, total;
};
__host__ __device__ inline SortLds sort_lds(uint32_t ntc, uint32_t nthr_pad, int ntypes, bool ovp)
{
    SortLds L;
    uint32_t o = (uint32_t)kSortPad * 4u;                    // keys
    o = (o + 15u) & ~15u;
    L.off_p4 = o;    o += (((uint32_t)kSortK >> sort_psh(ovp)) + 1u) * 8u;      // prefix sums, and the total
    o = (o + 15u) & ~15u;
    L.off_x = o;     o += sort_inline(ovp) ? ntc * 2u * 4u : ntc * nthr_pad * 4u;     // thresholds as keys (inline: the two outlier bounds only)
    L.off_s = o;     o += ntc * 4u;
    o = (o + 15u) & ~15u;
    L.off_v = o;     o += (uint32_t)ntypes * kSortTy * 4u;   // per type: values [66], thresholds T [64], n_thr, kout_pos, kout_neg, gmax, lim, m
    o = (o + 15u) & ~15u;
    L.off_misc = o;  o += 256u;                              // scan scratch, counters
    L.total = o;
    return L;
}

// ---- keys --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sort_key(float x)        // order of the unsigned keys == order of the (non-NaN) floats
{
    const uint32_t u = x == 0.0f ? 0u : f2u(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sort_unkey(uint32_t k) { return u2f((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
__device__ __forceinline__ long long sort_key_fixed(uint32_t k, double F)
{
    return k == kSortSent ? 0ll : sweep_fixed(sort_unkey(k), F);
}

// x_threshold (antq_k_fakequant.h) with the rounding boundary M = (pred(T) + T) / 2 and T's mantissa parity precomputed per
// threshold (they do not depend on the scale): the same float
__device__ __forceinline__ float sort_x_threshold(double M, bool t_even, double sd)
{
    const double prod = M * sd;                        // exact
    const float xf = (float)prod;
    const double back = (double)xf;
    const bool take = (back > prod) || (back == prod && t_even);
    return take ? xf : f_up(xf);
}
// a type's block: what every thread of a workgroup fills once (tid < 64: one threshold each; tid == 64: the scalars)
__device__ __forceinline__ void sort_fill_type(float *v, const SweepType &ty, uint32_t nneg, uint32_t tid)
{
    uint32_t *vu = reinterpret_cast<uint32_t *>(v);
    double *vm = reinterpret_cast<double *>(v + 140);
    if (tid < 64u) {
        const uint32_t k = tid;
        float T = 0.0f;
        if (k < ty.n_thr) {
            const uint4 th = ty.tlist[k];
            T = u2f(th.x);
            v[k + 1u] = u2f(th.z) + 0.0f;
            if (k == 0u) v[0] = u2f(th.y) + 0.0f;
        } else {                                         // beyond the last threshold: the last value again (A = B = 0)
            v[k + 1u] = ty.n_thr ? u2f(ty.tlist[ty.n_thr - 1u].z) + 0.0f : 0.0f;
        }
        v[66u + k] = T;
        vm[k] = 0.5 * ((double)f_dn(T) + (double)T);
        v[268u + k] = k < ty.m ? ty.grid[k] : 0.0f;
        const unsigned long long ev = __ballot((f2u(T) & 1u) == 0u);
        if (k == 0u) { vu[137] = (uint32_t)ev; vu[138] = (uint32_t)(ev >> 32); }
    }
    if (tid == 64u) {
        vu[130] = ty.n_thr;
        vu[131] = (uint32_t)ty.kout_pos;
        vu[132] = (uint32_t)ty.kout_neg;
        v[133] = ty.gmax;
        v[134] = ty.lim;
        vu[135] = ty.m;
        vu[136] = nneg < ty.n_thr ? nneg : ty.n_thr;
    }
}
__device__ __forceinline__ uint32_t sort_threshold_key(const float *v, uint32_t k, double sd)
{
    const uint32_t *vu = reinterpret_cast<const uint32_t *>(v);
    const double M = reinterpret_cast<const double *>(v + 140)[k];
    const bool even = ((vu[137u + (k >> 5)] >> (k & 31u)) & 1u) != 0u;
    return sort_key(sort_x_threshold(M, even, sd));
}

// the reference scan (quant_kernel.cu:25-37) for a literal element: the codebook from LDS where it fits (a scan that waits for
// a load from memory per value took 2000 cycles per element and candidate: 5 ms for a 64 x 1024 tensor with one Inf row)
__device__ __forceinline__ float sort_literal_q(float xv, float s, const float *v, const float *grid, int m, float &d)
{
    if (m > (int)kSortGridLds) return sweep_literal_q(xv, s, grid, m, d);
    d = xv / s;
    float sub_min = 102400.0f, z_min = 0.0f;
#pragma unroll 4
    for (int i = 0; i < m; i++) {
        const float g = v[268 + i];
        const float sub_v = fabsf(d - g);
        if (sub_v <= sub_min) { sub_min = sub_v; z_min = g; }
    }
    return z_min;
}

// ---- the sorting network -------------------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void sort_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sort_static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void sort_ce(uint32_t &a, uint32_t &b)
{
    const uint32_t lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}
// compare-exchange of the pairs that differ in register bit BIT
template <int BIT>
__device__ __forceinline__ void sort_ce_reg(uint32_t (&k)[kSortEPT])
{
#pragma unroll
    for (int r = 0; r < kSortEPT; r++)
        if (!((r >> BIT) & 1)) sort_ce(k[r], k[r | (1 << BIT)]);
}
// first step of a size-2^SB merge inside the registers: partner r ^ (2^SB - 1)
template <int SB>
__device__ __forceinline__ void sort_ce_reg_mirror(uint32_t (&k)[kSortEPT])
{
#pragma unroll
    for (int r = 0; r < kSortEPT; r++)
        if (!((r >> (SB - 1)) & 1)) sort_ce(k[r], k[r ^ ((1 << SB) - 1)]);
}
// CE on register bits TOP .. 0
template <int TOP>
__device__ __forceinline__ void sort_ce_down(uint32_t (&k)[kSortEPT])
{
    if constexpr (TOP >= 0) {
        sort_ce_reg<TOP>(k);
        sort_ce_down<TOP - 1>(k);
    }
}
// A (sub-)sort over BT index bits: the layout whose register bits are [lo, lo + R) and hold index bit b
template <int BT>
__host__ __device__ constexpr int sort_lay_lo(int b)
{
    return (b / kSortR) * kSortR < BT - kSortR ? (b / kSortR) * kSortR : BT - kSortR;
}
// dword address of the element (thread t, register 0) in layout lo; register r adds sort_lay_off(r, lo) (disjoint bit fields:
// the padding term i >> R splits over them)
__device__ __forceinline__ uint32_t sort_lay_base(uint32_t t, int lo)
{
    const uint32_t i = ((t >> lo) << (lo + kSortR)) | (t & ((1u << lo) - 1u));
    return i + (i >> kSortR);
}
__host__ __device__ constexpr uint32_t sort_lay_off(int r, int lo) { return ((uint32_t)r << lo) + (((uint32_t)r << lo) >> kSortR); }

template <int LO>
__device__ __forceinline__ void sort_store(const uint32_t (&k)[kSortEPT], uint32_t *sK, uint32_t t)
{
    uint32_t *p = sK + sort_lay_base(t, LO);
#pragma unroll
    for (int r = 0; r < kSortEPT; r++) p[sort_lay_off(r, LO)] = k[r];
}
// S > 0: the first read of the size-2^S merge -- the upper half block (index bit S - 1 set) comes in mirrored
template <int LO, int S>
__device__ __forceinline__ void sort_load(uint32_t (&k)[kSortEPT], const uint32_t *sK, uint32_t t)
{
    const uint32_t *p = sK + sort_lay_base(t, LO);
    if constexpr (S > 0) {
        const uint32_t *pm = sK + sort_lay_base(t ^ ((1u << LO) - 1u), LO);
#pragma unroll
        for (int r = 0; r < kSortEPT; r++) {
            if ((r >> (S - 1 - LO)) & 1) k[r] = pm[sort_lay_off(r ^ ((1 << (S - 1 - LO)) - 1), LO)];
            else k[r] = p[sort_lay_off(r, LO)];
        }
    } else {
#pragma unroll
        for (int r = 0; r < kSortEPT; r++) k[r] = p[sort_lay_off(r, LO)];
    }
}
// WAVE: the keys that meet belong to ONE wavefront (its own 1024 + 64 dwords of the buffer): its LDS traffic is in order,
// nothing to wait for but the compiler; otherwise the whole workgroup meets
template <bool WAVE>
__device__ __forceinline__ void sort_sync()
{
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): this wavefront's LDS traffic has landed
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}
// the strides 2^BB .. 1 of the size-2^S merge of a sort over BT bits (t: the thread's index among its 2^(BT - R)); the keys
// are in layout CUR
template <int BT, bool WAVE, int S, int BB, int CUR, bool FIRST>
__device__ __forceinline__ void sort_steps(uint32_t (&k)[kSortEPT], uint32_t *sK, uint32_t t)
{
    if constexpr (BB >= 0) {
        constexpr int lo = sort_lay_lo<BT>(BB);
        sort_store<CUR>(k, sK, t);
        sort_sync<WAVE>();
        sort_load<lo, FIRST ? S : 0>(k, sK, t);
        sort_sync<WAVE>();
        sort_ce_down<BB - lo>(k);
        sort_steps<BT, WAVE, S, lo - 1, lo, false>(k, sK, t);
    }
}
// ascending sort of the workgroup's 4096 keys; in: any arrangement; out: thread t holds sorted[16 t .. 16 t + 15]
// (the key buffer must be free when this is called, and is free again when it returns).  Sizes 2 .. 16 in the registers,
// 32 .. 1024 inside each wavefront (14 transposes through its own quarter of the buffer, no workgroup barrier), 2048 and
// 4096 across the workgroup (6 transposes, 12 barriers).
__device__ __forceinline__ void sort_wg(uint32_t (&k)[kSortEPT], uint32_t *sK, uint32_t t)
{
    constexpr int BW = 6 + kSortR;                          // index bits inside a wavefront
    sort_static_for<1, kSortR + 1>([&](auto s) {
        sort_ce_reg_mirror<decltype(s)::value>(k);
        sort_ce_down<decltype(s)::value - 2>(k);
    });
    uint32_t *sW = sK + (t >> 6) * ((1u << BW) + (1u << (BW - kSortR)));
    sort_static_for<kSortR + 1, BW + 1>([&](auto s) { sort_steps<BW, true, decltype(s)::value, decltype(s)::value - 1, 0, true>(k, sW, t & 63u); });
    sort_static_for<BW + 1, kSortB + 1>([&](auto s) { sort_steps<kSortB, false, decltype(s)::value, decltype(s)::value - 1, 0, true>(k, sK, t); });
}

// ---- the search kernel ---------------------------------------------------------------------------------------------------
// PT = false: one workgroup per row (rows grid-strided), sse[(type * ncand_all + c) * rows + row] (ncand of the ncand_all
//             candidates in this launch: `ratios` and `sse` point at the first of them).
// PT = true : a tensor with ONE scale; workgroup b walks the chunks b, b + gridDim.x, ... and leaves its partial terms in
//             slab b of `slabs` (doubles: item partials | literal terms | pair corrections | sum x^2); k_sort_pt_total adds the
//             slabs in slab order, k_sort_pt_finish forms the sums.
template <typename T, bool OVP, bool PT>
__global__ void __launch_bounds__(kSortNT, 3)        // (three wavefronts per SIMD: three workgroups per CU; four with the pair
                                                      //  rule -- 128 registers, 236 bytes of scratch -- measured 0.246 against 0.221 ms)
k_search_sorted(const uint4 *__restrict__ x, size_t vpr, size_t rows, const float *__restrict__ xmax,
                const float *__restrict__ ratios, double *__restrict__ sse, SortTypes st, uint32_t ncand, uint32_t ncand_all,
                double *__restrict__ slabs)
{
    constexpr int EPL = IO<T>::EPL;
    constexpr int VPT = kSortEPT / EPL;                     // vectors per thread and chunk
    constexpr size_t VPC = kSortK / EPL;                    // vectors per chunk
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t ntypes = (uint32_t)st.ntypes, ntc = ntypes * ncand, nthr_pad = st.nthr_pad, nkg = nthr_pad / (uint32_t)kSortKS;
    const uint32_t nitems = ntc * nkg;
    // (type, candidate) of a flat index without an integer division (~20 instructions each on this machine; two of them per
    // threshold were a tenth of the kernel): at most kMaxTypes = 4 types
    auto type_of = [&](uint32_t tc) { return (tc >= ncand ? 1u : 0u) + (tc >= 2u * ncand ? 1u : 0u) + (tc >= 3u * ncand ? 1u : 0u); };
    const SortLds L = sort_lds(ntc, nthr_pad, st.ntypes, OVP);
    char *base = reinterpret_cast<char *>(smem);
    uint32_t *sK = reinterpret_cast<uint32_t *>(base);
    long long *sP4 = reinterpret_cast<long long *>(base + L.off_p4);
    uint32_t *sX = reinterpret_cast<uint32_t *>(base + L.off_x);          // [ntc][nthr_pad] keys of the x-domain thresholds
    float *sS = reinterpret_cast<float *>(base + L.off_s);               // [ntc]
    float *sV = reinterpret_cast<float *>(base + L.off_v);               // per type: [0..65] values, [66..129] thresholds T
    unsigned long long *sScan = reinterpret_cast<unsigned long long *>(base + L.off_misc);      // [4] counts, [4..8] sums
    long long *sScanI = reinterpret_cast<long long *>(base + L.off_misc + 64);
    double *sScanD = reinterpret_cast<double *>(base + L.off_misc + 128);
    int *sFlag = reinterpret_cast<int *>(base + L.off_misc + 192);

    // per type: values, grid-domain thresholds, rounding boundaries and scalars (once per workgroup; constant indices into the
    // kernel argument)
#pragma unroll
    for (int t = 0; t < kMaxTypes; t++)
        if (t < st.ntypes) sort_fill_type(sV + (uint32_t)t * kSortTy, st.ty[t], st.nneg[t], tid);
    __syncthreads();
    auto ty_nthr = [&](uint32_t t) { return reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[130]; };
    auto ty_kpos = [&](uint32_t t) { return (int)reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[131]; };
    auto ty_kneg = [&](uint32_t t) { return (int)reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[132]; };
    auto ty_gmax = [&](uint32_t t) { return sV[t * kSortTy + 133u]; };
    auto ty_lim = [&](uint32_t t) { return sV[t * kSortTy + 134u]; };
    auto ty_nneg = [&](uint32_t t) { return reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[136]; };
    auto ty_m = [&](uint32_t t) { return (int)reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[135]; };
    auto ty_grid = [&](uint32_t t) {
        const float *g = st.ty[0].grid;
#pragma unroll
        for (int u = 1; u < kMaxTypes; u++) g = (uint32_t)u == t ? st.ty[u].grid : g;
        return g;
    };

    const size_t nchunks_row = (vpr + VPC - 1) / VPC;
    for (size_t row = PT ? 0 : blockIdx.x; row < rows; row += PT ? 1 : gridDim.x) {
        const float xm = xmax[row];
        const uint4 *xr = x + row * vpr;
        // A zero or NaN statistic (an all-zero row): every scale is 0 / NaN, x / s is NaN or +-Inf for every element, the scan
        // keeps its initial 0 and (0 - d) + d is NaN -- every candidate's sum is NaN whatever the row holds (quant_kernel.cu:25-37,
        // AQ:541-549).  Said directly instead of through 4096 literal evaluations per candidate.
        if (!PT && (xm == 0.0f || xm != xm)) {
            for (uint32_t tc = tid; tc < ntc; tc += kSortNT) {
                const uint32_t t = type_of(tc), c = tc - t * ncand;
                sse[((size_t)t * ncand_all + c) * rows + row] = __builtin_nan("");
            }
            continue;
        }
        // ---- 1. candidate scales (AQ:300, :536): s = fl32(fl32(x_max * ratio_c) / gmax_t), usable and non-decreasing along c
        bool ok = true;
        for (uint32_t tc = tid; tc < ntc; tc += kSortNT) {
            const uint32_t t = type_of(tc), c = tc - t * ncand;
            const Scale sc = make_scale(xm * ratios[c], ty_gmax(t));
            sS[tc] = sc.s;
            ok = ok && sc.ok && (sc.s > 0.0f);
        }
        __syncthreads();
        for (uint32_t tc = tid; tc < ntc; tc += kSortNT) {
            const uint32_t t = type_of(tc), c = tc - t * ncand;
            if (c > 0u) ok = ok && (sS[tc] >= sS[tc - 1u]);
        }
        if (tid == 0u) *sFlag = 0;
        __syncthreads();
        if (!ok) atomicOr(sFlag, 1);
        __syncthreads();
        const bool usable = *sFlag == 0;
        // ---- 2. thresholds in the x domain, as keys; partial sums cleared
        constexpr bool INL = sort_inline(OVP);
        if constexpr (INL) {
            // (per (type, candidate) only the two bounds of the outlier region: key >= kp a positive outlier, key < kn a negative one)
            for (uint32_t tc = tid; tc < ntc; tc += kSortNT) {
                const uint32_t t = type_of(tc);
                const float *v = sV + t * kSortTy;
                const double sd = (double)sS[tc];
                sX[2u * tc] = (usable && ty_kpos(t) >= 0) ? sort_threshold_key(v, (uint32_t)ty_kpos(t), sd) : kSortSent;
                sX[2u * tc + 1u] = (usable && ty_kneg(t) >= 0) ? sort_threshold_key(v, (uint32_t)ty_kneg(t), sd) : 0u;
            }
        }
        const uint32_t q_thr = kSortNT / nthr_pad, r_thr = kSortNT - q_thr * nthr_pad;      // (uniform: scalar divisions)
        for (uint32_t p = tid, tc = tid / nthr_pad, k = tid - (tid / nthr_pad) * nthr_pad; !INL && p < ntc * nthr_pad; p += kSortNT) {
            const uint32_t t = type_of(tc);
            uint32_t key = kSortSent;
            if (usable && k < ty_nthr(t)) {
                key = sort_threshold_key(sV + t * kSortTy, k, (double)sS[tc]);
            }
            sX[p] = key;
            k += r_thr;
            tc += q_thr + (k >= nthr_pad ? 1u : 0u);
            k -= k >= nthr_pad ? nthr_pad : 0u;
        }
        double acc[kSortNI], corr[kSortNC], lit[kSortNL];     // this thread's items it = tid + 256 u (statically indexed: registers)
#pragma unroll
        for (int u = 0; u < kSortNI; u++) acc[u] = 0.0;
#pragma unroll
        for (int u = 0; u < kSortNC; u++) corr[u] = 0.0;
#pragma unroll
        for (int u = 0; u < kSortNL; u++) lit[u] = 0.0;
        // ---- 3. per-row constants
        int ex = 0;
        (void)frexpf(xm, &ex);                              // x_max = f * 2^ex, f in [0.5, 1)
        const double F = __builtin_ldexp(1.0, 38 - ex), unit = __builtin_ldexp(1.0, ex - 38);
        float Lx = 0.0f;                                    // |x| < Lx: a step-function element for EVERY candidate of every type
        if (usable) {
            Lx = __builtin_ldexpf(0.999f, ex + 8);          // ... whose fixed-point image stays below 2^46
            for (uint32_t t = 0; t < ntypes; t++) Lx = fminf(Lx, ty_lim(t) * sS[t * ncand] * 0.999f);
        }
        __syncthreads();
        float XoP = __builtin_inff(), XoN = -__builtin_inff();            // OliVe: outlier under the smallest scale of SOME type
        if (OVP && usable) {
            for (uint32_t t = 0; t < ntypes; t++) {
                if (INL) {
                    if (ty_kpos(t) >= 0) XoP = fminf(XoP, sort_unkey(sX[2u * (t * ncand)]));
                    if (ty_kneg(t) >= 0) XoN = fmaxf(XoN, sort_unkey(sX[2u * (t * ncand) + 1u]));
                } else {
                    if (ty_kpos(t) >= 0) XoP = fminf(XoP, sort_unkey(sX[(t * ncand) * nthr_pad + (uint32_t)ty_kpos(t)]));
                    if (ty_kneg(t) >= 0) XoN = fmaxf(XoN, sort_unkey(sX[(t * ncand) * nthr_pad + (uint32_t)ty_kneg(t)]));
                }
            }
        }
        double Q = 0.0;
        uint32_t nch_done = 0;                              // (parity of the count scratch: a chunk with nothing to sort has one barrier only)

        // ---- 4. the chunks
        for (size_t ch = PT ? blockIdx.x : 0; ch < nchunks_row; ch += PT ? gridDim.x : 1) {
            uint32_t k[kSortEPT];
            float xs[kSortEPT];
            uint32_t nreg = 0, nlit = 0, ncap = 0;          // elements; literal elements (pairs count 2); capable pairs
            uint32_t litmask = 0, capmask = 0;              // per element / per pair (bit = index of the pair's first element)
#pragma unroll
            for (int j = 0; j < VPT; j++) {
                const size_t vi = ch * VPC + (size_t)j * kSortNT + tid;
                const bool live = vi < vpr;
                float xf[EPL];
                {
                    const uint4 v = live ? xr[vi] : make_uint4(0u, 0u, 0u, 0u);
                    IO<T>::unpack(v, xf);
                }
#pragma unroll
                for (int e = 0; e < EPL; e += 2) {
                    const int r = j * EPL + e;
                    const float a = xf[e], b = xf[e + 1];
                    bool la = live && !(fabsf(a) < Lx), lb = live && !(fabsf(b) < Lx);
                    bool cap = false;
                    if (OVP) {
                        la = lb = (la || lb);
                        cap = live && !la && ((a >= XoP) || (a < XoN) || (b >= XoP) || (b < XoN));
                    }
                    const bool ra = live && !la, rb = live && !lb;
                    k[r] = ra ? sort_key(a) : kSortSent;
                    k[r + 1] = rb ? sort_key(b) : kSortSent;
                    xs[r] = a;
                    xs[r + 1] = b;
                    Q = __builtin_fma((double)(ra ? a : 0.0f), (double)(ra ? a : 0.0f), Q);
                    Q = __builtin_fma((double)(rb ? b : 0.0f), (double)(rb ? b : 0.0f), Q);
                    nreg += (ra ? 1u : 0u) + (rb ? 1u : 0u);
                    nlit += (la ? 1u : 0u) + (lb ? 1u : 0u);
                    litmask |= (la ? 1u : 0u) << r | (lb ? 1u : 0u) << (r + 1);
                    if (OVP && cap) { ncap++; capmask |= 1u << r; }
                }
            }
            // counts of the chunk, and this thread's places in the lists: one packed scan (fields of 16 bits, <= 4096 each)
            unsigned long long pk = (unsigned long long)nreg | (unsigned long long)nlit << 16 | (unsigned long long)ncap << 32, inc = pk;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long tv = (unsigned long long)__shfl_up((long long)inc, off, 64);
                if (lane >= (uint32_t)off) inc += tv;
            }
            const uint32_t par = (nch_done & 1u) * 4u;
            nch_done++;
            if (lane == 63u) sScan[par + wave] = inc;
            __syncthreads();                                 // (also: the previous chunk's searches are done with sK / sP4)
            unsigned long long before = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < kSortNT / 64; w++) {
                const unsigned long long v = sScan[par + w];
                before += w < wave ? v : 0ull;
                total += v;
            }
            const uint32_t Kreg = (uint32_t)(total & 0xffffu), tot_lit = (uint32_t)((total >> 16) & 0xffffu), tot_cap = (uint32_t)((total >> 32) & 0xffffu);
            // ---- 4a. literal elements and outlier-capable pairs (rare): lists in the key buffer, then every (type, candidate)
            if (tot_lit | tot_cap) {
                const unsigned long long excl = before + inc - pk;
                uint32_t pl = (uint32_t)((excl >> 16) & 0xffffu), pc = (uint32_t)((excl >> 32) & 0xffffu);
                float *sList = reinterpret_cast<float *>(sK);             // literal elements from the front, capable pairs from the back
#pragma unroll
                for (int r = 0; r < kSortEPT; r++)
                    if ((litmask >> r) & 1u) sList[pl++] = xs[r];
                if (OVP) {
#pragma unroll
                    for (int r = 0; r < kSortEPT; r += 2)
                        if ((capmask >> r) & 1u) {
                            sK[kSortK - 2u - 2u * pc] = sort_key(xs[r]);         // (as keys: what every (type, candidate) compares)
                            sK[kSortK - 1u - 2u * pc] = sort_key(xs[r + 1]);
                            pc++;
                        }
                }
                __syncthreads();
                if (tot_lit) {
                    // An element that is a step-function element for THIS candidate (|x / s| < lim) contributes what the closed
                    // form would have given it -- (O_J - x)^2 in double -- so a candidate's sum does not depend (beyond 1e-13) on
                    // whether the launch's smallest scale sent the element here; elsewhere: the reference sequence.
                    for (uint32_t tc = tid, ui = 0; tc < ntc; tc += kSortNT, ui++) {
                        const uint32_t t = type_of(tc);
                        const float *grid = ty_grid(t);
                        const int gm = ty_m(t);
                        const float s = sS[tc], lim = usable ? ty_lim(t) : 0.0f;
                        const uint32_t *X = sX + (INL ? 0u : tc * nthr_pad);
                        const float *v = sV + t * kSortTy;
                        const uint32_t nthr_t = ty_nthr(t);
                        auto q_of = [&](float xv, float &d, bool &tab) -> float {
                            d = xv / s;
                            tab = fabsf(d) < lim;
                            if (tab) {
                                const uint32_t kx = sort_key(xv);
                                uint32_t lo = 0, hi = nthr_t;
                                while (lo < hi) {             // (inline thresholds: RN(x / s) >= T_k <=> x >= X_k)
                                    const uint32_t mid = (lo + hi) >> 1;
                                    if (INL ? d >= v[66u + mid] : kx >= X[mid]) lo = mid + 1u; else hi = mid;
                                }
                                return v[lo];
                            }
                            return sort_literal_q(xv, s, v, grid, gm, d);
                        };
                        auto term_of = [&](float q, float d, bool tab, float xv) -> double {
                            if (tab) {
                                const double e = (double)(q * s) - (double)xv;
                                return e * e;
                            }
                            return sweep_term(q, d, s, xv);
                        };
                        double sum_l = 0.0;
                        for (uint32_t i = 0; i < tot_lit; i += OVP ? 2u : 1u) {
                            const float xa_ = sList[i];
                            float da, db = 0.0f;
                            bool ta, tb = false;
                            float qa = q_of(xa_, da, ta), qb = 0.0f;
                            if (OVP) {
                                const float xb_ = sList[i + 1u];
                                qb = q_of(xb_, db, tb);
                                const bool me = fabsf(qa) > 32.0f, mo = fabsf(qb) > 32.0f;      // OQ:314
                                const bool ve = mo && !me;
                                qa = qa * (ve ? 0.0f : 1.0f);
                                qb = qb * (me ? 0.0f : 1.0f);
                                sum_l += term_of(qb, db, tb, xb_);
                            }
                            sum_l += term_of(qa, da, ta, xa_);
                        }
#pragma unroll
                        for (int u = 0; u < kSortNL; u++) lit[u] += (uint32_t)u == ui ? sum_l : 0.0;
                    }
                }
                if (OVP && tot_cap) {
                    // work item (type, candidate, j): the pairs j, j + 4, ... in list order
                    for (uint32_t it = tid, ui = 0; it < 4u * ntc; it += kSortNT, ui++) {
                        const uint32_t tc = it >> 2, j0 = it & 3u, t = type_of(tc);
                        const uint32_t *X = sX + (INL ? 0u : tc * nthr_pad);
                        const float s = sS[tc];
                        const float *v = sV + t * kSortTy;
                        const uint32_t nthr_t = ty_nthr(t);
                        const uint32_t kp = INL ? sX[2u * tc] : (ty_kpos(t) >= 0 ? X[ty_kpos(t)] : kSortSent);      // key >= kp: a positive outlier
                        const uint32_t kn = INL ? sX[2u * tc + 1u] : (ty_kneg(t) >= 0 ? X[ty_kneg(t)] : 0u);       // key <  kn: a negative outlier
                        double sum_c = 0.0;
                        for (uint32_t i = j0; i < tot_cap; i += 4u) {
                            const uint32_t ka = sK[kSortK - 2u - 2u * i], kb = sK[kSortK - 1u - 2u * i];
                            const bool me = ka >= kp || ka < kn, mo = kb >= kp || kb < kn;
                            if (me || mo) {
                                const uint32_t kv = me ? kb : ka;                                  // the victim (OQ:315-318)
                                const float vv = sort_unkey(kv);
                                const float dvv = vv / s;
                                uint32_t lo = 0, hi = nthr_t;
                                while (lo < hi) {
                                    const uint32_t mid = (lo + hi) >> 1;
                                    if (INL ? dvv >= v[66u + mid] : kv >= X[mid]) lo = mid + 1u; else hi = mid;
                                }
                                const double O = (double)(v[lo] * s), dv = (double)vv;
                                sum_c += dv * dv - (O - dv) * (O - dv);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < kSortNC; u++) corr[u] += (uint32_t)u == ui ? sum_c : 0.0;
                    }
                }
                __syncthreads();
            }
            if (Kreg == 0u) continue;                        // (uniform) nothing to sort
            // ---- 4b. sort; sorted keys and prefix sums to LDS
            sort_wg(k, sK, tid);
            {   // sorted keys, UNPADDED (the probes' addresses are then pos + constant): 64 contiguous bytes per thread
                uint4 *dst = reinterpret_cast<uint4 *>(sK + (size_t)kSortEPT * tid);
#pragma unroll
                for (int q = 0; q < kSortEPT / 4; q++) dst[q] = make_uint4(k[4 * q], k[4 * q + 1], k[4 * q + 2], k[4 * q + 3]);
            }
            constexpr int PSH = sort_psh(OVP), NG = kSortEPT >> PSH;
            long long g[NG], mine = 0;
#pragma unroll
            for (int q = 0; q < NG; q++) {
                g[q] = 0;
#pragma unroll
                for (int e = 0; e < (1 << PSH); e++) g[q] += sort_key_fixed(k[(q << PSH) + e], F);
                mine += g[q];
            }
            long long incs = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const long long tv = __shfl_up(incs, off, 64);
                if (lane >= (uint32_t)off) incs += tv;
            }
            if (lane == 63u) sScanI[wave] = incs;
            __syncthreads();
            long long pre = incs - mine;
#pragma unroll
            for (uint32_t w = 0; w < kSortNT / 64; w++) pre += w < wave ? sScanI[w] : 0ll;
#pragma unroll
            for (int q = 0; q < NG; q++) {
                sP4[(uint32_t)NG * tid + (uint32_t)q] = pre;
                pre += g[q];
            }
            if (tid == kSortNT - 1u) sP4[kSortK >> PSH] = pre;
            __syncthreads();
            // ---- 4c. the (type, candidate, threshold group) items
            const long long Stot_i = sP4[kSortK >> PSH];
            const double Stot = (double)Stot_i * unit, dn = (double)Kreg;
            const uint32_t q_it = kSortNT / nkg, r_it = kSortNT - q_it * nkg;
            for (uint32_t it = tid, ui = 0, tc = tid / nkg, kg = tid - (tid / nkg) * nkg; it < nitems; it += kSortNT, ui++) {
                const uint32_t t = type_of(tc);
                const float s = sS[tc];
                const float *v = sV + t * kSortTy + kg * (uint32_t)kSortKS;
                const uint32_t *X = sX + (INL ? 0u : tc * nthr_pad + kg * (uint32_t)kSortKS);
                // (the running ADDRESS is the search state: a probe is ds_read_b32 with an immediate offset, a step is compare +
                //  select + add -- an index would cost a shift-add per probe on top)
                uint32_t Xk[kSortKS], pos[kSortKS];
                const uint32_t *ap[kSortKS];
#pragma unroll
                for (int j = 0; j < kSortKS; j++) {
                    if constexpr (INL) {
                        const uint32_t kk = kg * (uint32_t)kSortKS + (uint32_t)j;
                        Xk[j] = (usable && kk < ty_nthr(t)) ? sort_threshold_key(sV + t * kSortTy, kk, (double)s) : kSortSent;
                    } else {
                        Xk[j] = X[j];
                    }
                    ap[j] = sK;
                }
#pragma unroll
                for (int step = kSortK / 2; step >= 1; step >>= 1) {
                    uint32_t kv[kSortKS];
#pragma unroll
                    for (int j = 0; j < kSortKS; j++) kv[j] = ap[j][step - 1];
#pragma unroll
                    for (int j = 0; j < kSortKS; j++) ap[j] += kv[j] < Xk[j] ? step : 0;
                }
#pragma unroll
                for (int j = 0; j < kSortKS; j++) pos[j] = (uint32_t)(ap[j] - sK) + (ap[j][0] < Xk[j] ? 1u : 0u);
                const uint32_t nb = ty_nneg(t);
                double part = 0.0;
                if (kg == 0u) {
                    const double Ob_ = (double)(sV[t * kSortTy + nb] * s);
                    part = dn * Ob_ * Ob_ - 2.0 * Ob_ * Stot;
                }
#pragma unroll
                for (int j = 0; j < kSortKS; j++) {
                    const uint32_t p = pos[j];
                    long long slt = sP4[p >> PSH];
                    if constexpr (PSH == 1) {
                        if (p & 1u) slt += sort_key_fixed(sK[p - 1u], F);
                    } else {
                        for (uint32_t i = p & ~3u; i < p; i++) slt += sort_key_fixed(sK[i], F);
                    }
                    const bool below = kg * (uint32_t)kSortKS + (uint32_t)j < nb;        // a threshold below zero: the elements BELOW it
                    const double N = below ? -(double)p : (double)(Kreg - p);
                    const double S = (double)(below ? -slt : Stot_i - slt) * unit;
                    const double Oa = (double)(v[j] * s), Ob = (double)(v[j + 1] * s);
                    part += (Ob - Oa) * ((Ob + Oa) * N - 2.0 * S);                          // (O_b^2 - O_a^2) N - 2 (O_b - O_a) S
                }
#pragma unroll
                for (int u = 0; u < kSortNI; u++) acc[u] += (uint32_t)u == ui ? part : 0.0;
                kg += r_it;
                tc += q_it + (kg >= nkg ? 1u : 0u);
                kg -= kg >= nkg ? nkg : 0u;
            }
            // (the next chunk's first barrier -- or the one below -- orders these reads before the key buffer is reused)
        }
        // ---- 5. sum x^2 of the step-function elements: a fixed tree
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) Q += __shfl_xor(Q, off, 64);
        __syncthreads();
        if (lane == 0u) sScanD[wave] = Q;
        __syncthreads();
        double Qw = 0.0;
#pragma unroll
        for (uint32_t w = 0; w < kSortNT / 64; w++) Qw += sScanD[w];
        // ---- 6. the sums -- or, for a tensor with ONE scale, this workgroup's partial terms to its slab
        if (PT) {
            const uint32_t ncell = nitems + ntc + (OVP ? 4u * ntc : 0u) + 1u;
            double *slab = slabs + (size_t)blockIdx.x * ncell;
#pragma unroll
            for (int u = 0; u < kSortNI; u++)
                if (tid + (uint32_t)u * kSortNT < nitems) slab[tid + (uint32_t)u * kSortNT] = acc[u];
#pragma unroll
            for (int u = 0; u < kSortNL; u++)
                if (tid + (uint32_t)u * kSortNT < ntc) slab[nitems + tid + (uint32_t)u * kSortNT] = lit[u];
            if (OVP) {
#pragma unroll
                for (int u = 0; u < kSortNC; u++)
                    if (tid + (uint32_t)u * kSortNT < 4u * ntc) slab[nitems + ntc + tid + (uint32_t)u * kSortNT] = corr[u];
            }
            if (tid == 0u) slab[ncell - 1u] = Qw;
            return;
        }
        // (the searches are done with the key buffer and the prefix sums: the partial terms meet there)
        double *fA = reinterpret_cast<double *>(sK), *fC = reinterpret_cast<double *>(sP4);
#pragma unroll
        for (int u = 0; u < kSortNI; u++)
            if (tid + (uint32_t)u * kSortNT < nitems) fA[tid + (uint32_t)u * kSortNT] = acc[u];
        if (OVP) {
#pragma unroll
            for (int u = 0; u < kSortNC; u++)
                if (tid + (uint32_t)u * kSortNT < 4u * ntc) fC[tid + (uint32_t)u * kSortNT] = corr[u];
        }
        __syncthreads();
        for (uint32_t tc = tid, ui = 0; tc < ntc; tc += kSortNT, ui++) {
            double sum = Qw;
            for (uint32_t kg = 0; kg < nkg; kg++) sum += fA[tc * nkg + kg];
            double l = 0.0;
#pragma unroll
            for (int u = 0; u < kSortNL; u++) l = (uint32_t)u == ui ? lit[u] : l;
            sum += l;
            if (OVP) sum += (fC[4u * tc] + fC[4u * tc + 1u]) + (fC[4u * tc + 2u] + fC[4u * tc + 3u]);
            const uint32_t t = type_of(tc), c = tc - t * ncand;       // (ratios / sse point at this piece's first candidate)
            sse[((size_t)t * ncand_all + c) * rows + row] = sum;
        }
        __syncthreads();
    }
}

// ---- short rows: one row per WAVEFRONT -----------------------------------------------------------------------------------
// A row of at most 1024 elements (BERT-base's 768-wide weights, most convolutions) wastes three quarters of the 4096-key
// sort above -- and the per-row work that does not shrink with the row (every threshold of every candidate moved into the x
// domain, the probes, the closed form) wants all four wavefronts busy.  Here each wavefront takes a row of its own: the
// sizes 2 .. 1024 of the same network (the in-wave phases of sort_wg, no workgroup barrier anywhere), 11-probe searches, a
// (codebook, candidate)'s thresholds moved into the x domain right where they are used (each is used once), one lane per
// (codebook, candidate) walking its threshold groups in order.  Literal elements (rare) are kept in a short list or read
// again from the row when a candidate's sum is formed.  With OliVe's pair rule (OVP): the outlier-capable pairs in the 64
// dwords behind the sorted keys (at most 32 of them, at most four per lane: a 3-sigma-clipped row of 1024 holds ~25), the
// victims' corrections from that list; more than that, and literal pairs, are read again from the row.
constexpr int kSortBS = 6 + kSortR, kSortKSh = 1 << kSortBS;          // 1024 keys per wavefront
__host__ __device__ inline uint32_t sort_short_wave_bytes(uint32_t ntc)
{
    return ((uint32_t)kSortKSh + (uint32_t)(kSortKSh >> kSortR)) * 4u + ((uint32_t)kSortKSh / 2u + 2u) * 8u + ((ntc + 3u) & ~3u) * 4u;
}
template <typename T, bool OVP>
__global__ void __launch_bounds__(256, 4)
k_search_sorted_short(const uint4 *__restrict__ x, uint32_t vpr, size_t rows, const float *__restrict__ xmax,
                      const float *__restrict__ ratios, double *__restrict__ sse, SortTypes st, uint32_t ncand, uint32_t ncand_all)
{
    constexpr int EPL = IO<T>::EPL;
    constexpr int VPT = kSortEPT / EPL;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t ntypes = (uint32_t)st.ntypes, ntc = ntypes * ncand, nthr_pad = st.nthr_pad, nkg = nthr_pad / (uint32_t)kSortKS;
    char *base = reinterpret_cast<char *>(smem);
    float *sV = reinterpret_cast<float *>(base);                                       // per type (as above), shared by the waves
    char *wb = base + (((uint32_t)st.ntypes * kSortTy * 4u + 15u) & ~15u) + wave * sort_short_wave_bytes(ntc);
    uint32_t *sK = reinterpret_cast<uint32_t *>(wb);                                   // this wavefront's keys
    long long *sP4 = reinterpret_cast<long long *>(wb + ((uint32_t)kSortKSh + (uint32_t)(kSortKSh >> kSortR)) * 4u);
    float *sS = reinterpret_cast<float *>(wb + ((uint32_t)kSortKSh + (uint32_t)(kSortKSh >> kSortR)) * 4u + ((uint32_t)kSortKSh / 2u + 2u) * 8u);
#pragma unroll
    for (int t = 0; t < kMaxTypes; t++)
        if (t < st.ntypes) sort_fill_type(sV + (uint32_t)t * kSortTy, st.ty[t], st.nneg[t], tid);
    __syncthreads();                                        // (the only workgroup barrier: from here on every wavefront is on its own)
    auto ty_nthr = [&](uint32_t t) { return reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[130]; };
    auto ty_kpos = [&](uint32_t t) { return (int)reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[131]; };
    auto ty_kneg = [&](uint32_t t) { return (int)reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[132]; };
    auto ty_gmax = [&](uint32_t t) { return sV[t * kSortTy + 133u]; };
    auto ty_lim = [&](uint32_t t) { return sV[t * kSortTy + 134u]; };
    auto ty_m = [&](uint32_t t) { return (int)reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[135]; };
    auto ty_nneg = [&](uint32_t t) { return reinterpret_cast<const uint32_t *>(sV + t * kSortTy)[136]; };
    auto ty_grid = [&](uint32_t t) {
        const float *g = st.ty[0].grid;
#pragma unroll
        for (int u = 1; u < kMaxTypes; u++) g = (uint32_t)u == t ? st.ty[u].grid : g;
        return g;
    };
    auto type_of = [&](uint32_t tc) { return (tc >= ncand ? 1u : 0u) + (tc >= 2u * ncand ? 1u : 0u) + (tc >= 3u * ncand ? 1u : 0u); };

    for (size_t row = (size_t)blockIdx.x * 4u + wave; row < rows; row += (size_t)gridDim.x * 4u) {
        const float xm = xmax[row];
        const uint4 *xr = x + row * (size_t)vpr;
        if (xm == 0.0f || xm != xm) {                        // (a zero / NaN statistic: NaN for every candidate, see above)
            for (uint32_t tc = lane; tc < ntc; tc += 64u) {
                const uint32_t t = type_of(tc), c = tc - t * ncand;
                sse[((size_t)t * ncand_all + c) * rows + row] = __builtin_nan("");
            }
            continue;
        }
        // ---- candidate scales
        bool ok = true;
        for (uint32_t tc = lane; tc < ntc; tc += 64u) {
            const uint32_t t = type_of(tc), c = tc - t * ncand;
            const Scale sc = make_scale(xm * ratios[c], ty_gmax(t));
            sS[tc] = sc.s;
            ok = ok && sc.ok && (sc.s > 0.0f);
        }
        sort_sync<true>();
        for (uint32_t tc = lane; tc < ntc; tc += 64u) {
            const uint32_t t = type_of(tc), c = tc - t * ncand;
            if (c > 0u) ok = ok && (sS[tc] >= sS[tc - 1u]);
        }
        const bool usable = __ballot(ok) == ~0ull;
        int ex = 0;
        (void)frexpf(xm, &ex);
        const double F = __builtin_ldexp(1.0, 38 - ex), unit = __builtin_ldexp(1.0, ex - 38);
        float Lx = 0.0f;
        if (usable) {
            Lx = __builtin_ldexpf(0.999f, ex + 8);
            for (uint32_t t = 0; t < ntypes; t++) Lx = fminf(Lx, ty_lim(t) * sS[t * ncand] * 0.999f);
        }
        float XoP = __builtin_inff(), XoN = -__builtin_inff();            // OliVe: an outlier under the smallest scale of SOME type
        if (OVP && usable) {
            for (uint32_t t = 0; t < ntypes; t++) {
                const double sd0 = (double)sS[t * ncand];
                if (ty_kpos(t) >= 0) XoP = fminf(XoP, sort_unkey(sort_threshold_key(sV + t * kSortTy, (uint32_t)ty_kpos(t), sd0)));
                if (ty_kneg(t) >= 0) XoN = fmaxf(XoN, sort_unkey(sort_threshold_key(sV + t * kSortTy, (uint32_t)ty_kneg(t), sd0)));
            }
        }
        // ---- the row: 16 elements per lane
        uint32_t k[kSortEPT];
        uint32_t nreg = 0, nlit = 0, ncap = 0;
        float lx0 = 0.0f, lx1 = 0.0f;                        // this lane's first two literal elements (a short list in LDS, below)
        uint32_t ca[4] = {0, 0, 0, 0}, cb[4] = {0, 0, 0, 0};   // ... and its first four outlier-capable pairs, as keys (statically indexed)
        double Q = 0.0;
#pragma unroll
        for (int j = 0; j < VPT; j++) {
            const uint32_t vi = (uint32_t)j * 64u + lane;
            const bool live = vi < vpr;
            float xf[EPL];
            {
                const uint4 v = live ? xr[vi] : make_uint4(0u, 0u, 0u, 0u);
                IO<T>::unpack(v, xf);
            }
            if constexpr (OVP) {
#pragma unroll
                for (int e = 0; e < EPL; e += 2) {
                    const float a = xf[e], b = xf[e + 1];
                    const bool lp = live && (!(fabsf(a) < Lx) || !(fabsf(b) < Lx)), rp = live && !lp;       // a literal pair / a regular one
                    const bool cap = rp && ((a >= XoP) || (a < XoN) || (b >= XoP) || (b < XoN));
                    const uint32_t ka = sort_key(a), kb = sort_key(b);
                    k[j * EPL + e] = rp ? ka : kSortSent;
                    k[j * EPL + e + 1] = rp ? kb : kSortSent;
                    Q = __builtin_fma((double)(rp ? a : 0.0f), (double)(rp ? a : 0.0f), Q);
                    Q = __builtin_fma((double)(rp ? b : 0.0f), (double)(rp ? b : 0.0f), Q);
                    nreg += rp ? 2u : 0u;
                    nlit += lp ? 2u : 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        ca[u] = (cap && ncap == (uint32_t)u) ? ka : ca[u];
                        cb[u] = (cap && ncap == (uint32_t)u) ? kb : cb[u];
                    }
                    ncap += cap ? 1u : 0u;
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    const float a = xf[e];
                    const bool la = live && !(fabsf(a) < Lx), ra = live && !la;
                    k[j * EPL + e] = ra ? sort_key(a) : kSortSent;
                    Q = __builtin_fma((double)(ra ? a : 0.0f), (double)(ra ? a : 0.0f), Q);
                    nreg += ra ? 1u : 0u;
                    lx0 = (la && nlit == 0u) ? a : lx0;
                    lx1 = (la && nlit == 1u) ? a : lx1;
                    nlit += la ? 1u : 0u;
                }
            }
        }
        const uint32_t my_lit = nlit, my_cap = ncap;
        uint32_t pk = nreg | nlit << 16, pc = ncap;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            pk += (uint32_t)__shfl_xor((int)pk, off, 64);
            if (OVP) pc += (uint32_t)__shfl_xor((int)pc, off, 64);
            Q += __shfl_xor(Q, off, 64);
        }
        const uint32_t Kreg = pk & 0xffffu, tot_lit = pk >> 16, tot_cap = OVP ? pc : 0u;
        long long Stot_i = 0;
        if (Kreg) {
            // ---- the in-wave sort (sizes 2 .. 16 in the registers, 32 .. 1024 through this wavefront's buffer), prefix sums
            sort_static_for<1, kSortR + 1>([&](auto s_) {
                sort_ce_reg_mirror<decltype(s_)::value>(k);
                sort_ce_down<decltype(s_)::value - 2>(k);
            });
            sort_static_for<kSortR + 1, kSortBS + 1>([&](auto s_) { sort_steps<kSortBS, true, decltype(s_)::value, decltype(s_)::value - 1, 0, true>(k, sK, lane); });
            {
                uint4 *dst = reinterpret_cast<uint4 *>(sK + (size_t)kSortEPT * lane);
#pragma unroll
                for (int q = 0; q < kSortEPT / 4; q++) dst[q] = make_uint4(k[4 * q], k[4 * q + 1], k[4 * q + 2], k[4 * q + 3]);
            }
            long long g[8], mine = 0;                        // prefix sums at every second sorted position
#pragma unroll
            for (int q = 0; q < 8; q++) {
                g[q] = sort_key_fixed(k[2 * q], F) + sort_key_fixed(k[2 * q + 1], F);
                mine += g[q];
            }
            long long incs = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const long long tv = __shfl_up(incs, off, 64);
                if (lane >= (uint32_t)off) incs += tv;
            }
            long long pre = incs - mine;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                sP4[8u * lane + (uint32_t)q] = pre;
                pre += g[q];
            }
            Stot_i = __shfl(incs, 63, 64);
            if (lane == 63u) sP4[kSortKSh / 2] = pre;
            sort_sync<true>();
        }
        // A few literal elements (a ratio list that starts below ~0.5: the row's largest elements): a list in the 64 dwords
        // behind the sorted keys, lane by lane; more than that -- or more than two in one lane -- are read again from the row
        const bool lit_list = !OVP && tot_lit != 0u && tot_lit <= 64u && __ballot(my_lit > 2u) == 0ull;
        const bool cap_list = OVP && tot_cap != 0u && tot_cap <= 32u && __ballot(my_cap > 4u) == 0ull;
        float *sL = reinterpret_cast<float *>(sK + kSortKSh);
        if (cap_list) {
            uint32_t inc = my_cap;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t tv = (uint32_t)__shfl_up((int)inc, off, 64);
                if (lane >= (uint32_t)off) inc += tv;
            }
            const uint32_t at = inc - my_cap;
            uint32_t *sC = sK + kSortKSh;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (my_cap > (uint32_t)u) { sC[2u * (at + (uint32_t)u)] = ca[u]; sC[2u * (at + (uint32_t)u) + 1u] = cb[u]; }
            sort_sync<true>();
        }
        if (lit_list) {
            uint32_t inc = my_lit;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t tv = (uint32_t)__shfl_up((int)inc, off, 64);
                if (lane >= (uint32_t)off) inc += tv;
            }
            const uint32_t at = inc - my_lit;
            if (my_lit > 0u) sL[at] = lx0;
            if (my_lit > 1u) sL[at + 1u] = lx1;
            sort_sync<true>();
        }
        const double Stot = (double)Stot_i * unit, dn = (double)Kreg;
        // ---- one lane per (codebook, candidate): its threshold groups in order, then the literal elements
        for (uint32_t tc = lane; tc < ntc; tc += 64u) {
            const uint32_t t = type_of(tc), c = tc - t * ncand;
            const float s = sS[tc];
            const double sd = (double)s;
            const float *v = sV + t * kSortTy;
            const uint32_t nthr_t = ty_nthr(t), nb = ty_nneg(t);
            double sum = Q;
            if (Kreg) {
                for (uint32_t kg = 0; kg < nkg; kg++) {
                    uint32_t Xk[kSortKS], pos[kSortKS];
#pragma unroll
                    for (int j = 0; j < kSortKS; j++) {
                        const uint32_t kk = kg * (uint32_t)kSortKS + (uint32_t)j;
                        Xk[j] = (usable && kk < nthr_t) ? sort_threshold_key(v, kk, sd) : kSortSent;
                    }
                    const uint32_t *ap[kSortKS];
#pragma unroll
                    for (int j = 0; j < kSortKS; j++) ap[j] = sK;
#pragma unroll
                    for (int step = kSortKSh / 2; step >= 1; step >>= 1) {
                        uint32_t kv[kSortKS];
#pragma unroll
                        for (int j = 0; j < kSortKS; j++) kv[j] = ap[j][step - 1];
#pragma unroll
                        for (int j = 0; j < kSortKS; j++) ap[j] += kv[j] < Xk[j] ? step : 0;
                    }
#pragma unroll
                    for (int j = 0; j < kSortKS; j++) pos[j] = (uint32_t)(ap[j] - sK) + (ap[j][0] < Xk[j] ? 1u : 0u);
                    double part = 0.0;
                    if (kg == 0u) {
                        const double Ob_ = (double)(v[nb] * s);
                        part = dn * Ob_ * Ob_ - 2.0 * Ob_ * Stot;
                    }
#pragma unroll
                    for (int j = 0; j < kSortKS; j++) {
                        const uint32_t kk = kg * (uint32_t)kSortKS + (uint32_t)j, p = pos[j];
                        long long slt = sP4[p >> 1];
                        if (p & 1u) slt += sort_key_fixed(sK[p - 1u], F);
                        const bool below = kk < nb;
                        const double N = below ? -(double)p : (double)(Kreg - p);
                        const double S = (double)(below ? -slt : Stot_i - slt) * unit;
                        const double Oa = (double)(v[kk] * s), Ob = (double)(v[kk + 1u] * s);
                        part += (Ob - Oa) * ((Ob + Oa) * N - 2.0 * S);
                    }
                    sum += part;
                }
            }
            if constexpr (OVP) {
                // the pair rule (OQ:311-320): victims' corrections from the list -- or, with the literal pairs, from the row
                const uint32_t kp = (usable && ty_kpos(t) >= 0) ? sort_threshold_key(v, (uint32_t)ty_kpos(t), sd) : kSortSent;
                const uint32_t kn = (usable && ty_kneg(t) >= 0) ? sort_threshold_key(v, (uint32_t)ty_kneg(t), sd) : 0u;
                auto cell_of = [&](float d) {                 // RN(x / s) >= T_k  <=>  x >= X_k
                    uint32_t lo = 0, hi = nthr_t;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (d >= v[66u + mid]) lo = mid + 1u; else hi = mid;
                    }
                    return lo;
                };
                auto correction = [&](uint32_t ka, uint32_t kb) -> double {
                    const bool me = ka >= kp || ka < kn, mo = kb >= kp || kb < kn;
                    if (!(me || mo)) return 0.0;
                    const float vv = sort_unkey(me ? kb : ka);                 // the victim (OQ:315-318)
                    const double O = (double)(v[cell_of(vv / s)] * s), dv = (double)vv;
                    return dv * dv - (O - dv) * (O - dv);
                };
                double sum_c = 0.0, sum_l = 0.0;
                if (cap_list) {
                    const uint32_t *sC = sK + kSortKSh;
                    for (uint32_t i = 0; i < tot_cap; i++) sum_c += correction(sC[2u * i], sC[2u * i + 1u]);
                }
                if (tot_lit != 0u || (tot_cap != 0u && !cap_list)) {
                    const float *grid = ty_grid(t);
                    const int gm = ty_m(t);
                    const float lim = usable ? ty_lim(t) : 0.0f;
                    auto q_of = [&](float xv, float &d, bool &tab) -> float {
                        d = xv / s;
                        tab = fabsf(d) < lim;
                        return tab ? v[cell_of(d)] : sort_literal_q(xv, s, v, grid, gm, d);
                    };
                    auto term_of = [&](float q, float d, bool tab, float xv) -> double {
                        if (tab) {
                            const double er = (double)(q * s) - (double)xv;
                            return er * er;
                        }
                        return sweep_term(q, d, s, xv);
                    };
                    for (uint32_t vi = 0; vi < vpr; vi++) {
                        float xf[EPL];
                        IO<T>::unpack(xr[vi], xf);
#pragma unroll
                        for (int e = 0; e < EPL; e += 2) {
                            const float a = xf[e], b = xf[e + 1];
                            if (!(fabsf(a) < Lx) || !(fabsf(b) < Lx)) {      // a literal pair: the reference sequence with the pair rule
                                float da, db;
                                bool ta, tb;
                                float qa = q_of(a, da, ta), qb = q_of(b, db, tb);
                                const bool me = fabsf(qa) > 32.0f, mo = fabsf(qb) > 32.0f;      // OQ:314
                                const bool ve = mo && !me;
                                qa = qa * (ve ? 0.0f : 1.0f);
                                qb = qb * (me ? 0.0f : 1.0f);
                                sum_l += term_of(qb, db, tb, b);
                                sum_l += term_of(qa, da, ta, a);
                            } else if (!cap_list) {
                                sum_c += correction(sort_key(a), sort_key(b));
                            }
                        }
                    }
                }
                sum += sum_c;
                sum += sum_l;
            } else if (tot_lit) {
                const float *grid = ty_grid(t);
                const int gm = ty_m(t);
                const float lim = usable ? ty_lim(t) : 0.0f;
                double sum_l = 0.0;
                auto lit_term = [&](float xv) {
                    float d = xv / s;
                    if (fabsf(d) < lim) {                     // a step-function element for THIS candidate: (O_J - x)^2 in double
                        uint32_t lo = 0, hi = nthr_t;
                        while (lo < hi) {                     // RN(x / s) >= T_k  <=>  x >= X_k: the cell from the grid-domain thresholds
                            const uint32_t mid = (lo + hi) >> 1;
                            if (d >= v[66u + mid]) lo = mid + 1u; else hi = mid;
                        }
                        const double er = (double)(v[lo] * s) - (double)xv;
                        sum_l += er * er;
                    } else {
                        const float q = sort_literal_q(xv, s, v, grid, gm, d);
                        sum_l += sweep_term(q, d, s, xv);
                    }
                };
                if (lit_list) {
                    for (uint32_t i = 0; i < tot_lit; i++) lit_term(sL[i]);
                } else {
                    // the literal elements of the row, in element order (every lane reads the same vector: a broadcast)
                    for (uint32_t vi = 0; vi < vpr; vi++) {
                        float xf[EPL];
                        IO<T>::unpack(xr[vi], xf);
#pragma unroll
                        for (int e = 0; e < EPL; e++)
                            if (!(fabsf(xf[e]) < Lx)) lit_term(xf[e]);
                    }
                }
                sum += sum_l;
            }
            sse[((size_t)t * ncand_all + c) * rows + row] = sum;
        }
        sort_sync<true>();
    }
}

// slabs of doubles added cell by cell in slab order (blockIdx.y: a group of `per_group` consecutive slabs; a second call
// with the groups as slabs finishes the sum)
static __global__ void __launch_bounds__(256)
k_sort_pt_total(const double *__restrict__ slabs, uint32_t nslab, uint32_t per_group, uint32_t ncell, double *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= ncell) return;
    const uint32_t g0 = blockIdx.y * per_group, g1 = min(nslab, g0 + per_group);
    const double *p = slabs + (size_t)g0 * ncell + i;
    double a = 0.0;
    for (uint32_t g = g0; g < g1; g++, p += ncell) a += p[0];
    out[(size_t)blockIdx.y * ncell + i] = a;
}
static __global__ void __launch_bounds__(256)
k_sort_pt_finish(const double *__restrict__ tot, uint32_t ntc, uint32_t nkg, int ovp, uint32_t ncand, uint32_t ncand_all,
                 double *__restrict__ sse)
{
    const uint32_t tc = blockIdx.x * 256u + threadIdx.x;
    if (tc >= ntc) return;
    const uint32_t nitems = ntc * nkg, ncell = nitems + ntc + (ovp ? 4u * ntc : 0u) + 1u;
    double sum = tot[ncell - 1u];
    for (uint32_t kg = 0; kg < nkg; kg++) sum += tot[tc * nkg + kg];
    sum += tot[nitems + tc];
    if (ovp) {
        const double *c = tot + nitems + ntc + 4u * tc;
        sum += (c[0] + c[1]) + (c[2] + c[3]);
    }
    const uint32_t t = tc / ncand, c = tc - t * ncand;
    sse[(size_t)t * ncand_all + c] = sum;
}

}  // namespace antq

#endif  // ANTQ_K_SORTSEARCH_H
