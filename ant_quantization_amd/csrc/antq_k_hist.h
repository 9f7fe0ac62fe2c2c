// antq_k_hist.h -- clip search of a 16-bit tensor with ONE scale on its histogram (round 5)
// Part of libantq's calibration translation unit (antq_search.hip includes it); gfx950 only.
//
// search_mse (AQ/quant_modules.py:287-326) of a per-tensor quantiser scores every clip candidate c (and, under
// search_adaptive_numeric_type, AQ:328-415, every candidate type t) by
//     sse[t, c] = sum_i fl32( fl32|fakequant_t(x_i; alpha_c) - x_i| ^ 2 )
// The direct kernels (antq_k_search.h) evaluate that element by element: n x T x C fake-quants (a BERT-base activation of
// 25 M elements under `ant-int-pot-flint`: 5.6 G of them, 2 ms).  But a bf16 / f16 tensor only takes 65 536 values, and with
// one scale for the whole tensor the term of an element depends on nothing but its bit pattern p (no pair rule: ANT, or OliVe
// with `no_outlier`):
//     sse[t, c] = sum_p count[p] * term(p; t, c)
// so ONE pass over the tensor builds count[] and 65 536 x T x C literal evaluations score it -- the reference's own sequence
// (x / s, scan, (q - d) + d, * s, |. - x|, square; the same fp32 operations the direct kernels perform per element, so the same
// term bit for bit), each multiplied by its count and added in double in one fixed order.  What differs from the direct
// kernels is only the ORDER of the additions (they add eight fp32 terms per vector before widening): relative 1e-7, three
// orders of magnitude inside the tie band the parity tests allow a pick to move in (tests/calib_check.py, 2e-5).
//
// OliVe's pair rule (round 5b).  With outlier-victim pairs (OQ:311-320) an element's output also depends on whether its pair
// partner quantises to an outlier: a victim's output is 0, its term fl32(x^2).  Outliers are the tails of the tensor, so
//     sse[t, c] = sum_p count[p] * term(p; t, c)  +  sum over victims v of ( fl32(x_v^2) - term(p_v; t, c) )
// where the victims of candidate (t, c) are found in a LIST of the pairs that hold an outlier-capable element -- a magnitude at
// or above a conservative bound of the smallest outlier threshold of any candidate (the smallest scale, the codebook with
// the lowest threshold).  k_hist16 writes that list while it counts: every wavefront of the non-negative workgroups compacts
// the qualifying 32-bit words (a word IS a flat pair) of its chunks into its own segment by ballot / prefix popcount, so the
// list's layout -- and with it the order of every double addition in the scoring kernel -- is the same on every run.  The
// scoring kernel learns the exact outlier thresholds of its candidate from the main pass (the smallest magnitude pattern
// per sign whose literal q has |q| > 32; q is monotone in x), then walks the list: two pattern compares per pair, one
// literal evaluation per victim.  A segment that overflows (a tensor with more than ~8 % outlier-capable pairs) raises a flag:
// the histogram kernels then write nothing and the direct kernels, enqueued behind them with that flag as their run condition,
// do the search -- no host decision, no synchronisation.
//
//   k_hist16        1024-thread workgroups, each with a private 32 768-bin histogram of ONE sign in its 128 KiB of LDS (a
//                   workgroup may own the whole 160 KiB of a CU): even workgroups count the non-negative patterns of their
//                   chunk, odd ones the negative patterns; ds_add_u32 without return; the zero patterns -- half of a ReLU
//                   output, all on one LDS address -- are counted by ballot instead.  Each workgroup dumps its bins to its
//                   own slab (plain stores: no global atomics anywhere).
//   k_hist_reduce   count[p] = sum of the slabs (integers: any order gives the same bits).
//   k_hist_score    one workgroup per (type, candidate): every pattern with a non-zero count through the literal sequence.
#ifndef ANTQ_K_HIST_H
#define ANTQ_K_HIST_H

#include "antq_device.h"
#include "antq_k_hrow.h"

namespace antq {

constexpr uint32_t kHistBins = 32768;            // magnitude patterns of one sign
constexpr int kHistMaxG = 128;                   // workgroups per sign (slabs: 2 * kHistMaxG * 128 KiB = 32 MiB)
constexpr size_t kHistSlabBytes = (size_t)kHistBins * 4;
constexpr uint32_t kHistSegCap = 1024;           // pair words per wavefront segment (OliVe's pair rule)
constexpr size_t kHistSlabsBytes = 2 * (size_t)kHistMaxG * kHistSlabBytes;
constexpr size_t kHistCountOff = kHistSlabsBytes;                                   // count[65536]
constexpr size_t kHistSegCountOff = kHistCountOff + 2 * kHistSlabBytes;             // seg_count[kHistMaxG * 16]
constexpr size_t kHistFlagsOff = kHistSegCountOff + (size_t)kHistMaxG * 16 * 4;     // flags[64] (bit 0 of word 0: a segment overflowed; word 1: length of list[])
constexpr size_t kHistSegOff = kHistFlagsOff + 256;                                 // segments[kHistMaxG * 16][kHistSegCap]
constexpr size_t kHistListOff = kHistSegOff + (size_t)kHistMaxG * 16 * kHistSegCap * 4;   // list[]: the segments, packed (k_hist_reduce)
constexpr size_t kHistWorkspaceBytes = kHistListOff + (size_t)kHistMaxG * 16 * kHistSegCap * 4;

// what the pair list needs to know (by value)
struct HistPairs {
    const float *xmax;         // device: the clip statistic (1 float)
    const float *ratios;       // device: the clip ratios (the smallest one gives the smallest scale)
    int ncand;
    float tmin_over_gmax;      // min over the candidate codebooks of (smallest outlier decision threshold / gmax), times 0.999
    uint32_t *seg;             // [G * 16][kHistSegCap]
    uint32_t *seg_count;       // [G * 16]
    int *flags;
    uint32_t *list;            // the segments one after the other (their order: the order of the scoring kernel's additions)
};

// XM: the tensor's abs-max on the way (antq_calibrate with the abs-max statistic: the separate pass over the tensor is not
// run) -- the non-negative workgroups keep the packed maximum of the 16-bit magnitudes they read and close with one
// atomicMax on the bits of the value as a float (non-negative floats order like unsigned integers, NaN above Inf: the very
// value k_absmax leaves, which folds the same converted patterns); *xmax_out is zeroed ahead of the launch.
template <typename T, bool PAIRS, bool XM = false>
__global__ void __launch_bounds__(1024)
k_hist16(const uint4 *__restrict__ x, size_t nv, uint32_t G, uint32_t *__restrict__ slabs, HistPairs hp, float *__restrict__ xmax_out = nullptr)
{
    __shared__ __attribute__((aligned(16))) uint32_t bins[kHistBins];        // 128 KiB, static: one workgroup per CU
    __shared__ uint32_t wmax[16];
    uint32_t pkmax = 0;                                                      // XM: two 16-bit magnitude maxima
    // The two workgroups of a chunk stream (one per sign) sit 8 apart in the grid: workgroup b runs on XCD b mod 8, so both read
    // through the SAME L2 and the tensor leaves HBM once (neighbours 2w / 2w + 1 sat on different XCDs: 2.0 x the tensor's
    // bytes fetched, profiles/r05_pmc_hist.txt).  G is a multiple of 8.
    const uint32_t sign = (blockIdx.x >> 3) & 1u, w = ((blockIdx.x >> 4) << 3) + (blockIdx.x & 7u);
    for (uint32_t i = threadIdx.x; i < kHistBins / 4; i += 1024u) reinterpret_cast<uint4 *>(bins)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    uint32_t zeros = 0;                                                      // this wavefront's count of the zero pattern (lane 0)
    // PAIRS: the magnitude pattern at or above which an element may quantise to an outlier under SOME candidate (a lower
    // bound: the list may hold pairs that never produce a victim, never the other way round); NaN / non-positive statistic:
    // 0 -- everything qualifies, the segments overflow, the direct kernels take over
    uint32_t theta = 0xffffffffu, nseg = 0;
    const uint32_t lane = threadIdx.x & 63u;
    // (the lister falls behind its partner, whose lines have left the L2 by then: 1.9 x the tensor fetched with the pair rule.
    //  Splitting the listing between the two -- 1.0 x -- measured 3-6 % SLOWER: the pass is VALU-bound with the list,
    //  profiles/r05_hist_split_listing.patch)
    const bool lister = PAIRS && sign == 0u;
    uint32_t *myseg = nullptr;
    if (lister) {
        float rmin = hp.ratios[0];
        for (int c = 1; c < hp.ncand; c++) rmin = fminf(rmin, hp.ratios[c]);      // (fminf drops a NaN ratio: checked below)
        for (int c = 0; c < hp.ncand; c++) rmin = hp.ratios[c] == hp.ratios[c] ? rmin : __builtin_nanf("");
        const float tv = hp.xmax[0] * rmin * hp.tmin_over_gmax;
        theta = (tv > 0.0f && tv < 3.0e38f) ? H16<T>::down(tv) : 0u;
        myseg = hp.seg + ((size_t)w * 16u + (threadIdx.x >> 6)) * kHistSegCap;
    }
    auto list_pair = [&](uint32_t word, bool live) {
        const bool hit = live && max(word & 0x7fffu, (word >> 16) & 0x7fffu) >= theta;
        const unsigned long long m = __ballot(hit);
        if (m) {
            const uint32_t off = nseg + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (hit && off < kHistSegCap) myseg[off] = word;
            nseg += (uint32_t)__builtin_popcountll(m);
        }
    };
    auto count = [&](uint32_t h) {
        const bool mine = (h >> 15) == sign;
        const uint32_t mag = h & 0x7fffu;
        const unsigned long long zmask = __ballot(mine && mag == 0u);
        zeros += (uint32_t)__builtin_popcountll(zmask);
        if (mine && mag != 0u) atomicAdd(&bins[mag], 1u);
    };
    // chunks of 1024 vectors, workgroup w takes chunks w, w + G, ...; four 16-byte loads in flight per lane.  The loop bounds
    // are workgroup-uniform (whole chunks): every lane of a wavefront takes part in every ballot.
    const size_t stride = (size_t)G * 1024u;
    size_t cb = (size_t)w * 1024u;
    for (; cb + 3 * stride + 1024u <= nv; cb += 4 * stride) {
        const size_t i = cb + threadIdx.x;
        const uint4 a0 = x[i], a1 = x[i + stride], a2 = x[i + 2 * stride], a3 = x[i + 3 * stride];
        const uint32_t ws[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
        for (int k = 0; k < 16; k++) { count(ws[k] & 0xffffu); count(ws[k] >> 16); }
        if (XM && sign == 0u) pkmax = IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(pkmax, a0), a1), a2), a3);
        if (lister) {
#pragma unroll
            for (int k = 0; k < 16; k++) list_pair(ws[k], true);
        }
    }
    for (; cb < nv; cb += stride) {
        const size_t j = cb + threadIdx.x;
        const bool live = j < nv;
        const uint4 a = live ? x[j] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t ws[4] = {a.x, a.y, a.z, a.w};
        // a lane past the end counts a pattern of the OTHER sign with a non-zero magnitude: nothing
        const uint32_t dead = sign ? 0x0001u : 0x8001u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            count(live ? (ws[k] & 0xffffu) : dead);
            count(live ? (ws[k] >> 16) : dead);
        }
        if (XM && sign == 0u) pkmax = IO<T>::amax_acc(pkmax, a);             // (a lane past the end holds zeros)
        if (lister) {
#pragma unroll
            for (int k = 0; k < 4; k++) list_pair(ws[k], live);
        }
    }
    if (lister && lane == 0u) {
        hp.seg_count[(size_t)w * 16u + (threadIdx.x >> 6)] = min(nseg, kHistSegCap);
        if (nseg > kHistSegCap) atomicOr(hp.flags, 1);
    }
    if ((threadIdx.x & 63u) == 0u && zeros) atomicAdd(&bins[0], zeros);
    if (XM && sign == 0u) {
        const uint32_t m = wave_max_u32(IO<T>::amax_bits(pkmax));            // (the bits of the maximum as a float)
        if (lane == 0u) wmax[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (XM && sign == 0u && threadIdx.x == 0u) {
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) m = max(m, wmax[k]);
        const uint32_t bits = m;
        unsigned int *dst = reinterpret_cast<unsigned int *>(xmax_out);
        if (bits && bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
    uint4 *slab = reinterpret_cast<uint4 *>(slabs + (size_t)(2u * w + sign) * kHistBins);
    for (uint32_t k = threadIdx.x; k < kHistBins / 4; k += 1024u) slab[k] = reinterpret_cast<const uint4 *>(bins)[k];
}

// count[sign * 32768 + b] = sum over the G slabs of that sign.  A workgroup owns 256 consecutive bins of one sign: every
// wavefront adds a quarter of the slabs (16-byte loads, four bins per lane, eight in flight), lane-wise sums through LDS.
// PAIRS: the workgroups also pack the pair list -- workgroup b copies segments [b * spw, (b + 1) * spw) behind the words of
// all earlier segments (their counts summed by every workgroup for itself: at most 2048 integers), so the scoring kernel
// walks ONE contiguous list with independent loads instead of 2048 short ones behind their counts.
template <bool PAIRS>
__global__ void __launch_bounds__(256)
k_hist_reduce(const uint32_t *__restrict__ slabs, uint32_t G, uint32_t *__restrict__ count, HistPairs hp, uint32_t n_seg)
{
    __shared__ uint4 part[4][64];
    __shared__ uint32_t pre[4];
    const uint32_t sign = blockIdx.x >> 7, base = (blockIdx.x & 127u) * 256u;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint4 *src = reinterpret_cast<const uint4 *>(slabs + (size_t)sign * kHistBins + base) + lane;
    constexpr size_t kStep = 2 * (size_t)kHistBins / 4;                       // uint4s from slab (2w + sign) to slab (2(w + 1) + sign)
    uint4 s = make_uint4(0u, 0u, 0u, 0u);
    uint32_t w = wv;
    for (; w + 28u < G; w += 32u) {
        uint4 a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = src[(size_t)(w + 4u * k) * kStep];
#pragma unroll
        for (int k = 0; k < 8; k++) { s.x += a[k].x; s.y += a[k].y; s.z += a[k].z; s.w += a[k].w; }
    }
    for (; w < G; w += 4u) {
        const uint4 a = src[(size_t)w * kStep];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    part[wv][lane] = s;
    __syncthreads();
    if (wv == 0u) {
        const uint4 b = part[1][lane], c = part[2][lane], d = part[3][lane];
        s.x += b.x + c.x + d.x; s.y += b.y + c.y + d.y; s.z += b.z + c.z + d.z; s.w += b.w + c.w + d.w;
        reinterpret_cast<uint4 *>(count + (size_t)sign * kHistBins + base)[lane] = s;
    }
    if (PAIRS) {
        if (hp.flags[0] & 1) return;                     // overflow: the direct kernels do the search
        const uint32_t spw = (n_seg + 255u) / 256u, first = min(blockIdx.x * spw, n_seg), last = min(first + spw, n_seg);
        uint32_t o = 0;
        for (uint32_t i = threadIdx.x; i < first; i += 256u) o += hp.seg_count[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) o += __shfl_xor(o, off, 64);
        if (lane == 0u) pre[wv] = o;
        __syncthreads();
        o = pre[0] + pre[1] + pre[2] + pre[3];
        for (uint32_t sg = first; sg < last; sg++) {
            const uint32_t cnt = hp.seg_count[sg];
            const uint32_t *sp = hp.seg + (size_t)sg * kHistSegCap;
            for (uint32_t i = threadIdx.x; i < cnt; i += 256u) hp.list[o + i] = sp[i];
            o += cnt;
        }
        if (last == n_seg && first < n_seg && threadIdx.x == 0u) hp.flags[1] = (int)o;      // (the workgroup of the last segment)
    }
}

struct HistTypes {
    PlanArgs pa[4];            // the codebooks' plans: the d-domain table decides where it is valid (same value as the scan,
    const uint4 *plan_tab[4];  //   proven by the plan builder's self-check and every kernel's parity tests), the literal scan elsewhere
    float gmax[4];
    int ntypes;
};

// the reference sequence for one pattern: q (before any pair rule) and the term of the element's own output.  The nearest
// grid value of d comes from the plan's table where d lies inside its domain (one LDS read instead of an m-step scan whose
// every step waits for the previous one: the scoring kernel is latency-bound), from the literal scan otherwise.
template <typename T>
__device__ __forceinline__ float hist_term(uint32_t p, const Scale &sc, const PlanArgs &pa, const PlanLds &L, float &q)
{
    const float xv = H16<T>::val(p);
    const float d = xv / sc.s;                   // AQ:541 / OQ:299
    if (pa.kind == kPlanLut && fabsf(d) < pa.fastlim) {          // false for NaN / Inf / beyond the table's domain
        const float dd[1] = {d};
        float qq[1];
        int jj[1];
        lut_lookup<1, false>(pa, L, dd, qq, jj);
        q = qq[0];
    } else {
        int jj;
        q = scan_lds(d, L.grid, (int)pa.m, jj);  // quant_kernel.cu:25-37
    }
    const float tt = (q - d) + d;                // AQ:547 / OQ:323
    const float df = fabsf(tt * sc.s - xv);      // AQ:549, :282
    return df * df;
}

// sse[t * ncand + c] for one (t, c) per workgroup.  Terms in ascending pattern order per thread (p = tid, tid + 1024, ...),
// then a fixed tree; PAIRS: the victims' corrections in the fixed order of the packed list: the same bits on every run.
template <typename T, bool PAIRS>
__global__ void __launch_bounds__(1024)
k_hist_score(const uint32_t *__restrict__ count, const float *__restrict__ xmax, const float *__restrict__ ratios, int ncand,
             HistTypes ht, double *__restrict__ sse, HistPairs hp)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];     // the codebook's plan table (stage_plan)
    __shared__ double part[16];
    __shared__ uint32_t thr[4];                          // per sign: smallest / largest magnitude pattern that quantises to an outlier
    if (PAIRS && (hp.flags[0] & 1)) return;              // the pair list overflowed: the direct kernels behind us do the search
    const int f = (int)blockIdx.x, t = f / ncand, c = f - t * ncand;
    const PlanArgs pa = ht.pa[t];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = ht.plan_tab[t][threadIdx.x];
    const PlanLds L = stage_plan(pa, ht.plan_tab[t], smem, tab0);
    if (threadIdx.x < 4) thr[threadIdx.x] = threadIdx.x < 2 ? 0xffffffffu : 0u;
    __syncthreads();
    const float a = xmax[0] * ratios[c];                 // AQ:300  new_alpha = base_alpha * fl32(i * 0.01)
    const Scale sc = make_scale(a, ht.gmax[t]);
    double acc = 0.0;
    // (eight counts in flight per lane: one at a time the loop was a chain of 64 L2 latencies, 22 us whatever the tensor)
#pragma unroll 1
    for (uint32_t p0 = threadIdx.x; p0 < 65536u; p0 += 8u * 1024u) {
        uint32_t nn[8];
#pragma unroll
        for (int k = 0; k < 8; k++) nn[k] = count[p0 + 1024u * k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t p = p0 + 1024u * k, n = nn[k];
            if (__ballot(n != 0u) == 0ull) continue;     // (whole exponent ranges no element of the tensor lies in)
            if (n != 0u) {
                float q;
                const float term = hist_term<T>(p, sc, pa, L, q);
                acc += (double)n * (double)term;
                if (PAIRS && fabsf(q) > 32.0f) {                                        // OQ:314
                    atomicMin(&thr[p >> 15], p & 0x7fffu);
                    atomicMax(&thr[2u + (p >> 15)], p & 0x7fffu);
                }
            }
        }
    }
    if (PAIRS) {
        __syncthreads();
        // (q is monotone in x up to the scan's horizon -- beyond it no entry lies within 102400 and q is 0 again: the
        //  outliers of a sign are the magnitudes in [lo, hi] among the patterns the tensor holds)
        const uint32_t lo0 = thr[0], lo1 = thr[1], hi0 = thr[2], hi1 = thr[3];
        double corr = 0.0;
        auto victim = [&](uint32_t wd) {
            const uint32_t pe = wd & 0xffffu, po = wd >> 16;
            const uint32_t ge = pe & 0x7fffu, go = po & 0x7fffu;
            const bool me = (pe >> 15) ? (ge >= lo1 && ge <= hi1) : (ge >= lo0 && ge <= hi0);
            const bool mo = (po >> 15) ? (go >= lo1 && go <= hi1) : (go >= lo0 && go <= hi0);
            // OQ:313-320: the odd element is a victim when its even partner is an outlier; the even one when its odd
            // partner is an outlier and it is not one itself
            if (me || mo) {
                const uint32_t v = me ? po : pe;
                float q;
                const float term = hist_term<T>(v, sc, pa, L, q);
                const float xv = H16<T>::val(v);
                corr += (double)(xv * xv) - (double)term;           // a victim's output is 0: its term is fl32(|0 - x|^2)
            }
        };
        // the packed list, eight words in flight per lane; a lane's additions in list order
        const uint32_t total = (uint32_t)hp.flags[1];
#pragma unroll 1
        for (uint32_t i0 = threadIdx.x; i0 < total; i0 += 8u * 1024u) {
            uint32_t wd[8];
#pragma unroll
            for (int k = 0; k < 8; k++) wd[k] = i0 + 1024u * k < total ? hp.list[i0 + 1024u * k] : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (i0 + 1024u * k < total) victim(wd[k]);
        }
        acc += corr;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < 16; k++) s += part[k];
        sse[f] = s;
    }
}

// the flag word of the pair list, zeroed ahead of k_hist16
static __global__ void k_hist_clear_flags(int *flags) { if (threadIdx.x < 64) flags[threadIdx.x] = 0; }

}  // namespace antq

#endif  // ANTQ_K_HIST_H
