// antq_k_sweep.h -- clip search of per-row scales WITHOUT re-evaluating every element for every candidate (round 6)
// Part of libantq's calibration translation unit (antq_search.hip includes it); gfx950 only.
//
// search_mse (AQ/quant_modules.py:287-326, OQ:189-233) scores C clip candidates per row -- and search_adaptive_numeric_type
// (AQ:328-415) does so for every candidate codebook -- by quantising the whole row once per candidate: the direct kernels
// (antq_k_search.h) spend ~12 instructions per element AND candidate (4096^2 fp32, 3 codebooks x 70 candidates: 1.29 ms
// against 8 us of HBM time; OPT-6.7B's 192 weights: 414 ms).  But for a fixed scale s the quantiser is a STEP FUNCTION of x
// (antq_k_hrow.h): out(x) = O_J, J = #{k : x >= X_k}, with the x-domain thresholds X_k = min{x : RN(x / s) >= T_k}
// (x_threshold: exact) and O_j = fl32((v_j + 0) * s), so
//     sum_x (O_J(x) - x)^2 = sum x^2 + n O_0^2 - 2 O_0 S + sum_k [ (O_{k+1}^2 - O_k^2) N_k - 2 (O_{k+1} - O_k) S_k ]
// where N_k, S_k = count and sum of the elements at or above threshold k.  And as the candidate scale grows every threshold
// moves monotonically (away from zero), so "x >= X_k(c)" flips AT MOST ONCE along the candidate list: an element contributes
//   * one entry to a static histogram (its interval at the first candidate), and
//   * one signed event (k, c*) per threshold that sweeps across it -- on average less than one --
// and N_k(c), S_k(c) for ALL candidates come out of a prefix sum along c.  Work per element: two small binary searches
// instead of C evaluations.  Counts and sums are INTEGERS (x in fixed point, 2^-38 of the row statistic's binade: exact for
// every element within 2^15 of it, 3.6e-12 of the statistic otherwise), so LDS atomics in any order give the same bits; the
// closed form is evaluated in double.  What the reference rounds per element -- fl32(out - x), fl32(e^2) -- is not rounded
// here: sums agree with the direct kernels' to ~1e-8 relative (the reference's own reduction noise is 3e-7,
// tests/calib_check.py).
//
// Not every element is a step-function element: NaN / Inf, elements clipped beyond the table's domain or beyond the range
// where the straight-through step (q - d) + d is exact (|x| >= lim * s_min), and -- OliVe's pair rule (OQ:311-320) -- both
// members of a pair one of which could quantise to an outlier under SOME candidate.  Those (planted outliers: ~0.1-0.3 % of
// a 3-sigma-clipped weight) are evaluated literally for every candidate, 64 candidates at a time over the lanes, in element
// order.  Rows whose candidate scales are unusable (zero / denormal / NaN statistic, a non-monotone ratio list) take the
// literal sequence for every element: slow, never wrong.
//
// One single-wavefront workgroup per row; the codebooks of a type selection one after the other (the row is re-read from L2).
#ifndef ANTQ_K_SWEEP_H
#define ANTQ_K_SWEEP_H

#include "antq_device.h"
#include "antq_k_fakequant.h"
#include "antq_k_search.h"

namespace antq {

constexpr int kSweepMaxThr = 64;
constexpr int kSweepMaxCand = 128;
constexpr int kSweepRep = 4;             // replicas of the static histogram (lane & 3): a quarter of the same-address conflicts

struct SweepType {
    const uint4 *tlist;      // device: HThr[n_thr]
    const float *grid;       // device: the codebook in scan order (literal path)
    uint32_t n_thr, m;
    float gmax, lim;         // lim: |x / s| below this -> the step function is the whole story (HArgs::lim)
    int kout_pos, kout_neg;  // OliVe: the threshold between the last normal value and the first outlier, per sign (-1: none)
};

__host__ __device__ inline size_t sweep_lds_bytes(uint32_t nthr, uint32_t cp)
{
    // sEs[nthr * cp] (i64) sHs[4][66] (i64) | sX[nthr * cp] (float) sEn[nthr * cp] (i32) sHn[4][66] (u32) sS[cp] sT[64] sV[66]
    return 8 * ((size_t)nthr * cp + 66 * kSweepRep) + 4 * (2 * (size_t)nthr * cp + 66 * kSweepRep + (size_t)cp + 64 + 66);
}

// the reference sequence for one element at one scale (quant_kernel.cu:25-37 scan, AQ:541-549): q before any pair rule
__device__ __forceinline__ float sweep_literal_q(float xv, float s, const float *__restrict__ grid, int m, float &d)
{
    d = xv / s;
    float sub_min = 102400.0f, z_min = 0.0f;
#pragma unroll 1
    for (int i = 0; i < m; i++) {
        const float g = grid[i];
        const float sub_v = fabsf(d - g);
        if (sub_v <= sub_min) { sub_min = sub_v; z_min = g; }
    }
    return z_min;
}
__device__ __forceinline__ double sweep_term(float q, float d, float s, float xv)
{
    const float tt = (q - d) + d;
    const float df = fabsf(tt * s - xv);
    return (double)(df * df);
}

// x in fixed point, units of 2^(ex - 38), as a 64-bit integer: |x| < 2^(ex + 8) -> |x * F| < 2^46, so adding 1.5 * 2^52 leaves
// rint(x * F) in the low 52 bits of the double (round to nearest even: one fixed rule -- every run forms the same integer)
__device__ __forceinline__ long long sweep_fixed(float xv, double F)
{
    const double d = (double)xv * F + 6755399441055744.0;
    return (long long)(__double_as_longlong(d) & 0x000fffffffffffffll) - 0x0008000000000000ll;
}

// cells of a per-tensor workgroup slab (64-bit each): Es, En, Hs[66], Hn[66] | doubles: Q, exc[128]
__host__ __device__ inline uint32_t sweep_slab_cells(uint32_t nthr, uint32_t cp) { return 2u * nthr * cp + 132u + 129u; }

// Phases 5 and 6 of the sweep: N_k(c), S_k(c) for every candidate (static suffix over the intervals + prefix of the events
// along c: lanes along the candidate axis, a wavefront scan per threshold), then the closed form per candidate.  Q: the
// wavefront's sum of x^2 already reduced over the lanes; exc0 / exc1: this lane's literal terms of candidates lane / lane + 64.
// hrep: replicas of the static histogram to add up (kSweepRep for the row kernel, 1 for totals).
__device__ __forceinline__ void sweep_finish(uint32_t lane, uint32_t nthr, uint32_t ncand, uint32_t cp, int *sEn, long long *sEs,
                                             const unsigned int *sHn, const long long *sHs, int hrep, const float *sS, const float *sV,
                                             double Q, double exc0, double exc1, double unit, double *__restrict__ out, size_t out_stride)
{
    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    };
    long long hn = 0, hs = 0;                                        // lane l: the histogram entry of interval l + 1
    if (lane < nthr)
        for (int r = 0; r < hrep; r++) { hn += (long long)sHn[r * 66 + lane + 1u]; hs += sHs[r * 66 + lane + 1u]; }
    long long bn = hn, bs = hs;                                      // suffix sums over the lanes: B_k = sum_{j > k} H[j] at lane k
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long tn = __shfl_down(bn, off, 64), ts = __shfl_down(bs, off, 64);
        if (lane + (uint32_t)off < 64u) { bn += tn; bs += ts; }
    }
    long long n_tot = __shfl(bn, 0, 64), s_tot = __shfl(bs, 0, 64);
    for (int r = 0; r < hrep; r++) { n_tot += (long long)sHn[r * 66]; s_tot += sHs[r * 66]; }
    for (uint32_t k = 0; k < nthr; k++) {
        long long cn = __shfl(bn, (int)k, 64), cs = __shfl(bs, (int)k, 64);              // carry: starts at the static part
        for (uint32_t c0 = 0; c0 < ncand; c0 += 64u) {
            const uint32_t c = c0 + lane;
            long long en = c < ncand ? (long long)sEn[k * cp + c] : 0, es = c < ncand ? sEs[k * cp + c] : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const long long tn = __shfl_up(en, off, 64), ts = __shfl_up(es, off, 64);
                if (lane >= (uint32_t)off) { en += tn; es += ts; }
            }
            en += cn;
            es += cs;
            if (c < ncand) { sEn[k * cp + c] = (int)en; sEs[k * cp + c] = es; }
            cn = __shfl(en, 63, 64);
            cs = __shfl(es, 63, 64);
        }
    }
    lds_sync();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t cc = lane + 64u * h;
        if (cc < ncand) {
            const float s = sS[cc];
            double Op = (double)(sV[0] * s);
            double sum = Q + (double)n_tot * Op * Op - 2.0 * Op * ((double)s_tot * unit);
            for (uint32_t k = 0; k < nthr; k++) {
                const double On = (double)(sV[k + 1u] * s);
                const double N = (double)sEn[k * cp + cc], S = (double)sEs[k * cp + cc] * unit;
                sum += (On * On - Op * Op) * N - 2.0 * (On - Op) * S;
                Op = On;
            }
            sum += h ? exc1 : exc0;
            out[(size_t)cc * out_stride] = sum;
        }
    }
}

// One codebook per launch; one single-wavefront workgroup per row (PT: per run of vectors of a tensor with ONE scale).
// sse: this codebook's [ncand][rows] block.
template <typename T, bool OVP, bool PT = false>
__global__ void __launch_bounds__(64)
k_search_sweep(const uint4 *__restrict__ x, size_t vpr, size_t rows, const float *__restrict__ xmax,
               const float *__restrict__ ratios, double *__restrict__ sse, SweepType ty, uint32_t ncand, uint32_t cp,
               long long *__restrict__ pt_slabs = nullptr, int fbits = 38)
{
    constexpr int EPL = IO<T>::EPL;
    constexpr int G = 4;                                   // elements handled side by side (a 16-bit vector: two groups)
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t lane = threadIdx.x;
    const uint32_t cl = ncand - 1u, nthr = ty.n_thr;
    // (64-bit arrays first: every pointer stays an LDS pointer -- an address laundered through an integer comes back as a
    //  generic one, and its atomics as FLAT atomics: 40 % of the first version's time)
    long long *sEs = reinterpret_cast<long long *>(smem);                 // [nthr * cp]
    long long *sHs = sEs + (size_t)nthr * cp;                             // [kSweepRep][66]
    float *sX = reinterpret_cast<float *>(sHs + 66 * kSweepRep);          // [nthr * cp]
    int *sEn = reinterpret_cast<int *>(sX + (size_t)nthr * cp);           // [nthr * cp]
    unsigned int *sHn = reinterpret_cast<unsigned int *>(sEn + (size_t)nthr * cp);      // [kSweepRep][66]
    float *sS = reinterpret_cast<float *>(sHn + 66 * kSweepRep);          // [cp]
    float *sT = sS + cp;                                                  // [64]
    float *sV = sT + 64;                                   // [66]: sV[0] = v below the first threshold, sV[k + 1] = v_hi of threshold k
    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): this wavefront's LDS traffic has landed
        __builtin_amdgcn_wave_barrier();
    };
    if (lane < nthr) {
        const uint4 th = ty.tlist[lane];
        sT[lane] = u2f(th.x);
        sV[lane + 1u] = u2f(th.z) + 0.0f;
        if (lane == 0u) sV[0] = u2f(th.y) + 0.0f;
    }
    uint32_t trips = 0;
    while ((1u << trips) < ncand) trips++;                 // bisection steps over [1, cl]
    const float r0 = ratios[0], r_last = ratios[cl];
    const float inv_dr = (cl > 0u && r_last > r0) ? (float)cl / (r_last - r0) : 0.0f;
    unsigned int *myHn = sHn + (lane & (kSweepRep - 1u)) * 66u;
    long long *myHs = sHs + (lane & (kSweepRep - 1u)) * 66u;

    for (size_t row = PT ? 0 : blockIdx.x; row < rows; row += gridDim.x) {
        const float xm = xmax[row];
        const uint4 *xr = x + row * vpr;
        // ---- 1. candidate scales (AQ:300, :536): s_c = fl32(fl32(x_max * ratio_c) / gmax), checked usable and non-decreasing
        bool ok = true;
        for (uint32_t c = lane; c < ncand; c += 64u) {
            const Scale sc = make_scale(xm * ratios[c], ty.gmax);
            sS[c] = sc.s;
            ok = ok && sc.ok && (sc.s > 0.0f);
        }
        lds_sync();
        for (uint32_t c = lane + 1u; c < ncand; c += 64u) ok = ok && (sS[c] >= sS[c - 1u]);
        const bool usable = __ballot(ok) == ~0ull && nthr > 0u;
        // ---- 2. the thresholds in the x domain for every candidate; events and histogram cleared
        const uint32_t cells = nthr * cp;
        for (uint32_t p = lane; p < cells; p += 64u) {
            const uint32_t k = p / cp, c = p - k * cp;
            float X = 0.0f;
            if (usable && c < ncand) {
                bool tok;
                X = x_threshold(sT[k], sS[c], 0.0f, tok);
            }
            sX[p] = X;
            sEn[p] = 0;
            sEs[p] = 0;
        }
        for (uint32_t j = lane; j < 66u * kSweepRep; j += 64u) { sHn[j] = 0u; sHs[j] = 0; }
        lds_sync();
        // ---- 3. per-row constants
        const float s0 = sS[0];
        int ex = 0;
        (void)frexpf(xm, &ex);                              // x_max = f * 2^ex, f in [0.5, 1)
        const double F = __builtin_ldexp(1.0, fbits - ex), unit = __builtin_ldexp(1.0, ex - fbits);      // (fbits: 38; fewer for a
                                                                // whole tensor, so that n * 2^(fbits + 8) stays below 2^62)
        float Lx = usable ? ty.lim * s0 * 0.999f : 0.0f;    // |x| < Lx: a step-function element for EVERY candidate
        Lx = fminf(Lx, __builtin_ldexpf(0.999f, ex + 8));   // ... and its fixed-point image stays below 2^46
        float Xo_pos = __builtin_inff(), Xo_neg = -__builtin_inff();
        if (OVP && usable) {
            if (ty.kout_pos >= 0) Xo_pos = sX[(uint32_t)ty.kout_pos * cp];       // x >= this: an outlier under the smallest scale
            if (ty.kout_neg >= 0) Xo_neg = sX[(uint32_t)ty.kout_neg * cp];       // x <  this: likewise
        }
        double Q = 0.0, exc0 = 0.0, exc1 = 0.0;
        const float s_a = lane < ncand ? sS[lane] : 1.0f, s_b = lane + 64u < ncand ? sS[lane + 64u] : 1.0f;
        // lane k keeps threshold k at the first and at the last candidate: the element loop reads them through readlane
        // (a uniform k: the value sits in an SGPR, the compare is one VALU instruction, nothing waits for LDS)
        const float X0v = lane < nthr ? sX[lane * cp] : __builtin_inff();
        const float Xlv = lane < nthr ? sX[lane * cp + cl] : __builtin_inff();
        const float rs_unit = __builtin_amdgcn_rcpf(xm / ty.gmax);
        // exact zeros (half of a ReLU output) all fall into ONE interval and no threshold ever sweeps across them: counted by
        // ballot -- one LDS address hammered by half the lanes was 40 % of a ReLU tensor's pass
        uint32_t J0z = 0, zeros = 0;
        for (uint32_t k = 0; k < nthr; k++)
            J0z += 0.0f >= __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, X0v), (int)k)) ? 1u : 0u;
        // q of one element at candidate cc, from the tables where the step function holds, literally elsewhere
        auto q_at = [&](float xv, float sc_, uint32_t cc, float &d) -> float {
            d = xv / sc_;
            if (usable && fabsf(d) < ty.lim) {
                uint32_t lo = 0, hi = nthr;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (xv >= sX[mid * cp + cc]) lo = mid + 1u; else hi = mid;
                }
                return sV[lo];
            }
            return sweep_literal_q(xv, sc_, ty.grid, (int)ty.m, d);
        };
        // G elements of one lane side by side
        auto group = [&](const float (&xf)[G], bool live) {
            bool lit[G];
#pragma unroll
            for (int e = 0; e < G; e++) lit[e] = live && !(fabsf(xf[e]) < Lx);
            if (OVP) {
#pragma unroll
                for (int p = 0; p < G / 2; p++) {
                    const float a = xf[2 * p], b = xf[2 * p + 1];
                    const bool cap = (a >= Xo_pos) || (a < Xo_neg) || (b >= Xo_pos) || (b < Xo_neg);
                    const bool both = live && (cap || lit[2 * p] || lit[2 * p + 1]);
                    lit[2 * p] = both;
                    lit[2 * p + 1] = both;
                }
            }
            // interval at the first candidate, J0 = #{k : x >= X[k][0]}, and at the last one, Je: one compare each per
            // threshold (positive thresholds move up with the scale, negative ones down: J only moves towards the middle)
            uint32_t J0[G], Je[G];
#pragma unroll
            for (int e = 0; e < G; e++) { J0[e] = 0u; Je[e] = 0u; }
            for (uint32_t k = 0; k < nthr; k++) {
                const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, X0v), (int)k));
                const float al = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Xlv), (int)k));
#pragma unroll
                for (int e = 0; e < G; e++) {
                    J0[e] += xf[e] >= a0 ? 1u : 0u;
                    Je[e] += xf[e] >= al ? 1u : 0u;
                }
            }
            long long xi[G];
            uint32_t ns[G];
            uint32_t nsmax = 0;
#pragma unroll
            for (int e = 0; e < G; e++) {
                const bool reg0 = live && !lit[e];
                const float xv = reg0 ? xf[e] : 0.0f;
                const bool reg = reg0 && xv != 0.0f;
                zeros += (uint32_t)__builtin_popcountll(__ballot(reg0 && xv == 0.0f));
                Q = __builtin_fma((double)xv, (double)xv, Q);
                xi[e] = sweep_fixed(xv, F);
                if (reg) {
                    atomicAdd(&myHn[J0[e]], 1u);
                    atomicAdd(reinterpret_cast<unsigned long long *>(&myHs[J0[e]]), (unsigned long long)xi[e]);
                }
                ns[e] = reg ? (J0[e] > Je[e] ? J0[e] - Je[e] : Je[e] - J0[e]) : 0u;
                nsmax = max(nsmax, ns[e]);
            }
            // the thresholds that sweep across an element: per (element, threshold) the candidate c* at which "x >= X" flips
            for (uint32_t slot = 0; __ballot(slot < nsmax) != 0ull; slot++) {
                uint32_t ka[G], lo[G], hi[G];
                bool act[G], pos[G];
                bool any_need = false;
#pragma unroll
                for (int e = 0; e < G; e++) {
                    act[e] = slot < ns[e];
                    pos[e] = J0[e] > Je[e];                         // x >= X[k][c] holds at the first candidate, fails from c* on
                    const uint32_t k = act[e] ? min(J0[e], Je[e]) + slot : 0u;
                    ka[e] = k * cp;
                    // X[k][c] ~ T_k * s_c and s_c ~ (x_max / gmax) * ratio_c with ratio_c ~ r0 + c * dr (the lists the
                    // calibration builds): the flip sits where ratio_c passes x / (T_k * x_max / gmax).  Four probes around
                    // the estimate settle it; a list that is no arithmetic progression (nothing brackets) is bisected
                    const float qx = xf[e] * rs_unit * __builtin_amdgcn_rcpf(sT[k]);
                    const float cf = (qx - r0) * inv_dr;
                    int c0 = (cf > -1.0f && cf < 1.0e6f) ? (int)cf + 1 : 1;
                    c0 = c0 < 2 ? 2 : (c0 > (int)cl - 1 ? (int)cl - 1 : c0);
                    // probes at c0 - 2 .. c0 + 1, clamped to [0, cl]; "flipped" at c: pos ? x < X : x >= X (false at 0, true at cl)
                    bool h[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        int c = c0 - 2 + j;
                        c = c < 0 ? 0 : (c > (int)cl ? (int)cl : c);
                        const bool ge = xf[e] >= sX[ka[e] + (uint32_t)c];
                        h[j] = pos[e] ? !ge : ge;
                    }
                    const bool bracket = cl >= 3u && !h[0] && h[3];
                    lo[e] = (uint32_t)(h[1] ? c0 - 1 : (h[2] ? c0 : c0 + 1));
                    hi[e] = lo[e];
                    const bool need = act[e] && !bracket;
                    if (need) { lo[e] = 1u; hi[e] = cl; }
                    any_need = any_need || need;
                }
                if (__ballot(any_need) != 0ull) {
                    for (uint32_t it = 0; it < trips; it++) {
                        float Xm[G];
                        uint32_t mid[G];
#pragma unroll
                        for (int e = 0; e < G; e++) { mid[e] = (lo[e] + hi[e]) >> 1; Xm[e] = sX[ka[e] + mid[e]]; }
#pragma unroll
                        for (int e = 0; e < G; e++) {
                            const bool ge = xf[e] >= Xm[e];
                            const bool hit = pos[e] ? !ge : ge;      // the flipped state holds at `mid`: c* <= mid
                            const bool open = lo[e] < hi[e];
                            hi[e] = (open && hit) ? mid[e] : hi[e];
                            lo[e] = (open && !hit) ? mid[e] + 1u : lo[e];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < G; e++) {
                    if (act[e]) {
                        atomicAdd(&sEn[ka[e] + lo[e]], pos[e] ? -1 : 1);
                        atomicAdd(reinterpret_cast<unsigned long long *>(&sEs[ka[e] + lo[e]]),
                                  (unsigned long long)(pos[e] ? -xi[e] : xi[e]));
                    }
                }
            }
            // the literal elements of this group: every candidate over the lanes, in element order
            constexpr int STEP = OVP ? 2 : 1;
#pragma unroll
            for (int e = 0; e < G; e += STEP) {
                unsigned long long m = __ballot(lit[e]);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const float xa_ = __shfl(xf[e], src, 64);
                    const float xb_ = OVP ? __shfl(xf[e + (OVP ? 1 : 0)], src, 64) : 0.0f;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t cc = lane + 64u * h;
                        if (cc < ncand) {
                            const float s = h ? s_b : s_a;
                            float da, db = 0.0f;
                            float qa = q_at(xa_, s, cc, da), qb = 0.0f;
                            if (OVP) {
                                qb = q_at(xb_, s, cc, db);
                                const bool me = fabsf(qa) > 32.0f, mo = fabsf(qb) > 32.0f;      // OQ:314
                                const bool ve = mo && !me;
                                qa = qa * (ve ? 0.0f : 1.0f);
                                qb = qb * (me ? 0.0f : 1.0f);
                            }
                            double term = sweep_term(qa, da, s, xa_);
                            if (OVP) term += sweep_term(qb, db, s, xb_);
                            if (h) exc1 += term; else exc0 += term;
                        }
                    }
                }
            }
        };
        // ---- 4. the elements (PT: this workgroup's share of the tensor, 64 vectors at a time, grid-strided)
        for (size_t v0 = PT ? (size_t)blockIdx.x * 64u : 0; v0 < vpr; v0 += PT ? (size_t)gridDim.x * 64u : 64u) {
            const size_t vi = v0 + lane;
            const bool live = vi < vpr;
            float xf[EPL];
            {
                const uint4 v = live ? xr[vi] : make_uint4(0u, 0u, 0u, 0u);
                IO<T>::unpack(v, xf);
            }
#pragma unroll
            for (int g = 0; g < EPL; g += G) {
                const float xg[G] = {xf[g], xf[g + 1], xf[g + 2], xf[g + 3]};
                group(xg, live);
            }
        }
        if (lane == 0u && zeros) atomicAdd(&sHn[J0z], zeros);
        // ---- 5 / 6. counts and sums for every candidate, the closed form (sweep_finish) -- or, for a tensor with ONE scale,
        //            this workgroup's integer tables and partial sums to its slab (k_sweep_pt_total adds the slabs)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) Q += __shfl_xor(Q, off, 64);
        lds_sync();
        if (PT) {
            long long *slab = pt_slabs + (size_t)blockIdx.x * sweep_slab_cells(nthr, cp);
            const uint32_t cells_ = nthr * cp;
            for (uint32_t p = lane; p < cells_; p += 64u) { slab[p] = sEs[p]; slab[cells_ + p] = (long long)sEn[p]; }
            for (uint32_t j = lane; j < 66u; j += 64u) {
                long long hn_ = 0, hs_ = 0;
                for (int r = 0; r < kSweepRep; r++) { hn_ += (long long)sHn[r * 66 + j]; hs_ += sHs[r * 66 + j]; }
                slab[2u * cells_ + j] = hs_;
                slab[2u * cells_ + 66u + j] = hn_;
            }
            double *dsl = reinterpret_cast<double *>(slab + 2u * cells_ + 132u);
            if (lane == 0u) dsl[0] = Q;
            dsl[1u + lane] = exc0;
            dsl[65u + lane] = exc1;
            return;
        }
        sweep_finish(lane, nthr, ncand, cp, sEn, sEs, sHn, sHs, kSweepRep, sS, sV, Q, exc0, exc1, unit, sse + row, rows);
        lds_sync();
    }
}

// ---- a tensor with ONE scale (per-tensor quantisers: every activation): the same pass spread over many workgroups ---------
// k_search_sweep<PT> leaves, per workgroup, its integer event / histogram tables and its partial sums in a slab;
// k_sweep_pt_total adds the slabs cell by cell (integers in any order; the doubles -- sum x^2 and the literal terms -- in slab
// order: one fixed order), k_sweep_pt_finish runs phases 5 / 6 on the totals.  Bit-reproducible like the row kernel.
static __global__ void __launch_bounds__(256)
k_sweep_pt_total(const long long *__restrict__ slabs, uint32_t nslab, uint32_t per_group, uint32_t ncells, uint32_t nint,
                 long long *__restrict__ out)
{
    // blockIdx.y: a group of `per_group` consecutive slabs -> out[group][cell] (a second call with the groups as slabs
    // finishes the sum: 2048 slabs added by ONE thread per cell were a chain of 2048 load latencies, 0.4 ms per codebook)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= ncells) return;
    const uint32_t g0 = blockIdx.y * per_group, g1 = min(nslab, g0 + per_group);
    const long long *p = slabs + (size_t)g0 * ncells + i;
    if (i < nint) {
        long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        uint32_t g = g0;
        for (; g + 3u < g1; g += 4u, p += 4 * (size_t)ncells) {
            a0 += p[0]; a1 += p[ncells]; a2 += p[2 * (size_t)ncells]; a3 += p[3 * (size_t)ncells];
        }
        for (; g < g1; g++, p += ncells) a0 += p[0];
        out[(size_t)blockIdx.y * ncells + i] = (a0 + a1) + (a2 + a3);
    } else {
        double a = 0.0;                                               // doubles: slab order, one fixed order
        for (uint32_t g = g0; g < g1; g++, p += ncells) a += __longlong_as_double(p[0]);
        out[(size_t)blockIdx.y * ncells + i] = __double_as_longlong(a);
    }
}

static __global__ void __launch_bounds__(64)
k_sweep_pt_finish(const long long *__restrict__ tot, const float *__restrict__ xmax, const float *__restrict__ ratios,
                  double *__restrict__ sse, SweepType ty, uint32_t ncand, uint32_t cp, int fbits)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t lane = threadIdx.x, nthr = ty.n_thr, cells = nthr * cp;
    long long *sEs = reinterpret_cast<long long *>(smem);
    long long *sHs = sEs + (size_t)nthr * cp;
    float *sX = reinterpret_cast<float *>(sHs + 66 * kSweepRep);
    int *sEn = reinterpret_cast<int *>(sX + (size_t)nthr * cp);
    unsigned int *sHn = reinterpret_cast<unsigned int *>(sEn + (size_t)nthr * cp);
    float *sS = reinterpret_cast<float *>(sHn + 66 * kSweepRep);
    float *sT = sS + cp;
    float *sV = sT + 64;
    const float xm = xmax[0];
    for (uint32_t c = lane; c < ncand; c += 64u) sS[c] = make_scale(xm * ratios[c], ty.gmax).s;
    if (lane < nthr) {
        const uint4 th = ty.tlist[lane];
        sV[lane + 1u] = u2f(th.z) + 0.0f;
        if (lane == 0u) sV[0] = u2f(th.y) + 0.0f;
    }
    for (uint32_t p = lane; p < cells; p += 64u) { sEs[p] = tot[p]; sEn[p] = (int)tot[cells + p]; }
    for (uint32_t j = lane; j < 66u; j += 64u) { sHs[j] = tot[2u * cells + j]; sHn[j] = (unsigned int)tot[2u * cells + 66u + j]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const long long *dt = tot + 2u * cells + 132u;
    int ex = 0;
    (void)frexpf(xm, &ex);
    sweep_finish(lane, nthr, ncand, cp, sEn, sEs, sHn, sHs, 1, sS, sV, __longlong_as_double(dt[0]), __longlong_as_double(dt[1u + lane]),
                 __longlong_as_double(dt[65u + lane]), __builtin_ldexp(1.0, ex - fbits), sse, 1);
}

}  // namespace antq

#endif  // ANTQ_K_SWEEP_H
