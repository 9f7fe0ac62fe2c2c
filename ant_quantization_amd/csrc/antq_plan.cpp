// antq_plan.cpp -- host-side construction of the table plan for one value grid.
//
// The reference kernel (ant_quantization/quant/quant_kernel.cu:25-37) maps a float d
// to  grid[j*],  j* = LAST j minimising fl32|d - grid[j]|  (update on `<=`), or to 0.0
// when no distance is <= 102400.  As a function of d this is a monotone step function:
// Voronoi cells in 1-D are intervals whatever the array order; the array order only
// decides exact / rounding-induced ties and which duplicate is reported.  The plan
// stores that step function EXACTLY:
//   * thresholds T_i = smallest float that the scan maps to the upper of two adjacent
//     distinct grid values, found by bisection over the float total order using the very
//     comparison the scan performs (so rounding-induced ties are reproduced, not modelled);
//   * a bucket table keyed by the top bits of |d| (exponent + a few mantissa bits) such
//     that no bucket holds more than one threshold -> one LDS read + one compare per
//     element on the device;
//   * the validity interval [lo_valid, hi_valid] outside which the scan yields 0.0.
// Conditions that make the step-function view provably complete (no "plateau" where
// far-apart entries tie after rounding) are checked; a grid that fails any of them, or
// fails the self-check against the literal scan, gets a scan plan (kind 0) and the
// kernels run the literal scan for it.  Pure CPU code: no HIP calls in this file.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/antq.h"
#include "antq_internal.h"

namespace antq {
namespace {

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// monotone map float -> uint32 (total order, -0 < +0)
inline uint32_t ord(float f)
{
    uint32_t u = f2u(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float unord(uint32_t o)
{
    return u2f((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
inline float next_up(float f) { return unord(ord(f) + 1); }
inline float next_dn(float f) { return unord(ord(f) - 1); }

// literal scan of quant_kernel.cu:25-37
inline float scan_one(float x, const float *y, int m, int *j_out)
{
    float sub_min = 102400.0f, z_min = 0.0f;
    int j = ANTQ_IDX_NONE;
    for (int i = 0; i < m; i++) {
        float sub_v = fabsf(x - y[i]);
        if (sub_v <= sub_min) { sub_min = sub_v; z_min = y[i]; j = i; }
    }
    *j_out = j;
    return z_min;
}

struct Distinct { float v; int win; };

inline bool pick_hi(float d, const Distinct &lo, const Distinct &hi)
{
    float r_lo = fabsf(d - lo.v);
    float r_hi = fabsf(d - hi.v);
    return r_hi < r_lo || (r_hi == r_lo && hi.win > lo.win);
}

inline uint32_t mag_key(float d, uint32_t shift) { return (f2u(d) & 0x7fffffffu) >> shift; }

void write_scan_plan(void *blob, const float *grid, int m)
{
    PlanHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = kPlanMagic;
    h.version = kPlanVersion;
    h.kind = kPlanScan;
    h.m = (uint32_t)m;
    h.m_pad = (uint32_t)((m + 3) & ~3);
    h.bytes = (uint32_t)(sizeof(PlanHeader) + sizeof(float) * h.m_pad);
    h.lo_valid = -INFINITY;
    h.hi_valid = INFINITY;
    memcpy(blob, &h, sizeof(h));
    float *g = reinterpret_cast<float *>(static_cast<char *>(blob) + sizeof(PlanHeader));
    for (uint32_t i = 0; i < h.m_pad; i++) g[i] = (i < (uint32_t)m) ? grid[i] : 0.0f;
}

// CPU model of the device table path (the kernels implement exactly this).
inline float eval_lut(const PlanHeader &h, const float *grid, const LutEntry *ent, float d, int *idx)
{
    uint32_t u = f2u(d);
    if (!(fabsf(d) < h.fastlim)) return scan_one(d, grid, (int)h.m, idx);  // slow path (also NaN)
    if (h.linear) {
        float kf = fmaf(d, h.lin_scale, h.lin_bias);
        kf = fminf(fmaxf(kf, 0.0f), (float)h.kmax);      // v_med3_f32 on the device (no NaN here)
        const LutEntry &e = ent[(uint32_t)kf];
        bool c = d >= e.T;
        *idx = (int)((c ? (e.idx >> 16) : e.idx) & kIdxMask);
        return c ? e.v_hi : e.v_lo;
    }
    int32_t ks = (int32_t)(((uint32_t)((int32_t)u >> h.shift)) & h.keymask);
    uint32_t k = (uint32_t)(std::min(std::max(ks, (int32_t)h.kmin), (int32_t)h.kmax) - (int32_t)h.kmin);
    const LutEntry &e = ent[k + ((u >> 31) ? h.nbneg : 0u)];
    bool c = d >= e.T;
    uint32_t id = (c ? (e.idx >> 16) : e.idx) & kIdxMask;
    *idx = (int)id;
    return c ? e.v_hi : e.v_lo;
}

}  // namespace
}  // namespace antq

using namespace antq;

extern "C" int antq_plan_build(const float *grid, int m, void *blob, size_t cap)
{
    if (!grid || !blob || m < 1 || m > ANTQ_MAX_GRID) return ANTQ_ERR_ARG;
    const size_t m_pad = (size_t)((m + 3) & ~3);
    if (cap < sizeof(PlanHeader) + 4 * m_pad) return ANTQ_ERR_PLAN;
    write_scan_plan(blob, grid, m);  // default; upgraded below when every check passes
    const int scan_bytes = (int)(sizeof(PlanHeader) + 4 * m_pad);

    for (int i = 0; i < m; i++)
        if (!isfinite(grid[i]) || fabsf(grid[i]) > 65536.0f) return scan_bytes;

    // distinct values in ascending order; win = last scan index holding that value
    std::vector<int> order(m);
    for (int i = 0; i < m; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return grid[a] < grid[b]; });
    std::vector<Distinct> dv;
    for (int t = 0; t < m; t++) {
        int i = order[t];
        if (!dv.empty() && dv.back().v == grid[i]) dv.back().win = std::max(dv.back().win, i);
        else dv.push_back({grid[i], i});
    }
    const int k = (int)dv.size();
    if (k < 2) return scan_bytes;

    // no-plateau conditions (see file header)
    std::vector<double> gap(k - 1);
    for (int i = 0; i + 1 < k; i++) gap[i] = (double)dv[i + 1].v - (double)dv[i].v;
    for (int i = 0; i + 2 < k; i++) {
        double r = gap[i] / gap[i + 1];
        if (r > 1048576.0 || r < 1.0 / 1048576.0) return scan_bytes;
    }
    // Outside [v_0, v_{k-1}] the two outermost entries on that side keep distinct rounded
    // distances while gap > ulp(distance), i.e. for |d| < gap * 2^23 - max|v|; the table path
    // is only used below that magnitude (fastlim), the literal scan beyond it.
    double vabs = std::max(fabs((double)dv[0].v), fabs((double)dv[k - 1].v));
    double d_plateau = std::min(gap[0], gap[k - 2]) * 8388608.0 - vabs;
    if (d_plateau < 4.0 * vabs) return scan_bytes;

    // thresholds
    std::vector<float> T(k - 1);
    for (int i = 0; i + 1 < k; i++) {
        uint32_t lo = ord(dv[i].v), hi = ord(dv[i + 1].v);  // pick_hi(lo)=false, pick_hi(hi)=true
        if (pick_hi(dv[i].v, dv[i], dv[i + 1]) || !pick_hi(dv[i + 1].v, dv[i], dv[i + 1])) return scan_bytes;
        while (hi - lo > 1) {
            uint32_t mid = lo + (hi - lo) / 2;
            if (pick_hi(unord(mid), dv[i], dv[i + 1])) hi = mid; else lo = mid;
        }
        T[i] = unord(hi);
        if (T[i] == 0.0f || fabsf(T[i]) < kSmallD) return scan_bytes;
        if (i > 0 && !(T[i] > T[i - 1])) return scan_bytes;
    }
    // the region containing d = 0 must dequantise to 0 (small-|d| shortcut of the kernels)
    {
        int rho = 0;
        for (int i = 0; i + 1 < k; i++) if (T[i] <= 0.0f) rho = i + 1;
        if (dv[rho].v != 0.0f) return scan_bytes;
    }
    // validity interval
    float hi_valid, lo_valid;
    {
        uint32_t lo = ord(dv[k - 1].v), hi = ord(3.0e38f);
        while (hi - lo > 1) {
            uint32_t mid = lo + (hi - lo) / 2;
            if (fabsf(unord(mid) - dv[k - 1].v) <= 102400.0f) lo = mid; else hi = mid;
        }
        hi_valid = unord(lo);
        lo = ord(-3.0e38f); hi = ord(dv[0].v);
        while (hi - lo > 1) {
            uint32_t mid = lo + (hi - lo) / 2;
            if (fabsf(unord(mid) - dv[0].v) <= 102400.0f) hi = mid; else lo = mid;
        }
        lo_valid = unord(hi);
    }

    // bucketisation: smallest number of mantissa bits that separates the thresholds
    PlanHeader h;
    memset(&h, 0, sizeof(h));
    bool found = false;
    const bool has_neg = T[0] < 0.0f;
    for (uint32_t mb = 0; mb <= 10 && !found; mb++) {
        uint32_t shift = 23 - mb;
        uint32_t kmin = 0xffffffffu, kmax = 0;
        bool ok = true;
        uint32_t prev_pos = 0xffffffffu, prev_neg = 0xffffffffu;
        for (int i = 0; i + 1 < k && ok; i++) {
            uint32_t key = mag_key(T[i], shift);
            kmin = std::min(kmin, key);
            kmax = std::max(kmax, key);
            if (T[i] > 0) { if (key == prev_pos) ok = false; prev_pos = key; }
            else          { if (key == prev_neg) ok = false; prev_neg = key; }
        }
        if (!ok) continue;
        uint32_t nb = kmax - kmin + 1;
        if (nb * (has_neg ? 2u : 1u) > 3072) break;  // 48 KiB of LDS for the table
        h.shift = shift; h.kmin = kmin; h.kmax = kmax; h.nb = nb;
        found = true;
    }
    // Uniformly spaced thresholds (int grids): a linear key needs one bucket per threshold where the float-bits key
    // needs 2^mb per octave (int-8: 255 instead of 2048 buckets, 4 KiB instead of 32 KiB of LDS per workgroup).  Used
    // when the float-bits table is too big for the per-row (x-domain) kernels anyway.  bucket(d) =
    // trunc(clamp(fma(d, scale, bias), 0, k-2)) is monotone in d, so "bucket(T[i]) == i for every i" is all it takes:
    // a d in bucket i then lies strictly between T[i-1] and T[i+1].
    bool linear = false;
    if (k >= 4 && (!found || h.nb * (has_neg ? 2u : 1u) > 128u)) {
        const double w = ((double)T[k - 2] - (double)T[0]) / (double)(k - 2);
        const float scale = (float)(1.0 / w);
        const float bias = (float)(0.5 - (double)T[0] * (double)scale);
        bool ok = w > 0.0 && isfinite(scale) && isfinite(bias);
        for (int i = 0; i + 1 < k && ok; i++) {
            float kf = fmaf(T[i], scale, bias);
            kf = fminf(fmaxf(kf, 0.0f), (float)(k - 2));
            if ((int)kf != i) ok = false;
        }
        if (ok) {
            linear = true;
            h.linear = 1u; h.lin_scale = scale; h.lin_bias = bias;
            h.shift = 0; h.kmin = 0; h.kmax = (uint32_t)(k - 2); h.nb = (uint32_t)(k - 1);
            found = true;
        }
    }
    if (!found) return scan_bytes;
    h.keymask = (has_neg && !linear) ? ((1u << (31 - h.shift)) - 1u) : 0xffffffffu;
    h.nbneg = (has_neg && !linear) ? h.nb : 0u;
    h.magic = kPlanMagic;
    h.version = kPlanVersion;
    h.kind = kPlanLut;
    h.m = (uint32_t)m;
    h.m_pad = (uint32_t)m_pad;
    h.n_entries = h.nb + h.nbneg;
    h.lo_valid = lo_valid;
    h.hi_valid = hi_valid;
    {
        double lim = std::min(std::min(fabs((double)lo_valid), fabs((double)hi_valid)), (double)kFastDMax);
        lim = std::min(lim, d_plateau);
        h.fastlim = (float)lim;
        if ((double)h.fastlim > lim) h.fastlim = next_dn(h.fastlim);
        // every threshold must be strictly inside the table's domain
        if (!(fabsf(T[0]) < h.fastlim) || !(fabsf(T[k - 2]) < h.fastlim)) return scan_bytes;
    }
    h.bytes = (uint32_t)(sizeof(PlanHeader) + 4 * m_pad + sizeof(LutEntry) * h.n_entries);
    if (cap < h.bytes) return ANTQ_ERR_PLAN;

    std::vector<LutEntry> ent(h.n_entries);
    auto region_of = [&](float d) {  // number of thresholds <= d
        return (int)(std::upper_bound(T.begin(), T.end(), d) - T.begin());
    };
    auto mk = [&](int rho_lo, int rho_hi, float thr) {
        LutEntry e;
        e.T = thr;
        e.v_lo = grid[dv[rho_lo].win];
        e.v_hi = grid[dv[rho_hi].win];
        e.idx = (uint32_t)dv[rho_lo].win | ((uint32_t)dv[rho_hi].win << 16);
        if (fabsf(e.v_lo) > 32.0f) e.idx |= 0x8000u;
        if (fabsf(e.v_hi) > 32.0f) e.idx |= 0x80000000u;
        return e;
    };
    for (uint32_t b = 0; linear && b < h.nb; b++) ent[b] = mk((int)b, (int)b + 1, T[b]);
    for (uint32_t b = 0; !linear && b < h.nb; b++) {
        uint32_t key = h.kmin + b;
        float edge = u2f(key << h.shift);  // smallest magnitude of the bucket
        int tp = -1, tn = -1;
        for (int i = 0; i + 1 < k; i++) {
            if (mag_key(T[i], h.shift) != key) continue;
            if (T[i] > 0) tp = i; else tn = i;
        }
        if (tp >= 0) ent[b] = mk(tp, tp + 1, T[tp]);
        else { int r = region_of(edge); ent[b] = mk(r, r, INFINITY); }
        if (has_neg) {
            if (tn >= 0) ent[h.nb + b] = mk(tn, tn + 1, T[tn]);
            else { int r = region_of(-edge); ent[h.nb + b] = mk(r, r, INFINITY); }
        }
    }

    // x-domain path eligibility (see PlanHeader::xdom).  The x-domain kernels pick the bucket from an
    // approximate quotient (within 2 ulp of fl(x/s)), so a threshold that sits within 2^-20 (relative) of
    // a bucket edge must also be found from the neighbouring bucket: it is duplicated into that
    // neighbour when the neighbour has no threshold of its own (harmless for the exact d-domain path:
    // every d of the neighbour lies on one side of it).  If the neighbour is taken, no x-domain path.
    {
        // wave-private table: at most two entries per lane (128), sign-interleaved for float-bits keys.  Measured: a
        // 255-bucket linear table (int-8, four entries per lane and task) is SLOWER per-row than the d-domain path
        // (67 vs 72 % batched bf16), the 63 / 127-bucket ones of int-6 / int-7 are faster (81 vs 75 %).
        bool ok = true;                                 // conditions (1) + (2): `adom`; plus the size and outlier tests: `xdom`
        if (linear) {
            // linear buckets have a threshold each, half a bucket away from both edges: a 2-ulp error of the approximate
            // quotient must not carry a d that is within 2^-20 of a threshold into another bucket
            for (int i = 0; i + 1 < k && ok; i++)
                for (float t : {T[i] * (1.0f + 0x1p-20f), T[i] * (1.0f - 0x1p-20f)}) {
                    float kf = fmaf(t, h.lin_scale, h.lin_bias);
                    kf = fminf(fmaxf(kf, 0.0f), (float)h.kmax);
                    if ((int)kf != i) ok = false;
                }
        }
        auto entry_of = [&](float t) -> LutEntry * {
            const uint32_t key = mag_key(t, h.shift);
            if (key < h.kmin || key > h.kmax) return nullptr;       // clamped region: same bucket as the edge one
            return &ent[(key - h.kmin) + ((t < 0.0f && has_neg) ? h.nb : 0u)];
        };
        for (int i = 0; i + 1 < k && ok && !linear; i++) {
            const float t = T[i];
            LutEntry *own = entry_of(t);
            for (float nb_t : {t * (1.0f + 0x1p-20f), t * (1.0f - 0x1p-20f)}) {
                LutEntry *nb = entry_of(nb_t);
                if (nb == nullptr || nb == own) continue;
                if (nb->T == INFINITY) *nb = *own;                  // duplicate {T, v_lo, v_hi, idx}
                else if (nb->T != t) ok = false;
            }
        }
        // The table stores fl(v*s) in place of ((q - d) + d) * s, which is only the same number when the
        // straight-through step is exact.  q - d is exact (Sterbenz) when q/2 <= d <= 2q, and then (q - d) + d == q;
        // q == 0 is always exact.  Every region of the step function has to satisfy this for all of its d:
        // true of every ANT / OliVe codebook (adjacent magnitudes within a factor of two, zero included), not of
        // arbitrary value lists -- those keep the d-domain kernels, which do the arithmetic literally.
        const bool ok_edges = ok;                        // condition (1): only the approximate-quotient paths need it
        ok = true;
        auto limit_of = [](float v_edge) -> double {      // how far the outermost region may extend
            if (v_edge == 0.0f) return INFINITY;
            return 2.0 * fabs((double)v_edge);
        };
        if (dv[k - 1].v < 0.0f || dv[0].v > 0.0f) ok = false;     // a one-signed grid without zero: d of the other sign
        double xl = std::min({(double)h.fastlim, limit_of(dv[k - 1].v), limit_of(dv[0].v)}) * (1.0 - 0x1p-18);
        h.xlim = (float)xl;
        if ((double)h.xlim > xl) h.xlim = next_dn(h.xlim);
        if (!(fabsf(T[0]) < h.xlim) || !(fabsf(T[k - 2]) < h.xlim)) ok = false;
        for (int i = 0; i < k && ok; i++) {
            const float v = dv[i].v;
            if (v == 0.0f) continue;
            // region of v: [T[i-1], T[i]) clipped to (-xlim, xlim)
            const double lo = i > 0 ? (double)T[i - 1] : -(double)h.xlim;
            const double hi = i + 1 < k ? (double)T[i] : (double)h.xlim;
            if (v > 0.0f) { if (!(lo >= 0.5 * v && hi <= 2.0 * v)) ok = false; }
            else          { if (!(hi <= 0.5 * v && lo >= 2.0 * v)) ok = false; }
        }
        // outlier test on the pre-multiplied outputs: |v| > 32  <=>  |fl(v*s)| >= fl(vout*s) needs the largest
        // normal magnitude and the smallest outlier magnitude to stay distinct after the multiply
        float vout = INFINITY, vnorm = 0.0f;
        for (int i = 0; i < k; i++) {
            const float a = fabsf(dv[i].v);
            if (a > 32.0f) vout = std::min(vout, a);
            else vnorm = std::max(vnorm, a);
        }
        const bool ok_ste = ok;                         // condition (2)
        ok = ok_edges && ok_ste;
        h.adom = ok ? 1u : 0u;                          // (the d-domain path reads the outlier flags of the entries)
        bool ok_out = true;                             // the outlier tests on pre-multiplied outputs
        if (vout < INFINITY && !((double)vout >= (double)vnorm * (1.0 + 0x1p-20))) ok_out = false;
        // ... and an element clipped beyond xlim keeps the table's decision (the extreme value of its sign) but redoes the
        // straight-through step, t = (q - d) + d = q -+ ulp(d) / 2: if that extreme value is an outlier, t has to stay
        // recognisable as one, |t| >= vout.  True of every reference codebook (384 against 48); a value list whose only
        // outlier magnitude IS its extreme (found by the 600-seed fuzz run of round 3: {-0, 35.8}, {-120.9, -0}) keeps the
        // d-domain kernels, whose entries carry the flag.
        for (const float ve : {dv[0].v, dv[k - 1].v}) {
            const double a = fabs((double)ve);
            if (a > 32.0 && !(a - (double)vout >= 0x1p-22 * (double)h.fastlim)) ok_out = false;
        }
        h.vout = vout;
        h.xdom = (ok && ok_out && h.n_entries <= 128) ? 1u : 0u;

        // 16-bit-domain row path (antq_k_hrow.h, PlanHeader::hdom).  The slot of an element comes from its own bit pattern,
        // so condition (1) is not needed; what is: every threshold -- and the row's limit, the sentinel -- in a slot of its
        // own whatever the scale.  A slot spans a factor of at most 1 + 2^-hmb; the 16-bit threshold patterns are within a
        // factor 1 + 2^-MANT above the thresholds themselves: consecutive same-sign thresholds a factor
        // (1 + 2^-hmb)(1 + 2^-MANT) apart can never share one.
        if (ok_ste && ok_out && k - 1 <= (int)kHMaxThr) {
            const double lim = std::min((double)h.fastlim * 0.99999, (double)h.xlim) * 0.999;     // (the kernel's margins)
            double rmin = INFINITY, tmin = INFINITY;
            int n_neg = 0;
            for (int i = 0; i + 1 < k; i++) {
                if (T[i] < 0.0f) n_neg++;
                tmin = std::min(tmin, fabs((double)T[i]));
                if (i + 2 < k && ((T[i] < 0.0f) == (T[i + 1] < 0.0f))) {
                    const double a = fabs((double)T[i]), b = fabs((double)T[i + 1]);
                    rmin = std::min(rmin, std::max(a, b) / std::min(a, b));
                }
            }
            if (n_neg > 0) rmin = std::min(rmin, lim / fabs((double)T[0]));              // the outermost thresholds against the limit
            if (n_neg < k - 1) rmin = std::min(rmin, lim / fabs((double)T[k - 2]));
            // 16-bit outputs of the largest normal magnitude and of the smallest outlier must stay apart
            auto outputs_apart = [&](int mant) { return !(vout < INFINITY) || (double)vout >= (double)vnorm * (1.0 + ldexp(1.0, -(mant - 1))); };
            uint32_t hshift = 0;
            int mant_of[2] = {7, 10};
            for (int t = 0; t < 2; t++) {
                const int mant = mant_of[t];
                int mb = 1;                                                       // (a whole octave per slot is never enough)
                while (mb <= mant && (1.0 + ldexp(1.0, -mb)) * (1.0 + ldexp(1.0, -mant)) > rmin * (1.0 - 1e-6)) mb++;
                if (mb > mant || !outputs_apart(mant)) continue;
                const double slots = ldexp(1.0, mb) * (log2(lim / tmin) + 1.0) + 2.0;       // keys between the innermost threshold and the limit
                if (!(slots <= (double)kHSlots)) continue;
                h.hdom |= 1u << t;
                hshift |= (uint32_t)(mant - mb) << (8 * t);
            }
            if (h.hdom) {
                h.h_nthr = (uint32_t)(k - 1);
                h.h_nneg = (uint32_t)n_neg;
                h.hshift = hshift;
            }
        }
    }

    // self-check against the literal scan
    {
        std::vector<float> pts;
        auto around = [&](float v) {
            float a = v, b = v;
            pts.push_back(v);
            for (int s = 0; s < 3; s++) { a = next_up(a); b = next_dn(b); pts.push_back(a); pts.push_back(b); }
        };
        for (int i = 0; i + 1 < k; i++) around(T[i]);
        for (int i = 0; i < k; i++) around(dv[i].v);
        for (uint32_t b = 0; b <= h.nb; b++) { float e = u2f((h.kmin + b) << h.shift); around(e); around(-e); }
        around(0.0f); around(-0.0f); around(lo_valid); around(hi_valid);
        around(h.fastlim); around(-h.fastlim);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        double span = (double)dv[k - 1].v - (double)dv[0].v;
        for (int i = 0; i < 8192; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            double u = (double)(s >> 11) / 9007199254740992.0;
            pts.push_back((float)((double)dv[0].v - 0.25 * span + 1.5 * span * u));
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            pts.push_back(u2f((uint32_t)s));  // arbitrary bit patterns incl. NaN / Inf / denormals
        }
        for (float d : pts) {
            int j_ref, j_lut;
            float z_ref = scan_one(d, grid, m, &j_ref);
            float z_lut = eval_lut(h, grid, ent.data(), d, &j_lut);
            if (j_ref != j_lut || f2u(z_ref) != f2u(z_lut)) return scan_bytes;
        }
    }

    // a-table (antq_k_approx.h): the same step function with the decision moved onto x, one slot per bucket
    std::vector<ATabEntry> atab;
    std::vector<uint32_t> aidx;
    if (h.adom) {
        const uint32_t nbp = h.n_entries - h.nbneg;
        const uint32_t slots = h.linear ? h.n_entries : 2u * nbp;
        if (slots > kATabMaxSlots) {
            h.adom = 0;
        } else {
            auto conv = [&](const LutEntry &e) {
                ATabEntry a;
                if (!(e.T < INFINITY)) a.Mp = (double)INFINITY;
                else {
                    const double M = 0.5 * ((double)next_dn(e.T) + (double)e.T);
                    a.Mp = (f2u(e.T) & 1u) ? nextafter(M, (double)INFINITY) : M;
                }
                a.v_lo = e.v_lo + 0.0f;
                a.v_hi = e.v_hi + 0.0f;
                return a;
            };
            atab.assign(slots, conv(ent[0]));
            aidx.assign((slots + 3u) & ~3u, ent[0].idx);
            for (uint32_t i = 0; i < h.n_entries; i++) {
                const uint32_t slot = h.linear ? i : (i < nbp ? 2u * i : 2u * (i - nbp) + 1u);
                atab[slot] = conv(ent[i]);
                aidx[slot] = ent[i].idx;
            }
            // (an unsigned grid has no negative buckets: its odd slots keep bucket 0, the region of every negative x)
            h.atab_slots = slots;
            h.bytes += (uint32_t)(sizeof(ATabEntry) * atab.size() + 4 * aidx.size());
            if (cap < h.bytes) return ANTQ_ERR_PLAN;
        }
    }

    std::vector<HThr> tlist;
    if (h.hdom) {
        for (int i = 0; i + 1 < k; i++) {
            HThr t;
            t.T = T[i];
            t.v_lo = grid[dv[i].win];
            t.v_hi = grid[dv[i + 1].win];
            t.flags = (fabsf(t.v_lo) > 32.0f ? 1u : 0u) | (fabsf(t.v_hi) > 32.0f ? 2u : 0u) |
                      ((uint32_t)dv[i].win << 8) | ((uint32_t)dv[i + 1].win << 18);      // scan-order indices (m <= 1024)
            tlist.push_back(t);
        }
        h.tlist_off = h.bytes;
        h.bytes += (uint32_t)(sizeof(HThr) * tlist.size());
        if (cap < h.bytes) return ANTQ_ERR_PLAN;
    }
    char *p = static_cast<char *>(blob);
    memcpy(p, &h, sizeof(h));
    if (!tlist.empty()) memcpy(p + h.tlist_off, tlist.data(), sizeof(HThr) * tlist.size());
    float *g = reinterpret_cast<float *>(p + sizeof(PlanHeader));
    for (size_t i = 0; i < m_pad; i++) g[i] = (i < (size_t)m) ? grid[i] : 0.0f;
    char *q = p + sizeof(PlanHeader) + 4 * m_pad;
    memcpy(q, ent.data(), sizeof(LutEntry) * h.n_entries);
    q += sizeof(LutEntry) * h.n_entries;
    if (!atab.empty()) {
        memcpy(q, atab.data(), sizeof(ATabEntry) * atab.size());
        memcpy(q + sizeof(ATabEntry) * atab.size(), aidx.data(), 4 * aidx.size());
    }
    return (int)h.bytes;
}

extern "C" int antq_plan_kind(const void *blob)
{
    if (!blob) return ANTQ_ERR_ARG;
    const PlanHeader *h = static_cast<const PlanHeader *>(blob);
    if (h->magic != kPlanMagic || h->version != kPlanVersion) return ANTQ_ERR_PLAN;
    return (int)h->kind;
}

extern "C" int antq_plan_bytes(const void *blob)
{
    if (!blob) return ANTQ_ERR_ARG;
    const PlanHeader *h = static_cast<const PlanHeader *>(blob);
    if (h->magic != kPlanMagic || h->version != kPlanVersion) return ANTQ_ERR_PLAN;
    return (int)h->bytes;
}

// Host model of the device element path: lets the CPU test-suite check the plan against
// the oracle scan without a GPU.  q/idx: n outputs for the n inputs d (grid domain).
extern "C" int antq_plan_eval_host(const void *blob, const float *d, float *q, int16_t *idx, size_t n)
{
    if (!blob || !d || !q) return ANTQ_ERR_ARG;
    const PlanHeader *h = static_cast<const PlanHeader *>(blob);
    if (h->magic != kPlanMagic || h->version != kPlanVersion) return ANTQ_ERR_PLAN;
    const float *grid = plan_grid(blob);
    const LutEntry *ent = plan_entries(blob);
    for (size_t i = 0; i < n; i++) {
        int j;
        q[i] = (h->kind == kPlanLut) ? eval_lut(*h, grid, ent, d[i], &j) : scan_one(d[i], grid, (int)h->m, &j);
        if (idx) idx[i] = (int16_t)j;
    }
    return ANTQ_OK;
}

// Host model of the approximate-quotient element path (quant_vec_a in antq_k_approx.h) for one quant group: out[i],
// idx[i] for the inputs x[i] at scale s = alpha / gmax: bucket from x * rs, decision from the sign of the exactly
// rounded x - M' * s in double (M' = the rounding boundary below the bucket's threshold, nudged up for odd thresholds).
// The device takes rs = v_rcp_f32(s), which is only specified to 1 ulp: `rs_ulps` moves the model's reciprocal that many
// ulps off RN(1/s) (tests sweep it and require identical results).  slow[i] (nullable) = 1 where the element took the
// literal sequence.  No OVP (pairs are a lane-local rule on top of q).  ANTQ_ERR_UNSUPPORTED when the plan has no `adom`.
extern "C" int antq_plan_eval_host_a(const void *blob, const float *x, size_t n, float alpha, float gmax, int rs_ulps,
                                     float *out, int16_t *idx, uint8_t *slow)
{
    if (!blob || !x || !out) return ANTQ_ERR_ARG;
    const PlanHeader *h = static_cast<const PlanHeader *>(blob);
    if (h->magic != kPlanMagic || h->version != kPlanVersion) return ANTQ_ERR_PLAN;
    if (h->kind != kPlanLut || !h->adom) return ANTQ_ERR_UNSUPPORTED;
    const float *grid = plan_grid(blob);
    const LutEntry *ent = plan_entries(blob);
    const float s = alpha / gmax;
    const bool ok = (s >= kScaleLo) && (s <= kScaleHi);
    float rs = 1.0f / s;
    for (int k = 0; k < abs(rs_ulps) && ok; k++) rs = rs_ulps > 0 ? next_up(rs) : next_dn(rs);
    const ATabEntry *atab = reinterpret_cast<const ATabEntry *>(ent + h->n_entries);
    const uint32_t *aidx = reinterpret_cast<const uint32_t *>(atab + h->atab_slots);
    for (size_t i = 0; i < n; i++) {
        const float dt = x[i] * rs;
        const bool fast = ok && (fabsf(dt) < h->fastlim * 0.99999f);
        const bool ste = fabsf(dt) < h->xlim;
        float q = 0.0f;
        int j = ANTQ_IDX_NONE;
        if (fast) {
            uint32_t slot;
            if (h->linear) {
                float kf = fmaf(dt, h->lin_scale, h->lin_bias);
                kf = fminf(fmaxf(kf, 0.0f), (float)h->kmax);
                slot = (uint32_t)kf;
            } else {
                const uint32_t u = f2u(dt);
                int32_t ks = (int32_t)(((uint32_t)((int32_t)u >> h->shift)) & h->keymask);
                const uint32_t k = (uint32_t)(std::min(std::max(ks, (int32_t)h->kmin), (int32_t)h->kmax) - (int32_t)h->kmin);
                slot = 2u * k + (u >> 31);
            }
            const ATabEntry &e = atab[slot];
            const bool c = fma(-e.Mp, (double)s, (double)x[i]) >= 0.0;
            q = c ? e.v_hi : e.v_lo;
            j = (int)((c ? (aidx[slot] >> 16) : aidx[slot]) & kIdxMask);
            if (ste) out[i] = q * s;
            else { const float d = x[i] / s; out[i] = ((q - d) + d) * s; }
        } else {
            const float d = x[i] / s;
            q = scan_one(d, grid, (int)h->m, &j);
            const float t = (q - d) + d;
            out[i] = t * s;
        }
        if (idx) idx[i] = (int16_t)j;
        if (slow) slow[i] = fast ? 0 : 1;
    }
    return ANTQ_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Host model of the 16-bit-domain row path (antq_k_hrow.h): one row of n 16-bit patterns x16 at scale alpha / gmax ->
// the output patterns, formed exactly the way the kernel forms them -- the per-row table of 8-byte slots keyed by the
// pattern's top bits, the sentinel slot, the far-clipped arithmetic, the literal sequence for what is left -- so that the
// CPU test-suite can hold the whole construction against the oracle without a GPU.  path[i] (nullable): 0 = table,
// 1 = far-clipped arithmetic, 2 = literal sequence.  ANTQ_ERR_UNSUPPORTED when the plan has no hdom for `dtype`.
// ------------------------------------------------------------------------------------------------------------------
namespace antq {
namespace {

inline uint32_t f32_to_f16(float f)          // round to nearest even, as v_cvt_f16_f32 / tensor.to(float16)
{
    const uint32_t u = f2u(f), sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return sign | 0x7e00u;                 // NaN
    if (a >= 0x47800000u) return sign | 0x7c00u;                // >= 65536 (or Inf): Inf; [65520, 65536) rounds up to Inf below
    if (a < 0x33000000u) return sign;                           // < 2^-25: zero
    const int e = (int)(a >> 23) - 127;
    uint32_t mant = (a & 0x7fffffu) | 0x800000u;
    int shift = e >= -14 ? 13 : (13 + (-14 - e));               // bits to drop
    const uint32_t half = 1u << (shift - 1), mask = (1u << shift) - 1u;
    uint32_t r = mant >> shift;
    const uint32_t rem = mant & mask;
    if (rem > half || (rem == half && (r & 1u))) r++;
    uint32_t h;
    if (e >= -14) h = (uint32_t)((e + 15) << 10) + (r - 0x400u);     // (r may carry into the exponent: still right)
    else h = r;                                                      // subnormal (r may become 0x400 = the smallest normal)
    return sign | h;
}
inline float f16_to_f32(uint32_t h)
{
    const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0x1fu) return u2f(sign | 0x7f800000u | (m << 13));
    if (e == 0) return u2f(sign) + (sign ? -1.0f : 1.0f) * ldexpf((float)m, -24);
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
inline uint32_t bf16_rne(float f)
{
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

struct H16Host {
    int dtype;
    uint32_t up(float U) const     // smallest magnitude pattern with value >= U (U > 0)
    {
        if (dtype == ANTQ_BF16) return (f2u(U) + 0xffffu) >> 16;
        uint32_t h = f32_to_f16(U);
        if (f16_to_f32(h) < U) h++;
        return h;
    }
    uint32_t down(float A) const   // largest magnitude pattern with value <= A (A > 0)
    {
        if (dtype == ANTQ_BF16) return f2u(A) >> 16;
        uint32_t h = f32_to_f16(A);
        if (f16_to_f32(h) > A) h--;
        return h;
    }
    uint32_t out(float o) const { return dtype == ANTQ_BF16 ? (bf16_rne(o) & 0xffffu) : f32_to_f16(o); }
    float val(uint32_t p) const { return dtype == ANTQ_BF16 ? u2f(p << 16) : f16_to_f32(p); }
};

// antq_k_fakequant.h: row_scale / x_threshold, restated on the host
inline float host_row_scale(float alpha, float gmax, bool &ok)
{
    const double inv = 1.0 / (double)gmax;
    float s = (float)((double)alpha * inv);
    ok = (inv != 0.0) && (s >= kScaleLo) && (s <= kScaleHi);
    if (!ok) {
        s = alpha / gmax;
        const float as = fabsf(s);
        ok = (as >= kScaleLo) && (as <= kScaleHi);
    }
    return s;
}
inline float host_x_threshold(float T, float s)
{
    const double M = 0.5 * ((double)next_dn(T) + (double)T);
    const double prod = M * (double)s;
    const float xf = (float)prod;
    const double back = (double)xf;
    const bool t_even = (f2u(T) & 1u) == 0u;
    return ((back > prod) || (back == prod && t_even)) ? xf : next_up(xf);
}

}  // namespace
}  // namespace antq

extern "C" int antq_plan_eval_host_h(const void *blob, const uint16_t *x16, size_t n, float alpha, float gmax, int dtype,
                                     unsigned flags, uint16_t *out16, uint8_t *path)
{
    if (!blob || !x16 || !out16) return ANTQ_ERR_ARG;
    const PlanHeader *h = static_cast<const PlanHeader *>(blob);
    if (h->magic != kPlanMagic || h->version != kPlanVersion) return ANTQ_ERR_PLAN;
    const int t = dtype == ANTQ_BF16 ? 0 : dtype == ANTQ_F16 ? 1 : -1;
    if (t < 0 || h->kind != kPlanLut || !(h->hdom & (1u << t))) return ANTQ_ERR_UNSUPPORTED;
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    const H16Host H{dtype};
    const float *grid = plan_grid(blob);
    const HThr *tl = plan_tlist(blob);
    const uint32_t n_thr = h->h_nthr, n_neg = h->h_nneg, hshift = (h->hshift >> (8 * t)) & 0xffu;
    const float flim = h->fastlim * 0.99999f, lim = fminf(flim, h->xlim);
    float vmin = grid[0], vmax = grid[0];
    for (uint32_t i = 1; i < h->m; i++) { vmin = std::min(vmin, grid[i]); vmax = std::max(vmax, grid[i]); }
    bool ok;
    const float s = host_row_scale(alpha, gmax, ok);
    ok = ok && (s > 0.0f);
    const float rs = 1.0f / s;                      // (the device takes v_rcp_f32: only used for the |x * rs| < flim test)
    std::vector<uint32_t> tab0(2 * kHSlots, 0u), tab1(2 * kHSlots, 0u);      // word 0 / word 1 of every slot
    uint32_t kmin = 0, klim = 0;
    if (ok) {
        const float limx = lim * s * 0.999f;
        const uint32_t lim16 = H.down(fminf(limx, 3.0e38f));
        klim = lim16 >> hshift;
        std::vector<uint32_t> t16(n_thr), key(n_thr), first(n_thr), second(n_thr);
        for (uint32_t i = 0; i < n_thr; i++) {
            const bool neg = i < n_neg;
            const float U = host_x_threshold(tl[i].T, s);
            t16[i] = neg ? H.down(-U) + 1u : H.up(U);
            key[i] = t16[i] >> hshift;
            const uint32_t o_lo = H.out((tl[i].v_lo + 0.0f) * s), o_hi = H.out((tl[i].v_hi + 0.0f) * s);
            first[i] = neg ? o_hi : o_lo;
            second[i] = neg ? o_lo : o_hi;
        }
        const uint32_t kpos = n_neg < n_thr ? key[n_neg] : 0xffffffffu, kneg = n_neg > 0 ? key[n_neg - 1] : 0xffffffffu;
        kmin = std::min(kpos, kneg);
        ok = (klim >= kmin) && (klim - kmin < kHSlots);
        for (uint32_t i = 0; i < n_thr && ok; i++) {
            const bool neg = i < n_neg;
            const bool last = neg ? (i == 0) : (i + 1 == n_thr);
            const uint32_t nxt = last ? klim : (neg ? key[i - 1] : key[i + 1]);
            if (!(key[i] < nxt)) ok = false;
        }
        if (ok && ovp) {
            const uint32_t othr = H.out(h->vout * s) & 0x7fffu;
            if (othr >= (dtype == ANTQ_BF16 ? 0x7f80u : 0x7c00u) && h->vout < INFINITY) ok = false;   // outputs of outliers overflow
        }
        if (ok) {
            auto put = [&](uint32_t k, bool ng, uint32_t w0, uint32_t w1) { const uint32_t j = ((k - kmin) << 1) + (ng ? 1u : 0u); tab0[j] = w0; tab1[j] = w1; };
            for (uint32_t i = 0; i < n_thr; i++) {
                const bool neg = i < n_neg;
                const bool last = neg ? (i == 0) : (i + 1 == n_thr), firsts = neg ? (i + 1 == n_neg) : (i == n_neg);
                const uint32_t nxt = last ? klim : (neg ? key[i - 1] : key[i + 1]);
                const uint32_t sbit = neg ? 0x80000000u : 0u;
                put(key[i], neg, sbit | (t16[i] << 16), first[i] | (second[i] << 16));
                for (uint32_t k = key[i] + 1u; k < nxt; k++) put(k, neg, 0xffffffffu, second[i] | (second[i] << 16));
                if (firsts) for (uint32_t k = kmin; k < key[i]; k++) put(k, neg, 0xffffffffu, first[i] | (first[i] << 16));
                if (last) put(klim, neg, sbit | (lim16 << 16), second[i] | (0xffffu << 16));
            }
            if (n_neg == 0 || n_neg == n_thr) {
                const bool ng = n_neg == 0;
                const uint32_t o = ng ? first[0] : first[n_thr - 1];
                for (uint32_t k = kmin; k < klim; k++) put(k, ng, 0xffffffffu, o | (o << 16));
                put(klim, ng, (ng ? 0x80000000u : 0u) | (lim16 << 16), o | (0xffffu << 16));
            }
        }
    }
    const uint32_t othr = (ok && ovp) ? (H.out(h->vout * s) & 0x7fffu) : 0u;
    auto lookup = [&](uint32_t pat) -> uint32_t {              // hrow_pair for one element (pattern in the high half)
        const uint32_t w = pat << 16;
        const uint32_t key = (w >> (16 + hshift)) & ((1u << (15 - hshift)) - 1u);
        const uint32_t ck = std::min(std::max(key, kmin), klim);
        const uint32_t sl = ((ck - kmin) << 1) | (w >> 31);
        return (w >= tab0[sl]) ? (tab1[sl] >> 16) : (tab1[sl] & 0xffffu);
    };
    auto exact_pair = [&](uint32_t p0, uint32_t p1, bool has1, uint16_t &o0, uint16_t &o1) {
        float x[2] = {H.val(p0), H.val(p1)}, d[2], q[2];
        for (int e = 0; e < 2; e++) { int j; d[e] = x[e] / s; q[e] = scan_one(d[e], grid, (int)h->m, &j); }
        if (ovp && has1) {
            const bool me = fabsf(q[0]) > 32.0f, mo = fabsf(q[1]) > 32.0f;
            q[0] = q[0] * ((mo && !me) ? 0.0f : 1.0f);
            q[1] = q[1] * (me ? 0.0f : 1.0f);
        }
        o0 = (uint16_t)H.out(((q[0] - d[0]) + d[0]) * s);
        o1 = (uint16_t)H.out(((q[1] - d[1]) + d[1]) * s);
    };
    // vectors of 8 elements, as the kernel takes them (a vector with an element in the sentinel slot is formed again as a
    // whole; n need not be a multiple of 8 here: the tail is a short vector)
    for (size_t b = 0; b < n; b += 8) {
        const size_t e1 = std::min(n, b + 8);
        bool sentinel = false, decided = true;
        uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint8_t pth[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) {
            for (size_t i = b; i < e1; i++) {
                o[i - b] = lookup(x16[i]);
                if (o[i - b] == 0xffffu) sentinel = true;
            }
            if (sentinel) {
                for (size_t i = b; i < e1; i++) {
                    if (o[i - b] != 0xffffu) continue;
                    const float xe = H.val(x16[i]);
                    decided = decided && (fabsf(xe * rs) < flim);
                    const float d = xe / s;
                    const float q = (d > 0.0f ? vmax : vmin) + 0.0f;
                    o[i - b] = H.out(((q - d) + d) * s);
                    pth[i - b] = 1;
                }
            }
            if (ovp && decided) {
                for (size_t i = b; i + 1 < e1; i += 2) {
                    const bool me = (o[i - b] & 0x7fffu) >= othr, mo = (o[i + 1 - b] & 0x7fffu) >= othr;
                    if (mo && !me) o[i - b] = 0;
                    if (me) o[i + 1 - b] = 0;
                }
            }
        }
        if (!ok || !decided) {
            for (size_t i = b; i < e1; i += 2) {
                uint16_t o0, o1;
                const bool has1 = i + 1 < e1;
                exact_pair(x16[i], has1 ? x16[i + 1] : 0, has1, o0, o1);
                o[i - b] = o0; pth[i - b] = 2;
                if (has1) { o[i + 1 - b] = o1; pth[i + 1 - b] = 2; }
            }
        }
        for (size_t i = b; i < e1; i++) { out16[i] = (uint16_t)o[i - b]; if (path) path[i] = pth[i - b]; }
    }
    return ANTQ_OK;
}
