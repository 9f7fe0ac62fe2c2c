// antq_internal.h -- plan blob layout shared by the host plan builder and the kernels.
#pragma once
#include <stdint.h>

namespace antq {

constexpr uint32_t kPlanMagic = 0x51544E41u;  // "ANTQ"
constexpr uint32_t kPlanVersion = 10;     // 10: HThr::flags carries the scan-order indices of v_lo / v_hi

constexpr uint32_t kPlanScan = 0;  // kernels run the literal scan for every element
constexpr uint32_t kPlanLut = 1;   // table plan: one LDS lookup + one compare per element

// Fast exact division (see div_fast in antq_kernels.hip) is proven for
// |scale| in [2^-40, 2^40]; with |d| < 2^20 this bounds |x| < 2^60, and any x the
// proof excludes on the small side (|x| < 2^-78) maps to |d| < 2^-38.
constexpr float kScaleLo = 0x1p-40f;
constexpr float kScaleHi = 0x1p+40f;
constexpr float kSmallD = 0x1p-37f;   // no threshold may be closer to zero than this
constexpr float kFastDMax = 0x1p+20f; // d at or above this magnitude takes the slow path

struct PlanHeader {      // 80 bytes
    uint32_t magic;
    uint32_t version;
    uint32_t kind;       // kPlanScan / kPlanLut
    uint32_t m;          // grid entries, scan order
    uint32_t m_pad;      // m rounded up to a multiple of 4 (grid area = m_pad floats)
    uint32_t shift;      // key = ((int32)bits(d) >> shift) & keymask
    uint32_t kmin, kmax; // clamp range of the key (signed clamp)
    uint32_t nb;         // buckets per sign = kmax - kmin + 1
    uint32_t keymask;    // 2^(31-shift)-1 when the grid has negative thresholds (key = magnitude key),
                         // 0xffffffff otherwise (negative d keeps a negative key and clamps to kmin)
    uint32_t nbneg;      // table offset of the negative half: nb, or 0 when there is none
    uint32_t n_entries;  // nb + nbneg  (positive buckets, then negative buckets)
    uint32_t bytes;      // total blob size
    float fastlim;       // |d| < fastlim -> table path; otherwise (huge, Inf, NaN) literal scan
    float lo_valid;      // d in [lo_valid, hi_valid] has a grid entry within 102400
    float hi_valid;
    // x-domain fast path (k_fq_xrow): usable when xdom != 0.  Per row the kernel rebuilds the
    // bucket table with thresholds moved into the x domain, U = min{x : fl(x/s) >= T}, and
    // outputs fl(v*s); buckets are then chosen from the APPROXIMATE quotient x*rcp(s) (within
    // 2 ulp of fl(x/s)), which is safe because no threshold lies within 2^-20 (relative) of a
    // bucket edge, and the straight-through step (q-d)+d is provably == q for |d| <= 2 max|v|.
    uint32_t xdom;
    float xlim;          // |x*rcp(s)| < xlim  ->  x-domain table path
    float vout;          // smallest |v| > 32 in the grid (+inf if none): in the x-domain table an output o = fl(v*s)
                         // belongs to an OliVe outlier (|v| > 32, OQ:314) iff |o| >= fl(vout*s)
    // linear key (uniformly spaced thresholds, e.g. the int grids of 8-bit layers: 255 buckets instead of the 2048
    // an exponent/mantissa key needs): bucket = trunc(clamp(fma(d, lin_scale, lin_bias), 0, kmax)); every bucket
    // holds exactly one threshold; no sign half (nbneg = 0), no x-domain variant.
    uint32_t linear;
    float lin_scale;
    float lin_bias;
    // approximate-quotient path in the d domain (quant_vec_a: small groups, tables too big for a per-row copy).  Usable
    // when adom != 0 -- the two conditions of `xdom` that do not depend on the table size: (1) the bucket picked from
    // x*rcp(s) (within 2^-22 of fl(x/s)) holds the threshold that decides fl(x/s) -- thresholds within 2^-20 of a
    // bucket edge are duplicated into the neighbour -- and (2) the straight-through step (q-d)+d is exact in every
    // region for |d| < xlim, so out = fl((q+0)*s).  Elements whose approximate quotient lies within 2^-20 (relative)
    // of their bucket's threshold are redone with the true division.
    uint32_t adom;
    uint32_t atab_slots;  // adom plans: number of a-table slots stored behind the entries (see blob layout below)
    // 16-bit-domain row path (antq_k_hrow.h): rows of bf16 / f16 elements quantised on their bit patterns.  hdom bit 0 / 1:
    // usable for bf16 / f16.  Needs what `xdom` needs except the table-size and bucket-edge conditions (the slot comes from
    // the element's own bits, exactly), plus: at most kHMaxThr thresholds, a key of hmb mantissa bits that puts every
    // threshold (and the row's limit) into a slot of its own for ANY scale, at most kHSlots slots per sign, and -- OliVe --
    // outputs of normal values and of outliers that stay apart after the rounding to 16 bits.
    uint32_t hdom;
    uint32_t h_nthr;      // decision thresholds in `tlist` (ascending; the first h_nneg are negative)
    uint32_t h_nneg;
    uint32_t hshift;      // byte 0: bf16 (7 - hmb), byte 1: f16 (10 - hmb): key = magnitude pattern >> hshift
    uint32_t tlist_off;   // byte offset of HThr tlist[h_nthr] in the blob
    uint32_t pad_[3];
};
static_assert(sizeof(PlanHeader) == 128, "PlanHeader must be 128 bytes");

// one decision threshold of the grid's step function (16-bit-domain row path)
struct HThr {
    float T;          // RN(x / s) >= T  ->  v_hi, else v_lo
    float v_lo;
    float v_hi;
    uint32_t flags;   // bit 0: |v_lo| > 32, bit 1: |v_hi| > 32 (OliVe outlier test, OQ:314); bits 8..17 / 18..27: the scan-order
                      // grid index of v_lo / v_hi (the last duplicate: what the reference scan's `<=` keeps) -- the packed codec's codes
};
static_assert(sizeof(HThr) == 16, "HThr must be 16 bytes");
constexpr uint32_t kHSlots = 128;         // slots per sign of the wave-private table (2 KiB per wavefront)
constexpr uint32_t kHMaxThr = 64;         // decision thresholds (one lane each)

// bits 15 / 31 of LutEntry::idx flag |v_lo| > 32 / |v_hi| > 32 (OliVe outlier test, OQ:314)
constexpr uint32_t kIdxMask = 0x7fffu;

// One bucket of the table: at most one decision threshold T lies inside it.
//   q   = (d >= T) ? v_hi : v_lo
//   idx = (d >= T) ? idx >> 16 : idx & 0xffff      (scan-order index, last duplicate)
struct LutEntry {
    float T;
    float v_lo;
    float v_hi;
    uint32_t idx;
};
static_assert(sizeof(LutEntry) == 16, "LutEntry must be 16 bytes");

// The decision table of the approximate-quotient path (antq_k_approx.h), one slot per bucket: slot = bucket for a linear
// key, 2 * (bucket - kmin) + sign for a float-bits key (an unsigned grid's odd slots all hold the lowest bucket).
//   q = (x - Mp * s >= 0, one correctly rounded fma in double) ? v_hi : v_lo
struct ATabEntry {
    double Mp;     // the rounding boundary below the bucket's threshold T (mid-point of pred(T) and T), moved to the next
                   // double above it when T's mantissa is odd (a tie then rounds DOWN to pred(T)); +inf: no threshold
    float v_lo;    // -0.0 stored as +0.0, as the reference's (q - d) + d yields
    float v_hi;
};
static_assert(sizeof(ATabEntry) == 16, "ATabEntry must be 16 bytes");
constexpr uint32_t kATabMaxSlots = 1024;   // 16 KiB of LDS; bigger tables keep the exact-division path

// blob layout:  PlanHeader | float grid[m_pad] | LutEntry entries[n_entries]
//               | ATabEntry atab[atab_slots] | uint32 aidx[atab_slots rounded up to 4]      (adom plans only)
//               | HThr tlist[h_nthr]                                                        (hdom plans only)
inline const float *plan_grid(const void *blob)
{
    return reinterpret_cast<const float *>(static_cast<const char *>(blob) + sizeof(PlanHeader));
}
inline const LutEntry *plan_entries(const void *blob)
{
    const PlanHeader *h = static_cast<const PlanHeader *>(blob);
    return reinterpret_cast<const LutEntry *>(static_cast<const char *>(blob) + sizeof(PlanHeader) +
                                              sizeof(float) * h->m_pad);
}

inline const HThr *plan_tlist(const void *blob)
{
    const PlanHeader *h = static_cast<const PlanHeader *>(blob);
    return reinterpret_cast<const HThr *>(static_cast<const char *>(blob) + h->tlist_off);
}

}  // namespace antq
