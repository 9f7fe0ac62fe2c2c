// antq_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ANT / OliVe
// fake-quant hot path + their C-ABI launchers (include/antq.h).
//
// Reference semantics being reproduced (bit-exact):
//   nearest-value scan   ant_quantization/quant/quant_kernel.cu:20-38
//   Quantizer._forward   ant_quantization/antquant/quant_modules.py:535-551
//   OliVe _forward + outlier-victim pairs  olive_quantization/antquant/quant_modules.py:294-330
//   AsymmetricQuantFunction  ant_quantization/antquant/quant_affine.py:95-115
//
// Design: the op is element-wise and HBM-bound, so the kernels are
// shaped by bytes, not flops: 16 B per lane per access (global_load_dwordx4), a
// wavefront owns a contiguous 1-4 KiB run of ONE quant group (row) so the scale is
// wave-uniform (SGPRs), the grid's decision table sits in LDS (one ds_read_b128 +
// one compare per element instead of the reference's M-step scan), the division
// x/scale is an exact 5-FMA sequence on a per-row reciprocal, and everything the
// reference does in 7-17 separate PyTorch kernels (div, scan, OVP mask ops, STE
// add, rescale) happens in registers between one load and one store.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (no fast-math: every
// float op below must round exactly as written).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>

#include "../../include/antq.h"
#include "antq_internal.h"

namespace antq {

// ------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// Streaming (nontemporal) 16-byte accesses: x is read once and out written once, so the
// lines are marked evict-first instead of thrashing L2 / MALL.  Measured on MI355X
// (tools/ubench.hip): a 4 KiB-per-wave copy runs 5.4 TB/s with nt, 2.9 TB/s without.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream(const uint4 *p)
{
    u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(uint4 *p, const uint4 &v)
{
    u32x4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t *>(p));
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float floatx2_t __attribute__((ext_vector_type(2)));

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2_t as_u16x2(uint32_t u) { return *reinterpret_cast<u16x2_t *>(&u); }
__device__ __forceinline__ uint32_t as_u32(u16x2_t v) { return *reinterpret_cast<uint32_t *>(&v); }

struct bf16_tag {};
struct f16_tag {};

// 16-byte vector <-> EPL floats.  Conversions to the storage type round to nearest-even
// (v_cvt_pk_bf16_f32 / v_cvt_f16_f32), which is what `tensor.to(dtype)` does.
template <typename T> struct IO;
template <> struct IO<float> {
    static constexpr int EPL = 4;
    static constexpr int ESIZE = 4;
    __device__ __forceinline__ static void unpack(const uint4 &v, float (&f)[4])
    {
        f[0] = u2f(v.x); f[1] = u2f(v.y); f[2] = u2f(v.z); f[3] = u2f(v.w);
    }
    __device__ __forceinline__ static uint4 pack(const float (&f)[4])
    {
        return make_uint4(f2u(f[0]), f2u(f[1]), f2u(f[2]), f2u(f[3]));
    }
    __device__ __forceinline__ static float load1(const void *p, size_t i) { return static_cast<const float *>(p)[i]; }
    __device__ __forceinline__ static void store1(void *p, size_t i, float v) { static_cast<float *>(p)[i] = v; }
    // running |x| maximum kept as fp32 magnitude bits (order like unsigned ints; NaN on top)
    __device__ __forceinline__ static uint32_t amax_acc(uint32_t m, const uint4 &v)
    {
        return max(max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, v.z & 0x7fffffffu)), v.w & 0x7fffffffu);
    }
    __device__ __forceinline__ static uint32_t amax_bits(uint32_t m) { return m; }
};
template <> struct IO<bf16_tag> {
    static constexpr int EPL = 8;
    static constexpr int ESIZE = 2;
    __device__ __forceinline__ static void unpack(const uint4 &v, float (&f)[8])
    {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f[2 * i] = u2f(w[i] << 16);
            f[2 * i + 1] = u2f(w[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ static uint32_t pk(float a, float b)
    {
        floatx2_t f = {a, b};
        bf16x2_t h = __builtin_convertvector(f, bf16x2_t);
        return *reinterpret_cast<uint32_t *>(&h);
    }
    __device__ __forceinline__ static uint4 pack(const float (&f)[8])
    {
        return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
    }
    __device__ __forceinline__ static float load1(const void *p, size_t i)
    {
        return u2f((uint32_t) static_cast<const uint16_t *>(p)[i] << 16);
    }
    __device__ __forceinline__ static void store1(void *p, size_t i, float v)
    {
        static_cast<uint16_t *>(p)[i] = (uint16_t)(pk(v, 0.0f) & 0xffffu);
    }
    // running |x| maximum on the packed 16-bit magnitudes (v_pk_max_u16: 2 elements per op)
    __device__ __forceinline__ static uint32_t amax_acc(uint32_t m, const uint4 &v)
    {
        u16x2_t a = as_u16x2(m);
        a = __builtin_elementwise_max(a, as_u16x2(v.x & 0x7fff7fffu));
        a = __builtin_elementwise_max(a, as_u16x2(v.y & 0x7fff7fffu));
        a = __builtin_elementwise_max(a, as_u16x2(v.z & 0x7fff7fffu));
        a = __builtin_elementwise_max(a, as_u16x2(v.w & 0x7fff7fffu));
        return as_u32(a);
    }
    __device__ __forceinline__ static uint32_t amax_bits(uint32_t m) { return max(m & 0xffffu, m >> 16) << 16; }
};
template <> struct IO<f16_tag> {
    static constexpr int EPL = 8;
    static constexpr int ESIZE = 2;
    __device__ __forceinline__ static float h2f(uint32_t bits16)
    {
        uint16_t b = (uint16_t)bits16;
        _Float16 h = *reinterpret_cast<_Float16 *>(&b);
        return (float)h;
    }
    __device__ __forceinline__ static uint32_t f2h(float f)
    {
        _Float16 h = (_Float16)f;
        return (uint32_t) * reinterpret_cast<uint16_t *>(&h);
    }
    __device__ __forceinline__ static void unpack(const uint4 &v, float (&f)[8])
    {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f[2 * i] = h2f(w[i] & 0xffffu);
            f[2 * i + 1] = h2f(w[i] >> 16);
        }
    }
    __device__ __forceinline__ static uint4 pack(const float (&f)[8])
    {
        return make_uint4(f2h(f[0]) | (f2h(f[1]) << 16), f2h(f[2]) | (f2h(f[3]) << 16),
                          f2h(f[4]) | (f2h(f[5]) << 16), f2h(f[6]) | (f2h(f[7]) << 16));
    }
    __device__ __forceinline__ static float load1(const void *p, size_t i)
    {
        return h2f(static_cast<const uint16_t *>(p)[i]);
    }
    __device__ __forceinline__ static void store1(void *p, size_t i, float v)
    {
        static_cast<uint16_t *>(p)[i] = (uint16_t)f2h(v);
    }
    __device__ __forceinline__ static uint32_t amax_acc(uint32_t m, const uint4 &v)
    {
        u16x2_t a = as_u16x2(m);
        a = __builtin_elementwise_max(a, as_u16x2(v.x & 0x7fff7fffu));
        a = __builtin_elementwise_max(a, as_u16x2(v.y & 0x7fff7fffu));
        a = __builtin_elementwise_max(a, as_u16x2(v.z & 0x7fff7fffu));
        a = __builtin_elementwise_max(a, as_u16x2(v.w & 0x7fff7fffu));
        return as_u32(a);
    }
    // half magnitude bits -> fp32 magnitude bits (exact widening)
    __device__ __forceinline__ static uint32_t amax_bits(uint32_t m) { return f2u(h2f(max(m & 0xffffu, m >> 16))); }
};

// Plan fields the kernels need, passed by value (lands in SGPRs).
struct PlanArgs {
    uint32_t kind;
    uint32_t m;
    uint32_t m_pad;
    uint32_t shift;
    uint32_t kmin;
    uint32_t kmax;
    uint32_t keymask;
    uint32_t nbneg;
    float fastlim;
    uint32_t n_entries;
    uint32_t tab_units;  // 16-byte units to stage into LDS: n_entries + m_pad/4
    uint32_t linear;     // PlanHeader::linear: bucket from fma(d, lin_scale, lin_bias) instead of the float's bits
    float lin_scale;
    float lin_bias;
};

// LDS view of the plan: [entries | grid]
struct PlanLds {
    const LutEntry *lut;
    const float *grid;
};

// Stage the table into LDS.  plan_tab points at the blob's grid area; the blob stores
// grid first, entries second, LDS wants entries first (16-byte aligned reads).
__device__ __forceinline__ PlanLds stage_plan(const PlanArgs &pa, const uint4 *__restrict__ plan_tab, uint4 *smem,
                                              uint4 first)
{
    // `first` = plan_tab[threadIdx.x], fetched by the caller ahead of its HBM loads so that
    // the (in-order) wait for it does not cover them.  Source unit i: [0, grid_units) is the
    // grid, the rest are table entries; LDS wants entries first (16-byte aligned b128 reads).
    const uint32_t grid_units = pa.m_pad >> 2;
    if (threadIdx.x < pa.tab_units) {
        const uint32_t i = threadIdx.x;
        smem[(i < grid_units) ? (pa.n_entries + i) : (i - grid_units)] = first;
    }
    for (uint32_t i = threadIdx.x + blockDim.x; i < pa.tab_units; i += blockDim.x)
        smem[(i < grid_units) ? (pa.n_entries + i) : (i - grid_units)] = plan_tab[i];
    PlanLds L;
    L.lut = reinterpret_cast<const LutEntry *>(smem);
    L.grid = reinterpret_cast<const float *>(smem + pa.n_entries);
    return L;
}

// ------------------------------------------------------------------------------------
// Scale of one quant group.  AQ/quant_modules.py:536: scale = alpha / max(grid)  (true
// fp32 division); rs = RN(1/scale) feeds the exact fast division below.
// ------------------------------------------------------------------------------------
struct Scale {
    float s;
    float rs;
    bool ok;  // |s| within [2^-40, 2^40]: div_fast is exact
};
__device__ __forceinline__ Scale make_scale(float alpha, float gmax)
{
    Scale sc;
    sc.s = alpha / gmax;
    float a = fabsf(sc.s);
    sc.ok = (a >= kScaleLo) && (a <= kScaleHi);
    sc.rs = 1.0f / sc.s;
    return sc;
}

// Correctly rounded x/s from rs = RN(1/s) with 1 mul + 4 fma (Markstein): q1 is a
// faithful quotient, the exact residual x - q1*s (one fma) times rs corrects it to
// RN(x/s).  Exact provided no intermediate under/overflows: |s| in [2^-40, 2^40] and
// x == 0 or |x| in [2^-78, 2^60]; callers guarantee that through Scale::ok and the
// kfast / kSmallD plan conditions (antq_internal.h).
__device__ __forceinline__ float div_fast(float x, float s, float rs)
{
    float q0 = x * rs;
    float e0 = __builtin_fmaf(-q0, s, x);
    float q1 = __builtin_fmaf(e0, rs, q0);
    float e1 = __builtin_fmaf(-q1, s, x);
    return __builtin_fmaf(e1, rs, q1);
}

// literal scan, quant_kernel.cu:25-37 (grid in LDS: every lane reads the same address,
// a broadcast).  Used for scan plans and for the rare lanes outside the table's domain.
__device__ __forceinline__ float scan_lds(float d, const float *grid, int m, int &j)
{
    float sub_min = 102400.0f, z_min = 0.0f;
    j = ANTQ_IDX_NONE;
#pragma unroll 1
    for (int i = 0; i < m; i++) {
        float g = grid[i];
        float sub_v = fabsf(d - g);
        if (sub_v <= sub_min) { sub_min = sub_v; z_min = g; j = i; }
    }
    return z_min;
}

// ------------------------------------------------------------------------------------
// Core: EPL elements of one lane, all from the same quant group.
//   in : x[e]           out: o[e] = fl(fl(fl(q-d)+d)*s), j[e] (if IDX)
// ------------------------------------------------------------------------------------
template <int EPL, bool OVP, bool IDX>
__device__ __forceinline__ void quant_vec(const PlanArgs &pa, const PlanLds &L, const Scale &sc,
                                          const float (&x)[EPL], float (&o)[EPL], int (&j)[EPL])
{
    float d[EPL], q[EPL];
    bool fast = (pa.kind == kPlanLut) && sc.ok;
    if (fast) {
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            d[e] = div_fast(x[e], sc.s, sc.rs);
            fast = fast && (fabsf(d[e]) < pa.fastlim);  // false for NaN / Inf / huge
        }
    }
    if (fast && pa.linear) {
        // uniformly spaced thresholds: one bucket per threshold, bucket = trunc(clamp(d * scale + bias))
        const float khi = (float)pa.kmax;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const float kf = __builtin_amdgcn_fmed3f(__builtin_fmaf(d[e], pa.lin_scale, pa.lin_bias), 0.0f, khi);
            const uint32_t k16 = (uint32_t)kf << 4;
            uint4 ent = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(L.lut) + k16);
            if (!IDX) asm volatile("" : "+v"(ent.w));
            const bool c = d[e] >= u2f(ent.x);
            q[e] = c ? u2f(ent.z) : u2f(ent.y);
            if (IDX) j[e] = (int)((c ? (ent.w >> 16) : ent.w) & kIdxMask);
        }
    } else if (fast) {
        // byte offset of the bucket: key*16 straight from the float's bits (exponent + top
        // mantissa bits, shifted so the key lands on bit 4), clamped, plus the negative half.
        const int32_t sh4 = (int32_t)pa.shift - 4;                 // shift >= 13 always
        const int32_t km16 = (int32_t)(pa.keymask << 4);
        const int32_t lo16 = (int32_t)(pa.kmin << 4), hi16 = (int32_t)(pa.kmax << 4);
        const uint32_t neg16 = pa.nbneg << 4;
        const char *lut0 = reinterpret_cast<const char *>(L.lut) - lo16;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int32_t u = (int32_t)f2u(d[e]);
            const int32_t t = (u >> sh4) & km16;
            const int32_t c16 = min(max(t, lo16), hi16);
            const uint32_t sg = (uint32_t)(u >> 31) & neg16;
            uint4 ent = *reinterpret_cast<const uint4 *>(lut0 + c16 + sg);
            if (!IDX) asm volatile("" : "+v"(ent.w));  // keep the read a single ds_read_b128 (b96 is 2x slower)
            const bool c = d[e] >= u2f(ent.x);
            q[e] = c ? u2f(ent.z) : u2f(ent.y);
            if (IDX) j[e] = (int)((c ? (ent.w >> 16) : ent.w) & kIdxMask);
        }
    } else {
        // exact slow path: true division + literal scan (scan plans, odd scales, huge/NaN/Inf)
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            d[e] = x[e] / sc.s;
            int jj;
            q[e] = scan_lds(d[e], L.grid, (int)pa.m, jj);
            if (IDX) j[e] = jj;
        }
    }
    if (OVP) {
        // OQ/quant_modules.py:313-320 on pairs (2p, 2p+1): the odd element is a victim when
        // its even partner is an outlier; the even one when its odd partner is an outlier and
        // it is not one itself.  q * (~victim) keeps the sign of zero, as float*bool does.
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
        float t = (q[e] - d[e]) + d[e];  // AQ:544 / OQ:323 straight-through form
        o[e] = t * sc.s;                 // AQ:546-549
    }
}

template <int EPL>
__device__ __forceinline__ void store_idx(int16_t *idx, size_t vec, const int (&j)[EPL])
{
    // EPL int16 = 8 or 16 bytes, naturally aligned at vec*EPL
    if (EPL == 8) {
        uint4 v = make_uint4(((uint32_t)j[0] & 0xffffu) | ((uint32_t)j[1] << 16),
                             ((uint32_t)j[2] & 0xffffu) | ((uint32_t)j[3] << 16),
                             ((uint32_t)j[4 % EPL] & 0xffffu) | ((uint32_t)j[5 % EPL] << 16),
                             ((uint32_t)j[6 % EPL] & 0xffffu) | ((uint32_t)j[7 % EPL] << 16));
        reinterpret_cast<uint4 *>(idx)[vec] = v;
    } else {
        uint2 v = make_uint2(((uint32_t)j[0] & 0xffffu) | ((uint32_t)j[1] << 16),
                             ((uint32_t)j[2] & 0xffffu) | ((uint32_t)j[3] << 16));
        reinterpret_cast<uint2 *>(idx)[vec] = v;
    }
}

// ------------------------------------------------------------------------------------
// K1a  wave-uniform scale.  A task = up to U*64 consecutive 16-byte vectors of ONE row
// (quant group) = one wavefront; the row's alpha is a scalar load and scale / reciprocal
// are wave-uniform.  All U loads of the task are issued before anything else; with 6-8
// resident wavefronts per SIMD that keeps > 100 KiB per CU in flight, which is what hides
// HBM latency (a persistent ping-pong variant measured slower: it doubles the registers).
// Rows need row_len % EPL == 0; lanes past the row end are masked.
//   vpr = vectors per row, tpr = tasks per row = ceil(vpr / (64*U)).
// DYN: alpha is not read but computed: alpha = max|row| * ratio (requires tpr == 1, the
// whole row sits in this wave's registers; one HBM read of x in total).
// Launch: 256 threads (4 wavefronts); grid = ceil(total_tasks / 4).
// ------------------------------------------------------------------------------------
template <typename T, int U>
__device__ __forceinline__ void task_load(const uint4 *__restrict__ x, const float *__restrict__ alpha, int per_row,
                                          uint32_t task, uint32_t vpr, uint32_t tpr, uint32_t lane, bool dyn,
                                          uint4 (&v)[U], float &a)
{
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint32_t v0 = g * (64u * U) + lane;
    const uint4 *p = x + (size_t)row * vpr;
    // Unconditional loads (lanes past the row end re-read the row's last vector and are
    // masked at the store): no exec-mask branches between the loads, so all U of them are
    // in flight together.
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ld_stream(p + min(v0 + 64u * u, vpr - 1u));
    a = 1.0f;
    if (!dyn) a = alpha[per_row ? row : 0];
}

// wave-wide max of a non-negative float (bit patterns order like integers); NaN propagates
// as in torch.max because a NaN's magnitude bits exceed every finite value's.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t m)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    return m;
}

template <typename T, bool OVP, bool IDX, int U, bool DYN>
__device__ __forceinline__ void task_run(uint4 *__restrict__ out, int16_t *__restrict__ idx,
                                         float *__restrict__ alpha_out, float ratio,
                                         uint32_t task, uint32_t vpr, uint32_t tpr, uint32_t lane, float gmax,
                                         const PlanArgs &pa, const PlanLds &L, const uint4 (&v)[U], float a)
{
    constexpr int EPL = IO<T>::EPL;
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint32_t v0 = g * (64u * U) + lane;
    const size_t base = (size_t)row * vpr + v0;
    if (DYN) {
        // alpha = fl32(max|x| * ratio): AQ/quant_modules.py:474 (x_max) and :300 (x_max * ratio)
        uint32_t m = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t mu = IO<T>::amax_acc(0u, v[u]);
            if (v0 + 64u * u < vpr) m = IO<T>::amax_acc(m, v[u]);  // lanes past the row end hold a duplicate
            (void)mu;
        }
        m = IO<T>::amax_bits(m);
        m = wave_max_u32(m);
        a = u2f(m) * ratio;
        if (alpha_out && lane == 0) alpha_out[row] = a;
    }
    const Scale sc = make_scale(a, gmax);
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (v0 + 64u * u < vpr) {
            float xf[EPL], of[EPL];
            int j[EPL];
            IO<T>::unpack(v[u], xf);
            quant_vec<EPL, OVP, IDX>(pa, L, sc, xf, of, j);
            st_stream(out + base + 64u * u, IO<T>::pack(of));
            if (IDX) store_idx<EPL>(idx, base + 64u * u, j);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep one vector's working set live at a time
    }
}

template <typename T, bool OVP, bool IDX, int U, bool DYN, bool LOOP = false>
__global__ void __launch_bounds__(256)
k_fq_uniform(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
             uint32_t total_tasks, uint32_t vpr, uint32_t tpr,
             const float *__restrict__ alpha, int per_row, float gmax, float ratio,
             float *__restrict__ alpha_out, PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t stride = gridDim.x * 4u;   // one-shot launch: stride >= total_tasks, the loop runs once

    // table fetch is issued FIRST (L2 hit) so that its wait (vmcnt is in-order) does not
    // also wait for the HBM loads of the task, which are issued right behind it
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];

    uint4 v[U];
    float a;
    bool active = task < total_tasks;
    task_load<T, U>(x, alpha, per_row, active ? task : total_tasks - 1u, vpr, tpr, lane, DYN, v, a);

    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    // Big tables (8-bit grids: up to 48 KiB) are staged once per workgroup and amortised over a
    // grid-stride loop of tasks; small tables use a one-shot grid (loop runs once).
    if (!LOOP) {
        if (active) task_run<T, OVP, IDX, U, DYN>(out, idx, alpha_out, ratio, task, vpr, tpr, lane, gmax, pa, L, v, a);
        return;
    }
    while (active) {
        task_run<T, OVP, IDX, U, DYN>(out, idx, alpha_out, ratio, task, vpr, tpr, lane, gmax, pa, L, v, a);
        task += stride;
        active = task < total_tasks;
        if (active) task_load<T, U>(x, alpha, per_row, task, vpr, tpr, lane, DYN, v, a);
    }
}

// ------------------------------------------------------------------------------------
// K1x  x-domain row kernel: the fast path for rows of >= 256 vectors (the headline shape).
//
// K1a spends most of its VALU time on per-element work that only depends on the ROW:
// dividing by the row's scale, and mapping the quotient back (straight-through add,
// multiply by the scale).  Here each wavefront first rebuilds the grid's bucket table for
// ITS row -- lane b owns bucket b:
//     U_b   = min { x : fl(x / s) >= T_b }       (threshold moved into the x domain, exact)
//     O_lo  = fl(v_lo * s),  O_hi = fl(v_hi * s)  (= the reference's output: (q-d)+d == q
//                                                  for |d| <= 2 max|v|, see antq_plan.cpp)
// into a wave-private 1 KiB LDS table (no workgroup barrier), then per element does
//     bucket from x * rcp(s)  (approximate quotient: only picks the bucket; thresholds keep
//                              2^-20 clear of bucket edges, so a 2-ulp error cannot matter)
//     out = (x >= U_b) ? O_hi : O_lo
// i.e. 1 mul + 5 integer ops + 1 LDS read + compare/select: ~10 VALU ops per element instead
// of ~18.  Lanes whose |x * rcp(s)| >= xlim (clipped far beyond the grid, Inf, NaN) and rows
// with an odd scale take the exact reference sequence (true division, literal scan).
// ------------------------------------------------------------------------------------
struct XArgs {
    uint32_t m;
    uint32_t shift;
    uint32_t kmin;
    uint32_t kmax;
    uint32_t keymask;
    uint32_t nbneg;
    uint32_t n_entries;
    float xlim;
    float vout;
};

__device__ __forceinline__ float f_up(float c)   // next float towards +inf (c != 0)
{
    const uint32_t u = f2u(c);
    return u2f((int32_t)u >= 0 ? u + 1u : u - 1u);
}
__device__ __forceinline__ float f_dn(float c)   // next float towards -inf (c != 0)
{
    const uint32_t u = f2u(c);
    return u2f((int32_t)u >= 0 ? u - 1u : u + 1u);
}

// U = min { x : RN(x / s) >= T } for a scale inside div_fast's domain, s > 0, T finite, non-zero.
__device__ __forceinline__ float x_threshold(float T, float s, float rs, bool &ok)
{
    float c = T * s;   // within an ulp or two of the boundary
#pragma unroll
    for (int it = 0; it < 2; it++) { const float p = f_dn(c); if (div_fast(p, s, rs) >= T) c = p; }
#pragma unroll
    for (int it = 0; it < 2; it++) { if (!(div_fast(c, s, rs) >= T)) c = f_up(c); }
    ok = (div_fast(c, s, rs) >= T) && !(div_fast(f_dn(c), s, rs) >= T);
    return c;
}

template <int EPL, bool OVP, bool IDX>
__device__ __forceinline__ void quant_vec_x(const XArgs &xa, const uint4 *wtab, const float *__restrict__ grid,
                                            const Scale &sc, bool rowfast, const float (&x)[EPL], float (&o)[EPL],
                                            int (&j)[EPL])
{
    bool fast = rowfast;
    float dt[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        dt[e] = x[e] * sc.rs;
        fast = fast && (fabsf(dt[e]) < xa.xlim);
    }
    if (fast) {
        // slot = 2 * clamp(key) + sign: positive and negative buckets interleaved, so that the sign costs one
        // v_alignbit and the clamp one v_med3 (an unsigned grid keeps a negative key: it clamps to kmin, slot 1)
        const int32_t sh = (int32_t)xa.shift, km = (int32_t)xa.keymask;
        const int32_t lo = (int32_t)xa.kmin, hi = (int32_t)xa.kmax;
        const char *t0 = reinterpret_cast<const char *>(wtab) - (lo << 5);
        bool isout[EPL];
        const float othr = xa.vout * sc.s;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int32_t u = (int32_t)f2u(dt[e]);
            const int32_t t = (u >> sh) & km;
            int32_t ck;
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(ck) : "v"(t), "v"(lo), "v"(hi));
            const uint32_t slot = __builtin_amdgcn_alignbit((uint32_t)ck, (uint32_t)u, 31);
            uint4 ent = *reinterpret_cast<const uint4 *>(t0 + (slot << 4));
            if (!IDX) asm volatile("" : "+v"(ent.w));
            const bool c = x[e] >= u2f(ent.x);
            o[e] = c ? u2f(ent.z) : u2f(ent.y);
            if (OVP) isout[e] = fabsf(o[e]) >= othr;      // |v| > 32 (PlanHeader::vout)
            if (IDX) j[e] = (int)((c ? (ent.w >> 16) : ent.w) & kIdxMask);
        }
        if (OVP) {
#pragma unroll
            for (int p = 0; p < EPL / 2; p++) {
                const bool me = isout[2 * p], mo = isout[2 * p + 1];
                const bool ve = mo && !me;
                o[2 * p] = ve ? 0.0f : o[2 * p];          // ((q*0 - d) + d) * s == +0 for s > 0
                o[2 * p + 1] = me ? 0.0f : o[2 * p + 1];
                if (IDX) {
                    if (ve) j[2 * p] = ANTQ_IDX_VICTIM;
                    if (me) j[2 * p + 1] = ANTQ_IDX_VICTIM;
                }
            }
        }
    } else {
        // exact reference sequence for this lane's EPL elements
        float d[EPL], q[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            d[e] = x[e] / sc.s;
            int jj;
            q[e] = scan_lds(d[e], grid, (int)xa.m, jj);
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

// Body of the x-domain row kernel for one wavefront task (shared by k_fq_xrow and k_fq_batch).
template <typename T, bool OVP, bool IDX, int U, bool DYN, int WPR>
__device__ __forceinline__ void xrow_task(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
                                          uint32_t task, uint32_t vpr, uint32_t tpr,
                                          const float *__restrict__ alpha, int per_row, float gmax, float ratio,
                                          float *__restrict__ alpha_out, const XArgs &xa,
                                          const uint4 *__restrict__ entries, const float *__restrict__ grid,
                                          uint4 *wtab, uint32_t lane, uint32_t wv)
{
    constexpr int EPL = IO<T>::EPL;
    // static bucket entries of this lane (L2 hits), issued ahead of the HBM loads; tables of
    // 65..128 buckets (e.g. unsigned int-4) give every lane a second entry
    uint4 ent = make_uint4(f2u(__builtin_inff()), 0u, 0u, 0u), ent2 = ent;
    if (lane < xa.n_entries) ent = entries[lane];
    const bool two = xa.n_entries > 64u;
    if (two && lane + 64u < xa.n_entries) ent2 = entries[lane + 64u];

    uint4 v[U];
    float a;
    task_load<T, U>(x, alpha, per_row, task, vpr, tpr, lane, DYN, v, a);

    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint32_t v0 = g * (64u * U) + lane;
    const size_t base = (size_t)row * vpr + v0;
    if (DYN) {
        uint32_t m = 0;
#pragma unroll
        for (int u = 0; u < U; u++)
            if (v0 + 64u * u < vpr) m = IO<T>::amax_acc(m, v[u]);
        m = wave_max_u32(IO<T>::amax_bits(m));
        if (WPR == 4) {
            // the row spans the 4 wavefronts of this workgroup (tpr == 4): combine their maxima
            __shared__ uint32_t wmax[4];
            if (lane == 0) wmax[wv] = m;
            __syncthreads();
            m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        }
        a = u2f(m) * ratio;
        if (alpha_out && lane == 0 && (WPR == 1 || wv == 0)) alpha_out[row] = a;
    }
    const Scale sc = make_scale(a, gmax);

    // per-row table: thresholds into the x domain, outputs pre-multiplied by the scale
    bool rowfast = sc.ok && (sc.s > 0.0f);
    {
        bool ok = true;
        float Ux = u2f(ent.x);
        if (rowfast && lane < xa.n_entries && Ux < __builtin_inff()) Ux = x_threshold(Ux, sc.s, sc.rs, ok);
        rowfast = rowfast && __all(ok);
        // (v + 0) * s: a -0.0 grid entry must come out as +0.0, like the reference's (q - d) + d
        // entry i is positive bucket i (slot 2i) or negative bucket i - nb (slot 2(i - nb) + 1)
        const uint32_t nbp = xa.n_entries - xa.nbneg;
        const uint4 w0 = make_uint4(f2u(Ux), f2u((u2f(ent.y) + 0.0f) * sc.s), f2u((u2f(ent.z) + 0.0f) * sc.s), ent.w);
        if (lane < xa.n_entries) wtab[lane < nbp ? 2u * lane : 2u * (lane - nbp) + 1u] = w0;
        if (xa.nbneg == 0u && lane == 0u) wtab[1] = w0;     // unsigned grid: every negative x lands in slot 1
        if (two) {
            bool ok2 = true;
            float U2 = u2f(ent2.x);
            const uint32_t i2 = lane + 64u;
            if (rowfast && i2 < xa.n_entries && U2 < __builtin_inff()) U2 = x_threshold(U2, sc.s, sc.rs, ok2);
            rowfast = rowfast && __all(ok2);
            if (i2 < xa.n_entries)
                wtab[i2 < nbp ? 2u * i2 : 2u * (i2 - nbp) + 1u] =
                    make_uint4(f2u(U2), f2u((u2f(ent2.y) + 0.0f) * sc.s), f2u((u2f(ent2.z) + 0.0f) * sc.s), ent2.w);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed

#pragma unroll
    for (int u = 0; u < U; u++) {
        if (v0 + 64u * u < vpr) {
            float xf[EPL], of[EPL];
            int j[EPL];
            IO<T>::unpack(v[u], xf);
            quant_vec_x<EPL, OVP, IDX>(xa, wtab, grid, sc, rowfast, xf, of, j);
            st_stream(out + base + 64u * u, IO<T>::pack(of));
            if (IDX) store_idx<EPL>(idx, base + 64u * u, j);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T, bool OVP, bool IDX, int U, bool DYN, int WPR = 1>
__global__ void __launch_bounds__(256)
k_fq_xrow(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
          uint32_t total_tasks, uint32_t vpr, uint32_t tpr,
          const float *__restrict__ alpha, int per_row, float gmax, float ratio,
          float *__restrict__ alpha_out, XArgs xa, const uint4 *__restrict__ entries,
          const float *__restrict__ grid)
{
    __shared__ __attribute__((aligned(16))) uint4 wtab_all[4][256];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wv);
    if (task >= total_tasks) return;   // no workgroup barrier in this kernel (WPR == 4: whole workgroups exit)
    xrow_task<T, OVP, IDX, U, DYN, WPR>(x, out, idx, task, vpr, tpr, alpha, per_row, gmax, ratio, alpha_out, xa, entries,
                                        grid, wtab_all[wv], lane, wv);
}

// ------------------------------------------------------------------------------------
// K1b  per-lane scale: small rows / small groups (vpr < 64: several quant groups share a
// wavefront, e.g. group-16 = 2 bf16 lanes or 4 fp32 lanes per group).  Each lane gathers
// its own alpha and builds its own scale.  vshift >= 0 when vpr is a power of two.
// ------------------------------------------------------------------------------------
template <typename T, bool OVP, bool IDX, int U, bool DYN>
__global__ void __launch_bounds__(256)
k_fq_lane(const uint4 *__restrict__ x, uint4 *__restrict__ out, int16_t *__restrict__ idx,
          size_t n_vec, uint32_t vpr, int vshift,
          const float *__restrict__ alpha, int per_row, float gmax, float ratio,
          float *__restrict__ alpha_out, PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    constexpr int EPL = IO<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const size_t first = ((size_t)blockIdx.x * U) * 256u + threadIdx.x;
    uint4 v[U];
    float a[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * 256u;
        v[u] = make_uint4(0, 0, 0, 0);
        a[u] = 1.0f;
        if (vi < n_vec) {
            v[u] = ld_stream(x + vi);
            if (!DYN) {
                size_t row = 0;
                if (per_row) row = (vshift >= 0) ? (vi >> vshift) : (vi / vpr);
                a[u] = alpha[row];
            }
        }
    }
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * 256u;
        float xf[EPL];
        IO<T>::unpack(v[u], xf);
        if (DYN) {
            // group = vpr (power of two <= 32) adjacent lanes; butterfly max inside the group.
            // Lanes past n_vec hold zeros and belong to no real group (n_vec % vpr == 0).
            uint32_t m = IO<T>::amax_bits(IO<T>::amax_acc(0u, v[u]));
            for (uint32_t off = 1; off < vpr; off <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, (int)off, 64));
            a[u] = u2f(m) * ratio;
            if (alpha_out && vi < n_vec && (vi & (vpr - 1)) == 0) alpha_out[vi >> vshift] = a[u];
        }
        if (vi < n_vec) {
            const Scale sc = make_scale(a[u], gmax);
            float of[EPL];
            int j[EPL];
            quant_vec<EPL, OVP, IDX>(pa, L, sc, xf, of, j);
            st_stream(out + vi, IO<T>::pack(of));
            if (IDX) store_idx<EPL>(idx, vi, j);
        }
    }
}

// ------------------------------------------------------------------------------------
// K1c  element-granular fallback: any row_len (e.g. conv1's K = 147), any alignment,
// and the < EPL tail of a per-tensor launch.  One thread per PAIR (2p, 2p+1) of the flat
// tensor so the OliVe victim rule stays inside a thread; with an odd element count the
// last element's "partner" is element 0 (torch.roll wrap-around, OQ:315-318).
//   elements [e0, e0 + n_here) of a tensor with n_total elements, e0 even.
// ------------------------------------------------------------------------------------
template <typename T, bool OVP, bool IDX>
__global__ void __launch_bounds__(256)
k_fq_scalar(const void *__restrict__ x, void *__restrict__ out, int16_t *__restrict__ idx,
            size_t e0, size_t n_here, size_t n_total, size_t row_len,
            const float *__restrict__ alpha, int per_row, float gmax,
            PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    const size_t p = (size_t)blockIdx.x * 256u + threadIdx.x;
    const size_t i0 = e0 + 2 * p;
    if (2 * p >= n_here) return;
    const bool has_odd = (2 * p + 1 < n_here);
    const size_t i1 = has_odd ? i0 + 1 : 0;  // wrap partner (only read when !has_odd && OVP)
    const bool need1 = has_odd || (OVP && i0 + 1 == n_total);

    float xs[2] = {IO<T>::load1(x, i0), need1 ? IO<T>::load1(x, i1) : 0.0f};
    float d[2], q[2];
    float s[2];
    int j[2] = {ANTQ_IDX_NONE, ANTQ_IDX_NONE};
    for (int e = 0; e < 2; e++) {
        const size_t ii = e ? i1 : i0;
        const float a = alpha[per_row ? (ii / row_len) : 0];
        s[e] = a / gmax;
        d[e] = xs[e] / s[e];
        int jj;
        q[e] = scan_lds(d[e], L.grid, (int)pa.m, jj);
        j[e] = jj;
    }
    if (OVP && need1) {
        const bool me = fabsf(q[0]) > 32.0f;
        const bool mo = fabsf(q[1]) > 32.0f;
        if (has_odd) {
            const bool ve = mo && !me;
            q[0] = q[0] * (ve ? 0.0f : 1.0f);
            q[1] = q[1] * (me ? 0.0f : 1.0f);
            if (ve) j[0] = ANTQ_IDX_VICTIM;
            if (me) j[1] = ANTQ_IDX_VICTIM;
        } else {
            // odd numel: the last (even-indexed) element is zeroed iff element 0 is an outlier
            q[0] = q[0] * (mo ? 0.0f : 1.0f);
            if (mo) j[0] = ANTQ_IDX_VICTIM;
        }
    }
    {
        float t = (q[0] - d[0]) + d[0];
        IO<T>::store1(out, i0, t * s[0]);
        if (IDX) idx[i0] = (int16_t)j[0];
    }
    if (has_odd) {
        float t = (q[1] - d[1]) + d[1];
        IO<T>::store1(out, i1, t * s[1]);
        if (IDX) idx[i1] = (int16_t)j[1];
    }
}

// ------------------------------------------------------------------------------------
// quant_cuda.quant replacement: literal scan, grid arrives as a device array of unknown
// content (no host plan possible without a sync).  Grid -> LDS once per workgroup, four
// elements per thread, every LDS read is a wave-wide broadcast.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_nearest(const T *__restrict__ x, T *__restrict__ z, int16_t *__restrict__ idx, size_t n,
          const T *__restrict__ grid, int m)
{
    __shared__ float y[ANTQ_MAX_GRID];
    for (int i = threadIdx.x; i < m; i += 256) y[i] = (float)grid[i];  // quant_kernel.cu:23 narrows to float
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 1024u + threadIdx.x;
    float xv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const size_t i = base + 256u * u;
        xv[u] = (i < n) ? (float)x[i] : 0.0f;  // :28 narrows x to float
    }
    float sub_min[4], z_min[4];
    int jm[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { sub_min[u] = 102400.0f; z_min[u] = 0.0f; jm[u] = ANTQ_IDX_NONE; }
    for (int i = 0; i < m; i++) {
        const float g = y[i];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float sub_v = fabsf(xv[u] - g);
            if (sub_v <= sub_min[u]) { sub_min[u] = sub_v; z_min[u] = g; jm[u] = i; }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const size_t i = base + 256u * u;
        if (i < n) {
            z[i] = (T)z_min[u];
            if (idx) idx[i] = (int16_t)jm[u];
        }
    }
}

// ------------------------------------------------------------------------------------
// quant_cuda.quant, fast variant.  The grid is only known on the device, so every workgroup
// analyses it itself (M <= 256 threads, a few hundred instructions, amortised over 1024
// elements): rank-sorts it (any order for M <= 64, e.g. OliVe's cat(normal, outliers);
// larger grids must already be non-decreasing), records for every distinct value the LAST
// scan index holding it, and derives the magnitude `fastlim` below which the scan's result
// is decided by the two neighbouring values alone (no rounding plateau: all non-zero gaps
// within 2^19 of each other, edge gaps > ulp of any distance below fastlim).  Elements then
// binary-search their neighbours and apply the scan's own comparison to the two candidates
// (ties -> later scan index); everything else falls back to the literal scan.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_nearest_fast(const T *__restrict__ x, T *__restrict__ z, int16_t *__restrict__ idx, size_t n,
               const T *__restrict__ grid, int m)
{
    __shared__ float y[256];      // scan order
    __shared__ float sv[256];     // sorted values
    __shared__ int16_t win[256];  // sorted position -> last scan index with that value
    __shared__ int s_bad;
    __shared__ float s_mingap, s_maxgap, s_fastlim;
    const int t = threadIdx.x;
    if (t == 0) { s_bad = 0; s_mingap = 3.0e38f; s_maxgap = 0.0f; }
    sv[t] = __builtin_inff();                          // padding for the fixed-step search
    if (t < m) y[t] = (float)grid[t];
    __syncthreads();
    if (t < m) {
        const float v = y[t];
        if (!(fabsf(v) <= 65536.0f)) s_bad = 1;          // NaN / Inf / huge entries: literal scan
        int rank = t;
        if (m <= 64) {
            rank = 0;
            for (int j = 0; j < m; j++) rank += (y[j] < v || (y[j] == v && j < t)) ? 1 : 0;
        } else if (t + 1 < m && !(v <= y[t + 1])) {
            s_bad = 1;                                   // big grids must arrive sorted
        }
        sv[rank] = v;
        win[rank] = (int16_t)t;
    }
    __syncthreads();
    // last scan index among equal values (equal values are adjacent and in scan order)
    int w = 0;
    float g = 0.0f;
    if (t < m) {
        w = win[t];
        for (int j = t + 1; j < m && sv[j] == sv[t]; j++) w = max(w, (int)win[j]);
        for (int j = t - 1; j >= 0 && sv[j] == sv[t]; j--) w = max(w, (int)win[j]);
        g = (t + 1 < m) ? sv[t + 1] - sv[t] : 0.0f;
    }
    __syncthreads();
    if (t < m) {
        win[t] = (int16_t)w;
        if (g > 0.0f) {
            atomicMin(reinterpret_cast<unsigned int *>(&s_mingap), f2u(g));   // positive floats order like uints
            atomicMax(reinterpret_cast<unsigned int *>(&s_maxgap), f2u(g));
        }
    }
    __syncthreads();
    if (t == 0) {
        float lim = 0.0f;
        if (!s_bad && s_maxgap > 0.0f && s_maxgap <= s_mingap * 524288.0f) {
            // first / last non-zero gap
            float g0 = 0.0f, g1 = 0.0f;
            for (int j = 0; j + 1 < m && g0 == 0.0f; j++) g0 = sv[j + 1] - sv[j];
            for (int j = m - 1; j > 0 && g1 == 0.0f; j--) g1 = sv[j] - sv[j - 1];
            const float vabs = fmaxf(fabsf(sv[0]), fabsf(sv[m - 1]));
            lim = fminf(fminf(g0, g1) * 4194304.0f - vabs, 65536.0f);        // gap * 2^22 (one bit of margin)
            lim = fminf(lim, 102399.0f - vabs);                              // every |x| < lim has an entry within 102400
            if (!(lim > 2.0f * vabs)) lim = 0.0f;
        }
        s_fastlim = lim;
    }
    __syncthreads();
    const float fastlim = s_fastlim;
    // branch-free upper bound with a fixed number of steps (sv[] is padded with +inf beyond m), four
    // elements per thread in flight so the dependent LDS reads of one element overlap the others'
    int top = 1;
    while (top * 2 <= m) top *= 2;
    constexpr int E = 4;
    const size_t base = (size_t)blockIdx.x * (256u * E * 2) + threadIdx.x;
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        float xv[E];
        int p[E];
        bool ok[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const size_t i = base + 256u * (half * E + e);
            xv[e] = (i < n) ? (float)x[i] : 0.0f;
            p[e] = 0;
            ok[e] = fabsf(xv[e]) < fastlim;
        }
        for (int st = top; st >= 1; st >>= 1) {
#pragma unroll
            for (int e = 0; e < E; e++)
                if (sv[p[e] + st - 1] <= xv[e]) p[e] += st;       // p = number of sorted entries <= x
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            const size_t i = base + 256u * (half * E + e);
            if (i >= n) continue;
            int j;
            float zq;
            if (ok[e]) {
                const int pl = max(p[e] - 1, 0), ph = min(p[e], m - 1);
                const float r_lo = fabsf(xv[e] - sv[pl]);
                const float r_hi = fabsf(xv[e] - sv[ph]);
                const int w_lo = win[pl], w_hi = win[ph];
                // p == 0 / p == m: pl == ph, both candidates are the same entry
                j = (r_hi < r_lo || (r_hi == r_lo && w_hi > w_lo)) ? w_hi : w_lo;
                zq = y[j];
            } else {
                zq = scan_lds(xv[e], y, m, j);
            }
            z[i] = (T)zq;
            if (idx) idx[i] = (int16_t)j;
        }
    }
}

// bf16 / f16 storage variant of k_nearest (grid is float)
template <typename T>
__global__ void __launch_bounds__(256)
k_nearest16(const void *__restrict__ x, void *__restrict__ z, int16_t *__restrict__ idx, size_t n,
            const float *__restrict__ grid, int m)
{
    __shared__ float y[ANTQ_MAX_GRID];
    for (int i = threadIdx.x; i < m; i += 256) y[i] = grid[i];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float xv = IO<T>::load1(x, i);
    int j;
    const float q = scan_lds(xv, y, m, j);
    IO<T>::store1(z, i, q);
    if (idx) idx[i] = (int16_t)j;
}

// ------------------------------------------------------------------------------------
// AsymmetricQuantFunction.forward, quant_affine.py:95-115.  fp32, element-wise; rintf is
// round-half-to-even like torch.round.  Expression order follows the reference exactly:
//   scale*x - zp  (linear_quantize :39), (q + zp) / scale  (linear_dequantize :62).
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_affine(const float *__restrict__ x, float *__restrict__ out, int32_t *__restrict__ qout,
         size_t n, size_t row_len, int k,
         const float *__restrict__ xmin, const float *__restrict__ xmax, int per_row)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const size_t r = per_row ? i / row_len : 0;
    const float nlev = (float)((1 << k) - 1);
    const float half = (float)(1 << (k - 1));
    float range = xmax[r] - xmin[r];
    if (range < 1e-8f) range = 1e-8f;   // torch.clamp(min=1e-8): NaN stays NaN
    const float scale = (1.0f / range) * nlev;  // `n / tensor` is reciprocal(tensor) * n in torch (__rtruediv__)
    float zp = rintf(scale * xmin[r]);
    zp = zp + half;
    float q = rintf(scale * x[i] - zp);
    if (q < -half) q = -half;           // torch.clamp(q, -n, n-1): NaN stays NaN
    if (q > half - 1.0f) q = half - 1.0f;
    if (qout) qout[i] = (int32_t)q;
    out[i] = (q + zp) / scale;
}

// 16 B per lane streaming copy: the empirical HBM ceiling for this access pattern.
__global__ void __launch_bounds__(256)
k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_vec)
{
    const size_t first = ((size_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 256u + (threadIdx.x & 63u);
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const size_t i = first + 64u * u;
        if (i < n_vec) v[u] = ld_stream(src + i);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const size_t i = first + 64u * u;
        if (i < n_vec) st_stream(dst + i, v[u]);
    }
}

__global__ void __launch_bounds__(256) k_scale_inplace(float *__restrict__ a, size_t n, float ratio)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) a[i] = a[i] * ratio;
}

// ------------------------------------------------------------------------------------
// Row abs-max (the x_max of search_mse, AQ:289 / AQ:308).  One wavefront per row; rows
// with row_len % EPL == 0 and 16-byte alignment use vector loads, anything else element
// loads.  per_row == 0: every wavefront folds its strip into amax[0] with atomicMax on the
// float's bit pattern (non-negative floats order like unsigned ints; NaN sorts above Inf,
// so a NaN anywhere yields NaN like torch.max).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_absmax(const void *__restrict__ x, float *__restrict__ amax, size_t rows, size_t row_len, int per_row, int vec_ok)
{
    constexpr int EPL = IO<T>::EPL;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4u;
    if (per_row) {
        for (size_t r = wave; r < rows; r += nwaves) {
            uint32_t m = 0;
            if (vec_ok) {
                const uint4 *p = static_cast<const uint4 *>(x) + r * (row_len / EPL);
                uint32_t mp = 0;
                for (size_t i = lane; i < row_len / EPL; i += 64) mp = IO<T>::amax_acc(mp, p[i]);
                m = IO<T>::amax_bits(mp);
            } else {
                for (size_t i = lane; i < row_len; i += 64) m = max(m, f2u(IO<T>::load1(x, r * row_len + i)) & 0x7fffffffu);
            }
            m = wave_max_u32(m);
            if (lane == 0) amax[r] = u2f(m);
        }
    } else {
        // one scale for the whole tensor: block-strided, four independent 16-byte loads in flight per lane, one
        // atomicMax per workgroup (plain loads: the clip search reads the same bytes next, out of the Infinity Cache)
        const size_t n = rows * row_len;
        const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, stride = (size_t)gridDim.x * 256u;
        uint32_t m = 0;
        if (vec_ok) {
            const uint4 *p = static_cast<const uint4 *>(x);
            const size_t nv = n / EPL;
            uint32_t mp = 0;
            size_t i = tid;
            for (; i + 3 * stride < nv; i += 4 * stride) {
                const uint4 a0 = p[i], a1 = p[i + stride], a2 = p[i + 2 * stride], a3 = p[i + 3 * stride];
                mp = IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(mp, a0), a1), a2), a3);
            }
            for (; i < nv; i += stride) mp = IO<T>::amax_acc(mp, p[i]);
            m = IO<T>::amax_bits(mp);
            for (size_t k = nv * EPL + tid; k < n; k += stride) m = max(m, f2u(IO<T>::load1(x, k)) & 0x7fffffffu);
        } else {
            for (size_t k = tid; k < n; k += stride) m = max(m, f2u(IO<T>::load1(x, k)) & 0x7fffffffu);
        }
        m = wave_max_u32(m);
        __shared__ uint32_t wm[4];
        if (lane == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
            if (m) atomicMax(reinterpret_cast<unsigned int *>(amax), m);
        }
    }
}

// ------------------------------------------------------------------------------------
// Backward of the fused fake-quant w.r.t. alpha (QAT, AQ:39 alpha is a Parameter; AQ:544-549 straight-through
// graph): d out / d alpha = (q - d) / gmax = (out - x) / alpha, so
//     gsum[r] = sum_c fl32( gout[r,c] * fl32(out[r,c] - x[r,c]) )          (the caller divides by alpha[r])
// fp32 terms, fp64 accumulation.  One wavefront per row; one scale per tensor: block-strided with one atomic per
// workgroup.  d out / d x is the identity (no clip mask in the reference), so there is no kernel for it.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_alpha_grad(const void *__restrict__ x, const void *__restrict__ out, const void *__restrict__ gout,
             double *__restrict__ gsum, size_t rows, size_t row_len, int per_row, int vec_ok)
{
    constexpr int EPL = IO<T>::EPL;
    const uint32_t lane = threadIdx.x & 63u;
    auto vec_term = [](const uint4 &xv, const uint4 &ov, const uint4 &gv) -> float {
        float xf[EPL], of[EPL], gf[EPL];
        IO<T>::unpack(xv, xf);
        IO<T>::unpack(ov, of);
        IO<T>::unpack(gv, gf);
        float part = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; e++) part += gf[e] * (of[e] - xf[e]);
        return part;
    };
    auto one_term = [&](size_t i) -> float {
        return IO<T>::load1(gout, i) * (IO<T>::load1(out, i) - IO<T>::load1(x, i));
    };
    if (per_row) {
        const size_t wave = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
        const size_t nwaves = (size_t)gridDim.x * 4u;
        for (size_t r = wave; r < rows; r += nwaves) {
            double acc = 0.0;
            if (vec_ok) {
                const size_t vpr = row_len / EPL;
                const uint4 *px = static_cast<const uint4 *>(x) + r * vpr;
                const uint4 *po = static_cast<const uint4 *>(out) + r * vpr;
                const uint4 *pg = static_cast<const uint4 *>(gout) + r * vpr;
                for (size_t i = lane; i < vpr; i += 64) acc += (double)vec_term(px[i], po[i], pg[i]);
            } else {
                for (size_t i = lane; i < row_len; i += 64) acc += (double)one_term(r * row_len + i);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) gsum[r] = acc;
        }
    } else {
        const size_t n = rows * row_len;
        const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, stride = (size_t)gridDim.x * 256u;
        double acc = 0.0;
        if (vec_ok) {
            const size_t nv = n / EPL;
            const uint4 *px = static_cast<const uint4 *>(x), *po = static_cast<const uint4 *>(out);
            const uint4 *pg = static_cast<const uint4 *>(gout);
            size_t i = tid;
            for (; i + stride < nv; i += 2 * stride) {
                const uint4 x0 = px[i], o0 = po[i], g0 = pg[i];
                const uint4 x1 = px[i + stride], o1 = po[i + stride], g1 = pg[i + stride];
                acc += (double)vec_term(x0, o0, g0);
                acc += (double)vec_term(x1, o1, g1);
            }
            for (; i < nv; i += stride) acc += (double)vec_term(px[i], po[i], pg[i]);
            for (size_t k = nv * EPL + tid; k < n; k += stride) acc += (double)one_term(k);
        } else {
            for (size_t k = tid; k < n; k += stride) acc += (double)one_term(k);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        __shared__ double wsum[4];
        if (lane == 0) wsum[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(gsum, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    }
}

// ------------------------------------------------------------------------------------
// Clip search (search_mse, AQ:287-326): for every candidate ratio the squared error of the
// fake-quantised row against the row itself, WITHOUT writing the quantised tensor: x is
// read once into registers and all `ncand` candidates are evaluated on it.
//   sse[c, r] += sum_col fl32( fl32|out - x| ^ 2 )      (fp32 terms, fp64 accumulation)
// Same task decomposition as K1a (U*64 vectors of one row per task); tasks of one row add
// their partial sums with a double atomicAdd.
// ------------------------------------------------------------------------------------
constexpr int kPtCand = 128;   // candidates per workgroup in the one-scale-per-tensor mode (LDS accumulators)

template <typename T, bool OVP, int U, bool PT>
__global__ void __launch_bounds__(256)
k_search_sse(const uint4 *__restrict__ x, uint32_t total_tasks, uint32_t vpr, uint32_t tpr, size_t rows,
             const float *__restrict__ xmax, int per_row, const float *__restrict__ ratios, int ncand, float gmax,
             double *__restrict__ sse, PlanArgs pa, const uint4 *__restrict__ plan_tab, int cand_chunk)
{
    constexpr int EPL = IO<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t lane = threadIdx.x & 63u;
    // small tensors do not have enough rows to fill the chip: blockIdx.y splits the candidate list
    const int c_begin = (int)blockIdx.y * cand_chunk;
    const int c_end = min(ncand, c_begin + cand_chunk);
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    // PT (one scale for the whole tensor): every task adds to the same ncand sums.  Global atomics on 75 addresses
    // from every task serialise in L2 (measured: 4x the arithmetic), so each wavefront keeps its sums in LDS and the
    // workgroup issues one atomic per candidate at the end.
    __shared__ double wacc[PT ? 4 : 1][PT ? kPtCand : 1];
    if (PT)
        for (int c = (int)lane; c < kPtCand; c += 64) wacc[threadIdx.x >> 6][c] = 0.0;
    __syncthreads();
    const size_t na = per_row ? rows : 1;
    for (uint32_t task = blockIdx.x * 4u + (threadIdx.x >> 6); task < total_tasks; task += gridDim.x * 4u) {
        uint4 v[U];
        float xm;
        task_load<T, U>(x, xmax, per_row, task, vpr, tpr, lane, false, v, xm);
        uint32_t row = task, g = 0;
        if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
        const uint32_t v0 = g * (64u * U) + lane;
        for (int c = c_begin; c < c_end; c++) {
            const float a = xm * ratios[c];  // AQ:300  new_alpha = base_alpha * fl32(i*0.01)
            const Scale sc = make_scale(a, gmax);
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (v0 + 64u * u < vpr) {
                    float xf[EPL], of[EPL];
                    int j[EPL];
                    IO<T>::unpack(v[u], xf);
                    quant_vec<EPL, OVP, false>(pa, L, sc, xf, of, j);
                    float part = 0.0f;
#pragma unroll
                    for (int e = 0; e < EPL; e++) {
                        const float df = fabsf(of[e] - xf[e]);  // AQ:282 (q - x).abs().pow(2)
                        part += df * df;
                    }
                    acc += (double)part;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (PT) {
                if (lane == 0) wacc[threadIdx.x >> 6][c - c_begin] += acc;
            } else if (lane == 0) {
                double *dst = sse + (size_t)c * na + (per_row ? row : 0);
                if (per_row && tpr == 1) *dst = acc; else atomicAdd(dst, acc);
            }
        }
    }
    if (PT) {
        __syncthreads();
        for (int c = (int)threadIdx.x; c < c_end - c_begin; c += 256)
            atomicAdd(sse + (size_t)(c_begin + c), (wacc[0][c] + wacc[1][c]) + (wacc[2][c] + wacc[3][c]));
    }
}

// Element-granular clip search for ragged rows (row_len % EPL != 0, e.g. 3x3x3 conv rows) or
// unaligned buffers: one wavefront per row (per strip of 16 Ki elements for a per-tensor
// scale), exact slow-path arithmetic (true division + literal scan).  With OVP the partner
// element (i ^ 1, or element 0 for the last element of an odd-sized tensor) is quantised
// with ITS row's candidate alpha, as the reference does when it quantises the whole tensor.
template <typename T, bool OVP>
__global__ void __launch_bounds__(256)
k_search_sse_scalar(const void *__restrict__ x, size_t rows, size_t row_len, const float *__restrict__ xmax,
                    int per_row, const float *__restrict__ ratios, int ncand, float gmax, double *__restrict__ sse,
                    PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const size_t n = rows * row_len;
    const size_t strip = per_row ? row_len : (size_t)16384;
    const size_t nstrips = per_row ? rows : (n + strip - 1) / strip;
    const size_t na = per_row ? rows : 1;
    for (size_t st = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6); st < nstrips; st += (size_t)gridDim.x * 4u) {
        const size_t b = st * strip;
        const size_t e = per_row ? b + row_len : (b + strip < n ? b + strip : n);
        for (int c = 0; c < ncand; c++) {
            const float r = ratios[c];
            double acc = 0.0;
            for (size_t i = b + lane; i < e; i += 64) {
                const float xv = IO<T>::load1(x, i);
                const float s0 = (xmax[per_row ? i / row_len : 0] * r) / gmax;
                const float d = xv / s0;
                int j;
                float q = scan_lds(d, L.grid, (int)pa.m, j);
                if (OVP) {
                    size_t ip = i ^ (size_t)1;
                    if (ip >= n) ip = 0;                       // odd numel: torch.roll wrap-around
                    const bool has_partner = (ip != i);
                    if (has_partner) {
                        const float s1 = (xmax[per_row ? ip / row_len : 0] * r) / gmax;
                        int jp;
                        const float qp = scan_lds(IO<T>::load1(x, ip) / s1, L.grid, (int)pa.m, jp);
                        const bool me = fabsf(q) > 32.0f, mp = fabsf(qp) > 32.0f;
                        bool victim;
                        if (i & 1) victim = mp;                 // odd element: victim iff its even partner is an outlier
                        else if ((i ^ 1) < n) victim = mp && !me;  // even element with a real odd partner
                        else victim = mp;                       // last element of an odd-sized tensor
                        q = q * (victim ? 0.0f : 1.0f);
                    }
                }
                const float t = (q - d) + d;
                const float df = fabsf(t * s0 - xv);
                acc += (double)(df * df);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) {
                double *dst = sse + (size_t)c * na + (per_row ? st : 0);
                if (per_row) *dst = acc; else atomicAdd(dst, acc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------
static bool plan_args_from_host(const void *plan_host, PlanArgs &pa)
{
    const PlanHeader *h = static_cast<const PlanHeader *>(plan_host);
    if (h->magic != kPlanMagic || h->version != kPlanVersion) return false;
    if (h->m < 1 || h->m > ANTQ_MAX_GRID || h->m_pad != ((h->m + 3) & ~3u)) return false;
    pa.kind = h->kind;
    pa.m = h->m;
    pa.m_pad = h->m_pad;
    pa.shift = h->shift;
    pa.kmin = h->kmin;
    pa.kmax = h->kmax;
    pa.keymask = h->keymask;
    pa.nbneg = h->nbneg;
    pa.fastlim = h->fastlim;
    pa.n_entries = (h->kind == kPlanLut) ? h->n_entries : 0;
    pa.tab_units = pa.n_entries + (pa.m_pad >> 2);
    pa.linear = h->linear;
    pa.lin_scale = h->lin_scale;
    pa.lin_bias = h->lin_bias;
    return true;
}

static inline const uint4 *plan_tab_ptr(const void *plan_dev)
{
    return reinterpret_cast<const uint4 *>(static_cast<const char *>(plan_dev) + sizeof(PlanHeader));
}

// tuning knobs (dev / bench only; see antq_debug_set)
static int g_knob_u = 0;        // force U of the uniform kernel (0 = heuristic)
static int g_knob_blocks = 0;   // (unused since the kernels are one-shot)
static int g_knob_x = 1;        // 0 disables the x-domain row kernel (A/B measurements)
static int g_knob_nearest_fast = 1;   // 0: antq_nearest always runs the literal scan

template <typename T, bool OVP, bool IDX, bool DYN>
static int launch_uniform(const void *x, void *out, int16_t *idx, size_t rows, size_t vpr, const float *alpha,
                          int per_row, float gmax, float ratio, float *alpha_out, const PlanArgs &pa,
                          const void *plan_host, const void *plan_dev, size_t lds, hipStream_t st)
{
    const PlanHeader *ph = static_cast<const PlanHeader *>(plan_host);
    const uint4 *tab = plan_tab_ptr(plan_dev);
    const uint4 *xv = static_cast<const uint4 *>(x);
    uint4 *ov = static_cast<uint4 *>(out);
    const bool use_x = (g_knob_x != 0) && pa.kind == kPlanLut && ph->xdom && vpr >= 256 && (!DYN || vpr <= 2048);
    if (use_x) {
        // x-domain row kernel: 4 or 8 KiB of one row per wavefront (the per-row table is rebuilt per task)
        int U = 4;   // 4 KiB of the row per wavefront measured best at steady clocks (79 % of 8 TB/s on 1 GiB)
        if (DYN) U = (vpr <= 256 || (vpr > 512 && vpr <= 1024)) ? 4 : 8;
        if (g_knob_u) U = DYN ? U : g_knob_u;
        const bool wpr4 = DYN && vpr > 512;            // one row per workgroup: 4 wavefronts x U x 64 vectors
        const size_t tpr = wpr4 ? 4 : (vpr + (size_t)64 * U - 1) / ((size_t)64 * U);
        const size_t total = rows * tpr;
        if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
        XArgs xa;
        xa.m = pa.m; xa.shift = pa.shift; xa.kmin = pa.kmin; xa.kmax = pa.kmax; xa.keymask = pa.keymask;
        xa.nbneg = pa.nbneg; xa.n_entries = pa.n_entries; xa.xlim = ph->xlim; xa.vout = ph->vout;
        const uint4 *entries = tab + (pa.m_pad >> 2);
        const float *grid = reinterpret_cast<const float *>(tab);
        const dim3 grid_dim((unsigned)((total + 3) / 4)), block(256);
#define ANTQ_LAUNCH_X(UU)                                                                                           \
    hipLaunchKernelGGL((k_fq_xrow<T, OVP, IDX, UU, DYN>), grid_dim, block, 0, st, xv, ov, idx, (uint32_t)total,     \
                       (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out, xa, entries, grid)
        if (wpr4) {
            if (U == 8)
                hipLaunchKernelGGL((k_fq_xrow<T, OVP, IDX, 8, DYN, DYN ? 4 : 1>), grid_dim, block, 0, st, xv, ov, idx,
                                   (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out,
                                   xa, entries, grid);
            else
                hipLaunchKernelGGL((k_fq_xrow<T, OVP, IDX, 4, DYN, DYN ? 4 : 1>), grid_dim, block, 0, st, xv, ov, idx,
                                   (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out,
                                   xa, entries, grid);
            return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
        }
        switch (U) {
        case 8: ANTQ_LAUNCH_X(8); break;
        case 4: ANTQ_LAUNCH_X(4); break;
        case 2: ANTQ_LAUNCH_X(2); break;
        default: ANTQ_LAUNCH_X(1); break;
        }
#undef ANTQ_LAUNCH_X
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
    // U: 1 .. 4 KiB of one row per task, keeping lane utilisation high at the row tail
    int U = 4;
    if (DYN) {
        U = vpr <= 64 ? 1 : vpr <= 128 ? 2 : vpr <= 256 ? 4 : 8;
        if (vpr > 512) return ANTQ_ERR_UNSUPPORTED;  // caller falls back to absmax + static
    } else {
        double best = -1.0;
        for (int cand : {4, 2, 1}) {
            const size_t span = (size_t)64 * cand;
            const double util = (double)vpr / (double)(((vpr + span - 1) / span) * span);
            if (util > best + 0.05) { best = util; U = cand; }
        }
        if (g_knob_u) U = g_knob_u;
    }
    const size_t tpr = (vpr + (size_t)64 * U - 1) / ((size_t)64 * U);
    const size_t total = rows * tpr;
    if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
    size_t blocks = (total + 3) / 4;
    const bool loop = !DYN && lds > 16384;
    if (loop) {
        // staging a big table per 16 KiB of data would dominate: persistent workgroups instead
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>(8, (size_t)(144 * 1024) / lds));
        blocks = std::min(blocks, (size_t)256 * per_cu);
        const dim3 grid_l((unsigned)blocks), block_l(256);
        hipLaunchKernelGGL((k_fq_uniform<T, OVP, IDX, 4, false, true>), grid_l, block_l, lds, st, xv, ov, idx,
                           (uint32_t)((rows * ((vpr + 255) / 256))), (uint32_t)vpr, (uint32_t)((vpr + 255) / 256), alpha,
                           per_row, gmax, ratio, alpha_out, pa, tab);
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)blocks), block(256);
#define ANTQ_LAUNCH_U(UU)                                                                                          \
    hipLaunchKernelGGL((k_fq_uniform<T, OVP, IDX, UU, DYN>), grid, block, lds, st, xv, ov, idx, (uint32_t)total,  \
                       (uint32_t)vpr, (uint32_t)tpr, alpha, per_row, gmax, ratio, alpha_out, pa, tab)
    switch (U) {
    case 8: ANTQ_LAUNCH_U(8); break;
    case 4: ANTQ_LAUNCH_U(4); break;
    case 2: ANTQ_LAUNCH_U(2); break;
    default: ANTQ_LAUNCH_U(1); break;
    }
#undef ANTQ_LAUNCH_U
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T, bool OVP, bool IDX>
static int launch_fq(const void *x, void *out, int16_t *idx, size_t rows, size_t row_len,
                     const float *alpha, int per_row, float gmax, const PlanArgs &pa,
                     const void *plan_host, const void *plan_dev, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const size_t n = rows * row_len;
    const size_t lds = (size_t)pa.tab_units * 16;
    const uint4 *tab = plan_tab_ptr(plan_dev);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                         (!idx || reinterpret_cast<uintptr_t>(idx) % 16 == 0);
    if (!per_row) { rows = 1; row_len = n; }

    if (aligned && row_len % EPL == 0) {
        const size_t vpr = row_len / EPL;
        if (vpr >= 64) {
            if (vpr > 0xffffffffull) return ANTQ_ERR_UNSUPPORTED;
            return launch_uniform<T, OVP, IDX, false>(x, out, idx, rows, vpr, alpha, per_row, gmax, 1.0f, nullptr, pa,
                                                      plan_host, plan_dev, lds, st);
        } else {
            const size_t n_vec = n / EPL;
            int vshift = -1;
            if ((vpr & (vpr - 1)) == 0) { vshift = 0; while (((size_t)1 << vshift) < vpr) vshift++; }
            constexpr int U = 2;
            const size_t blocks = (n_vec + 256 * U - 1) / (256 * U);
            if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
            hipLaunchKernelGGL((k_fq_lane<T, OVP, IDX, U, false>), dim3((unsigned)blocks), dim3(256), lds, st,
                               static_cast<const uint4 *>(x), static_cast<uint4 *>(out), idx, n_vec, (uint32_t)vpr,
                               vshift, alpha, per_row, gmax, 1.0f, (float *)nullptr, pa, tab);
        }
    } else if (aligned && !per_row && n >= (size_t)64 * EPL) {
        // per-tensor scale with a ragged tail: vector body + element tail
        const size_t n_body = (n / EPL) * EPL;
        int rc = launch_fq<T, OVP, IDX>(x, out, idx, 1, n_body, alpha, 0, gmax, pa, plan_host, plan_dev, st);
        if (rc != ANTQ_OK) return rc;
        const size_t n_tail = n - n_body;
        const size_t pairs = (n_tail + 1) / 2;
        hipLaunchKernelGGL((k_fq_scalar<T, OVP, IDX>), dim3((unsigned)((pairs + 255) / 256)), dim3(256), lds, st, x, out,
                           idx, n_body, n_tail, n, n, alpha, 0, gmax, pa, tab);
    } else {
        const size_t pairs = (n + 1) / 2;
        const size_t blocks = (pairs + 255) / 256;
        if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((k_fq_scalar<T, OVP, IDX>), dim3((unsigned)blocks), dim3(256), lds, st, x, out, idx,
                           (size_t)0, n, n, row_len, alpha, per_row, gmax, pa, tab);
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T>
static int launch_fq_flags(const void *x, void *out, int16_t *idx, size_t rows, size_t row_len,
                           const float *alpha, int per_row, float gmax, const PlanArgs &pa,
                           const void *plan_host, const void *plan_dev, unsigned flags, hipStream_t st)
{
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    if (ovp) {
        if (idx) return launch_fq<T, true, true>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, st);
        return launch_fq<T, true, false>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, st);
    }
    if (idx) return launch_fq<T, false, true>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, st);
    return launch_fq<T, false, false>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, st);
}

}  // namespace antq

// ======================================================================================
// C ABI
// ======================================================================================
using namespace antq;

extern "C" int antq_abi_version(void) { return ANTQ_ABI_VERSION; }

extern "C" const char *antq_strerror(int code)
{
    switch (code) {
    case ANTQ_OK: return "ok";
    case ANTQ_ERR_ARG: return "invalid argument";
    case ANTQ_ERR_UNSUPPORTED: return "unsupported dtype / size for this entry point";
    case ANTQ_ERR_PLAN: return "malformed or undersized plan blob";
    case ANTQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case ANTQ_ERR_ALIGN: return "pointer not aligned to the element size";
    default: return "unknown antq error";
    }
}

extern "C" int antq_nearest(const void *x, void *z, int16_t *idx, size_t n, const void *grid, int m, int dtype,
                            void *stream)
{
    if (n == 0) return ANTQ_OK;
    if (!x || !z || !grid || m < 1) return ANTQ_ERR_ARG;
    if (m > ANTQ_MAX_GRID) return ANTQ_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ANTQ_F32 || dtype == ANTQ_F64) {
        const size_t blocks = (n + 1023) / 1024;
        if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        if (dtype == ANTQ_F32) {
            if (m <= 256 && g_knob_nearest_fast)
                hipLaunchKernelGGL((k_nearest_fast<float>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, st,
                                   static_cast<const float *>(x), static_cast<float *>(z), idx, n,
                                   static_cast<const float *>(grid), m);
            else
                hipLaunchKernelGGL((k_nearest<float>), dim3((unsigned)blocks), dim3(256), 0, st,
                                   static_cast<const float *>(x), static_cast<float *>(z), idx, n,
                                   static_cast<const float *>(grid), m);
        } else {
            if (m <= 256 && g_knob_nearest_fast)
                hipLaunchKernelGGL((k_nearest_fast<double>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, st,
                                   static_cast<const double *>(x), static_cast<double *>(z), idx, n,
                                   static_cast<const double *>(grid), m);
            else
                hipLaunchKernelGGL((k_nearest<double>), dim3((unsigned)blocks), dim3(256), 0, st,
                                   static_cast<const double *>(x), static_cast<double *>(z), idx, n,
                                   static_cast<const double *>(grid), m);
        }
    } else if (dtype == ANTQ_BF16 || dtype == ANTQ_F16) {
        const size_t blocks = (n + 255) / 256;
        if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        if (dtype == ANTQ_BF16)
            hipLaunchKernelGGL((k_nearest16<bf16_tag>), dim3((unsigned)blocks), dim3(256), 0, st, x, z, idx, n,
                               static_cast<const float *>(grid), m);
        else
            hipLaunchKernelGGL((k_nearest16<f16_tag>), dim3((unsigned)blocks), dim3(256), 0, st, x, z, idx, n,
                               static_cast<const float *>(grid), m);
    } else {
        return ANTQ_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

extern "C" int antq_fakequant(const void *x, void *out, int16_t *idx, size_t rows, size_t row_len,
                              const float *alpha, int alpha_per_row, float gmax, const void *plan_host,
                              const void *plan_dev, unsigned flags, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !alpha || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_row = alpha_per_row ? 1 : 0;
    switch (dtype) {
    case ANTQ_F32:
        if (reinterpret_cast<uintptr_t>(x) % 4 || reinterpret_cast<uintptr_t>(out) % 4) return ANTQ_ERR_ALIGN;
        return launch_fq_flags<float>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, flags, st);
    case ANTQ_BF16:
        if (reinterpret_cast<uintptr_t>(x) % 2 || reinterpret_cast<uintptr_t>(out) % 2) return ANTQ_ERR_ALIGN;
        return launch_fq_flags<bf16_tag>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, flags, st);
    case ANTQ_F16:
        if (reinterpret_cast<uintptr_t>(x) % 2 || reinterpret_cast<uintptr_t>(out) % 2) return ANTQ_ERR_ALIGN;
        return launch_fq_flags<f16_tag>(x, out, idx, rows, row_len, alpha, per_row, gmax, pa, plan_host, plan_dev, flags, st);
    default:
        return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_affine(const float *x, float *out, int32_t *q, size_t rows, size_t row_len, int k,
                           const float *xmin, const float *xmax, int per_row, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !xmin || !xmax || k < 1 || k > 24) return ANTQ_ERR_ARG;
    const size_t n = rows * row_len;
    const size_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_affine, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, q, n,
                       row_len, k, xmin, xmax, per_row ? 1 : 0);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

extern "C" int antq_copy(const void *src, void *dst, size_t bytes, void *stream)
{
    if (bytes == 0) return ANTQ_OK;
    if (!src || !dst || bytes % 16 || reinterpret_cast<uintptr_t>(src) % 16 || reinterpret_cast<uintptr_t>(dst) % 16)
        return ANTQ_ERR_ARG;
    const size_t n_vec = bytes / 16;
    const size_t blocks = (n_vec + 1023) / 1024;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_copy, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n_vec);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

extern "C" int antq_debug_set(int key, int value)
{
    if (key == 0) g_knob_u = value;
    else if (key == 1) g_knob_blocks = value;
    else if (key == 2) g_knob_x = value;
    else if (key == 3) g_knob_nearest_fast = value;
    else return ANTQ_ERR_ARG;
    return ANTQ_OK;
}

namespace antq {

template <typename T>
static int launch_absmax(const void *x, float *amax, size_t rows, size_t row_len, int per_row, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const bool al = reinterpret_cast<uintptr_t>(x) % 16 == 0;
    const int vec_ok = per_row ? (al && row_len % EPL == 0) : al;
    size_t waves = per_row ? rows : (rows * row_len + 64 * EPL * 4 - 1) / (64 * EPL * 4);
    size_t blocks = (waves + 3) / 4;
    if (blocks > (per_row ? 4096u : 1024u)) blocks = per_row ? 4096 : 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_absmax<T>), dim3((unsigned)blocks), dim3(256), 0, st, x, amax, rows, row_len, per_row, vec_ok);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

template <typename T, bool OVP, bool IDX>
static int launch_dynamic(const void *x, void *out, int16_t *idx, float *alpha_out, size_t rows, size_t row_len,
                          float ratio, float gmax, const PlanArgs &pa, const void *plan_host, const void *plan_dev, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const size_t lds = (size_t)pa.tab_units * 16;
    const uint4 *tab = plan_tab_ptr(plan_dev);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                         (!idx || reinterpret_cast<uintptr_t>(idx) % 16 == 0);
    if (aligned && row_len % EPL == 0) {
        const size_t vpr = row_len / EPL;
        const bool pow2 = (vpr & (vpr - 1)) == 0;
        if (vpr <= 32 && pow2) {
            // several groups per wavefront: butterfly max over vpr adjacent lanes
            int vshift = 0;
            while (((size_t)1 << vshift) < vpr) vshift++;
            const size_t n_vec = rows * vpr;
            constexpr int U = 2;
            const size_t blocks = (n_vec + 256 * U - 1) / (256 * U);
            if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
            hipLaunchKernelGGL((k_fq_lane<T, OVP, IDX, U, true>), dim3((unsigned)blocks), dim3(256), lds, st,
                               static_cast<const uint4 *>(x), static_cast<uint4 *>(out), idx, n_vec, (uint32_t)vpr,
                               vshift, (const float *)nullptr, 1, gmax, ratio, alpha_out, pa, tab);
            return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
        }
        if (vpr <= 2048) {
            // one quant group (row) per wavefront (<= 512 vectors) or per workgroup (<= 2048): the row
            // lives in registers, single HBM read.  Plans without the x-domain table only have the
            // wavefront variant; longer rows fall through to the two-pass scheme.
            int rc = launch_uniform<T, OVP, IDX, true>(x, out, idx, rows, vpr, nullptr, 1, gmax, ratio, alpha_out, pa,
                                                       plan_host, plan_dev, lds, st);
            if (rc != ANTQ_ERR_UNSUPPORTED) return rc;
        }
    }
    // long or ragged rows: abs-max pass (read) + static pass (read + write)
    if (!alpha_out) return ANTQ_ERR_ARG;
    int rc = launch_absmax<T>(x, alpha_out, rows, row_len, 1, st);
    if (rc != ANTQ_OK) return rc;
    hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, alpha_out, rows, ratio);
    return launch_fq<T, OVP, IDX>(x, out, idx, rows, row_len, alpha_out, 1, gmax, pa, plan_host, plan_dev, st);
}

template <typename T>
static int launch_dynamic_flags(const void *x, void *out, int16_t *idx, float *alpha_out, size_t rows, size_t row_len,
                                float ratio, float gmax, const PlanArgs &pa, const void *plan_host, const void *plan_dev, unsigned flags,
                                hipStream_t st)
{
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    if (ovp) {
        if (idx) return launch_dynamic<T, true, true>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, st);
        return launch_dynamic<T, true, false>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, st);
    }
    if (idx) return launch_dynamic<T, false, true>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, st);
    return launch_dynamic<T, false, false>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, st);
}

template <typename T, bool OVP>
static int launch_search(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                         const float *ratios, int ncand, float gmax, const PlanArgs &pa, const void *plan_dev,
                         double *sse, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const size_t lds = (size_t)pa.tab_units * 16;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0 || (per_row ? row_len : rows * row_len) % EPL != 0) {
        size_t strips = per_row ? rows : (rows * row_len + 16383) / 16384;
        size_t blocks = (strips + 3) / 4;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL((k_search_sse_scalar<T, OVP>), dim3((unsigned)blocks), dim3(256), lds, st, x, rows, row_len,
                           xmax, per_row, ratios, ncand, gmax, sse, pa, plan_tab_ptr(plan_dev));
        return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
    }
    if (!per_row) { row_len = rows * row_len; rows = 1; }
    const size_t vpr = row_len / EPL;
    if (vpr > 0xffffffffull) return ANTQ_ERR_UNSUPPORTED;
    constexpr int U = 4;
    const size_t tpr = (vpr + 64 * U - 1) / (64 * U);
    const size_t total = rows * tpr;
    if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
    const bool pt = rows == 1;
    size_t blocks = (total + 3) / 4;
    const size_t cap = pt ? 256 * 4 : 256 * 8;
    if (blocks > cap) blocks = cap;
    // enough wavefronts to fill 256 CUs x 8 waves/SIMD: split the candidates when there are few rows
    int chunks = (int)std::min<size_t>((size_t)ncand, std::max<size_t>(1, (size_t)2048 / blocks));
    if (pt) chunks = std::max(chunks, (ncand + kPtCand - 1) / kPtCand);
    const int cand_chunk = (ncand + chunks - 1) / chunks;
    chunks = (ncand + cand_chunk - 1) / cand_chunk;
    if (pt)
        hipLaunchKernelGGL((k_search_sse<T, OVP, U, true>), dim3((unsigned)blocks, (unsigned)chunks), dim3(256), lds, st,
                           static_cast<const uint4 *>(x), (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, rows, xmax,
                           per_row, ratios, ncand, gmax, sse, pa, plan_tab_ptr(plan_dev), cand_chunk);
    else
        hipLaunchKernelGGL((k_search_sse<T, OVP, U, false>), dim3((unsigned)blocks, (unsigned)chunks), dim3(256), lds, st,
                           static_cast<const uint4 *>(x), (uint32_t)total, (uint32_t)vpr, (uint32_t)tpr, rows, xmax,
                           per_row, ratios, ncand, gmax, sse, pa, plan_tab_ptr(plan_dev), cand_chunk);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

}  // namespace antq

extern "C" int antq_fakequant_dynamic(const void *x, void *out, int16_t *idx, float *alpha_out, size_t rows,
                                      size_t row_len, float ratio, float gmax, const void *plan_host,
                                      const void *plan_dev, unsigned flags, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_dynamic_flags<float>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, flags, st);
    case ANTQ_BF16: return launch_dynamic_flags<bf16_tag>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, flags, st);
    case ANTQ_F16: return launch_dynamic_flags<f16_tag>(x, out, idx, alpha_out, rows, row_len, ratio, gmax, pa, plan_host, plan_dev, flags, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

namespace antq {
template <typename T>
static int launch_alpha_grad(const void *x, const void *out, const void *gout, double *gsum, size_t rows, size_t row_len,
                             int per_row, hipStream_t st)
{
    constexpr int EPL = IO<T>::EPL;
    const bool al = (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gout)) % 16 == 0;
    const int vec_ok = per_row ? (al && row_len % EPL == 0) : al;
    size_t waves = per_row ? rows : (rows * row_len + 64 * EPL * 2 - 1) / (64 * EPL * 2);
    size_t blocks = (waves + 3) / 4;
    const size_t cap = per_row ? 4096 : 1024;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_alpha_grad<T>), dim3((unsigned)blocks), dim3(256), 0, st, x, out, gout, gsum, rows, row_len, per_row,
                       vec_ok);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq

extern "C" int antq_alpha_grad(const void *x, const void *out, const void *gout, size_t rows, size_t row_len, int per_row,
                               double *gsum, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !out || !gout || !gsum) return ANTQ_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_alpha_grad<float>(x, out, gout, gsum, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_BF16: return launch_alpha_grad<bf16_tag>(x, out, gout, gsum, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_F16: return launch_alpha_grad<f16_tag>(x, out, gout, gsum, rows, row_len, per_row ? 1 : 0, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_absmax(const void *x, float *amax, size_t rows, size_t row_len, int per_row, int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !amax) return ANTQ_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_absmax<float>(x, amax, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_BF16: return launch_absmax<bf16_tag>(x, amax, rows, row_len, per_row ? 1 : 0, st);
    case ANTQ_F16: return launch_absmax<f16_tag>(x, amax, rows, row_len, per_row ? 1 : 0, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_search_sse(const void *x, size_t rows, size_t row_len, const float *xmax, int per_row,
                               const float *ratios, int ncand, float gmax, const void *plan_host, const void *plan_dev,
                               unsigned flags, int dtype, double *sse, void *stream)
{
    if (rows == 0 || row_len == 0 || ncand == 0) return ANTQ_OK;
    if (!x || !xmax || !ratios || !plan_host || !plan_dev || !sse || ncand < 0) return ANTQ_ERR_ARG;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    const int pr = per_row ? 1 : 0;
    switch (dtype) {
    case ANTQ_F32:
        return ovp ? launch_search<float, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_dev, sse, st)
                   : launch_search<float, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_dev, sse, st);
    case ANTQ_BF16:
        return ovp ? launch_search<bf16_tag, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_dev, sse, st)
                   : launch_search<bf16_tag, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_dev, sse, st);
    case ANTQ_F16:
        return ovp ? launch_search<f16_tag, true>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_dev, sse, st)
                   : launch_search<f16_tag, false>(x, rows, row_len, xmax, pr, ratios, ncand, gmax, pa, plan_dev, sse, st);
    default:
        return ANTQ_ERR_UNSUPPORTED;
    }
}

// ======================================================================================
// Batched launch (antq_batch_build / antq_fakequant_batch)
// ======================================================================================
namespace antq {

constexpr uint32_t kBatchMagic = 0x42544E41u;  // "ANTB"
constexpr int kBatchU = 4;                      // vectors per lane per task (4 KiB per wavefront: best measured)

struct BatchDesc {   // 144 bytes, device-visible
    const uint4 *x;
    uint4 *out;
    const float *alpha;
    const uint4 *plan_tab;
    uint64_t n_vec;        // lane kind: number of 16-byte vectors
    uint32_t total_tasks;  // row kind: wavefront tasks
    uint32_t vpr;
    uint32_t tpr;
    int32_t vshift;
    uint32_t first_block;
    uint32_t kind;         // 0 = row-run per wavefront, d-domain table; 1 = per-lane scale (vpr < 64);
                           // 2 = row-run per wavefront, x-domain table (pad[] = xlim bits, grid offset)
    int32_t per_row;
    float gmax;
    PlanArgs pa;
    uint32_t pad[4];
};
static_assert(sizeof(BatchDesc) == 144, "BatchDesc must be 144 bytes");

struct BatchHeader {   // 32 bytes
    uint32_t magic, n, total_blocks, dtype, flags, lds_bytes, map_offset, bytes;
};

template <typename T, bool OVP>
__global__ void __launch_bounds__(256)
k_fq_batch(const BatchDesc *__restrict__ descs, const uint32_t *__restrict__ block_map)
{
    constexpr int EPL = IO<T>::EPL;
    constexpr int U = kBatchU;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t j = block_map[blockIdx.x];
    const BatchDesc &D = descs[j];
    const PlanArgs pa = D.pa;
    const uint32_t lb = blockIdx.x - D.first_block;
    const uint32_t lane = threadIdx.x & 63u;
    const uint4 *plan_tab = D.plan_tab;

    if (D.kind == 2) {
        // x-domain rows: wave-private table, no workgroup barrier
        __shared__ __attribute__((aligned(16))) uint4 wtab_all[4][256];
        const uint32_t wv = threadIdx.x >> 6;
        const uint32_t task = __builtin_amdgcn_readfirstlane(lb * 4u + wv);
        if (task >= D.total_tasks) return;
        XArgs xa;
        xa.m = pa.m; xa.shift = pa.shift; xa.kmin = pa.kmin; xa.kmax = pa.kmax; xa.keymask = pa.keymask;
        xa.nbneg = pa.nbneg; xa.n_entries = pa.n_entries; xa.xlim = u2f(D.pad[0]); xa.vout = u2f(D.pad[1]);
        xrow_task<T, OVP, false, U, false, 1>(D.x, D.out, nullptr, task, D.vpr, D.tpr, D.alpha, D.per_row, D.gmax, 1.0f,
                                              nullptr, xa, plan_tab + (pa.m_pad >> 2), reinterpret_cast<const float *>(plan_tab),
                                              wtab_all[wv], lane, wv);
        return;
    }
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    if (D.kind == 0) {
        const uint32_t total = D.total_tasks, vpr = D.vpr, tpr = D.tpr;
        const uint32_t task = __builtin_amdgcn_readfirstlane(lb * 4u + (threadIdx.x >> 6));
        const bool active = task < total;
        uint4 v[U];
        float a;
        task_load<T, U>(D.x, D.alpha, D.per_row, active ? task : total - 1u, vpr, tpr, lane, false, v, a);
        const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
        __syncthreads();
        if (active)
            task_run<T, OVP, false, U, false>(D.out, nullptr, nullptr, 1.0f, task, vpr, tpr, lane, D.gmax, pa, L, v, a);
    } else {
        const size_t n_vec = D.n_vec;
        const size_t first = ((size_t)lb * U) * 256u + threadIdx.x;
        uint4 v[U];
        float a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t vi = first + (size_t)u * 256u;
            v[u] = make_uint4(0, 0, 0, 0);
            a[u] = 1.0f;
            if (vi < n_vec) {
                v[u] = ld_stream(D.x + vi);
                size_t row = 0;
                if (D.per_row) row = (D.vshift >= 0) ? (vi >> D.vshift) : (vi / D.vpr);
                a[u] = D.alpha[row];
            }
        }
        const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t vi = first + (size_t)u * 256u;
            if (vi < n_vec) {
                const Scale sc = make_scale(a[u], D.gmax);
                float xf[EPL], of[EPL];
                int jj[EPL];
                IO<T>::unpack(v[u], xf);
                quant_vec<EPL, OVP, false>(pa, L, sc, xf, of, jj);
                st_stream(D.out + vi, IO<T>::pack(of));
            }
        }
    }
}

static int epl_of(int dtype) { return dtype == ANTQ_F32 ? 4 : (dtype == ANTQ_BF16 || dtype == ANTQ_F16) ? 8 : 0; }

// blocks a job needs, or 0 if it cannot be expressed (ragged / unaligned)
static size_t job_blocks(const antq_job &J, int epl, BatchDesc *d)
{
    size_t rows = J.rows, row_len = J.row_len;
    const size_t n = rows * row_len;
    if (!J.alpha_per_row) { rows = 1; row_len = n; }
    if (n == 0 || row_len % epl != 0) return 0;
    if (reinterpret_cast<uintptr_t>(J.x_dev) % 16 || reinterpret_cast<uintptr_t>(J.out_dev) % 16) return 0;
    const size_t vpr = row_len / epl;
    if (vpr > 0xffffffffull) return 0;
    size_t blocks;
    if (vpr >= 64) {
        const size_t tpr = (vpr + 64 * kBatchU - 1) / (64 * kBatchU);
        const size_t total = rows * tpr;
        if (total > 0xfffffff0ull) return 0;
        blocks = (total + 3) / 4;
        if (d) { d->kind = 0; d->total_tasks = (uint32_t)total; d->vpr = (uint32_t)vpr; d->tpr = (uint32_t)tpr; d->vshift = -1; d->n_vec = n / epl; }
    } else {
        const size_t n_vec = n / epl;
        blocks = (n_vec + 256 * kBatchU - 1) / (256 * kBatchU);
        int vshift = -1;
        if ((vpr & (vpr - 1)) == 0) { vshift = 0; while (((size_t)1 << vshift) < vpr) vshift++; }
        if (d) { d->kind = 1; d->total_tasks = 0; d->vpr = (uint32_t)vpr; d->tpr = 1; d->vshift = vshift; d->n_vec = n_vec; }
    }
    return blocks;
}

}  // namespace antq

extern "C" size_t antq_batch_capacity(const antq_job *jobs, int n, int dtype)
{
    const int epl = epl_of(dtype);
    if (!jobs || n < 1 || !epl) return 0;
    size_t blocks = 0;
    for (int i = 0; i < n; i++) blocks += job_blocks(jobs[i], epl, nullptr);
    return sizeof(BatchHeader) + sizeof(BatchDesc) * (size_t)n + 4 * blocks;
}

extern "C" int antq_batch_build(const antq_job *jobs, int n, int dtype, unsigned flags, void *blob, size_t cap)
{
    const int epl = epl_of(dtype);
    if (!jobs || !blob || n < 1 || n > 65535) return ANTQ_ERR_ARG;
    if (!epl) return ANTQ_ERR_UNSUPPORTED;
    char *p = static_cast<char *>(blob);
    BatchHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = kBatchMagic; h.n = (uint32_t)n; h.dtype = (uint32_t)dtype; h.flags = flags;
    h.map_offset = (uint32_t)(sizeof(BatchHeader) + sizeof(BatchDesc) * (size_t)n);
    if (cap < h.map_offset) return ANTQ_ERR_PLAN;
    BatchDesc *descs = reinterpret_cast<BatchDesc *>(p + sizeof(BatchHeader));
    uint32_t *map = reinterpret_cast<uint32_t *>(p + h.map_offset);
    size_t total_blocks = 0, lds = 0;
    for (int i = 0; i < n; i++) {
        const antq_job &J = jobs[i];
        if (!J.x_dev || !J.out_dev || !J.alpha_dev || !J.plan_host || !J.plan_dev) return ANTQ_ERR_ARG;
        BatchDesc d;
        memset(&d, 0, sizeof(d));
        const size_t blocks = job_blocks(J, epl, &d);
        if (blocks == 0) return ANTQ_ERR_UNSUPPORTED;
        if (!plan_args_from_host(J.plan_host, d.pa)) return ANTQ_ERR_PLAN;
        {
            const PlanHeader *ph = static_cast<const PlanHeader *>(J.plan_host);
            if (d.kind == 0 && g_knob_x && d.pa.kind == kPlanLut && ph->xdom && d.vpr >= 256) {
                d.kind = 2;
                memcpy(&d.pad[0], &ph->xlim, 4);
                memcpy(&d.pad[1], &ph->vout, 4);
            }
        }
        if (total_blocks + blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
        if (cap < h.map_offset + 4 * (total_blocks + blocks)) return ANTQ_ERR_PLAN;
        d.x = static_cast<const uint4 *>(J.x_dev);
        d.out = static_cast<uint4 *>(J.out_dev);
        d.alpha = J.alpha_dev;
        d.plan_tab = plan_tab_ptr(J.plan_dev);
        d.first_block = (uint32_t)total_blocks;
        d.per_row = J.alpha_per_row ? 1 : 0;
        d.gmax = J.gmax;
        descs[i] = d;
        for (size_t b = 0; b < blocks; b++) map[total_blocks + b] = (uint32_t)i;
        total_blocks += blocks;
        lds = std::max(lds, (size_t)d.pa.tab_units * 16);
    }
    h.total_blocks = (uint32_t)total_blocks;
    h.lds_bytes = (uint32_t)lds;
    h.bytes = (uint32_t)(h.map_offset + 4 * total_blocks);
    memcpy(p, &h, sizeof(h));
    return (int)h.bytes;
}

extern "C" int antq_fakequant_batch(const void *batch_host, const void *batch_dev, void *stream)
{
    if (!batch_host || !batch_dev) return ANTQ_ERR_ARG;
    const BatchHeader *h = static_cast<const BatchHeader *>(batch_host);
    if (h->magic != kBatchMagic) return ANTQ_ERR_PLAN;
    if (h->total_blocks == 0) return ANTQ_OK;
    const char *pd = static_cast<const char *>(batch_dev);
    const BatchDesc *descs = reinterpret_cast<const BatchDesc *>(pd + sizeof(BatchHeader));
    const uint32_t *map = reinterpret_cast<const uint32_t *>(pd + h->map_offset);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(h->total_blocks), block(256);
    const bool ovp = (h->flags & ANTQ_FLAG_OVP) != 0;
#define ANTQ_LAUNCH_B(TT)                                                                                         \
    do {                                                                                                          \
        if (ovp) hipLaunchKernelGGL((k_fq_batch<TT, true>), grid, block, h->lds_bytes, st, descs, map);           \
        else hipLaunchKernelGGL((k_fq_batch<TT, false>), grid, block, h->lds_bytes, st, descs, map);              \
    } while (0)
    switch (h->dtype) {
    case ANTQ_F32: ANTQ_LAUNCH_B(float); break;
    case ANTQ_BF16: ANTQ_LAUNCH_B(bf16_tag); break;
    case ANTQ_F16: ANTQ_LAUNCH_B(f16_tag); break;
    default: return ANTQ_ERR_UNSUPPORTED;
    }
#undef ANTQ_LAUNCH_B
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

// ======================================================================================
// Packed 4-bit codec (antq_encode4 / antq_decode4)
// ======================================================================================
namespace antq {

// One lane: 8 consecutive elements (4 pairs) of one row -> 4 bytes of codes: the fused
// quantiser (quant_vec with the index output), then every index is folded into a nibble.
template <typename T, bool OVP>
__global__ void __launch_bounds__(256)
k_encode4(const void *__restrict__ x, uint32_t *__restrict__ codes, size_t n_oct, size_t row_len,
          const float *__restrict__ alpha, int per_row, float gmax, int n_normal, int zero_code,
          PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    const size_t o = (size_t)blockIdx.x * 256u + threadIdx.x;   // octet index: elements [8o, 8o+8)
    if (o >= n_oct) return;
    const size_t e0 = o * 8;
    // row_len % 8 == 0: an octet (4 pairs) lies inside one row -> one scale
    const float a = alpha[per_row ? (e0 / row_len) : 0];
    const Scale sc = make_scale(a, gmax);
    float xf[8], of[8];
    int j[8];
#pragma unroll
    for (int e = 0; e < 8; e++) xf[e] = IO<T>::load1(x, e0 + e);
    quant_vec<8, OVP, true>(pa, L, sc, xf, of, j);
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

template <typename T, bool OVP>
__global__ void __launch_bounds__(256)
k_decode4(const uint32_t *__restrict__ codes, void *__restrict__ out, size_t n_oct, size_t row_len,
          const float *__restrict__ alpha, int per_row, float gmax, int n_normal,
          const float *__restrict__ grid, int m)
{
    __shared__ float g[32];
    if (threadIdx.x < 32) g[threadIdx.x] = ((int)threadIdx.x < m) ? grid[threadIdx.x] : 0.0f;
    __syncthreads();
    const size_t o = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (o >= n_oct) return;
    const uint32_t word = codes[o];
    const size_t e0 = o * 8;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const size_t i = e0 + 2 * p;
        const float s = alpha[per_row ? (i / row_len) : 0] / gmax;
        const uint32_t c0 = (word >> (8 * p)) & 15u, c1 = (word >> (8 * p + 4)) & 15u;
        float q0, q1;
        if (OVP) {
            // identifier 15 in one nibble: that element is the victim (0), its partner an outlier
            q0 = (c0 == 15u) ? 0.0f : ((c1 == 15u) ? g[n_normal + c0] : g[c0]);
            q1 = (c1 == 15u) ? 0.0f : ((c0 == 15u) ? g[n_normal + c1] : g[c1]);
        } else {
            q0 = g[c0];
            q1 = g[c1];
        }
        IO<T>::store1(out, i, q0 * s);
        IO<T>::store1(out, i + 1, q1 * s);
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
    const size_t blocks = (n_oct + 255) / 256;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    const float *grid_host = plan_grid(plan_host);
    int zero_code = 0;
    for (int i = 0; i < (ovp ? n_normal : m); i++) if (grid_host[i] == 0.0f) zero_code = i;
    if (enc) {
        const size_t lds = (size_t)pa.tab_units * 16;
        uint32_t *codes = static_cast<uint32_t *>(codes_or_out);
        if (ovp) hipLaunchKernelGGL((k_encode4<T, true>), dim3((unsigned)blocks), dim3(256), lds, st, x, codes, n_oct, row_len,
                                    alpha, per_row, gmax, n_normal, zero_code, pa, plan_tab_ptr(plan_dev));
        else hipLaunchKernelGGL((k_encode4<T, false>), dim3((unsigned)blocks), dim3(256), lds, st, x, codes, n_oct, row_len,
                                alpha, per_row, gmax, n_normal, zero_code, pa, plan_tab_ptr(plan_dev));
    } else {
        const float *grid_dev = reinterpret_cast<const float *>(plan_tab_ptr(plan_dev));
        const uint32_t *codes = reinterpret_cast<const uint32_t *>(codes_in);
        if (ovp) hipLaunchKernelGGL((k_decode4<T, true>), dim3((unsigned)blocks), dim3(256), 0, st, codes, codes_or_out, n_oct,
                                    row_len, alpha, per_row, gmax, n_normal, grid_dev, m);
        else hipLaunchKernelGGL((k_decode4<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, codes, codes_or_out, n_oct,
                                row_len, alpha, per_row, gmax, n_normal, grid_dev, m);
    }
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

}  // namespace antq

extern "C" int antq_encode4(const void *x, uint8_t *codes, size_t rows, size_t row_len, const float *alpha, int per_row,
                            float gmax, const void *plan_host, const void *plan_dev, int n_normal, unsigned flags,
                            int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!x || !codes || !alpha || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(codes) % 4) return ANTQ_ERR_ALIGN;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_codec<float>(true, x, codes, nullptr, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    case ANTQ_BF16: return launch_codec<bf16_tag>(true, x, codes, nullptr, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    case ANTQ_F16: return launch_codec<f16_tag>(true, x, codes, nullptr, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}

extern "C" int antq_decode4(const uint8_t *codes, void *out, size_t rows, size_t row_len, const float *alpha, int per_row,
                            float gmax, const void *plan_host, const void *plan_dev, int n_normal, unsigned flags,
                            int dtype, void *stream)
{
    if (rows == 0 || row_len == 0) return ANTQ_OK;
    if (!codes || !out || !alpha || !plan_host || !plan_dev) return ANTQ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(codes) % 4) return ANTQ_ERR_ALIGN;
    PlanArgs pa;
    if (!plan_args_from_host(plan_host, pa)) return ANTQ_ERR_PLAN;
    const bool ovp = (flags & ANTQ_FLAG_OVP) != 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case ANTQ_F32: return launch_codec<float>(false, nullptr, out, codes, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    case ANTQ_BF16: return launch_codec<bf16_tag>(false, nullptr, out, codes, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    case ANTQ_F16: return launch_codec<f16_tag>(false, nullptr, out, codes, rows, row_len, alpha, per_row ? 1 : 0, gmax, pa, plan_host, plan_dev, n_normal, ovp, st);
    default: return ANTQ_ERR_UNSUPPORTED;
    }
}


namespace antq {
// search_mse's selection loop, one thread per row (AQ:299-306): strict '<' keeps the earliest best.
__global__ void __launch_bounds__(256)
k_search_pick(const double *__restrict__ sse, const float *__restrict__ xmax, const float *__restrict__ ratios,
              int ncand, size_t na, double row_len, float *__restrict__ best_score, float *__restrict__ best_alpha)
{
    const size_t r = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= na) return;
    float best = 1e10f;
    const float xm = xmax[r];
    float alpha = xm;
    for (int c = 0; c < ncand; c++) {
        const float score = (float)(sse[(size_t)c * na + r] / row_len);
        if (score < best) { best = score; alpha = xm * ratios[c]; }
    }
    best_score[r] = best;
    best_alpha[r] = alpha;
}
}  // namespace antq

extern "C" int antq_search_pick(const double *sse, const float *xmax, const float *ratios, int ncand, size_t na,
                                size_t row_len, float *best_score, float *best_alpha, void *stream)
{
    if (na == 0) return ANTQ_OK;
    if (!sse || !xmax || !ratios || !best_score || !best_alpha || ncand < 0 || row_len == 0) return ANTQ_ERR_ARG;
    const size_t blocks = (na + 255) / 256;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(antq::k_search_pick, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), sse, xmax,
                       ratios, ncand, na, (double)row_len, best_score, best_alpha);
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
