// antq_device.h -- element I/O traits, the plan in LDS, exact division, the per-lane quantiser core (d-domain)
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_DEVICE_H
#define ANTQ_DEVICE_H

namespace antq {

// Rows of at least this many 16-byte vectors get a wavefront (or several) to themselves: wave-uniform scale, per-row
// x-domain table.  Shorter rows / groups share a wavefront with a per-lane scale (k_fq_lane).  Measured on batches of
// 16 x 4096^2: at 64 vectors per row the row kernels reach 47-55 % (bf16) / 56-70 % (fp32) because a wavefront then has
// one vector per lane in flight, the lane kernel 60 / 79 %; from 128 vectors up the row-table kernel wins (73-80 %).
constexpr unsigned kRowKernelMinVpr = 128;

// vectors per lane and task for rows of `vpr` vectors handled by whole wavefronts: the U in {4, 3, 2} with the best lane
// utilisation (ties: the larger one, fewer tasks)
__host__ __device__ inline uint32_t row_task_u(uint32_t vpr)
{
    uint32_t best_u = 4;
    double best = -1.0;
    for (uint32_t u = 4; u >= 2; u--) {
        const uint32_t span = 64u * u, tasks = (vpr + span - 1u) / span;
        const double util = (double)vpr / (double)(tasks * span);
        if (util > best + 1e-9) { best = util; best_u = u; }
    }
    return best_u;
}
// The same with ties going to the SMALLER task: the batched launch's one-wavefront workgroups stream best with 2 KiB per
// wavefront (32 x 4096^2 bf16, same box: 2 vectors per lane 81.4 %, 4: 79.7 %, 3: 80.8 % at 89 % lane use, 1: 70.4 % --
// the per-task table build then costs more than the finer granularity gains; profiles/r03_launch_shapes_*.log)
__host__ __device__ inline uint32_t row_task_u_small(uint32_t vpr)
{
    uint32_t best_u = 2;
    double best = -1.0;
    for (uint32_t u = 2; u <= 4; u++) {
        const uint32_t span = 64u * u, tasks = (vpr + span - 1u) / span;
        const double util = (double)vpr / (double)(tasks * span);
        if (util > best + 1e-9) { best = util; best_u = u; }
    }
    return best_u;
}


// ------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// Streaming (nontemporal) 16-byte accesses: x is read once and out written once, so the
// lines are marked evict-first instead of thrashing L2 / MALL.  Measured on MI355X
// (tools/ubench.hip): a 4 KiB-per-wave copy runs 5.4 TB/s with nt, 2.9 TB/s without.
// Every pointer these kernels dereference is device (global) memory.  Pointers that arrive through a descriptor in
// memory (the batched launch) are "generic" to the compiler, which then emits FLAT instructions -- those tick BOTH the
// vector-memory and the LDS counters, so every wait for a table read also waits for them.  Saying "address space 1"
// turns them into global_load / global_store (round 3: the batched kernels had 45 flat loads and no global one).
#define ANTQ_GLOBAL __attribute__((address_space(1)))
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream(const uint4 *p)
{
    u32x4_t v = __builtin_nontemporal_load((const ANTQ_GLOBAL u32x4_t *)(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(uint4 *p, const uint4 &v)
{
    u32x4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (ANTQ_GLOBAL u32x4_t *)(p));
}
// plain (cached) global loads: tables, scales
__device__ __forceinline__ uint4 ld_global(const uint4 *p)
{
    u32x4_t v = *(const ANTQ_GLOBAL u32x4_t *)(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld_global(const float *p) { return *(const ANTQ_GLOBAL float *)(p); }
__device__ __forceinline__ void st_global(float *p, float v) { *(ANTQ_GLOBAL float *)(p) = v; }

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
    static constexpr int DTYPE = ANTQ_F32;
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
    // "every magnitude of the vector is below lim" on the raw words: amax_acc(0, v) against lim_key(lim).  NaN / Inf are
    // never below (their magnitude bits exceed every finite value's).
    __device__ __forceinline__ static uint32_t lim_key(float lim) { return f2u(lim); }
    __device__ __forceinline__ static bool all_below(uint32_t acc, uint32_t key) { return acc < key; }
};
template <> struct IO<bf16_tag> {
    static constexpr int DTYPE = ANTQ_BF16;
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
    // both 16-bit magnitudes <= (lim's bf16 bits, truncated) - 1, i.e. strictly below lim: one v_pk_max_u16 + one compare
    __device__ __forceinline__ static uint32_t lim_key(float lim) { return (max(f2u(lim) >> 16, 1u) - 1u) * 0x10001u; }
    __device__ __forceinline__ static bool all_below(uint32_t acc, uint32_t key)
    {
        return as_u32(__builtin_elementwise_max(as_u16x2(acc), as_u16x2(key))) == key;
    }
};
template <> struct IO<f16_tag> {
    static constexpr int DTYPE = ANTQ_F16;
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
        // The fp32 result is rounded to half as a SECOND rounding (what `tensor.to(torch.float16)` does to an fp32
        // tensor).  Without the barrier LLVM folds "fp32 multiply, then convert" into v_fma_mixlo_f16, which rounds
        // the exact product to half once -- a different number about once in 2^13 elements.
        asm volatile("" : "+v"(f));
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
    // half magnitudes order like their bits too; lim rounded to half (nearest), one step taken off: strictly below lim
    __device__ __forceinline__ static uint32_t lim_key(float lim)
    {
        _Float16 h = (_Float16)lim;                    // beyond 65504 -> Inf (0x7c00): every finite half passes
        return (max((uint32_t) * reinterpret_cast<uint16_t *>(&h), 2u) - 2u) * 0x10001u;
    }
    __device__ __forceinline__ static bool all_below(uint32_t acc, uint32_t key)
    {
        return as_u32(__builtin_elementwise_max(as_u16x2(acc), as_u16x2(key))) == key;
    }
};

// floor(o / opr) without an integer division: one f64 multiply by the reciprocal (inv = 1.0 / opr) and a +-1 fix-up.
// Exact for every o < 2^32: the estimate is off by at most one, and the fix-up compares in 64 bits (q * opr can exceed
// 2^32 when the estimate is one too high and o is close to 2^32).
__device__ __forceinline__ uint32_t oct_row(uint32_t o, uint32_t opr, double inv)
{
    uint32_t q = (uint32_t)((double)o * inv);
    const uint64_t qo = (uint64_t)q * opr;
    if (qo > o) q--;
    else if ((uint64_t)o - qo >= opr) q++;
    return q;
}

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
    uint32_t adom;       // PlanHeader::adom: the approximate-quotient path (quant_vec_a) may be used
    float xlim;          // PlanHeader::xlim: |x * rcp(s)| below this -> table path of quant_vec_a
    uint32_t atab_slots; // PlanHeader::atab_slots: entries of the a-table behind the plan's entries (adom plans)
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
        smem[(i < grid_units) ? (pa.n_entries + i) : (i - grid_units)] = ld_global(plan_tab + i);
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
// Table lookup for EPL grid-domain values d (all inside the table's domain, |d| < fastlim): q = nearest grid value of
// the reference scan, j = its scan-order index.  One ds_read_b128 and one compare per element.
template <int EPL, bool IDX>
__device__ __forceinline__ void lut_lookup(const PlanArgs &pa, const PlanLds &L, const float (&d)[EPL], float (&q)[EPL],
                                           int (&j)[EPL])
{
    if (pa.linear) {
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
    } else {
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
    }
}

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
    if (fast) {
        lut_lookup<EPL, IDX>(pa, L, d, q, j);
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

}  // namespace antq

#endif  // ANTQ_DEVICE_H
