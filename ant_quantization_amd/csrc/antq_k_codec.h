// antq_k_codec.h -- packed 4-bit codec kernels and their launcher
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_CODEC_H
#define ANTQ_K_CODEC_H

#include "antq_device.h"
#include "antq_k_approx.h"

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
    if (threadIdx.x < 32) g[threadIdx.x] = ((int)threadIdx.x < m) ? grid[threadIdx.x] : 0.0f;
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
