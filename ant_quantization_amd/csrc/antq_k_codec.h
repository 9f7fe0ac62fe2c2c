// antq_k_codec.h -- packed 4-bit codec kernels and their launcher
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_CODEC_H
#define ANTQ_K_CODEC_H

#include "antq_device.h"
#include "antq_k_approx.h"

namespace antq {

// One lane: 8 consecutive elements (4 pairs) of one row -> 4 bytes of codes: the fused quantiser (quant_vec with the
// index output), then every index is folded into a nibble.  VEC (16-byte aligned tensors): the octet is one (bf16 / f16)
// or two (fp32) 16-byte loads and each thread has 4 octets in flight; otherwise element loads.
template <typename T> struct Oct;                      // 8 elements <-> 16-byte vectors
template <> struct Oct<float> {
    static constexpr int NV = 2;
    __device__ __forceinline__ static void load(const void *p, size_t o, float (&f)[8])
    {
        const uint4 *v = static_cast<const uint4 *>(p) + 2 * o;
        const uint4 a = ld_stream(v), b = ld_stream(v + 1);
        f[0] = u2f(a.x); f[1] = u2f(a.y); f[2] = u2f(a.z); f[3] = u2f(a.w);
        f[4] = u2f(b.x); f[5] = u2f(b.y); f[6] = u2f(b.z); f[7] = u2f(b.w);
    }
    __device__ __forceinline__ static void store(void *p, size_t o, const float (&f)[8])
    {
        uint4 *v = static_cast<uint4 *>(p) + 2 * o;
        st_stream(v, make_uint4(f2u(f[0]), f2u(f[1]), f2u(f[2]), f2u(f[3])));
        st_stream(v + 1, make_uint4(f2u(f[4]), f2u(f[5]), f2u(f[6]), f2u(f[7])));
    }
};
template <typename T> struct Oct {
    static constexpr int NV = 1;
    __device__ __forceinline__ static void load(const void *p, size_t o, float (&f)[8])
    {
        IO<T>::unpack(ld_stream(static_cast<const uint4 *>(p) + o), f);
    }
    __device__ __forceinline__ static void store(void *p, size_t o, const float (&f)[8])
    {
        st_stream(static_cast<uint4 *>(p) + o, IO<T>::pack(f));
    }
};

template <typename T, bool OVP, bool VEC, int UE>
__global__ void __launch_bounds__(256)
k_encode4(const void *__restrict__ x, uint32_t *__restrict__ codes, size_t n_oct, size_t row_len,
          const float *__restrict__ alpha, int per_row, float gmax, int n_normal, int zero_code,
          PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    constexpr int U = VEC ? UE : 1;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (pa.adom) tab0 = atab_prefetch<true>(pa, plan_tab);         // exact decision on x (antq_k_approx.h), with indices
    else if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const size_t first = ((size_t)blockIdx.x * U) * 256u + threadIdx.x;   // octet index: elements [8o, 8o+8)
    float xf[U][8], a[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t o = first + (size_t)u * 256u;
        a[u] = 1.0f;
#pragma unroll
        for (int e = 0; e < 8; e++) xf[u][e] = 0.0f;
        if (o < n_oct) {
            // row_len % 8 == 0: an octet (4 pairs) lies inside one row -> one scale
            a[u] = alpha[per_row ? (o * 8 / row_len) : 0];
            if (VEC) Oct<T>::load(x, o, xf[u]);
            else {
#pragma unroll
                for (int e = 0; e < 8; e++) xf[u][e] = IO<T>::load1(x, o * 8 + e);
            }
        }
    }
    PlanLds L;
    ATab A;
    if (pa.adom) A = stage_atab<true>(pa, plan_tab, smem, tab0);
    else L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t o = first + (size_t)u * 256u;
        if (o >= n_oct) continue;
        float of[8];
        int j[8];
        if (pa.adom) {
            const ScaleA sc = make_scale_a(a[u], gmax);
            quant_vec_a<8, OVP, true>(pa, A, sc, xf[u], of, j);
        } else {
            const Scale sc = make_scale(a[u], gmax);
            quant_vec<8, OVP, true>(pa, L, sc, xf[u], of, j);
        }
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
    constexpr int kEncU = 1;                           // octets per thread when encoding: VALU-bound, more in flight did not pay
    const size_t per_block = vec ? (enc ? 256 * kEncU : 1024 * IO<T>::EPL / 8) : 256;     // octets per workgroup
    const size_t blocks = (n_oct + per_block - 1) / per_block;
    if (blocks > 0x7fffffffull) return ANTQ_ERR_UNSUPPORTED;
    const float *grid_host = plan_grid(plan_host);
    int zero_code = 0;
    for (int i = 0; i < (ovp ? n_normal : m); i++) if (grid_host[i] == 0.0f) zero_code = i;
    const dim3 gd((unsigned)blocks), bd(256);
    if (enc) {
        const size_t lds = lds_table(pa, true);
        uint32_t *codes = static_cast<uint32_t *>(codes_or_out);
#define ANTQ_ENC(O, V) hipLaunchKernelGGL((k_encode4<T, O, V, kEncU>), gd, bd, lds, st, x, codes, n_oct, row_len, alpha, per_row, gmax, \
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
