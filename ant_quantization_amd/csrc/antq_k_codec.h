// antq_k_codec.h -- packed 4-bit codec kernels and their launcher
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_CODEC_H
#define ANTQ_K_CODEC_H

#include "antq_device.h"

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

#endif  // ANTQ_K_CODEC_H
