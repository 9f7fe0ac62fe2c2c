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
constexpr size_t kHistWorkspaceBytes = 2 * (size_t)kHistMaxG * kHistSlabBytes + 2 * kHistSlabBytes;   // slabs, then count[65536]

static __global__ void __launch_bounds__(1024)
k_hist16(const uint4 *__restrict__ x, size_t nv, uint32_t G, uint32_t *__restrict__ slabs)
{
    __shared__ __attribute__((aligned(16))) uint32_t bins[kHistBins];        // 128 KiB, static: one workgroup per CU
    const uint32_t sign = blockIdx.x & 1u, w = blockIdx.x >> 1;
    for (uint32_t i = threadIdx.x; i < kHistBins / 4; i += 1024u) reinterpret_cast<uint4 *>(bins)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    uint32_t zeros = 0;                                                      // this wavefront's count of the zero pattern (lane 0)
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
    }
    if ((threadIdx.x & 63u) == 0u && zeros) atomicAdd(&bins[0], zeros);
    __syncthreads();
    uint4 *slab = reinterpret_cast<uint4 *>(slabs + (size_t)blockIdx.x * kHistBins);
    for (uint32_t k = threadIdx.x; k < kHistBins / 4; k += 1024u) slab[k] = reinterpret_cast<const uint4 *>(bins)[k];
}

// count[sign * 32768 + b] = sum over the G slabs of that sign
static __global__ void __launch_bounds__(256)
k_hist_reduce(const uint32_t *__restrict__ slabs, uint32_t G, uint32_t *__restrict__ count)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;                      // 0 .. 65535
    const uint32_t sign = p >> 15, b = p & 0x7fffu;
    uint32_t s = 0;
    for (uint32_t w = 0; w < G; w++) s += slabs[((size_t)(2u * w + sign)) * kHistBins + b];
    count[p] = s;
}

struct HistTypes {
    const float *grid[4];      // device pointers: m floats each (the plan blob's copy of the grid)
    int m[4];
    float gmax[4];
    int ntypes;
};

// sse[t * ncand + c] for one (t, c) per workgroup.  Terms in ascending pattern order per thread (p = tid, tid + 1024, ...),
// then a fixed tree: the same bits on every run.
template <typename T>
__global__ void __launch_bounds__(1024)
k_hist_score(const uint32_t *__restrict__ count, const float *__restrict__ xmax, const float *__restrict__ ratios, int ncand,
             HistTypes ht, double *__restrict__ sse)
{
    __shared__ float g[ANTQ_MAX_GRID];
    __shared__ double part[16];
    const int f = (int)blockIdx.x, t = f / ncand, c = f - t * ncand;
    const int m = ht.m[t];
    for (int i = (int)threadIdx.x; i < m; i += 1024) g[i] = ht.grid[t][i];
    __syncthreads();
    const float a = xmax[0] * ratios[c];                 // AQ:300  new_alpha = base_alpha * fl32(i * 0.01)
    const Scale sc = make_scale(a, ht.gmax[t]);
    double acc = 0.0;
#pragma unroll 1
    for (uint32_t p = threadIdx.x; p < 65536u; p += 1024u) {
        const uint32_t n = count[p];
        if (__ballot(n != 0u) == 0ull) continue;         // (whole exponent ranges no element of the tensor lies in)
        if (n != 0u) {
            const float xv = H16<T>::val(p);
            const float d = xv / sc.s;                   // AQ:541
            int jj;
            const float q = scan_lds(d, g, m, jj);       // quant_kernel.cu:25-37
            const float tt = (q - d) + d;                // AQ:547
            const float df = fabsf(tt * sc.s - xv);      // AQ:549, :282
            acc += (double)n * (double)(df * df);
        }
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

}  // namespace antq

#endif  // ANTQ_K_HIST_H
