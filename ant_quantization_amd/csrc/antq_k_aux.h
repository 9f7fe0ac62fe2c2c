// antq_k_aux.h -- quant_affine, copy, abs-max, alpha gradient
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_AUX_H
#define ANTQ_K_AUX_H

#include "antq_device.h"

namespace antq {

// ------------------------------------------------------------------------------------
// AsymmetricQuantFunction.forward, quant_affine.py:95-115.  fp32, element-wise; rintf is
// round-half-to-even like torch.round.  Expression order follows the reference exactly:
//   scale*x - zp  (linear_quantize :39), (q + zp) / scale  (linear_dequantize :62).
// ------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
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

// Vector variant (16-byte aligned, row_len % 4 == 0): 4 floats per lane per access, 4 accesses in flight, the row's
// scale / zero point computed once per vector, and the final (q + zp) / scale as the exact 5-FMA division when the
// operands are inside its domain (true division otherwise).  Same arithmetic, same order, same bits as k_affine.
#ifndef ANTQ_AFFINE_U
#define ANTQ_AFFINE_U 2
#endif
constexpr int kAffineU = ANTQ_AFFINE_U;   // vectors per lane (2 measured best for one-launch-per-tensor kernels)
static __global__ void __launch_bounds__(256)
k_affine_vec(const uint4 *__restrict__ x, uint4 *__restrict__ out, int4 *__restrict__ qout, size_t n_vec, size_t vpr, int k,
             const float *__restrict__ xmin, const float *__restrict__ xmax, int per_row)
{
    constexpr int U = kAffineU;
    const size_t first = ((size_t)blockIdx.x * U) * 256u + threadIdx.x;
    const float nlev = (float)((1 << k) - 1);
    const float half = (float)(1 << (k - 1));
    uint4 v[U];
    float mn[U], mx[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * 256u;
        v[u] = make_uint4(0, 0, 0, 0);
        mn[u] = 0.0f;
        mx[u] = 1.0f;
        if (vi < n_vec) {
            v[u] = ld_stream(x + vi);
            const size_t r = per_row ? vi / vpr : 0;
            mn[u] = xmin[r];
            mx[u] = xmax[r];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * 256u;
        if (vi >= n_vec) continue;
        float range = mx[u] - mn[u];
        if (range < 1e-8f) range = 1e-8f;
        const float scale = (1.0f / range) * nlev;
        float zp = rintf(scale * mn[u]);
        zp = zp + half;
        const float rs = 1.0f / scale;
        const float as = fabsf(scale);
        const bool dom = (as >= kScaleLo) && (as <= kScaleHi);
        const float xs[4] = {u2f(v[u].x), u2f(v[u].y), u2f(v[u].z), u2f(v[u].w)};
        float o[4];
        int qi[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float q = rintf(scale * xs[e] - zp);
            if (q < -half) q = -half;
            if (q > half - 1.0f) q = half - 1.0f;
            qi[e] = (int)q;
            const float num = q + zp;
            const float an = fabsf(num);
            const bool fast = dom && (num == 0.0f || (an >= 0x1p-78f && an <= 0x1p60f));
            o[e] = fast ? div_fast(num, scale, rs) : num / scale;
        }
        st_stream(out + vi, make_uint4(f2u(o[0]), f2u(o[1]), f2u(o[2]), f2u(o[3])));
        if (qout) qout[vi] = make_int4(qi[0], qi[1], qi[2], qi[3]);
    }
}

// 16 B per lane streaming copy: the empirical HBM ceiling for this access pattern.
static __global__ void __launch_bounds__(256)
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

static __global__ void __launch_bounds__(256) k_scale_inplace(float *__restrict__ a, size_t n, float ratio)
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
static __global__ void k_zero_f32(float *p) { if (threadIdx.x == 0) *p = 0.0f; }      // (cheaper than hipMemsetAsync of 4 bytes)

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
                const size_t vpr = row_len / EPL;
                uint32_t mp = 0;
                size_t i = lane;
                for (; i + 192 < vpr; i += 256) {          // four 16-byte loads in flight per lane
                    const uint4 a0 = p[i], a1 = p[i + 64], a2 = p[i + 128], a3 = p[i + 192];
                    mp = IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(mp, a0), a1), a2), a3);
                }
                for (; i < vpr; i += 64) mp = IO<T>::amax_acc(mp, p[i]);
                m = IO<T>::amax_bits(mp);
            } else {
                for (size_t i = lane; i < row_len; i += 64) m = max(m, f2u(IO<T>::load1(x, r * row_len + i)) & 0x7fffffffu);
            }
            m = wave_max_u32(m);
            if (lane == 0) amax[r] = u2f(m);
        }
    } else {
        // one scale for the whole tensor: block-strided, eight independent 16-byte loads in flight per lane, one
        // atomicMax per workgroup (plain loads: the clip search reads the same bytes next, out of the Infinity Cache)
        const size_t n = rows * row_len;
        const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, stride = (size_t)gridDim.x * 256u;
        uint32_t m = 0;
        if (vec_ok) {
            const uint4 *p = static_cast<const uint4 *>(x);
            const size_t nv = n / EPL;
            uint32_t mp = 0;
            size_t i = tid;
            for (; i + 7 * stride < nv; i += 8 * stride) {         // eight 16-byte loads in flight per lane
                const uint4 a0 = p[i], a1 = p[i + stride], a2 = p[i + 2 * stride], a3 = p[i + 3 * stride];
                const uint4 a4 = p[i + 4 * stride], a5 = p[i + 5 * stride], a6 = p[i + 6 * stride], a7 = p[i + 7 * stride];
                mp = IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(mp, a0), a1), a2), a3);
                mp = IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(IO<T>::amax_acc(mp, a4), a5), a6), a7);
            }
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
            // (a look before the atomic: after the first few workgroups the running maximum already covers most of the
            //  others, whose atomics -- serialised on ONE address -- then never happen: lets the launch use every CU slot)
            unsigned int *dst = reinterpret_cast<unsigned int *>(amax);
            if (m && m > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, m);
        }
    }
}

// Small groups (a power of two <= 64 of 16-byte vectors per group, e.g. group-16): several groups per wavefront, four
// vectors per lane, butterfly max over the group's lanes (DPP) -- a wavefront per 2-vector group would idle 62 lanes.
// Streaming (nontemporal) loads: 16384^2 bf16 in groups of 16 64.9 -> 72.5 % of 8 TB/s, fp32 73.3 -> 79.8 %, one 33.5 MB tensor
// 53.7 -> 56.2 %; eight vectors per lane instead of four: no better (72.0 / 77.9 %) -- profiles/r05_absmax_experiments.log.
template <typename T, int U = 4, bool NT = true>
__global__ void __launch_bounds__(256)
k_absmax_groups(const uint4 *__restrict__ x, float *__restrict__ amax, size_t n_vec, uint32_t vpr, int vshift)
{
    const size_t first = ((size_t)blockIdx.x * U) * 256u + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * 256u;
        v[u] = make_uint4(0, 0, 0, 0);
        if (vi < n_vec) v[u] = NT ? ld_stream(x + vi) : x[vi];
    }
    uint32_t m[U];
#pragma unroll
    for (int u = 0; u < U; u++) m[u] = IO<T>::amax_bits(IO<T>::amax_acc(0u, v[u]));
    group_max_multi<U>(m, vpr);
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * 256u;
        if (vi < n_vec && (vi & (vpr - 1)) == 0) amax[vi >> vshift] = u2f(m[u]);
    }
}

// ------------------------------------------------------------------------------------
// Backward of the fused fake-quant w.r.t. alpha (QAT, AQ:39 alpha is a Parameter; AQ:544-549 straight-through
// graph): d out / d alpha = (q - d) / gmax = (out - x) / alpha, so
//     gsum[r] = sum_c fl32( gout[r,c] * fl32(out[r,c] - x[r,c]) )          (the caller divides by alpha[r])
// fp32 terms, fp64 accumulation.  One wavefront per row; one scale per tensor: block-strided, every workgroup writes its
// partial to the caller's workspace and k_sum_partials adds them in a fixed order (no floating-point atomics).  d out / d x is the identity (no clip mask in the reference), so there is no kernel for it.
// ------------------------------------------------------------------------------------
constexpr int kPartialStride = 128;   // doubles between two workgroup partials in the workspace (= kPtCand, antq_k_search.h)

template <typename T>
__global__ void __launch_bounds__(256)
k_alpha_grad(const void *__restrict__ x, const void *__restrict__ out, const void *__restrict__ gout,
             double *__restrict__ gsum, double *__restrict__ ws, size_t rows, size_t row_len, int per_row, int vec_ok)
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
                size_t i = lane;
                for (; i + 64 < vpr; i += 128) {           // six 16-byte streaming loads in flight per lane
                    const uint4 x0 = ld_stream(px + i), o0 = ld_stream(po + i), g0 = ld_stream(pg + i);
                    const uint4 x1 = ld_stream(px + i + 64), o1 = ld_stream(po + i + 64), g1 = ld_stream(pg + i + 64);
                    acc += (double)vec_term(x0, o0, g0);
                    acc += (double)vec_term(x1, o1, g1);
                }
                for (; i < vpr; i += 64) acc += (double)vec_term(ld_stream(px + i), ld_stream(po + i), ld_stream(pg + i));
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
                const uint4 x0 = ld_stream(px + i), o0 = ld_stream(po + i), g0 = ld_stream(pg + i);
                const uint4 x1 = ld_stream(px + i + stride), o1 = ld_stream(po + i + stride), g1 = ld_stream(pg + i + stride);
                acc += (double)vec_term(x0, o0, g0);
                acc += (double)vec_term(x1, o1, g1);
            }
            for (; i < nv; i += stride) acc += (double)vec_term(ld_stream(px + i), ld_stream(po + i), ld_stream(pg + i));
            for (size_t k = nv * EPL + tid; k < n; k += stride) acc += (double)one_term(k);
        } else {
            for (size_t k = tid; k < n; k += stride) acc += (double)one_term(k);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        __shared__ double wsum[4];
        if (lane == 0) wsum[threadIdx.x >> 6] = acc;
        __syncthreads();
        // this workgroup's partial; k_sum_partials (antq_k_search.h) adds the workgroups in a fixed tree
        if (threadIdx.x == 0) ws[(size_t)blockIdx.x * kPartialStride] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    }
}

// ------------------------------------------------------------------------------------
// OliVe's clip statistic (OQ/quant_modules.py:193-197, :213-218): x_max = max(|mean + 3 std|, |mean - 3 std|) per row or
// per tensor, unbiased std.  The reference runs t.mean() and t.std() (two reductions, >= 3 reads of the tensor) before a
// search that itself needs one read; here ONE read-only pass leaves (sum x, sum x^2) in double per row / tensor --
// x^2 is exact in double, every sum is formed in one fixed order (a wavefront owns a row; whole-tensor sums from
// workgroup partials through k_sum_partials) -- and k_xmax_3sigma turns them into the fp32 x_max with the roundings the
// reference's dtype would apply (fp32: mean / std rounded to float; bf16 / fp16: mean, std, 3 * std, the sum and the
// difference each rounded to the tensor's dtype, as torch's element-wise kernels do).  The sums themselves are what a
// row-sharded per-tensor quantiser all-reduces across ranks ((sum x, sum x^2, n)).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_moments(const void *__restrict__ x, double *__restrict__ sums, double *__restrict__ ws, size_t rows, size_t row_len,
          int per_row, int vec_ok)
{
    constexpr int EPL = IO<T>::EPL;
    const uint32_t lane = threadIdx.x & 63u;
    double s1 = 0.0, s2 = 0.0;
    auto acc_vec = [&](const uint4 &v) {
        float f[EPL];
        IO<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const double d = (double)f[e];
            s1 += d;
            s2 = __builtin_fma(d, d, s2);
        }
    };
    auto acc_one = [&](size_t i) {
        const double d = (double)IO<T>::load1(x, i);
        s1 += d;
        s2 = __builtin_fma(d, d, s2);
    };
    if (per_row) {
        const size_t wave = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
        const size_t nwaves = (size_t)gridDim.x * 4u;
        for (size_t r = wave; r < rows; r += nwaves) {
            s1 = 0.0;
            s2 = 0.0;
            if (vec_ok) {
                const size_t vpr = row_len / EPL;
                const uint4 *p = static_cast<const uint4 *>(x) + r * vpr;
                size_t i = lane;
                for (; i + 192 < vpr; i += 256) {          // four 16-byte loads in flight per lane
                    const uint4 a0 = p[i], a1 = p[i + 64], a2 = p[i + 128], a3 = p[i + 192];
                    acc_vec(a0); acc_vec(a1); acc_vec(a2); acc_vec(a3);
                }
                for (; i < vpr; i += 64) acc_vec(p[i]);
            } else {
                for (size_t i = lane; i < row_len; i += 64) acc_one(r * row_len + i);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
            if (lane == 0) { sums[2 * r] = s1; sums[2 * r + 1] = s2; }
        }
    } else {
        const size_t n = rows * row_len;
        const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, stride = (size_t)gridDim.x * 256u;
        if (vec_ok) {
            const size_t nv = n / EPL;
            const uint4 *p = static_cast<const uint4 *>(x);
            size_t i = tid;
            for (; i + 3 * stride < nv; i += 4 * stride) {
                const uint4 a0 = p[i], a1 = p[i + stride], a2 = p[i + 2 * stride], a3 = p[i + 3 * stride];
                acc_vec(a0); acc_vec(a1); acc_vec(a2); acc_vec(a3);
            }
            for (; i < nv; i += stride) acc_vec(p[i]);
            for (size_t k = nv * EPL + tid; k < n; k += stride) acc_one(k);
        } else {
            for (size_t k = tid; k < n; k += stride) acc_one(k);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
        __shared__ double w1[4], w2[4];
        if (lane == 0) { w1[threadIdx.x >> 6] = s1; w2[threadIdx.x >> 6] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            ws[(size_t)blockIdx.x * kPartialStride] = (w1[0] + w1[1]) + (w1[2] + w1[3]);
            ws[(size_t)blockIdx.x * kPartialStride + 1] = (w2[0] + w2[1]) + (w2[2] + w2[3]);
        }
    }
}

// x_max[r] from (sum x, sum x^2) of n elements.  T: the dtype whose roundings the reference would apply.
template <typename T> struct RoundTo { __device__ __forceinline__ static float r(float v) { return v; } };
template <> struct RoundTo<bf16_tag> {
    __device__ __forceinline__ static float r(float v) { return u2f(IO<bf16_tag>::pk(v, 0.0f) << 16); }
};
template <> struct RoundTo<f16_tag> {
    __device__ __forceinline__ static float r(float v) { return IO<f16_tag>::h2f(IO<f16_tag>::f2h(v)); }
};
template <typename T>
__global__ void __launch_bounds__(256)
k_xmax_3sigma(const double *__restrict__ sums, size_t na, double n, float *__restrict__ xmax)
{
    const size_t r = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= na) return;
    const double s1 = sums[2 * r], s2 = sums[2 * r + 1];
    const double mean = s1 / n;
    double var = (s2 - s1 * mean) / (n - 1.0);           // unbiased (torch.std default); n == 1: 0 / 0 = NaN, like torch
    if (var < 0.0) var = 0.0;                            // (cancellation on a constant row)
    const float m = RoundTo<T>::r((float)mean), sd = RoundTo<T>::r((float)__builtin_sqrt(var));
    const float t3 = RoundTo<T>::r(3.0f * sd);
    const float a = fabsf(RoundTo<T>::r(m + t3)), b = fabsf(RoundTo<T>::r(m - t3));
    xmax[r] = (a != a || b != b) ? __builtin_nanf("") : __builtin_fmaxf(a, b);     // torch.maximum propagates NaN
}

}  // namespace antq

#endif  // ANTQ_K_AUX_H
