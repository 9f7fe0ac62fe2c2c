// antq_k_reduce.h -- whole-tensor reductions in ONE launch (round 6, ABI 7): abs-max and the alpha gradient
// Part of libantq's antq_kernels.hip translation unit; gfx950 only.
//
// A whole-tensor reduction needs every workgroup's partial in one place.  Until round 5: abs-max = one atomicMax per
// workgroup on ONE address (256 of them arrive together at the end of a 5 us stream and serialise: 9.3 us for a 33.5 MB
// bf16 tensor, 45 % of the roofline) and it needs its accumulator zeroed first; alpha gradient = a second launch that adds
// the partials.  Here: hierarchical last-arriver tickets in a small caller-owned block (ANTQ_REDUCE_WS_BYTES, zeroed ONCE by
// the caller; every call leaves it zeroed, so calls issued one after the other on one stream share it):
//   * workgroup b stores its partial, fences, and takes a ticket of its GROUP (kGroup workgroups, one counter per group,
//     128 bytes apart: the counters of different groups live in different cache lines / channels and do not queue up);
//   * the last arriver of a group folds the group's partials IN INDEX ORDER (so the result does not depend on who came
//     last), stores the group partial, fences, and takes a ticket of the launch;
//   * the last arriver of the launch folds the group partials in index order and writes the result.
// At most kGroup atomics queue on any one address (16 or 32 instead of 256 or 1024), sums are formed in one fixed tree
// (bit-reproducible, no floating-point atomics), nothing has to be zeroed per call and there is no second launch.
#ifndef ANTQ_K_REDUCE_H
#define ANTQ_K_REDUCE_H

#include "antq_device.h"

namespace antq {

constexpr uint32_t kTkStride = 32;                 // uint32 between two counters (128 bytes)
constexpr uint32_t kTkMaxGroups = 64;
constexpr uint32_t kTkCounterBytes = (1 + kTkMaxGroups) * kTkStride * 4;          // 8320: the part that must be zero
constexpr uint32_t kTkPartialOffset = 16384;       // bytes: workgroup partials (<= 1024 x 16 B), any content
constexpr uint32_t kTkGroupOffset = 16384 + 16384 + 8192;   // bytes: group partials (<= 64 x 16 B), any content
static_assert(kTkGroupOffset + kTkMaxGroups * 16 <= ANTQ_REDUCE_WS_BYTES, "reduce workspace layout");

// One ticket.  Returns true in EVERY lane of the calling wavefront iff this workgroup is the last of `count` to arrive at
// `counter` (which it then resets).  Called by wavefront 0 after the workgroup's partial has been stored by its lane 0.
// NO __threadfence(): a device-scope fence on gfx950 writes back and invalidates the XCD's whole L2 (measured: 1024
// workgroups each fencing twice took 68 us for an 18 us kernel).  Instead every partial is written and read with
// device-scope ATOMIC stores / loads (sc1: written through to, and read from, the memory side, past the per-XCD L2), and
// the producer waits for its store's acknowledgement (s_waitcnt 0) before it takes the ticket -- the ticket itself is a
// device-scope atomic, so whoever sees the final count runs after every member's partial has reached memory.
__device__ __forceinline__ bool ticket_last(uint32_t *counter, uint32_t count, uint32_t lane)
{
    uint32_t last = 0;
    if (lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);                                         // the partial's write-through has been acknowledged
        const uint32_t old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == count - 1u) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left zeroed for the next call
            last = 1;
        }
    }
    return __shfl((int)last, 0, 64) != 0;
}

struct RedLayout {
    uint32_t *counters;          // [0]: launch ticket; [(1 + g) * kTkStride]: group g
    char *partials, *groups;
    __device__ __forceinline__ explicit RedLayout(void *ws)
        : counters(static_cast<uint32_t *>(ws)), partials(static_cast<char *>(ws) + kTkPartialOffset),
          groups(static_cast<char *>(ws) + kTkGroupOffset) {}
};

// ---- fold policies: how partials of type V combine, lane-parallel then across the wavefront, in a fixed order ----------
struct FoldMaxU32 {
    typedef uint32_t V;
    __device__ __forceinline__ static V zero() { return 0u; }
    __device__ __forceinline__ static V add(V a, V b) { return max(a, b); }
    __device__ __forceinline__ static V ld(const V *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ static void st(V *p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ static V wave(V v) { return wave_max_u32(v); }
};
struct FoldSumF64 {
    typedef double V;
    __device__ __forceinline__ static V zero() { return 0.0; }
    __device__ __forceinline__ static V add(V a, V b) { return a + b; }
    __device__ __forceinline__ static V ld(const V *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ static void st(V *p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ static V wave(V v)
    {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    }
};

// The tail every workgroup runs after its own reduction: `mine` = the workgroup's partial (valid in lane 0 of wavefront 0).
// Only wavefront 0 calls.  Returns true (in every lane) in the ONE workgroup that holds the final value, which it gets in
// `result`.  group = workgroups per group (<= 64: one lane per member when folding).
template <typename F>
__device__ __forceinline__ bool reduce_tail(void *ws, typename F::V mine, uint32_t lane, uint32_t group, typename F::V &result)
{
    typedef typename F::V V;
    RedLayout L(ws);
    V *part = reinterpret_cast<V *>(L.partials), *gp = reinterpret_cast<V *>(L.groups);
    const uint32_t b = blockIdx.x, nb = gridDim.x;
    const uint32_t g = b / group, ng = (nb + group - 1u) / group;
    const uint32_t in_group = min(group, nb - g * group);
    if (lane == 0) F::st(part + b, mine);
    if (!ticket_last(L.counters + (1u + g) * kTkStride, in_group, lane)) return false;
    V v = F::zero();
    for (uint32_t i = lane; i < in_group; i += 64u) v = F::add(v, F::ld(part + g * group + i));      // (index order, then a
    v = F::wave(v);                                                                                   //  fixed butterfly)
    if (ng == 1u) { result = v; return true; }
    if (lane == 0) F::st(gp + g, v);
    if (!ticket_last(L.counters, ng, lane)) return false;
    V w = F::zero();
    for (uint32_t i = lane; i < ng; i += 64u) w = F::add(w, F::ld(gp + i));
    result = F::wave(w);
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// abs-max of a whole tensor, one launch: 256 workgroups (one per CU) walk the tensor block-strided with eight 16-byte
// loads in flight per lane (the shape that reads 80 % of the roofline on a 0.5 GB tensor); the result is WRITTEN to
// amax[0] (not accumulated: nothing to zero).  NaN anywhere yields NaN, like torch.max (bit-pattern order).
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_absmax_t(const void *__restrict__ x, float *__restrict__ amax, size_t n, int vec_ok, void *__restrict__ ws, uint32_t group)
{
    constexpr int EPL = IO<T>::EPL;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, stride = (size_t)gridDim.x * 256u;
    uint32_t m = 0;
    if (vec_ok) {
        const uint4 *p = static_cast<const uint4 *>(x);
        const size_t nv = n / EPL;
        uint32_t mp = 0;
        size_t i = tid;
        for (; i + 7 * stride < nv; i += 8 * stride) {
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
    if (threadIdx.x >= 64u) return;
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    uint32_t res;
    if (reduce_tail<FoldMaxU32>(ws, m, lane, group, res) && lane == 0) amax[0] = u2f(res);
}

// ------------------------------------------------------------------------------------------------------------------
// Row abs-max, rows of 64 * U vectors (U = 2, 4, 8, 16): one wavefront = one workgroup per row, the whole row in flight
// at once through streaming loads (the tensor is read once here; the shape that took groups of 16 from 65 to 72 %).
// 16 x 4096^2, one launch per tensor: bf16 56 -> 62 %, fp32 63 -> 73 % of 8 TB/s.
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int U>
__global__ void __launch_bounds__(64)
k_absmax_rows(const uint4 *__restrict__ x, float *__restrict__ amax, uint32_t rows)
{
    const uint32_t lane = threadIdx.x;
    for (uint32_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const uint4 *p = x + (size_t)r * (64u * U) + lane;
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld_stream(p + 64 * u);
        uint32_t mp = 0;
#pragma unroll
        for (int u = 0; u < U; u++) mp = IO<T>::amax_acc(mp, v[u]);
        const uint32_t m = wave_max_u32(IO<T>::amax_bits(mp));
        if (lane == 0) amax[r] = u2f(m);
    }
}
// (measured and not kept, profiles/r06_aux_kernels.log: 2 / 4 wavefronts per workgroup 61.7 / 62.3 % against 62.0 % for bf16
//  rows of 4096, two rows per wavefront in flight 59.9 % -- a 33.5 MB read-only launch does not get under ~6.7 us here
//  whatever its shape: the copy-shaped chunk kernel below reads the same bytes in 6.5 us)

// (antq_absmax_into keeps the round-5 kernel.  Measured with a FRESH zero in the slot -- a slot that already holds the
//  maximum makes every workgroup skip its atomic and flatters any shape -- 16 x 4096^2 bf16, profiles/r06_absmax_fresh.log:
//  round-5 kernel 9.2-9.4 us (45 %: 7.4 us of streaming + the 256 closing atomics); 8 KiB chunks with one single-wavefront
//  workgroup each 12.6 us at 512 workgroups, 65 us at 4096 (the atomics serialise at ~15 ns); an early fire-and-forget
//  maximum after the first chunk + a closing look 10.3 us (vmcnt is in order: the issuing wavefront's later loads wait for
//  the atomic's turn in the queue); the ticket kernel above 9.6 us.  A grid-wide meeting point costs ~2 us here whichever
//  way it is built; profiles/r06_absmax_early.patch has the kernels.)

// ------------------------------------------------------------------------------------------------------------------
// Alpha gradient of a tensor with ONE scale, one launch (AQ:39, :544-549; see k_alpha_grad): every wavefront walks
// contiguous chunks of 128 vectors of x / out / gout (six 16-byte streaming loads in flight per lane: the per-row kernel's
// access shape, 74.5 % where the block-strided walk read 57 %), fp32 terms, fp64 accumulation, fixed-order tree.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_alpha_grad_t(const void *__restrict__ x, const void *__restrict__ out, const void *__restrict__ gout,
               double *__restrict__ gsum, size_t n, int vec_ok, void *__restrict__ ws, uint32_t group)
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
    double acc = 0.0;
    const size_t wave = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4u;
    if (vec_ok) {
        const size_t nv = n / EPL, nchunks = (nv + 127u) / 128u;
        const uint4 *px = static_cast<const uint4 *>(x), *po = static_cast<const uint4 *>(out);
        const uint4 *pg = static_cast<const uint4 *>(gout);
        for (size_t c = wave; c < nchunks; c += nwaves) {
            const size_t i0 = c * 128u + lane, i1 = i0 + 64u;
            if (i1 < nv) {
                const uint4 x0 = ld_stream(px + i0), o0 = ld_stream(po + i0), g0 = ld_stream(pg + i0);
                const uint4 x1 = ld_stream(px + i1), o1 = ld_stream(po + i1), g1 = ld_stream(pg + i1);
                acc += (double)vec_term(x0, o0, g0);
                acc += (double)vec_term(x1, o1, g1);
            } else if (i0 < nv) {
                acc += (double)vec_term(ld_stream(px + i0), ld_stream(po + i0), ld_stream(pg + i0));
            }
        }
        const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, stride = (size_t)gridDim.x * 256u;
        for (size_t k = nv * EPL + tid; k < n; k += stride) acc += (double)one_term(k);
    } else {
        const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, stride = (size_t)gridDim.x * 256u;
        for (size_t k = tid; k < n; k += stride) acc += (double)one_term(k);
    }
    acc = FoldSumF64::wave(acc);
    __shared__ double wsum[4];
    if (lane == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x >= 64u) return;
    const double mine = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    double res;
    if (reduce_tail<FoldSumF64>(ws, mine, lane, group, res) && lane == 0) gsum[0] = res;
}

}  // namespace antq

#endif  // ANTQ_K_REDUCE_H
