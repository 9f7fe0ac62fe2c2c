// antq_k_nearest.h -- quant_cuda.quant replacement: literal scan and the binary-search fast path
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_NEAREST_H
#define ANTQ_K_NEAREST_H

#include "antq_device.h"

namespace antq {

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
    __shared__ float sv[512];     // sorted values, +inf beyond m (m = 256: the search probes up to index 2 m - 2)
    __shared__ int16_t win[256];  // sorted position -> last scan index with that value
    __shared__ int s_bad;
    __shared__ float s_mingap, s_maxgap, s_fastlim;
    const int t = threadIdx.x;
    if (t == 0) { s_bad = 0; s_mingap = 3.0e38f; s_maxgap = 0.0f; }
    sv[t] = __builtin_inff();                          // padding for the fixed-step search
    sv[t + 256] = __builtin_inff();
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
// Nearest value through a plan (antq_nearest_plan): when the caller knows the grid on the host (a quantiser's static
// codebook), the M-step scan is the same one-LDS-read lookup the fused kernels use, on d = x itself.  16 bytes per lane,
// 4 vectors in flight; elements beyond the table's domain (|x| >= fastlim, NaN, Inf) and scan plans run the literal scan.
//
// HINTED (antq_nearest_hinted): the operator's contract is "nearest value of the DEVICE array `gcheck`", and the plan
// is only what the host believes that array holds.  Every workgroup compares the m device values with the plan's own
// copy of the grid, bit for bit (m <= 1024 L2-resident floats, fetched ahead of the HBM loads); on any difference it
// runs the literal scan on the device values instead -- a stale plan costs time, never a wrong result -- and raises
// *stale so that the host can drop its belief without ever synchronising.
// ------------------------------------------------------------------------------------
// U vectors per lane: 2 measured best for the plain table kernel (23.1 vs 24.4 us per 4096^2 fp32, as k_fq_lane); the hinted
// form checks the device grid once per workgroup and big tables are staged per workgroup: 4
template <typename T, bool IDX, bool HINTED, int U>
__global__ void __launch_bounds__(256)
k_nearest_plan(const uint4 *__restrict__ x, uint4 *__restrict__ z, int16_t *__restrict__ idx, size_t n_vec,
               PlanArgs pa, const uint4 *__restrict__ plan_tab, const float *__restrict__ gcheck, int *__restrict__ stale)
{
    constexpr int EPL = IO<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    float g0 = 0.0f;
    if (HINTED && threadIdx.x < pa.m) g0 = gcheck[threadIdx.x];
    const size_t first = ((size_t)blockIdx.x * U) * 256u + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * 256u;
        v[u] = make_uint4(0, 0, 0, 0);
        if (vi < n_vec) v[u] = ld_stream(x + vi);
    }
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    bool hinted_ok = true;
    if (HINTED) {
        int diff = 0;
        if (threadIdx.x < pa.m) diff = f2u(g0) != f2u(L.grid[threadIdx.x]);
        for (uint32_t i = threadIdx.x + 256u; i < pa.m; i += 256u) diff |= f2u(gcheck[i]) != f2u(L.grid[i]);
        if (__syncthreads_or(diff)) {
            // the plan describes another grid: literal scan on what the device array holds now
            float *lg = const_cast<float *>(L.grid);
            for (uint32_t i = threadIdx.x; i < pa.m; i += 256u) lg[i] = gcheck[i];
            __syncthreads();
            hinted_ok = false;
            if (stale && blockIdx.x == 0 && threadIdx.x == 0) *stale = 1;
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * 256u;
        if (vi >= n_vec) continue;
        float d[EPL], q[EPL];
        int j[EPL];
        IO<T>::unpack(v[u], d);
        bool fast = pa.kind == kPlanLut && hinted_ok;
#pragma unroll
        for (int e = 0; e < EPL; e++) fast = fast && (fabsf(d[e]) < pa.fastlim);
        if (fast) {
            lut_lookup<EPL, IDX>(pa, L, d, q, j);
        } else {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                int jj;
                q[e] = scan_lds(d[e], L.grid, (int)pa.m, jj);
                j[e] = jj;
            }
        }
        st_stream(z + vi, IO<T>::pack(q));
        if (IDX) store_idx<EPL>(idx, vi, j);
    }
}

}  // namespace antq

#endif  // ANTQ_K_NEAREST_H
