// antq_k_f64.h -- fused Quantizer._forward of a float64 tensor (round 5)
// Part of libantq's device translation units (antq_kernels.hip includes it); gfx950 only.
//
// The reference's operator is dispatched for double and narrows to float INSIDE the kernel (KQ/quant_kernel.cu:51, :28:
// `float x = in[i]`), while the tensor ops around it run in double on a `.double()` model (AQ/quant_modules.py:535-551,
// OQ:294-330):
//     scale = alpha / max(grid)                    double / double
//     d     = x / scale                            double
//     q     = grid[nearest(float(d))]              the scan on the NARROWED value, result widened back
//     [OliVe] pair rule on the flat tensor         q * 0 for a victim
//     out   = ((q - d) + d) * scale                double
// Until round 5 the module surface composed this from seven launches around antq_nearest (core.fake_quant_f64, kept for
// calls that need autograd through the torch ops); this kernel is the same sequence fused: one read, one write, 16 B per
// element.  Double precision is not a throughput path -- a thread per flat pair (2k, 2k + 1), the plan's d-domain table in
// LDS where the narrowed value lies inside it, the literal scan otherwise.
#ifndef ANTQ_K_F64_H
#define ANTQ_K_F64_H

#include "antq_device.h"

namespace antq {

__device__ __forceinline__ double f64_nearest(const PlanArgs &pa, const PlanLds &L, double d)
{
    const float f = (float)d;                                  // quant_kernel.cu:28 (round to nearest; beyond FLT_MAX: Inf)
    float q;
    if (pa.kind == kPlanLut && fabsf(f) < pa.fastlim) {        // false for NaN / Inf / beyond the table's domain
        const float dd[1] = {f};
        float qq[1];
        int jj[1];
        lut_lookup<1, false>(pa, L, dd, qq, jj);
        q = qq[0];
    } else {
        int j;
        q = scan_lds(f, L.grid, (int)pa.m, j);
    }
    return (double)q;
}

template <bool OVP>
__global__ void __launch_bounds__(256)
k_fq_f64(const double *__restrict__ x, double *__restrict__ out, size_t n, size_t row_len, const double *__restrict__ alpha,
         int per_row, double gmax, PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < pa.tab_units) tab0 = plan_tab[threadIdx.x];
    const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
    __syncthreads();
    const size_t npairs = (n + 1) / 2;
    for (size_t p = (size_t)blockIdx.x * 256u + threadIdx.x; p < npairs; p += (size_t)gridDim.x * 256u) {
        const size_t i0 = 2 * p, i1 = i0 + 1;
        const bool has1 = i1 < n;
        const double x0 = x[i0], x1 = has1 ? x[i1] : 0.0;
        const double s0 = alpha[per_row ? i0 / row_len : 0] / gmax;
        const double s1 = has1 ? alpha[per_row ? i1 / row_len : 0] / gmax : s0;
        const double d0 = x0 / s0, d1 = x1 / s1;
        double q0 = f64_nearest(pa, L, d0), q1 = has1 ? f64_nearest(pa, L, d1) : 0.0;
        if (OVP) {
            // OQ:311-320 on the flat tensor: the odd element is a victim when its even partner is an outlier, the even one when
            // its odd partner is an outlier and it is not one itself; the last element of an odd-sized tensor pairs with element
            // 0 (torch.roll wraps)
            const bool me = fabs(q0) > 32.0;
            bool mo = fabs(q1) > 32.0;
            if (!has1) mo = fabs(f64_nearest(pa, L, x[0] / (alpha[0] / gmax))) > 32.0;
            const bool ve = has1 ? (mo && !me) : mo;
            q0 = q0 * (ve ? 0.0 : 1.0);
            q1 = q1 * (me ? 0.0 : 1.0);
        }
        out[i0] = ((q0 - d0) + d0) * s0;
        if (has1) out[i1] = ((q1 - d1) + d1) * s1;
    }
}

}  // namespace antq

#endif  // ANTQ_K_F64_H
