/*
 * antq_oracle.c -- CPU restatement of the ANT / OliVe fake-quant hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link
 * or execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * Each function cites the reference lines it restates (paths relative to the
 * upstream repository):
 *   KQ  = ant_quantization/quant/            (identical copy in olive_quantization/quant/)
 *   AQ  = ant_quantization/antquant/
 *   OQ  = olive_quantization/antquant/
 *
 * Pinning: the reference CUDA kernel cannot be built in this image (no nvcc,
 * no NVIDIA device), so there is no oracle/_ref build.  The restatement of the
 * kernel below is literal (KQ/quant_kernel.cu:20-38); everything around it is
 * pinned by tests/golden/ fixtures that were produced by importing the
 * reference's own Python (AQ/quant_modules.py, OQ/quant_modules.py,
 * AQ/quant_affine.py) with THIS file's scan standing in for `quant_cuda.quant`
 * (tests/golden/make_golden.py).  The reference ships no tests or golden
 * vectors of its own for this path.
 *
 * All arithmetic is IEEE binary32 unless stated; build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math  (see oracle/Makefile)
 * so that no FMA contraction or reassociation changes a result.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ANTQ_ORACLE_IDX_NONE   (-1)  /* no grid entry within 102400 (or NaN/Inf input) */
#define ANTQ_ORACLE_IDX_VICTIM (-2)  /* OliVe victim: value forced to zero            */

/* ------------------------------------------------------------------ */
/* a1. KQ/quant_kernel.cu:25-37 -- one element of the scan.            */
/*   float sub_min = 102400.0; float z_min = 0.0;                      */
/*   for i in 0..y_size: sub_v = fabsf(x_v - y[i]);                    */
/*        if (sub_v <= sub_min) { sub_min = sub_v; z_min = y[i]; }     */
/* The reference never materialises the index; we return it as the     */
/* canonical "integer quant index" (SURVEY A.3): last minimum wins.    */
/* ------------------------------------------------------------------ */
static inline float scan_one(float x_v, const float *y, int m, int *j_out)
{
    float sub_min = 102400.0f;
    float z_min = 0.0f;
    int j = ANTQ_ORACLE_IDX_NONE;
    for (int i = 0; i < m; i++) {
        float sub_v = fabsf(x_v - y[i]);
        if (sub_v <= sub_min) {
            sub_min = sub_v;
            z_min = y[i];
            j = i;
        }
    }
    *j_out = j;
    return z_min;
}

/* a1/a2 for float tensors (AT_DISPATCH_FLOATING_TYPES, scalar_t=float).
 * idx may be NULL. */
void antq_oracle_nearest_f32(const float *x, float *z, int32_t *idx, size_t n,
                             const float *grid, int m)
{
    for (size_t i = 0; i < n; i++) {
        int j;
        z[i] = scan_one(x[i], grid, m, &j);
        if (idx) idx[i] = j;
    }
}

/* a1/a2 for double tensors (scalar_t=double): KQ/quant_kernel.cu:23 narrows
 * the grid to float in shared memory, :28 narrows x to float, :36 widens the
 * selected float back to double. */
void antq_oracle_nearest_f64(const double *x, double *z, int32_t *idx, size_t n,
                             const double *grid, int m)
{
    float y[1024];
    if (m > 1024) m = 1024;
    for (int i = 0; i < m; i++) y[i] = (float)grid[i];
    for (size_t i = 0; i < n; i++) {
        int j;
        z[i] = (double)scan_one((float)x[i], y, m, &j);
        if (idx) idx[i] = j;
    }
}

/* ------------------------------------------------------------------ */
/* bf16 helpers (extension: oracle for bf16 I/O is "reference on        */
/* x.float(), result cast to bf16").  Round-to-nearest-even   */
/* as torch's float->bfloat16 conversion (c10/util/BFloat16.h           */
/* round_to_nearest_even: NaN -> 0x7FC0).                               */
/* ------------------------------------------------------------------ */
static inline float bf16_to_f32(uint16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if (f != f) return 0x7FC0;
    uint32_t rounding_bias = ((u >> 16) & 1u) + 0x7FFFu;
    return (uint16_t)((u + rounding_bias) >> 16);
}
void antq_oracle_bf16_to_f32(const uint16_t *in, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = bf16_to_f32(in[i]);
}
void antq_oracle_f32_to_bf16(const float *in, uint16_t *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = f32_to_bf16(in[i]);
}

/* ------------------------------------------------------------------ */
/* a4 / a5. Quantizer._forward                                          */
/*   ANT   AQ/quant_modules.py:535-551                                  */
/*   OliVe OQ/quant_modules.py:294-330                                  */
/*                                                                      */
/*   scale = alpha / max(quant_grid)           AQ:536  OQ:296           */
/*   data  = x / scale   (row-broadcast)       AQ:538-541 OQ:298-301    */
/*   quant = nearest(data, grid)               AQ:543  OQ:308           */
/*   [OliVe] outlier-victim pairs on the flat tensor   OQ:311-320       */
/*   tensor = (quant - data) + data            AQ:544  OQ:323           */
/*   out   = tensor * scale                    AQ:546-549 OQ:325-328    */
/*                                                                      */
/* x is [rows, row_len] row-major.  alpha has `rows` entries when       */
/* alpha_per_row != 0 (is_perchannel, alpha shape [C,1]) else 1 entry.  */
/* grid  : the array actually passed to the kernel (ANT: quant_grid;    */
/*         OliVe: cat(quant_grid, outliers) unless no_outlier).         */
/* gmax  : torch.max(quant_grid) -- of the NORMAL grid only (OQ:296).   */
/* ovp   : 0 = ANT / OliVe no_outlier, 1 = OliVe pair masking.          */
/* idx   : optional; grid index j* of each element, -1 if none,         */
/*         -2 for victims.                                              */
/* ------------------------------------------------------------------ */
static void forward_row_major(const float *x, float *out, int32_t *idx,
                              size_t rows, size_t row_len,
                              const float *alpha, int alpha_per_row,
                              const float *grid, int m, float gmax, int ovp)
{
    size_t n = rows * row_len;
    /* pass 1: data = x/scale, quant = nearest(data); keep both (the OVP mask
       is defined on the flat quant tensor, OQ:313). */
    float *data = out; /* reuse: out holds `data`, then is overwritten */
    float *quant = (float *)__builtin_malloc(n * sizeof(float) + 4);
    for (size_t r = 0; r < rows; r++) {
        float scale = (alpha_per_row ? alpha[r] : alpha[0]) / gmax;
        for (size_t c = 0; c < row_len; c++) {
            size_t i = r * row_len + c;
            float d = x[i] / scale;
            int j;
            quant[i] = scan_one(d, grid, m, &j);
            data[i] = d;
            if (idx) idx[i] = j;
        }
    }
    if (ovp && n > 0) {
        /* OQ:313-320, element-for-element:
         *   mask        = abs(q) > 32
         *   victim_odd  = roll(mask, 1);  victim_odd[::2]  = 0
         *   victim_even = roll(mask & ~victim_odd, -1); victim_even[1::2] = 0
         *   victim      = victim_even | victim_odd
         *   q           = q * (~victim)                                  */
        unsigned char *mask = (unsigned char *)__builtin_malloc(n);
        unsigned char *vodd = (unsigned char *)__builtin_malloc(n);
        unsigned char *tmp = (unsigned char *)__builtin_malloc(n);
        for (size_t i = 0; i < n; i++) mask[i] = fabsf(quant[i]) > 32.0f;
        for (size_t i = 0; i < n; i++) vodd[i] = mask[(i + n - 1) % n];
        for (size_t i = 0; i < n; i += 2) vodd[i] = 0;
        for (size_t i = 0; i < n; i++) tmp[i] = mask[i] & (unsigned char)!vodd[i];
        for (size_t i = 0; i < n; i++) {
            unsigned char veven = tmp[(i + 1) % n];
            if (i & 1) veven = 0;
            unsigned char victim = veven | vodd[i];
            /* float * bool: q * 1.0f or q * 0.0f (keeps the sign of zero) */
            quant[i] = quant[i] * (victim ? 0.0f : 1.0f);
            if (idx && victim) idx[i] = ANTQ_ORACLE_IDX_VICTIM;
        }
        __builtin_free(mask);
        __builtin_free(vodd);
        __builtin_free(tmp);
    }
    for (size_t r = 0; r < rows; r++) {
        float scale = (alpha_per_row ? alpha[r] : alpha[0]) / gmax;
        for (size_t c = 0; c < row_len; c++) {
            size_t i = r * row_len + c;
            float d = data[i];
            float t = (quant[i] - d) + d;
            out[i] = t * scale;
        }
    }
    __builtin_free(quant);
}

void antq_oracle_forward_f32(const float *x, float *out, int32_t *idx,
                             size_t rows, size_t row_len,
                             const float *alpha, int alpha_per_row,
                             const float *grid, int m, float gmax, int ovp)
{
    forward_row_major(x, out, idx, rows, row_len, alpha, alpha_per_row, grid, m, gmax, ovp);
}

/* bf16 I/O extension: reference on x.float(), result cast to bf16. */
void antq_oracle_forward_bf16(const uint16_t *x, uint16_t *out, int32_t *idx,
                              size_t rows, size_t row_len,
                              const float *alpha, int alpha_per_row,
                              const float *grid, int m, float gmax, int ovp)
{
    size_t n = rows * row_len;
    float *xf = (float *)__builtin_calloc(n + 1, sizeof(float));
    float *of = (float *)__builtin_malloc(n * sizeof(float) + 4);
    for (size_t i = 0; i < n; i++) xf[i] = bf16_to_f32(x[i]);
    forward_row_major(xf, of, idx, rows, row_len, alpha, alpha_per_row, grid, m, gmax, ovp);
    for (size_t i = 0; i < n; i++) out[i] = f32_to_bf16(of[i]);
    __builtin_free(xf);
    __builtin_free(of);
}

/* ------------------------------------------------------------------ */
/* Dynamic abs-max alpha (the initial alpha of _init_quant_para,        */
/* AQ/quant_modules.py:473-477, OQ:265-269):                            */
/*   per-channel: alpha[r] = max_c |x[r,c]| ; per-tensor: max |x|.      */
/* `ratio` reproduces search_mse's candidate alpha = x_max * fl32(r)    */
/* (AQ:300).  NaN propagates as in torch.max (any NaN -> NaN).          */
/* ------------------------------------------------------------------ */
void antq_oracle_absmax_f32(const float *x, float *alpha, size_t rows, size_t row_len,
                            int per_row, float ratio)
{
    if (per_row) {
        for (size_t r = 0; r < rows; r++) {
            float mx = 0.0f; int nan = 0;
            for (size_t c = 0; c < row_len; c++) {
                float a = fabsf(x[r * row_len + c]);
                if (a != a) nan = 1;
                if (a > mx) mx = a;
            }
            alpha[r] = nan ? NAN : mx * ratio;
        }
    } else {
        float mx = 0.0f; int nan = 0;
        for (size_t i = 0; i < rows * row_len; i++) {
            float a = fabsf(x[i]);
            if (a != a) nan = 1;
            if (a > mx) mx = a;
        }
        alpha[0] = nan ? NAN : mx * ratio;
    }
}

/* ------------------------------------------------------------------ */
/* a9. mse_loss  AQ/quant_modules.py:280-285, OQ:181-187                */
/*   per-channel: mean_c |q - x|^2 per row ; else mean over the tensor. */
/* torch evaluates (q-x).abs().pow(2) in fp32 per element and reduces   */
/* in fp32 with a vectorised multi-accumulator order that we do not     */
/* reproduce; we accumulate the fp32 per-element terms in double and    */
/* round once, so comparisons against torch carry rel. tolerance 1e-5   */
/* (SURVEY 8c "Third-party arithmetic").                                */
/* ------------------------------------------------------------------ */
void antq_oracle_mse_f32(const float *q, const float *x, float *mse,
                         size_t rows, size_t row_len, int per_row)
{
    if (per_row) {
        for (size_t r = 0; r < rows; r++) {
            double acc = 0.0;
            for (size_t c = 0; c < row_len; c++) {
                float d = fabsf(q[r * row_len + c] - x[r * row_len + c]);
                float p = d * d;
                acc += (double)p;
            }
            mse[r] = (float)(acc / (double)row_len);
        }
    } else {
        double acc = 0.0;
        size_t n = rows * row_len;
        for (size_t i = 0; i < n; i++) {
            float d = fabsf(q[i] - x[i]);
            float p = d * d;
            acc += (double)p;
        }
        mse[0] = (float)(acc / (double)n);
    }
}

/* ------------------------------------------------------------------ */
/* a10. search_mse  AQ/quant_modules.py:287-326 ; OliVe OQ:189-233      */
/*                                                                      */
/*   x_max   : [rows] (per_row) or [1]; the caller supplies it (ANT:    */
/*             row abs-max AQ:289,308; OliVe: 3-sigma rule OQ:193-197). */
/*   for i in range(lb, ub, step):                                      */
/*       new_alpha = x_max * fl32(i*0.01)        AQ:300,318             */
/*       q = _forward(x) ; score = mse_loss      AQ:302-304             */
/*       strict '<' keeps the earliest best      AQ:305-306,322-324     */
/*   returns best_score (per row / scalar), alpha.                      */
/* trace (optional) receives every candidate's score:                   */
/*   [ncand, rows] (per_row) or [ncand].                                */
/* ------------------------------------------------------------------ */
int antq_oracle_search_mse_f32(const float *x, size_t rows, size_t row_len, int per_row,
                               const float *x_max, int lb, int ub, int step,
                               const float *grid, int m, float gmax, int ovp,
                               float *best_score, float *best_alpha, float *trace)
{
    size_t na = per_row ? rows : 1;
    size_t n = rows * row_len;
    float *alpha = (float *)__builtin_malloc(na * sizeof(float));
    float *score = (float *)__builtin_malloc(na * sizeof(float));
    float *q = (float *)__builtin_malloc(n * sizeof(float) + 4);
    for (size_t r = 0; r < na; r++) {
        best_score[r] = 1e10f;
        best_alpha[r] = x_max[r];
    }
    int ncand = 0;
    for (int i = lb; i < ub; i += step) {
        float ratio = (float)((double)i * 0.01);
        for (size_t r = 0; r < na; r++) alpha[r] = x_max[r] * ratio;
        forward_row_major(x, q, NULL, rows, row_len, alpha, per_row, grid, m, gmax, ovp);
        antq_oracle_mse_f32(q, x, score, rows, row_len, per_row);
        for (size_t r = 0; r < na; r++) {
            if (trace) trace[(size_t)ncand * na + r] = score[r];
            if (score[r] < best_score[r]) {
                best_score[r] = score[r];
                best_alpha[r] = alpha[r];
            }
        }
        ncand++;
    }
    __builtin_free(alpha);
    __builtin_free(score);
    __builtin_free(q);
    return ncand;
}

/* ------------------------------------------------------------------ */
/* a14. AsymmetricQuantFunction.forward  AQ/quant_affine.py:95-115      */
/*   n     = 2^k - 1                                         :75        */
/*   scale = n / clamp(max - min, 1e-8)                      :76        */
/*           python-int / tensor = Tensor.__rtruediv__, i.e.            */
/*           reciprocal(range) * n  -> fl(fl(1/range) * n)              */
/*   zp    = round(scale * min) + 2^(k-1)                    :77-85     */
/*   q     = clamp(round(scale*x - zp), -2^(k-1), 2^(k-1)-1) :108-110   */
/*   out   = (q + zp) / scale                                :111-114   */
/* torch.round is round-half-to-even = nearbyintf under the default     */
/* rounding mode.  x_min/x_max have `rows` entries (per_row) or one.    */
/* ------------------------------------------------------------------ */
void antq_oracle_affine_f32(const float *x, float *out, int32_t *qout,
                            size_t rows, size_t row_len, int k,
                            const float *x_min, const float *x_max, int per_row)
{
    float nlev = (float)((1 << k) - 1);
    float half = (float)(1 << (k - 1));
    for (size_t r = 0; r < rows; r++) {
        float mn = per_row ? x_min[r] : x_min[0];
        float mx = per_row ? x_max[r] : x_max[0];
        float range = mx - mn;
        if (range < 1e-8f) range = 1e-8f;
        float scale = (1.0f / range) * nlev;
        float zp = nearbyintf(scale * mn);
        zp = zp + half;
        for (size_t c = 0; c < row_len; c++) {
            size_t i = r * row_len + c;
            float q = nearbyintf(scale * x[i] - zp);
            if (q < -half) q = -half;
            if (q > half - 1.0f) q = half - 1.0f;
            if (qout) qout[i] = (int32_t)q;
            out[i] = (q + zp) / scale;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Literal PyTorch-op-sequence variant of a4 used only as the timed CPU */
/* baseline ("kind": "port"): identical results to forward_row_major    */
/* for ovp==0, written so that OpenMP can split rows.                   */
/* ------------------------------------------------------------------ */
void antq_oracle_forward_rows_f32(const float *x, float *out, size_t row_begin, size_t row_end,
                                  size_t row_len, const float *alpha, int alpha_per_row,
                                  const float *grid, int m, float gmax)
{
    for (size_t r = row_begin; r < row_end; r++) {
        float scale = (alpha_per_row ? alpha[r] : alpha[0]) / gmax;
        for (size_t c = 0; c < row_len; c++) {
            size_t i = r * row_len + c;
            float d = x[i] / scale;
            int j;
            float q = scan_one(d, grid, m, &j);
            float t = (q - d) + d;
            out[i] = t * scale;
        }
    }
}

void antq_oracle_forward_rows_bf16(const uint16_t *x, uint16_t *out, size_t row_begin, size_t row_end,
                                   size_t row_len, const float *alpha, int alpha_per_row,
                                   const float *grid, int m, float gmax)
{
    for (size_t r = row_begin; r < row_end; r++) {
        float scale = (alpha_per_row ? alpha[r] : alpha[0]) / gmax;
        for (size_t c = 0; c < row_len; c++) {
            size_t i = r * row_len + c;
            float d = bf16_to_f32(x[i]) / scale;
            int j;
            float q = scan_one(d, grid, m, &j);
            float t = (q - d) + d;
            out[i] = f32_to_bf16(t * scale);
        }
    }
}

/* The timed CPU baseline of bench.py ("antq_cpu ... OpenMP, at 1 thread and at all host cores"): `reps`
 * sweeps of forward_rows_bf16 over all rows, rows split statically over `threads` OpenMP threads (0 = the runtime's
 * default).  Returns the number of threads that actually ran. */
#ifdef _OPENMP
#include <omp.h>
#endif
int antq_oracle_forward_omp_bf16(const uint16_t *x, uint16_t *out, size_t rows, size_t row_len, const float *alpha,
                                 int alpha_per_row, const float *grid, int m, float gmax, int reps, int threads)
{
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
        for (int rep = 0; rep < reps; rep++) {
#pragma omp for schedule(static) nowait
            for (long long r = 0; r < (long long)rows; r++)
                antq_oracle_forward_rows_bf16(x, out, (size_t)r, (size_t)r + 1, row_len, alpha, alpha_per_row, grid, m, gmax);
        }
    }
#else
    (void)threads;
    for (int rep = 0; rep < reps; rep++)
        antq_oracle_forward_rows_bf16(x, out, 0, rows, row_len, alpha, alpha_per_row, grid, m, gmax);
#endif
    return used;
}

int antq_oracle_abi_version(void) { return 1; }
