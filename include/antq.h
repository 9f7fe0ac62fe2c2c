/*
 * antq.h -- C ABI of libantq.so: MI355X (gfx950) fake-quant hot path of ANT / OliVe.
 *
 * This is the drop-in boundary for the reference's ONLY native operator,
 *     quant_cuda.quant(x, grid) -> (z, idx)
 *         ant_quantization/quant/quant.cpp:17-29        (pybind entry)
 *         ant_quantization/quant/quant_kernel.cu:11-62  (kernel + launcher)
 * plus fused entry points that replace the PyTorch op sequences the reference
 * runs around that operator on every forward:
 *     Quantizer._forward            ant_quantization/antquant/quant_modules.py:535-551
 *     Quantizer._forward (OliVe)    olive_quantization/antquant/quant_modules.py:294-330
 *     AsymmetricQuantFunction       ant_quantization/antquant/quant_affine.py:95-115
 *     mse_loss / search_mse         ant_quantization/antquant/quant_modules.py:280-326
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary;
 *   - every *_dev pointer is device (HBM) memory owned by the caller; the library
 *     never allocates, frees or synchronises; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream);
 *   - return 0 on success, a negative ANTQ_ERR_* otherwise; no exceptions;
 *   - re-entrant and thread-safe: no process-global mutable state (the antq_debug_set development knobs are
 *     thread-local: they only steer the calling thread's own later calls);
 *     nothing here allocates or synchronises, so every entry point can be captured into a hipGraph;
 *   - results: grid index bit-exact with the reference scan; dequantised floats
 *     bit-identical to the reference's fp32 op sequence (bf16/f16 outputs are the
 *     fp32 result rounded to nearest-even).
 */
#ifndef ANTQ_H
#define ANTQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): antq_search_sse / antq_alpha_grad take a caller workspace before `stream`, antq_absmax initialises its
 * output, the batch blob changed (plan version 8), ANTQ_FLAG_UNORDERED.  3: antq_calibrate / antq_calibrate_workspace_bytes
 * added (nothing else changed).  4 (round 4): the plan blob grew (version 9: 128-byte header + the threshold list of the
 * 16-bit-domain kernels; ANTQ_PLAN_MAX_BYTES with it), the batch blob changed, antq_plan_eval_host_h and
 * antq_prefetch_kernels added.  5: antq_calibrate_batch / antq_calibrate_batch_workspace_bytes and antq_absmax_into added (nothing else changed).
 * 7 (round 6): antq_absmax_t / antq_alpha_grad_t (whole-tensor reductions in ONE launch through a caller-owned ticket block,
 * ANTQ_REDUCE_WS_BYTES) and antq_calibrate_install added (nothing else changed).
 * A caller built against another version must not call in: the blobs / argument lists differ. */
#define ANTQ_ABI_VERSION 7

/* element types of x / out */
#define ANTQ_F32  0
#define ANTQ_BF16 1
#define ANTQ_F16  2
#define ANTQ_F64  3   /* antq_nearest only (AT_DISPATCH_FLOATING_TYPES: float, double) */

/* error codes */
#define ANTQ_OK               0
#define ANTQ_ERR_ARG        (-1)   /* null pointer, bad size, bad enum               */
#define ANTQ_ERR_UNSUPPORTED (-2)  /* dtype / grid size not supported by this entry  */
#define ANTQ_ERR_PLAN       (-3)   /* plan blob malformed or too small               */
#define ANTQ_ERR_LAUNCH     (-4)   /* hipLaunchKernel failed (hipGetLastError)       */
#define ANTQ_ERR_ALIGN      (-5)   /* pointer not aligned to the element size        */

/* flags of antq_fakequant* */
#define ANTQ_FLAG_OVP        1u    /* OliVe outlier-victim pair masking (OQ:311-320) */
#define ANTQ_FLAG_DYNAMIC    2u    /* antq_batch_build only: alpha computed in the kernel */
#define ANTQ_FLAG_UNORDERED  4u    /* antq_fakequant only: this launch does not depend on the launches queued before it on
                                    * the stream (its inputs are at rest: calibrated weights), so it may START while they are
                                    * still draining -- the dispatch packet goes out without the barrier bit
                                    * (hipExtAnyOrderLaunch).  Ordinary launches queued after it still wait for it.  Worth
                                    * 2-3 us per 33.5 MB tensor when many weight tensors are quantised one launch each.
                                    * The OUTPUT buffer must be at rest too: not one a stream-ordered allocator has just
                                    * recycled from a kernel that may still be running (torch's caching allocator does that:
                                    * the Python binding refuses the flag without a caller-owned `out`).
                                    * A HINT: honoured by the table / lane kernels that serve 4-bit codebooks (the cases it was
                                    * measured on); the uniform-grid, scalar and literal-scan paths launch ordered whatever
                                    * the flag says.  Not with an index output: ANTQ_ERR_ARG. */

/* values written to the optional int16 index output */
#define ANTQ_IDX_NONE    (-1)      /* no grid entry within 102400 (NaN/Inf/huge)     */
#define ANTQ_IDX_VICTIM  (-2)      /* OliVe victim: value forced to zero             */

#define ANTQ_MAX_GRID     1024     /* entries; the reference's LDS array holds 256   */
#define ANTQ_PLAN_MAX_BYTES (128 + 4 * ANTQ_MAX_GRID + 16 * 3072 + 20 * 1024 + 16 * 64)

int         antq_abi_version(void);
const char *antq_strerror(int code);

/* ---------------------------------------------------------------------------
 * quant_cuda.quant replacement (ant_quantization/quant/quant_kernel.cu:11-62).
 *   z[i] = grid[j*],  j* = last j minimising fl32|fl32(x[i]) - fl32(grid[j])|,
 *   z[i] = 0 when no distance is <= 102400 (NaN / Inf / huge inputs).
 * x, z: n elements of `dtype` (F32, F64, BF16, F16).  grid_dev: m entries of the
 * SAME dtype as x for F32/F64 (the reference casts grid.type_as(x)); float for
 * BF16/F16.  idx_dev (nullable): int16 j* per element (the reference allocates an
 * index tensor but never writes it; see quant_kernel.cu:18,49).
 * ------------------------------------------------------------------------- */
int antq_nearest(const void *x_dev, void *z_dev, int16_t *idx_dev, size_t n,
                 const void *grid_dev, int m, int dtype, void *stream);

/* The same operator for a grid the caller knows on the host (a quantiser's static codebook: build the plan once, see
 * below): one table lookup per element instead of the m-step scan -- same z, same j*.  x, z: n elements of F32 / BF16 /
 * F16, 16-byte aligned, n a multiple of 4 (F32) or 8; anything else: ANTQ_ERR_UNSUPPORTED, use antq_nearest. */
int antq_nearest_plan(const void *x_dev, void *z_dev, int16_t *idx_dev, size_t n,
                      const void *plan_host, const void *plan_dev, int dtype, void *stream);

/* quant_cuda.quant (quant_kernel.cu:11-62) with a plan as a HINT.  Contract = antq_nearest on the device array
 * grid_dev (m floats; x of F32 / BF16 / F16): the plan is only what the host believes grid_dev holds (built from an
 * earlier read-back of the same buffer).  Every workgroup compares the m device values with the plan's own copy of the
 * grid bit for bit; if they differ it runs the literal scan on the device values -- a stale plan costs time, never a
 * wrong result -- and stores 1 to *stale (nullable; device-accessible memory, e.g. a pinned host int the caller polls
 * without synchronising).  m must equal the plan's grid size (ANTQ_ERR_ARG otherwise); size / alignment rules of
 * antq_nearest_plan. */
int antq_nearest_hinted(const void *x_dev, void *z_dev, int16_t *idx_dev, size_t n,
                        const float *grid_dev, int m, const void *plan_host, const void *plan_dev,
                        int *stale, int dtype, void *stream);

/* ---------------------------------------------------------------------------
 * Plan: host-side, exact pre-computation of the decision thresholds of the
 * reference scan for one grid (pure CPU, no HIP calls).  The caller keeps the
 * host blob and uploads a copy to the device; both are passed to the fused
 * entry points.  Returns the number of bytes written (>0) or ANTQ_ERR_*.
 *   grid_host : m floats in the order the reference would pass them to the
 *               kernel (ANT: sorted quant_grid; OliVe: cat(quant_grid, outliers)).
 * antq_plan_kind(): 1 = table plan (fast path), 0 = scan plan (the grid is not
 * representable as a table; the kernels fall back to the literal scan).
 * ------------------------------------------------------------------------- */
int antq_plan_build(const float *grid_host, int m, void *plan_host, size_t plan_capacity);
int antq_plan_kind(const void *plan_host);
int antq_plan_bytes(const void *plan_host);

/* Host model of the device element path for one plan (pure CPU): q[i], idx[i] for the
 * grid-domain inputs d[i].  The kernels implement exactly this function; the CPU test
 * suite uses it to check a plan against the literal scan without a GPU. */
int antq_plan_eval_host(const void *plan_host, const float *d, float *q, int16_t *idx, size_t n);

/* Host model of the approximate-quotient element path the small-group / big-table kernels run for plans that allow it
 * (pure CPU): out[i] = fake-quant of x[i] at scale alpha / gmax, idx[i] its grid index, slow[i] (nullable) = 1 where the
 * margin test sent the element to the exact sequence.  rs_ulps: offset of the modelled reciprocal from RN(1 / scale) in
 * ulps (the device's v_rcp_f32 is specified to 1 ulp; results must not depend on it).  ANTQ_ERR_UNSUPPORTED for plans
 * without that path. */
int antq_plan_eval_host_a(const void *plan_host, const float *x, size_t n, float alpha, float gmax, int rs_ulps,
                          float *out, int16_t *idx, uint8_t *slow);

/* Host model of the 16-bit-domain row path (bf16 / f16 rows of at least 128 vectors with a 4- / 5-bit codebook: the
 * headline kernels; pure CPU): one row of n 16-bit patterns x16 at scale alpha / gmax -> the output patterns out16,
 * through the same per-row slot table, sentinel slot, far-clipped arithmetic and literal sequence the kernel uses.
 * path[i] (nullable): 0 = table, 1 = far-clipped arithmetic, 2 = literal sequence.  flags: ANTQ_FLAG_OVP (pairs inside
 * the row: n even).  ANTQ_ERR_UNSUPPORTED when the plan does not allow that path for `dtype` (ANTQ_BF16 / ANTQ_F16). */
int antq_plan_eval_host_h(const void *plan_host, const uint16_t *x16, size_t n, float alpha, float gmax, int dtype,
                          unsigned flags, uint16_t *out16, uint8_t *path);

/* ---------------------------------------------------------------------------
 * Fused Quantizer._forward with a calibrated (static) alpha:
 *     scale = alpha / gmax ; d = x / scale ; q = nearest(d) ; [OVP] ;
 *     out = ((q - d) + d) * scale
 * x/out: [rows, row_len] row-major, dtype F32 / BF16 / F16 (in place allowed).
 * alpha_dev: `rows` floats when alpha_per_row != 0 (is_perchannel), else 1.
 * Group-G quantisation = rows := numel/G, row_len := G on the same buffer.
 * gmax: max of the NORMAL grid (OliVe: without outliers, OQ:296).
 * idx_dev nullable.  flags: ANTQ_FLAG_OVP.
 * ------------------------------------------------------------------------- */
int antq_fakequant(const void *x_dev, void *out_dev, int16_t *idx_dev,
                   size_t rows, size_t row_len,
                   const float *alpha_dev, int alpha_per_row, float gmax,
                   const void *plan_host, const void *plan_dev,
                   unsigned flags, int dtype, void *stream);

/* ---------------------------------------------------------------------------
 * The same for a float64 tensor (a `.double()` model).  The reference's operator narrows to float INSIDE the kernel
 * (quant_kernel.cu:51, :28) while the tensor ops around it run in double: scale = alpha / gmax, d = x / scale,
 * q = grid[nearest(float(d))], [OVP], out = ((q - d) + d) * scale, all in double -- fused here (one read, one write).
 * alpha_dev: doubles (a double model's alpha Parameter is double).  flags: ANTQ_FLAG_OVP only.  8-byte aligned pointers.
 * Bit-identical to that op sequence; not a throughput path (16 B per element, a thread per flat pair).
 * ------------------------------------------------------------------------- */
int antq_fakequant_f64(const double *x_dev, double *out_dev, size_t rows, size_t row_len,
                       const double *alpha_dev, int alpha_per_row, double gmax,
                       const void *plan_host, const void *plan_dev, unsigned flags, void *stream);

/* ---------------------------------------------------------------------------
 * Same, with alpha computed in the kernel from the data (the initial alpha of
 * _init_quant_para, AQ:473-477, times one clip ratio as in search_mse AQ:300):
 *     alpha[r] = fl32(max_c |x[r,c]| * ratio)
 * One quant group (row) per wavefront / workgroup, single read of x.
 * alpha_out_dev (nullable): receives the `rows` alphas.  Per-row only.
 * ------------------------------------------------------------------------- */
int antq_fakequant_dynamic(const void *x_dev, void *out_dev, int16_t *idx_dev,
                           float *alpha_out_dev, size_t rows, size_t row_len,
                           float ratio, float gmax,
                           const void *plan_host, const void *plan_dev,
                           unsigned flags, int dtype, void *stream);

/* ---------------------------------------------------------------------------
 * Row abs-max (AQ:289,474 / per-tensor AQ:308,477):  amax[r] = max_c |x[r,c]|.
 * per_row == 0: one value for the whole tensor (workgroup maxima combined with an
 * integer atomicMax on the float's bits: order-independent, so bit-reproducible).
 * amax_dev need not be initialised in either mode.
 * ------------------------------------------------------------------------- */
int antq_absmax(const void *x_dev, float *amax_dev, size_t rows, size_t row_len,
                int per_row, int dtype, void *stream);

/* The whole-tensor abs-max as ONE launch: *amax_dev = max(*amax_dev, max_i |x[i]|) over n elements.  The caller provides
 * the initial value -- 0 for a fresh maximum (e.g. a slot of a buffer zeroed once for many calls: antq_absmax has to spend
 * a launch on that zero, 10.8 -> 7.6 us for a 4096 x 4096 bf16 tensor), or the running maximum of the blocks seen so far
 * (a tensor that arrives in pieces, a row-sharded quantiser's local blocks).  Same combination rule as antq_absmax
 * (integer atomicMax on the float's bits: order-independent, NaN wins). */
int antq_absmax_into(const void *x_dev, float *amax_dev, size_t n, int dtype, void *stream);

/* Whole-tensor reductions in ONE launch with nothing to zero per call (ABI 7).  reduce_ws_dev: ANTQ_REDUCE_WS_BYTES of
 * device memory owned by the caller, 16-byte aligned, ZEROED ONCE after allocation (hipMemsetAsync); every call leaves it
 * zeroed again, so calls issued one after the other on ONE stream share one block -- two streams never do.  Inside:
 * last-arriver tickets, one counter per group of workgroups (no more than 32 atomics ever queue on one address) and the
 * workgroup / group partials, folded in index order: results do not depend on which workgroup finishes last.
 *   antq_absmax_t     : *amax_dev = max_i |x[i]| over n elements, WRITTEN (not accumulated; amax_dev need not be
 *                       initialised).  Same value as antq_absmax(per_row = 0), NaN wins.  A 33.5 MB bf16 tensor: 9.3 us
 *                       (antq_absmax: zeroing launch + 256 atomics on one address) -> see profiles/r06_aux_kernels.log.
 *   antq_alpha_grad_t : *gsum_dev = sum_i fl32(gout[i] * fl32(out[i] - x[i])) in double, the per-tensor form of
 *                       antq_alpha_grad as one launch (that entry point: a second launch adds the partials).  Sums are
 *                       formed in one fixed tree -- bit-reproducible, but not the tree of antq_alpha_grad: the two may
 *                       differ in the last bits of the double. */
#define ANTQ_REDUCE_WS_BYTES 65536
int antq_absmax_t(const void *x_dev, float *amax_dev, size_t n, int dtype, void *reduce_ws_dev, void *stream);
int antq_alpha_grad_t(const void *x_dev, const void *out_dev, const void *gout_dev, size_t n, double *gsum_dev, int dtype,
                      void *reduce_ws_dev, void *stream);

/* ---------------------------------------------------------------------------
 * Backward of antq_fakequant with respect to alpha (QAT: alpha is a Parameter, AQ:39; the autograd graph of
 * AQ:535-551 gives d out / d alpha = (q - d) / max(grid) = (out - x) / alpha and d out / d x = 1):
 *     gsum[r] = sum_c fl32( gout[r,c] * fl32(out[r,c] - x[r,c]) )     fp32 terms, fp64 accumulation
 * The caller divides by alpha[r].  x / out / gout: [rows, row_len] of `dtype` (F32 / BF16 / F16); gsum_dev: `rows`
 * doubles (alpha_per_row) or 1, not initialised.  The per-tensor sum is formed from workgroup partials in
 * workspace_dev (antq_search_workspace_bytes() bytes of scratch, as for antq_search_sse; may be NULL per row) in a
 * fixed order: no floating-point atomics, the same bits on every run.
 * ------------------------------------------------------------------------- */
int antq_alpha_grad(const void *x_dev, const void *out_dev, const void *gout_dev, size_t rows, size_t row_len,
                    int alpha_per_row, double *gsum_dev, void *workspace_dev, int dtype, void *stream);

/* ---------------------------------------------------------------------------
 * mse_loss of a fake-quantised tensor against its source without materialising
 * it (AQ:280-285 applied to AQ:535-551): for every candidate c in [0,ncand)
 *     alpha_c[r] = fl32(xmax[r] * ratios[c])
 *     sse[c, r]  = sum_col ( fl32|fakequant(x; alpha_c)[r,col] - x[r,col]| )^2
 * accumulated in fp32 per lane and fp64 across lanes.  The caller divides by
 * row_len and picks the arg-min (strict '<', first best), as search_mse does.
 * sse_dev: [ncand, rows] doubles (per_row) or [ncand] (per tensor); it need not
 * be initialised.  ratios_dev: ncand floats.  xmax_dev: rows floats or 1.
 * Every sum is formed in one fixed order (no floating-point atomics): the result
 * is bit-identical from run to run, so near-tied candidates resolve the same way
 * every time and on every rank.  A sum over the whole tensor (per_row == 0, or a
 * single row) is formed from workgroup partials kept in workspace_dev:
 * antq_search_workspace_bytes() bytes of device memory owned by the caller, not
 * shared with a call running concurrently on another stream; contents are
 * scratch (no initialisation).  May be NULL for per_row with rows > 1.
 * A bf16 / f16 tensor with ONE scale (per_row == 0 or rows == 1) and at least 2^19 elements (2^20 with ANTQ_FLAG_OVP) is scored on its 65 536-bin
 * histogram (one pass counts the bit patterns, then every pattern with a non-zero count goes through the literal reference
 * sequence for every candidate: count x term, summed in double in one fixed order): the same terms as the element-by-element
 * kernels, added in another order (relative 1e-7 on a sum).  With ANTQ_FLAG_OVP the pass also lists the pairs that hold an
 * outlier-capable element and the scoring corrects the victims' terms from that list; a tensor with too many such pairs
 * (> ~8 %) is searched by the element-by-element kernels instead (decided on the device, no synchronisation).  The workspace
 * holds the slabs and the list: always size it with antq_search_workspace_bytes() (48.3 MiB since ABI 6; 8 MiB before).
 * Per-row searches on rows of >= 128 elements (>= 256 with ANTQ_FLAG_OVP) of codebooks with a threshold list, and fp32
 * tensors with ONE scale from 2^20 elements, are scored from the SORTED row / 4096-element chunk (round 6): counts and sums
 * of the elements beyond every x-domain threshold by binary search into integer prefix sums, the squared error in closed
 * form in double -- the terms are not rounded to fp32 one by one, so these sums lie within 1e-14 of the exact float64 sum of
 * the reference's per-element outputs where the element-by-element kernels lie within 2e-7 (the reference's own fp32
 * reduction: 3e-7).  Same arguments, same workspace, bit-identical from run to run.
 * ------------------------------------------------------------------------- */
size_t antq_search_workspace_bytes(void);
int antq_search_sse(const void *x_dev, size_t rows, size_t row_len,
                    const float *xmax_dev, int per_row,
                    const float *ratios_dev, int ncand, float gmax,
                    const void *plan_host, const void *plan_dev,
                    unsigned flags, int dtype, double *sse_dev, void *workspace_dev, void *stream);

/* The type selection (search_adaptive_numeric_type, AQ:328-415 / OQ:235-256) on ONE read of the tensor: the sums of
 * antq_search_sse for `ntypes` (<= 4) codebooks at once,
 *     sse[t, c, r]  = sum_col ( fl32|fakequant_t(x; xmax[r] * ratios[c])[r,col] - x[r,col]| )^2
 * sse_dev: [ntypes, ncand, rows] doubles (per_row) or [ntypes, ncand], not initialised; workspace_dev as for
 * antq_search_sse (same order-fixed sums).  gmax_host, plan_host, plan_dev: host arrays of ntypes entries.  Every plan must have the x-domain path (all ANT / OliVe codebooks up to 128
 * buckets) and rows must be whole 16-byte vectors, at least 64 of them (1 KiB): otherwise ANTQ_ERR_UNSUPPORTED and the
 * caller issues one antq_search_sse per type. */
int antq_search_sse_multi(const void *x_dev, size_t rows, size_t row_len, const float *xmax_dev, int per_row,
                          const float *ratios_dev, int ncand, int ntypes, const float *gmax_host,
                          const void *const *plan_host, const void *const *plan_dev,
                          unsigned flags, int dtype, double *sse_dev, void *workspace_dev, void *stream);

/* The selection step of search_mse on the device (AQ:299-306 / :317-324): for every row r
 *     score_c = fl32(sse[c, r] / row_len), c ascending; best starts at 1e10 and is replaced on a strict `<`
 *     best_alpha[r] = fl32(xmax[r] * ratios[c*]) of the first best candidate, xmax[r] if none qualifies.
 * na = rows (per-channel) or 1.  Keeps the whole calibration free of device->host syncs. */
int antq_search_pick(const double *sse_dev, const float *xmax_dev, const float *ratios_dev, int ncand,
                     size_t na, size_t row_len, float *best_score_dev, float *best_alpha_dev, void *stream);

/* ---------------------------------------------------------------------------
 * The whole first-call calibration of one quantiser in ONE call, stream-ordered, without a device->host sync
 * (Quantizer._init_quant_para's search: search_mse AQ:287-326 per candidate type, the type choice of
 * search_adaptive_numeric_type AQ:328-415; OliVe OQ:189-256) -- the composition of the entry points above that
 * ant/quant_modules.py and olive/quant_modules.py perform step by step:
 *   x_max      ANTQ_XMAX_ABSMAX: row / tensor abs-max (ANT, AQ:289 / :308) written to xmax_dev;
 *              ANTQ_XMAX_3SIGMA: max(|mean + 3 std|, |mean - 3 std|) (OliVe, OQ:193-197 / :213-218) written to xmax_dev;
 *              ANTQ_XMAX_GIVEN : xmax_dev is an input (a statistic all-reduced over ranks, `no_outlier`, ...)
 *   candidates i = lb, lb + step, ... < ub:  alpha_i[r] = fl32(xmax[r] * fl32(i * 0.01))   (ANT: step 1; OliVe: step 2)
 *   per type t (ntypes codebooks: plan_host / plan_dev / gmax_host arrays), per row r (alpha_per_row) or for the tensor:
 *              alpha_dev[t * na + r] = the candidate with the smallest mean squared error, the first one on ties, x_max
 *              when the candidate list is empty or no error is below 1e10 (AQ:299-306)
 *   score_dev[t] = sum over rows of the best mean squared error (what search_mse returns, AQ:326), summed in double in
 *              one fixed order and rounded to float
 *   type_dev[0]  = the t with the smallest score, the first on ties, NaN scores last (np.argsort(mse)[0], AQ:413-415)
 * na = rows (alpha_per_row) or 1.  flags: ANTQ_FLAG_OVP searches with OliVe's outlier-victim pairs applied.
 * workspace_dev: antq_calibrate_workspace_bytes(...) bytes of caller-owned scratch, 16-byte aligned, not shared with a
 * call running concurrently on another stream.  The host decides lb / ub (ANT: lb = 95 when bit > 6, AQ:291-292) and which
 * codebooks are candidates (the `-float1..4` quirk of AQ:370-397 included: pass float_value(1)'s plan for each).
 * ------------------------------------------------------------------------- */
#define ANTQ_XMAX_GIVEN  0
#define ANTQ_XMAX_ABSMAX 1
#define ANTQ_XMAX_3SIGMA 2
size_t antq_calibrate_workspace_bytes(size_t rows, int alpha_per_row, int lb, int ub, int step, int ntypes);
int antq_calibrate(const void *x_dev, size_t rows, size_t row_len, int alpha_per_row, int dtype,
                   int xmax_mode, float *xmax_dev, int lb, int ub, int step,
                   int ntypes, const float *gmax_host, const void *const *plan_host, const void *const *plan_dev,
                   unsigned flags, float *alpha_dev, float *score_dev, int32_t *type_dev,
                   void *workspace_dev, size_t workspace_bytes, void *stream);

/* What _init_quant_para leaves behind (AQ:468-533, OQ:258-292) for a quantiser with ONE scale whose type was picked ON THE
 * DEVICE by antq_calibrate -- without the host learning the pick (ABI 7).  With t = *type_dev, stream-ordered, two launches:
 *   alpha_out[0] = alpha_dev[t];  mse_out[0] = score_dev[t] (the reference's log value, AQ:519-520);
 *   grid_out[0 .. grid_len) = grids_dev[t][..] (the candidates' codebooks, padded to one length, stacked by the caller;
 *   float32);  outl_out likewise from outl_dev (OliVe; NULL: none);
 *   out = the steady-state _forward of x with codebook t at alpha_dev[t] (AQ:535-551 / OQ:294-330; ANTQ_FLAG_OVP: pairs) --
 *   the calibrating call's own output, one pass over the tensor instead of one per candidate and a gather.
 * The host needs the pick only to NAME the mode (it copies *type_dev to pinned memory and reads it when the model's forward
 * has returned).  x: n elements, 16-byte aligned, n a multiple of the vector's element count (4 fp32 / 8 16-bit), otherwise
 * ANTQ_ERR_UNSUPPORTED (the caller quantises with every candidate and gathers, as before).  ntypes <= 4. */
int antq_calibrate_install(const void *x_dev, void *out_dev, size_t n, int dtype, int ntypes, const float *gmax_host,
                           const void *const *plan_host, const void *const *plan_dev, unsigned flags, const int32_t *type_dev,
                           const float *alpha_dev, const float *score_dev, const float *grids_dev, int grid_len, float *grid_out,
                           const float *outl_dev, int outl_len, float *outl_out, float *alpha_out, float *mse_out, void *stream);

/* The calibration of MANY quantisers in one call: job i is exactly antq_calibrate(jobs[i]...) -- same kernels, same order,
 * same bits -- all enqueued on `stream` one after the other through ONE workspace (antq_calibrate_batch_workspace_bytes: the
 * largest job's need; the jobs are stream-ordered, so they can share it).  The weight quantisers of a model do not depend
 * on any activation: a model calibrates all of them with this call before its first layer runs, reads the n type picks
 * back in ONE copy, and the per-layer host work and read-backs of the reference's first forward (one `if cuda_tensor:` per
 * quantiser, AQ:470, and np.argsort(...cpu()) per type selection, AQ:413) are gone for them.  dtype / flags are common to
 * the batch (a model's weights share a dtype; ANT or OliVe decides the pair rule).  Stops at the first job that fails and
 * returns its error; jobs before it are enqueued. */
typedef struct antq_calib_job {
    const void *x_dev;            /* rows x row_len elements */
    size_t rows, row_len;
    int alpha_per_row;
    int xmax_mode;                /* ANTQ_XMAX_* */
    float *xmax_dev;              /* na floats: output (ABSMAX / 3SIGMA) or input (GIVEN) */
    int lb, ub, step;
    int ntypes;
    const float *gmax_host;       /* ntypes */
    const void *const *plan_host; /* ntypes */
    const void *const *plan_dev;  /* ntypes */
    float *alpha_dev;             /* ntypes x na */
    float *score_dev;             /* ntypes */
    int32_t *type_dev;            /* 1 */
} antq_calib_job;
size_t antq_calibrate_batch_workspace_bytes(const antq_calib_job *jobs, int n);     /* 0: a job has a bad range / type count */
int antq_calibrate_batch(const antq_calib_job *jobs, int n, int dtype, unsigned flags, void *workspace_dev,
                         size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * AsymmetricQuantFunction.forward (ant_quantization/antquant/quant_affine.py:95-115)
 *     scale = (2^k-1)/clamp(max-min,1e-8); zp = round(scale*min) + 2^(k-1)
 *     q = clamp(round(scale*x - zp), -2^(k-1), 2^(k-1)-1); out = (q+zp)/scale
 * xmin_dev/xmax_dev: `rows` floats (per_row) or 1.  q_dev nullable (int32).
 * F32 only (the reference path is fp32 PyTorch).
 * ------------------------------------------------------------------------- */
int antq_affine(const float *x_dev, float *out_dev, int32_t *q_dev,
                size_t rows, size_t row_len, int k,
                const float *xmin_dev, const float *xmax_dev, int per_row,
                void *stream);

/* ---------------------------------------------------------------------------
 * Batched launch: many independent fake-quant jobs (e.g. the 54 weight tensors of a ResNet-50,
 * SURVEY 8a C1) in ONE kernel launch.  Small tensors are launch-bound when issued one by one
 * (~6 us of host time each vs < 1 us of HBM time); here every workgroup looks up its job in a
 * descriptor table that the caller builds once (weights and alphas are static between
 * forwards) and keeps resident on the device.
 *   antq_batch_build : pure host code; writes the descriptor blob (returns its size in bytes).  All jobs share dtype
 *                      and flags.  Jobs whose rows are not a multiple of 16 bytes (conv1: K = 147) or whose
 *                      buffers are not 16-byte aligned run element-granular inside the same launch.
 *                      With ANTQ_FLAG_DYNAMIC every job's alpha_dev is an OUTPUT (NULL: the scales are not stored):
 *                      alpha[r] = max_c |x[r,c]| is computed in the kernel from the registers holding the group / row
 *                      (antq_fakequant_dynamic with ratio 1, many tensors, one launch); per row only, groups of a
 *                      power of two of 16-byte vectors up to 64, or rows of 128 .. 8192 vectors (beyond 512 the plan
 *                      must be x-domain eligible: every ANT / OliVe 4-bit codebook), otherwise ANTQ_ERR_UNSUPPORTED.
 *   antq_fakequant_batch : one launch for all jobs; batch_dev is the caller's device copy.
 * ------------------------------------------------------------------------- */
typedef struct antq_job {
    const void  *x_dev;
    void        *out_dev;
    const float *alpha_dev;
    size_t       rows, row_len;
    int          alpha_per_row;
    float        gmax;
    const void  *plan_host;
    const void  *plan_dev;
} antq_job;

size_t antq_batch_capacity(const antq_job *jobs, int n, int dtype);   /* bytes antq_batch_build needs */
int antq_batch_build(const antq_job *jobs, int n, int dtype, unsigned flags, void *batch_host, size_t capacity);
int antq_fakequant_batch(const void *batch_host, const void *batch_dev, void *stream);

/* ---------------------------------------------------------------------------
 * Packed 4-bit codec: the quantised tensor itself instead of its fake-quant image
 * (SURVEY 8f N4).  Two codes per byte, element 2k in the low nibble.
 *   code = scan-order grid index of the element (what antq_fakequant's idx output holds),
 *   m <= 16 for plain grids (ANT 4-bit: the 16-entry grid incl. its duplicate zero).
 * With ANTQ_FLAG_OVP the grid is OliVe's cat(normal[n_normal <= 15], outliers[<= 15]):
 *   normal value  -> its index 0..n_normal-1
 *   victim        -> 15, the outlier identifier (the paper's reserved code)
 *   outlier       -> index INTO THE OUTLIER LIST; the partner nibble of the pair is 15,
 *                    which is what tells the decoder to use the outlier codebook.
 * antq_decode4 writes fl(v * scale) for the decoded grid value v.  That is bit-identical to antq_fakequant(x) whenever
 * the reference's straight-through step ((q - d) + d) is exact, which holds for every ANT / OliVe 4-bit codebook
 * (adjacent magnitudes within a factor of two: the condition the plan builder records as its x-domain eligibility)
 * and finite inputs within twice the outermost grid values; for arbitrary value lists the two may differ by one ulp.
 * Elements outside the scan's validity range (code would be ANTQ_IDX_NONE) encode as the
 * grid entry holding 0.0.  row_len must be a multiple of 8; x: F32 / BF16 / F16; codes: rows*row_len/2 bytes.
 * (BF16 / F16 rows of at least 1024 elements are encoded in the tensor's own 16-bit domain: the per-row slot table of the
 *  headline kernels holding code bytes instead of output patterns, antq_k_codec.h: k_encode4_hrow.)
 * ------------------------------------------------------------------------- */
int antq_encode4(const void *x_dev, uint8_t *codes_dev, size_t rows, size_t row_len,
                 const float *alpha_dev, int alpha_per_row, float gmax,
                 const void *plan_host, const void *plan_dev, int n_normal,
                 unsigned flags, int dtype, void *stream);
int antq_decode4(const uint8_t *codes_dev, void *out_dev, size_t rows, size_t row_len,
                 const float *alpha_dev, int alpha_per_row, float gmax,
                 const void *plan_host, const void *plan_dev, int n_normal,
                 unsigned flags, int dtype, void *stream);

/* -------------------------------------------------------------------------
 * OliVe's clip statistic on ONE read (replaces t.mean() + t.std() of olive_quantization/antquant/quant_modules.py:193-197
 * and :213-218, which read the tensor at least three times before the clip search reads it again).
 * antq_moments: sums_dev[2 r] = sum of x, sums_dev[2 r + 1] = sum of x^2 over row r (alpha_per_row) or over the whole
 *   tensor (one pair), in double, formed in one fixed order (bit-reproducible; no atomics).  workspace_dev: required for
 *   the whole-tensor form, antq_search_workspace_bytes() bytes.  These are also the numbers a row-sharded per-tensor
 *   quantiser all-reduces across ranks (sum x, sum x^2, n).
 * antq_xmax_3sigma: xmax_dev[r] = max(|mean + 3 std|, |mean - 3 std|) from na pairs of sums over n_per elements each,
 *   unbiased std, with the roundings of `dtype` applied where the reference's tensor ops round (fp32: mean and std to
 *   float; bf16 / fp16: mean, std, 3 * std, sum and difference each rounded to the tensor's dtype).  Agrees with the
 *   reference's torch reductions to their own summation-order noise (relative 1e-6 in fp32).
 * ------------------------------------------------------------------------- */
int antq_moments(const void *x_dev, size_t rows, size_t row_len, int alpha_per_row, int dtype,
                 double *sums_dev, void *workspace_dev, void *stream);
int antq_xmax_3sigma(const double *sums_dev, size_t na, size_t n_per, int dtype, float *xmax_dev, void *stream);

/* Plain device copy with the fake-quant kernels' access pattern (16 B per lane):
 * used by bench.py to measure the empirical HBM ceiling on the same buffers. */
int antq_copy(const void *src_dev, void *dst_dev, size_t bytes, void *stream);

/* Development / benchmark tuning knobs (thread-local: only the calling thread's later calls are affected; not part of
 * the stable surface):
 *   key 0: force the vectors per lane and task (U) of the row kernels -- batched launch 1..4, row-table encoder 2 / 4 / 8,
 *          calibration kernels 4 / 8 (0 = heuristic)
 *   key 1: number of persistent workgroups of antq_encode4's element encoder (0 = default, 2048)
 *   key 2: 0 disables the per-row (x-domain) table kernels, the d-domain kernels run instead (A/B measurements)
 *   key 3: 0 disables the binary-search path of antq_nearest (literal scan only)
 *   key 4: 0 makes the element encoder divide exactly instead of using the approximate-quotient decision
 *   key 5: 0 sends long rows through the per-row table kernels, 2 through the lane kernel in batches too (1 = the measured
 *          default: lane kernel for one-launch-per-tensor calls and fp32 batches)
 *   key 6: wavefronts per workgroup (1 / 2 / 4) of the batched row-table launch and of the one-launch lane kernel (0 = heuristic)
 *   key 7: vectors per lane of the one-launch lane kernel (0 = heuristic)
 *   key 8: workgroup -> task map rotation of the batched row-table launch: 0 per job (default), 1 always, 2 never
 *   key 9: 0 disables the 16-bit-domain row kernels and the 16-bit-domain encoder (bf16 / f16 rows of >= 128 vectors go back to
 *          the fp32-domain row table / row encoder)
 *   key 10: dynamic LDS bytes per workgroup of those kernels (occupancy A/B; -1 = default: 24 workgroups per CU)
 *   key 11: extra dynamic LDS bytes per workgroup of the batched lane-job kernels (occupancy A/B; -1 = default: fp32 static
 *           launches of >= 32768 workgroups are held at 6 workgroups = 24 wavefronts per CU, nothing otherwise)
 *   key 12: clip search: number of candidate-list chunks (0 = cost model)
 *   key 14: the histogram clip search of 16-bit tensors with one scale (antq_search_sse / _multi / antq_calibrate, no pair
 *           rule, or OliVe's pairs through a list of the outlier-capable pairs): 0 off, 1 (default) when the tensor is large
 *           enough for it to pay, 2 for every eligible tensor, 3 as 1 but never with the pair rule
 *   key 15: antq_calibrate with the abs-max statistic on a tensor the histogram search takes: 1 (default) the counting pass
 *           also finds the maximum (no separate abs-max pass), 0 the abs-max pass runs as everywhere else (A/B; same value)
 *   key 13: experiment switch of the kernel under development (0 = off; 1: the 16-bit-domain encoder's 8-vector tasks store
 *           their codes nontemporally instead of through the cache)
 *   key 16-18: the streaming row abs-max (16) and the ticket groups / workgroups of the one-launch reductions (17, 18), A/B
 *   key 19: the threshold-sweep clip search (round 6, first generation): 0 off, 1 (default) where it paid, 2 every eligible
 *           launch -- runs only where key 20 leaves a launch to it
 *   key 20: the sorted-row clip search (round 6: per-row searches from rows of 128 elements, OliVe's pair rule from 576, fp32
 *           tensors with one scale from 1 M elements): 0 off, 1 (default), 2 every eligible launch (rows from 128 elements
 *           with or without the pair rule, one-scale fp32 tensors from 4096)
 *   key 21: 0 sends rows of <= 1024 elements through the 4096-key kernel instead of one row per wavefront (A/B) */
int antq_debug_set(int key, int value);

/* Load the library's GPU code objects for the current device now (HIP would load each of them at the first launch of one
 * of its kernels, ~50 ms in total, i.e. inside the first calibrating forward).  Launches nothing; idempotent. */
int antq_prefetch_kernels(void);

#ifdef __cplusplus
}
#endif
#endif /* ANTQ_H */
