// antq_k_batch.h -- batched launch: descriptor table, block -> job map, the multi-tensor kernels
// Part of libantq's single device translation unit (antq_kernels.hip includes it); gfx950 only.
#ifndef ANTQ_K_BATCH_H
#define ANTQ_K_BATCH_H

#include "antq_k_fakequant.h"
#include "antq_k_hrow.h"

namespace antq {

constexpr uint32_t kBatchMagic = 0x33544E41u;  // "ANT3"
constexpr int kBatchU = 4;                      // vectors per lane per task (4 KiB per wavefront: best measured)

struct BatchDesc {   // 192 bytes, device-visible
    const uint4 *x;
    uint4 *out;
    const float *alpha;    // ANTQ_FLAG_DYNAMIC: an OUTPUT
    const uint4 *plan_tab;
    uint64_t n_vec;        // lane kind: number of 16-byte vectors
    uint32_t total_tasks;  // row kind: wavefront tasks
    uint32_t vpr;
    uint32_t tpr;
    int32_t vshift;
    uint32_t first_block;  // first block of this job INSIDE ITS FAMILY'S LAUNCH
    uint32_t kind;         // 0 = row-run per wavefront, d-domain table; 1 = per-lane scale (vpr < kRowKernelMinVpr);
                           // 2 = row-run per wavefront, x-domain table
                           // 3 = element-granular (ragged rows / unaligned buffers): n_vec = elements, vpr = row_len
                           // 4..7 = kind 2 with the alpha computed in the kernel (ANTQ_FLAG_DYNAMIC, k_fq_batch_dyn): the row in
                           //         one wavefront, 4 / 8 vectors per lane (<= 256 / <= 512 vectors: kinds 4 / 6), or spread
                           //         over the 4 wavefronts of the workgroup (<= 1024 / <= 2048: kinds 5 / 7; <= 512 with 2 vectors per lane: kind 12)
                           // 13 = 16-bit rows through the table in their own 16-bit domain (antq_k_hrow.h): tasks of u vectors per lane
                           // 14 / 15 / 16 = the same with the alpha computed in the kernel: the row in 1 / 4 / 16 wavefronts, u vectors per lane
    int32_t per_row;
    float gmax;
    PlanArgs pa;
    float vout;            // PlanHeader::vout (x-domain kinds)
    float ratio;           // ANTQ_FLAG_DYNAMIC: alpha = max|row| * ratio
    uint32_t u;            // x-domain row kinds (2, 4): vectors per lane and task, 2 / 3 / 4 -- the value that leaves the fewest
                           // idle lanes for this row length (rows of 128 vectors: 2; ResNet's 3x3 rows of 144 / 288 / 576: 3)
    uint32_t rot;          // x-domain rows, one- / two-wavefront workgroups: rotate the workgroup -> task map of every group of 8
                           // workgroups by the group's number.  Workgroup b runs on XCD b % 8; when a row's last task is a
                           // partial one and the tasks per row share a factor with 8, task fullness would line up with the XCD
                           // number (4096 x 2048 bf16 in 192-vector tasks: 71 % instead of 81 %).  Only set for such jobs:
                           // full tasks stream 1-4 points better on the fixed map (profiles/r03_batch_rotation.log)
    float vmin, vmax;      // the grid's extreme values (XArgs::vmin / vmax: what far-clipped elements quantise to)
    uint32_t pad_;
    double inv_gmax;       // 1.0 / (double)gmax (XArgs::inv_gmax: the row's scale without a division)
    const uint4 *tlist;    // kind 13: the plan's threshold list (HThr[h_nthr]) on the device
    uint32_t h_n;          // kind 13: thresholds | negative ones << 16
    uint32_t hshift;       // kind 13: key = magnitude pattern >> hshift
};
static_assert(sizeof(BatchDesc) == 192, "BatchDesc must be 192 bytes");

// A batch is up to four launches, one per kernel FAMILY, so that no kernel carries the registers of code paths its
// jobs never take (the headline x-domain row kernel keeps its 80 VGPRs whatever else a batch may contain):
//   family 0  k_fq_batch        x-domain row tables: kind 2
//   family 1  k_fq_batch_d<AD>  d-domain table, approximate-quotient elements (plans with adom): kinds 0, 1 (+ 3)
//   family 2  k_fq_batch_d      d-domain table, exact division (scan plans, arbitrary value lists): kinds 0, 1, 3
//   family 3  k_fq_batch_dyn    ANTQ_FLAG_DYNAMIC rows of >= 128 vectors with an x-domain plan: kinds 4..7
//   family 4  k_fq_batch_dyn16  the same for rows of 2049..8192 vectors, one row per 1024-thread workgroup: kinds 9, 10
//   family 5  k_fq_hbatch       16-bit rows in their own 16-bit domain (bf16 / f16, 4- / 5-bit codebooks): kind 13
//   family 6 / 7 / 8  k_fq_hbatch_dyn<.., 1 / 4 / 16>  the same with ANTQ_FLAG_DYNAMIC: a row of 128..512 vectors per
//             wavefront (one-wavefront workgroups; map entries of 4 rows), of <= 2048 / <= 8192 vectors per workgroup of
//             4 / 16 wavefronts (one map entry per row): kinds 14 / 15 / 16
// (with ANTQ_FLAG_DYNAMIC families 1 / 2 run their DYN instantiation: groups of <= 64 vectors, rows of <= 256 vectors)
constexpr int kBatchFamilies = 9;

struct BatchHeader {   // 80 bytes
    uint32_t magic, n, dtype, flags, lds_bytes, map_offset, bytes, total_blocks;
    uint32_t fam_blocks[kBatchFamilies];
    uint32_t pad;          // bit 0: mixed static batch (one launch of the all-in-one kernel); bits 8..: wavefronts per workgroup of family 0
    uint32_t pad2[2];
};
static_assert(sizeof(BatchHeader) == 80, "BatchHeader must be 80 bytes");

__device__ __forceinline__ XArgs xargs_of(const BatchDesc &D)
{
    const PlanArgs &pa = D.pa;
    XArgs xa;
    xa.m = pa.m; xa.shift = pa.shift; xa.kmin = pa.kmin; xa.kmax = pa.kmax; xa.keymask = pa.keymask;
    xa.nbneg = pa.nbneg; xa.n_entries = pa.n_entries; xa.xlim = pa.xlim; xa.vout = D.vout;
    xa.linear = pa.linear; xa.lin_scale = pa.lin_scale; xa.lin_bias = pa.lin_bias;
    xa.flim = pa.fastlim * 0.99999f; xa.vmin = D.vmin; xa.vmax = D.vmax; xa.inv_gmax = D.inv_gmax;
    return xa;
}

// Family 0: the x-domain paths only.
// (A/B builds: tools/probe_ab_lib.py.  Round 2, same box, three interleaved rounds: plain 5 / 6 / 7 waves and pairs
//  6 / 7 / 8 waves all within 79.4-80.3 % -- since the d-domain paths left this kernel its occupancy no longer matters;
//  7 for the pair rule is the highest value that does not spill (8 needed 44 bytes of scratch).)
#ifndef ANTQ_OVP_WAVES
#define ANTQ_OVP_WAVES 8
#endif
#ifndef ANTQ_PLAIN_WAVES
#define ANTQ_PLAIN_WAVES 6
#endif
// WAVES: wavefronts per workgroup (4, 2 or 1).  The block -> job map and first_block stay in units of 4 tasks; a launch
// with fewer wavefronts per workgroup takes 4 / WAVES workgroups per map entry.  The tables are wave-private and there is
// no workgroup barrier, so the only thing the workgroup size changes is how wavefronts are admitted and retired: four at
// a time, or one by one -- which is what a streaming kernel wants (a plain copy of 1 GiB: 76.5 -> 80.3 % / 81.9 -> 83.5 %
// of 8 TB/s on two boxes with 64-thread workgroups, profiles/r03_stream_shapes_*.log).
template <typename T, bool OVP, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES, OVP ? ANTQ_OVP_WAVES : ANTQ_PLAIN_WAVES)
k_fq_batch(const BatchDesc *__restrict__ descs, const uint32_t *__restrict__ block_map, uint32_t rotate)
{
    constexpr uint32_t SPLIT = 4u / WAVES;                  // workgroups per map entry
    // Workgroup b runs on XCD b % 8.  When a row's last task is a partial one and the tasks per row divide 8 (or share a
    // factor with it), task "fullness" would line up with the XCD number -- half the XCDs get the full tasks, the others
    // the nearly empty ones (measured: 4096 x 2048 bf16 in 192-vector tasks 71 % instead of 82 %).  Rotating the eight
    // workgroups of every group of eight by the group's number keeps each group on the same eight tasks (same locality)
    // and gives every XCD every residue in turn.
    uint32_t blk = blockIdx.x;
    if (WAVES < 4) {
        const uint32_t g8 = blk >> 3;
        // (the decision belongs to the GROUP -- all eight of its workgroups must take the same one: the job of its first task)
        if ((g8 << 3) + 8u <= gridDim.x && (rotate == 1u || (rotate == 0u && descs[block_map[(g8 << 3) / SPLIT]].rot)))
            blk = (g8 << 3) + ((blk + g8) & 7u);
    }
    const uint32_t b4 = blk / SPLIT, sub = blk % SPLIT;
    const uint32_t j = block_map[b4];
    const BatchDesc &D = descs[j];
    const uint32_t lb = b4 - D.first_block;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint4 *plan_tab = D.plan_tab;
    const XArgs xa = xargs_of(D);

    __shared__ __attribute__((aligned(16))) uint4 wtab_all[WAVES][256];     // wave-private row tables
    // x-domain rows: wave-private table, no workgroup barrier
    const uint32_t task = __builtin_amdgcn_readfirstlane(lb * 4u + sub * WAVES + wv);
    if (task >= D.total_tasks) return;
#define ANTQ_XROW(UU)                                                                                                   \
    xrow_task<T, OVP, false, UU, false, 1>(D.x, D.out, nullptr, task, D.vpr, D.tpr, D.alpha, D.per_row, D.gmax, 1.0f, nullptr, \
                                           xa, plan_tab + (D.pa.m_pad >> 2), reinterpret_cast<const float *>(plan_tab),        \
                                           wtab_all[wv], lane, wv)
    if (D.u == 4u) ANTQ_XROW(4); else if (D.u == 3u) ANTQ_XROW(3); else if (D.u == 1u) ANTQ_XROW(1); else ANTQ_XROW(2);
#undef ANTQ_XROW
}

// Family 5: 16-bit rows in their own domain.  One wavefront per workgroup; the block -> job map is in units of 4 tasks
// like family 0's (4 workgroups per map entry); the same per-group rotation for jobs with a partial last task.
template <typename T, bool OVP>
__global__ void __launch_bounds__(64)
k_fq_hbatch(const BatchDesc *__restrict__ descs, const uint32_t *__restrict__ block_map, uint32_t rotate)
{
    __shared__ __attribute__((aligned(16))) uint2 tab[kHSlots * 2];
    uint32_t blk = blockIdx.x;
    const uint32_t g8 = blk >> 3;
    if ((g8 << 3) + 8u <= gridDim.x && (rotate == 1u || (rotate == 0u && descs[block_map[(g8 << 3) >> 2]].rot)))
        blk = (g8 << 3) + ((blk + g8) & 7u);
    const uint32_t b4 = blk >> 2, sub = blk & 3u;
    const BatchDesc &D = descs[block_map[b4]];
    const uint32_t task = (b4 - D.first_block) * 4u + sub;
    if (task >= D.total_tasks) return;
    HArgs ha;
    ha.n_thr = D.h_n & 0xffffu; ha.n_neg = D.h_n >> 16; ha.hshift = D.hshift; ha.m = D.pa.m;
    ha.flim = D.pa.fastlim * 0.99999f; ha.lim = fminf(ha.flim, D.pa.xlim);
    ha.vmin = D.vmin; ha.vmax = D.vmax; ha.vout = D.vout; ha.inv_gmax = D.inv_gmax;
    const float *grid = reinterpret_cast<const float *>(D.plan_tab);
#define ANTQ_HROW(UU) hrow_wave_task<T, OVP, UU>(D.x, D.out, task, D.vpr, D.tpr, D.alpha, D.per_row, D.gmax, ha, D.tlist, grid, tab, threadIdx.x)
    if (D.u == 4u) ANTQ_HROW(4); else if (D.u == 3u) ANTQ_HROW(3); else ANTQ_HROW(2);
#undef ANTQ_HROW
}

// Families 6 / 7 / 8: 16-bit rows in their own domain, the scale computed from the row (ANTQ_FLAG_DYNAMIC).
template <typename T, bool OVP, int WPR>
__global__ void __launch_bounds__(64 * WPR)
k_fq_hbatch_dyn(const BatchDesc *__restrict__ descs, const uint32_t *__restrict__ block_map)
{
    __shared__ __attribute__((aligned(16))) uint2 tab[WPR][kHSlots * 2];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // WPR = 1: one-wavefront workgroups, a map entry per 4 rows; else one workgroup (= one map entry) per row
    const uint32_t ent = WPR == 1 ? (blockIdx.x >> 2) : blockIdx.x;
    const BatchDesc &D = descs[block_map[ent]];
    const uint32_t rowi = WPR == 1 ? (ent - D.first_block) * 4u + (blockIdx.x & 3u) : ent - D.first_block;
    if (rowi * (uint32_t)WPR >= D.total_tasks) return;
    const uint32_t task = __builtin_amdgcn_readfirstlane(rowi * WPR + wv);
    HArgs ha;
    ha.n_thr = D.h_n & 0xffffu; ha.n_neg = D.h_n >> 16; ha.hshift = D.hshift; ha.m = D.pa.m;
    ha.flim = D.pa.fastlim * 0.99999f; ha.lim = fminf(ha.flim, D.pa.xlim);
    ha.vmin = D.vmin; ha.vmax = D.vmax; ha.vout = D.vout; ha.inv_gmax = D.inv_gmax;
    const float *grid = reinterpret_cast<const float *>(D.plan_tab);
    float *alpha_out = const_cast<float *>(D.alpha);
    hrow_wave_task_dyn<T, OVP, WPR>(D.x, D.out, task, D.vpr, D.u, D.ratio, alpha_out, D.gmax, ha, D.tlist, grid, tab[wv], lane, wv);
}

// Families 1 / 2: the plan's own (d-domain) table staged per workgroup.
template <typename T, bool OVP, bool AD, bool DYN>
__global__ void __launch_bounds__(256)
k_fq_batch_d(const BatchDesc *__restrict__ descs, const uint32_t *__restrict__ block_map)
{
    constexpr int U = kBatchU;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t j = block_map[blockIdx.x];
    const BatchDesc &D = descs[j];
    const PlanArgs pa = D.pa;
    const uint32_t lb = blockIdx.x - D.first_block;
    const uint32_t lane = threadIdx.x & 63u;
    const uint4 *plan_tab = D.plan_tab;
    float *alpha_out = DYN ? const_cast<float *>(D.alpha) : nullptr;
    if (D.kind == 1) {
        // 4 vectors per lane and workgroup, or 2 when the whole batch is only a few rounds of workgroups (antq_batch_build)
        if (DYN && AD && D.vpr > 64u)                    // 16-bit rows of 128 vectors: groups of 2 wavefronts
            lane_task<T, OVP, false, U, DYN, AD, true>(D.x, D.out, nullptr, (size_t)D.n_vec, D.vpr, D.vshift, D.alpha, D.per_row,
                                                       D.gmax, D.ratio, alpha_out, pa, plan_tab, smem,
                                                       ((size_t)lb * U) * 256u + threadIdx.x);
        else if (D.u == 2u)
            lane_task<T, OVP, false, 2, DYN, AD>(D.x, D.out, nullptr, (size_t)D.n_vec, D.vpr, D.vshift, D.alpha, D.per_row, D.gmax,
                                                 D.ratio, alpha_out, pa, plan_tab, smem, ((size_t)lb * 2) * 256u + threadIdx.x);
        else
            lane_task<T, OVP, false, U, DYN, AD>(D.x, D.out, nullptr, (size_t)D.n_vec, D.vpr, D.vshift, D.alpha, D.per_row, D.gmax,
                                                 D.ratio, alpha_out, pa, plan_tab, smem, ((size_t)lb * U) * 256u + threadIdx.x);
        return;
    }
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (AD && D.kind == 0) tab0 = atab_prefetch<false>(pa, plan_tab);
    else if (threadIdx.x < pa.tab_units) tab0 = ld_global(plan_tab + threadIdx.x);
    if (D.kind == 0) {
        const uint32_t total = D.total_tasks, vpr = D.vpr, tpr = D.tpr;
        const uint32_t task = __builtin_amdgcn_readfirstlane(lb * 4u + (threadIdx.x >> 6));
        const bool active = task < total;
        uint4 v[U];
        float a;
        task_load<T, U>(D.x, D.alpha, D.per_row, active ? task : total - 1u, vpr, tpr, lane, DYN, v, a);
        PlanLds L;
        ATab A;
        if (AD) A = stage_atab<false>(pa, plan_tab, smem, tab0);
        else L = stage_plan(pa, plan_tab, smem, tab0);
        __syncthreads();
        if (active)
            task_run<T, OVP, false, U, DYN, AD ? 1 : 0>(D.out, nullptr, alpha_out, D.ratio, task, vpr, tpr, lane, D.gmax, pa, L, A, v, a);
    } else if (!DYN) {
        const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
        __syncthreads();
        scalar_pair<T, OVP, false>(D.x, D.out, nullptr, (size_t)lb * 256u + threadIdx.x, 0, (size_t)D.n_vec, (size_t)D.n_vec,
                                   (size_t)D.vpr, D.alpha, D.per_row, D.gmax, pa, L);
    }
}

// A MIXED static batch (jobs of more than one of the families 0 / 1 / 2, e.g. ResNet-50 per channel: rows of >= 128
// vectors, shorter rows and conv1's ragged K = 147) goes out as ONE launch of this all-in-one kernel: a family per launch
// would pay two or three ramps, tails and kernel boundaries on a pass of ~35 us (measured: 66 % against 75 % in one
// launch).  The price is the union of all paths' registers, which is why single-family batches -- the headline -- do
// not use it.
template <typename T, bool OVP>
__global__ void __launch_bounds__(256)
k_fq_batch_all(const BatchDesc *__restrict__ descs, const uint32_t *__restrict__ block_map)
{
    constexpr int U = kBatchU;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    __shared__ __attribute__((aligned(16))) uint4 wtab_all[4][256];
    const uint32_t j = block_map[blockIdx.x];
    const BatchDesc &D = descs[j];
    const PlanArgs pa = D.pa;
    const uint32_t lb = blockIdx.x - D.first_block;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint4 *plan_tab = D.plan_tab;
    if (D.kind == 2) {
        const uint32_t task = __builtin_amdgcn_readfirstlane(lb * 4u + wv);
        if (task >= D.total_tasks) return;
        const XArgs xa = xargs_of(D);
#define ANTQ_XROW(UU)                                                                                                   \
    xrow_task<T, OVP, false, UU, false, 1>(D.x, D.out, nullptr, task, D.vpr, D.tpr, D.alpha, D.per_row, D.gmax, 1.0f, nullptr, \
                                           xa, plan_tab + (pa.m_pad >> 2), reinterpret_cast<const float *>(plan_tab),          \
                                           wtab_all[wv], lane, wv)
        if (D.u == 4u) ANTQ_XROW(4); else if (D.u == 3u) ANTQ_XROW(3); else ANTQ_XROW(2);
#undef ANTQ_XROW
    } else if (D.kind == 1) {
        const size_t first = ((size_t)lb * D.u) * 256u + threadIdx.x;
        if (pa.adom) {
            if (D.u == 2u)
                lane_task<T, OVP, false, 2, false, true>(D.x, D.out, nullptr, (size_t)D.n_vec, D.vpr, D.vshift, D.alpha, D.per_row,
                                                         D.gmax, 1.0f, nullptr, pa, plan_tab, smem, first);
            else
                lane_task<T, OVP, false, U, false, true>(D.x, D.out, nullptr, (size_t)D.n_vec, D.vpr, D.vshift, D.alpha, D.per_row,
                                                         D.gmax, 1.0f, nullptr, pa, plan_tab, smem, first);
        } else {
            lane_task<T, OVP, false, U, false, false>(D.x, D.out, nullptr, (size_t)D.n_vec, D.vpr, D.vshift, D.alpha, D.per_row,
                                                      D.gmax, 1.0f, nullptr, pa, plan_tab, smem, first);
        }
    } else if (D.kind == 0) {
        const uint32_t total = D.total_tasks, vpr = D.vpr, tpr = D.tpr;
        const uint32_t task = __builtin_amdgcn_readfirstlane(lb * 4u + wv);
        const bool active = task < total;
        uint4 tab0 = make_uint4(0, 0, 0, 0);
        if (pa.adom) tab0 = atab_prefetch<false>(pa, plan_tab);
        else if (threadIdx.x < pa.tab_units) tab0 = ld_global(plan_tab + threadIdx.x);
        uint4 v[U];
        float a;
        task_load<T, U>(D.x, D.alpha, D.per_row, active ? task : total - 1u, vpr, tpr, lane, false, v, a);
        PlanLds L;
        ATab A;
        if (pa.adom) A = stage_atab<false>(pa, plan_tab, smem, tab0);
        else L = stage_plan(pa, plan_tab, smem, tab0);
        __syncthreads();
        if (active) task_run<T, OVP, false, U, false, -1>(D.out, nullptr, nullptr, 1.0f, task, vpr, tpr, lane, D.gmax, pa, L, A, v, a);
    } else {
        uint4 tab0 = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < pa.tab_units) tab0 = ld_global(plan_tab + threadIdx.x);
        const PlanLds L = stage_plan(pa, plan_tab, smem, tab0);
        __syncthreads();
        scalar_pair<T, OVP, false>(D.x, D.out, nullptr, (size_t)lb * 256u + threadIdx.x, 0, (size_t)D.n_vec, (size_t)D.n_vec,
                                   (size_t)D.vpr, D.alpha, D.per_row, D.gmax, pa, L);
    }
}

// Family 3.  ANTQ_FLAG_DYNAMIC rows: alpha = row abs-max (x ratio) computed from the registers that hold the row -- one
// HBM read -- for every tensor of the batch in one launch.  A kernel of its own so that the 8-vectors-per-lane variants
// do not raise the register count (and lower the occupancy) of the static kernel above.
template <typename T, bool OVP>
__global__ void __launch_bounds__(256)
k_fq_batch_dyn(const BatchDesc *__restrict__ descs, const uint32_t *__restrict__ block_map)
{
    __shared__ __attribute__((aligned(16))) uint4 wtab_all[4][256];
    const uint32_t j = block_map[blockIdx.x];
    const BatchDesc &D = descs[j];
    const PlanArgs pa = D.pa;
    const uint32_t lb = blockIdx.x - D.first_block;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint4 *plan_tab = D.plan_tab;
    const uint32_t task = __builtin_amdgcn_readfirstlane(lb * 4u + wv);
    const XArgs xa = xargs_of(D);
    float *alpha_out = const_cast<float *>(D.alpha);
    const uint4 *entries = plan_tab + (pa.m_pad >> 2);
    const float *grid = reinterpret_cast<const float *>(plan_tab);
    const float ratio = D.ratio;
    if (D.kind == 4) {
        if (task < D.total_tasks) {
#define ANTQ_XROW(UU)                                                                                                   \
    xrow_task<T, OVP, false, UU, true, 1>(D.x, D.out, nullptr, task, D.vpr, 1u, nullptr, 1, D.gmax, ratio, alpha_out, xa,  \
                                          entries, grid, wtab_all[wv], lane, wv)
            if (D.u == 4u) ANTQ_XROW(4); else if (D.u == 3u) ANTQ_XROW(3); else ANTQ_XROW(2);
#undef ANTQ_XROW
        }
    } else if (D.kind == 6) {
        if (task < D.total_tasks)
            xrow_task<T, OVP, false, 8, true, 1>(D.x, D.out, nullptr, task, D.vpr, 1u, nullptr, 1, D.gmax, ratio, alpha_out, xa,
                                                 entries, grid, wtab_all[wv], lane, wv);
    } else if (D.kind == 12) {
        // rows of 257..512 vectors spread over the 4 wavefronts of the workgroup, 2 vectors per lane
        xrow_task<T, OVP, false, 2, true, 4>(D.x, D.out, nullptr, task, D.vpr, 4u, nullptr, 1, D.gmax, ratio, alpha_out, xa,
                                             entries, grid, wtab_all[wv], lane, wv);
    } else if (D.kind == 5) {
        xrow_task<T, OVP, false, 4, true, 4>(D.x, D.out, nullptr, task, D.vpr, 4u, nullptr, 1, D.gmax, ratio, alpha_out, xa,
                                             entries, grid, wtab_all[wv], lane, wv);
    } else {
        xrow_task<T, OVP, false, 8, true, 4>(D.x, D.out, nullptr, task, D.vpr, 4u, nullptr, 1, D.gmax, ratio, alpha_out, xa,
                                             entries, grid, wtab_all[wv], lane, wv);
    }
}

// Family 4.  Rows of 2049..8192 vectors (C4's 28 672-wide rows): one row per 1024-thread workgroup, 4 or 8 vectors per lane.
template <typename T, bool OVP>
__global__ void __launch_bounds__(1024)
k_fq_batch_dyn16(const BatchDesc *__restrict__ descs, const uint32_t *__restrict__ block_map)
{
    __shared__ __attribute__((aligned(16))) uint4 wtab_all[16][256];
    const uint32_t j = block_map[blockIdx.x];
    const BatchDesc &D = descs[j];
    const PlanArgs pa = D.pa;
    const uint32_t lb = blockIdx.x - D.first_block;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint4 *plan_tab = D.plan_tab;
    const uint32_t task = __builtin_amdgcn_readfirstlane(lb * 16u + wv);
    const XArgs xa = xargs_of(D);
    float *alpha_out = const_cast<float *>(D.alpha);
    const uint4 *entries = plan_tab + (pa.m_pad >> 2);
    const float *grid = reinterpret_cast<const float *>(plan_tab);
    if (D.kind == 9)
        xrow_task<T, OVP, false, 4, true, 16>(D.x, D.out, nullptr, task, D.vpr, 16u, nullptr, 1, D.gmax, D.ratio, alpha_out, xa,
                                              entries, grid, wtab_all[wv], lane, wv);
    else
        xrow_task<T, OVP, false, 8, true, 16>(D.x, D.out, nullptr, task, D.vpr, 16u, nullptr, 1, D.gmax, D.ratio, alpha_out, xa,
                                              entries, grid, wtab_all[wv], lane, wv);
}

static int epl_of(int dtype) { return dtype == ANTQ_F32 ? 4 : (dtype == ANTQ_BF16 || dtype == ANTQ_F16) ? 8 : 0; }

// blocks a job needs, or 0 if it cannot be expressed
static size_t job_blocks(const antq_job &J, int epl, BatchDesc *d)
{
    size_t rows = J.rows, row_len = J.row_len;
    const size_t n = rows * row_len;
    if (!J.alpha_per_row) { rows = 1; row_len = n; }
    if (n == 0) return 0;
    if (row_len % epl != 0 || reinterpret_cast<uintptr_t>(J.x_dev) % 16 || reinterpret_cast<uintptr_t>(J.out_dev) % 16) {
        // ragged rows (conv1: K = 147) or unaligned buffers: one thread per flat pair, exact reference arithmetic
        if (row_len > 0xffffffffull) return 0;
        if (d) { d->kind = 3; d->total_tasks = 0; d->vpr = (uint32_t)row_len; d->tpr = 1; d->vshift = -1; d->n_vec = n; }
        return ((n + 1) / 2 + 255) / 256;
    }
    const size_t vpr = row_len / epl;
    if (vpr > 0xffffffffull) return 0;
    size_t blocks;
    if (vpr >= kRowKernelMinVpr) {
        const size_t tpr = (vpr + 64 * kBatchU - 1) / (64 * kBatchU);
        const size_t total = rows * tpr;
        if (total > 0xfffffff0ull) return 0;
        blocks = (total + 3) / 4;
        if (d) { d->kind = 0; d->total_tasks = (uint32_t)total; d->vpr = (uint32_t)vpr; d->tpr = (uint32_t)tpr; d->vshift = -1; d->n_vec = n / epl; }
    } else {
        const size_t n_vec = n / epl;
        blocks = (n_vec + 256 * kBatchU - 1) / (256 * kBatchU);
        int vshift = -1;
        if ((vpr & (vpr - 1)) == 0) { vshift = 0; while (((size_t)1 << vshift) < vpr) vshift++; }
        if (d) { d->kind = 1; d->total_tasks = 0; d->vpr = (uint32_t)vpr; d->tpr = 1; d->vshift = vshift; d->n_vec = n_vec; }
    }
    return blocks;
}

}  // namespace antq

#endif  // ANTQ_K_BATCH_H
