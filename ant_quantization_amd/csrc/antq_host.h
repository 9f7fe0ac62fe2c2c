// antq_host.h -- host-side helpers shared by libantq's translation units (launchers of antq_fq.hip, antq_batch.hip,
// antq_search.hip, antq_kernels.hip): the development knobs and the plan-blob accessors.  gfx950 only.
#ifndef ANTQ_HOST_H
#define ANTQ_HOST_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/antq.h"
#include "antq_internal.h"
#include "antq_device.h"
#include "antq_k_approx.h"

namespace antq {

// tuning knobs (dev / bench only; see antq_debug_set).  THREAD-LOCAL: they change the dispatch of the calling thread's
// later calls only, so a probe that forgets to reset them cannot change which kernel another thread's calls run, and the
// library keeps no process-global mutable state.  Defined in antq_kernels.hip.
extern thread_local int g_knob_u;             // force U of the uniform kernel (0 = heuristic)
extern thread_local int g_knob_encwg;         // persistent workgroups of the 4-bit encoder (256 CUs x 8)
extern thread_local int g_knob_x;             // 0 disables the x-domain row kernel (A/B measurements)
extern thread_local int g_knob_nearest_fast;  // 0: antq_nearest always runs the literal scan
extern thread_local int g_knob_lane_rows;     // 0: rows of a power of two of vectors through the per-row table kernels (A/B)
extern thread_local int g_knob_a;             // 0 disables the approximate-quotient element path (quant_vec_a): exact division
extern thread_local int g_knob_waves;         // wavefronts per workgroup of the streaming kernels: 0 = the measured default, 1 / 2 / 4 (A/B)
extern thread_local int g_knob_rot;            // 1: rotate the workgroup -> task map of the batched row kernel per group of 8 (XCD balance)
extern thread_local int g_knob_lane_u;        // vectors per lane of the one-launch-per-tensor lane kernel: 0 = default (A/B)
extern thread_local int g_knob_h;             // 0 disables the 16-bit-domain row kernels (antq_k_hrow.h; A/B measurements)
extern thread_local int g_knob_hist;          // clip search of 16-bit per-tensor quantisers on the tensor's histogram: 0 off, 1 when it pays, 2 always (tests)
extern thread_local int g_knob_sweep;         // 0: per-row clip searches through the direct kernels instead of the threshold sweep (A/B, tests)
extern thread_local int g_knob_sort_short;    // 0: rows of <= 1024 elements through the 4096-key sorted search instead of the one-row-per-wavefront kernel (A/B)
extern thread_local int g_knob_sort;          // clip searches from the sorted row (antq_k_sortsearch.h): 0 off, 1 the default rule, 2 every eligible launch (tests)
extern thread_local int g_knob_tk_group, g_knob_tk_blocks;   // A/B of the one-launch reductions (antq_k_reduce.h)
extern thread_local int g_knob_rows_stream;   // 0: row abs-max of 128..1024-vector rows through the round-5 kernel (A/B)
extern thread_local int g_knob_hist_xmax;     // antq_calibrate: the abs-max statistic of a histogram-searched tensor from the counting pass (1) or from its own pass (0)
extern thread_local int g_knob_exp;           // experiment switch of the kernel under development (A/B; 0 = off)
extern thread_local int g_knob_schunks;       // clip search: candidate-list chunks (blockIdx.y) forced to this many (A/B; 0 = cost model)
extern thread_local int g_knob_dlds;          // extra dynamic LDS bytes of the d-domain batched kernels' workgroups (occupancy A/B)
extern thread_local int g_knob_hlds;          // dynamic LDS of those kernels' workgroups (occupancy, A/B): -1 = kHRowLdsPad

static inline bool plan_args_from_host(const void *plan_host, PlanArgs &pa)
{
    const PlanHeader *h = static_cast<const PlanHeader *>(plan_host);
    if (h->magic != kPlanMagic || h->version != kPlanVersion) return false;
    if (h->m < 1 || h->m > ANTQ_MAX_GRID || h->m_pad != ((h->m + 3) & ~3u)) return false;
    pa.kind = h->kind;
    pa.m = h->m;
    pa.m_pad = h->m_pad;
    pa.shift = h->shift;
    pa.kmin = h->kmin;
    pa.kmax = h->kmax;
    pa.keymask = h->keymask;
    pa.nbneg = h->nbneg;
    pa.fastlim = h->fastlim;
    pa.n_entries = (h->kind == kPlanLut) ? h->n_entries : 0;
    pa.tab_units = pa.n_entries + (pa.m_pad >> 2);
    pa.linear = h->linear;
    pa.lin_scale = h->lin_scale;
    pa.lin_bias = h->lin_bias;
    pa.adom = (h->kind == kPlanLut && g_knob_a != 0) ? h->adom : 0u;
    pa.xlim = h->xlim;
    pa.atab_slots = h->atab_slots;
    return true;
}

static inline const uint4 *plan_tab_ptr(const void *plan_dev)
{
    return reinterpret_cast<const uint4 *>(static_cast<const char *>(plan_dev) + sizeof(PlanHeader));
}

// dynamic LDS of a kernel that stages the plan's table (stage_plan) or, for plans with adom, its converted image (stage_atab)
static inline size_t lds_table(const PlanArgs &pa, bool idx)
{
    const size_t plain = (size_t)pa.tab_units * 16;
    return pa.adom ? std::max(plain, (size_t)atab_units(pa.atab_slots, pa.m_pad, idx) * 16) : plain;
}

}  // namespace antq

#endif  // ANTQ_HOST_H
