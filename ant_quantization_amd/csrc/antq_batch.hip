// antq_batch.hip -- the batched launch of libantq: antq_batch_capacity / antq_batch_build / antq_fakequant_batch
// (many tensors' Quantizer._forward, AQ/quant_modules.py:535-551 / OQ:294-330, in one launch).  gfx950 only.
#include "antq_host.h"
#include "antq_k_batch.h"

using namespace antq;

extern "C" size_t antq_batch_capacity(const antq_job *jobs, int n, int dtype)
{
    const int epl = epl_of(dtype);
    if (!jobs || n < 1 || !epl) return 0;
    size_t blocks = 0;
    // (the dynamic variant gives rows of 257..1024 vectors a workgroup each: never more than max(static, rows))
    // (x-domain rows may be cut into tasks of 1, 2 or 3 vectors per lane instead of 4: at most four times the blocks)
    for (int i = 0; i < n; i++) blocks += std::max(4 * job_blocks(jobs[i], epl, nullptr) + 1, jobs[i].rows);
    return sizeof(BatchHeader) + sizeof(BatchDesc) * (size_t)n + 4 * blocks;
}

extern "C" int antq_batch_build(const antq_job *jobs, int n, int dtype, unsigned flags, void *blob, size_t cap)
{
    const int epl = epl_of(dtype);
    if (!jobs || !blob || n < 1 || n > 65535) return ANTQ_ERR_ARG;
    if (!epl) return ANTQ_ERR_UNSUPPORTED;
    const bool dyn = (flags & ANTQ_FLAG_DYNAMIC) != 0;
    char *p = static_cast<char *>(blob);
    BatchHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = kBatchMagic; h.n = (uint32_t)n; h.dtype = (uint32_t)dtype; h.flags = flags;
    h.map_offset = (uint32_t)(sizeof(BatchHeader) + sizeof(BatchDesc) * (size_t)n);
    if (cap < h.map_offset) return ANTQ_ERR_PLAN;
    BatchDesc *descs = reinterpret_cast<BatchDesc *>(p + sizeof(BatchHeader));
    uint32_t *map = reinterpret_cast<uint32_t *>(p + h.map_offset);
    std::vector<uint8_t> fam((size_t)n);
    std::vector<size_t> nblk((size_t)n);
    // bytes the launch will move (in + out).  Whole models (OPT-6.7B: 25.8 GB, the 70 B stack: 137 GB in place) stream with
    // a longer average memory latency than a few GB do (address translation), and the best launch shape moves with it: below
    // ~8 GB one-wavefront workgroups win (81.4 against 80.5 %, tools/probe_footprint.py); beyond, 4 wavefronts per workgroup
    // with the same 2-vector tasks: OPT-6.7B 79.4 against 78.3 % (one wavefront, 2 or 4 vectors), the 70 B stack 77.9
    // against 77.3 % (tools/bench_sharded.py --knobs in steady state; tools/probe_sharded_ab.py on five boxes: +1.0 ... +1.6).
    // (An earlier choice -- 4-vector tasks in one-wavefront workgroups -- came from a harness that timed 5 passes after 2:
    //  before the clocks had settled, where fewer, longer tasks look better.)
    double batch_bytes = 0.0;
    for (int i = 0; i < n; i++) batch_bytes += 2.0 * (double)jobs[i].rows * (double)jobs[i].row_len * (dtype == ANTQ_F32 ? 4.0 : 2.0);
    const bool big_footprint = batch_bytes >= 8.0 * 1073741824.0;
    size_t fam_blocks[kBatchFamilies] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, lds = 0;
    bool any_da = false;
    for (int i = 0; i < n; i++) {
        const antq_job &J = jobs[i];
        if (!J.x_dev || !J.out_dev || (!J.alpha_dev && !dyn) || !J.plan_host || !J.plan_dev) return ANTQ_ERR_ARG;
        const uintptr_t esz = (dtype == ANTQ_F32) ? 4 : 2;
        if (reinterpret_cast<uintptr_t>(J.x_dev) % esz || reinterpret_cast<uintptr_t>(J.out_dev) % esz) return ANTQ_ERR_ALIGN;
        BatchDesc d;
        memset(&d, 0, sizeof(d));
        size_t blocks = job_blocks(J, epl, &d);
        if (blocks == 0) return ANTQ_ERR_UNSUPPORTED;
        if (!plan_args_from_host(J.plan_host, d.pa)) return ANTQ_ERR_PLAN;
        const PlanHeader *ph = static_cast<const PlanHeader *>(J.plan_host);
        const bool xdom = g_knob_x && d.pa.kind == kPlanLut && ph->xdom;
        d.vout = ph->vout;
        {
            const XArgs xe = xargs_from_plan(J.plan_host, d.pa);
            d.vmin = xe.vmin; d.vmax = xe.vmax;
        }
        d.ratio = 1.0f;
        int f;
        d.u = (uint32_t)kBatchU;
        if (!dyn) {
            if (d.kind == 0 && d.pa.adom && (dtype == ANTQ_F32 || g_knob_lane_rows == 2) && g_knob_lane_rows != 0) {
                // fp32 long rows as lane jobs (alpha index = a shift, or the f64-reciprocal quotient): 16 x 4096^2 78.9 -> 81.0 %,
                // with OliVe's pairs 79.4 -> 81.0 %, BERT-base's 768 / 3072-wide rows 79.0 -> 80.8 %, ResNet-50 75.2 -> 76.0 %
                // against the per-row table kernel.  16-bit rows stay on the table kernel: equal without the pair rule (80.3 vs
                // 80.8, 81.1 vs 81.2 %), 0.6-1.3 points ahead with it, 0.7 ahead on BERT's shapes (tools/probe_batch_lane.py;
                // knob 5 = 0 restores the table kernel for fp32 too, 2 makes every long row a lane job)
                d.kind = 1; d.total_tasks = 0; d.tpr = 1; d.vshift = -1;
                if ((d.vpr & (d.vpr - 1u)) == 0u) { d.vshift = 0; while ((1u << d.vshift) < d.vpr) d.vshift++; }
                blocks = (size_t)((d.n_vec + 256u * kBatchU - 1u) / (256u * kBatchU));
            }
            HArgs hx;
            const uint32_t hu = (d.kind == 0 && g_knob_h != 0 && hargs_from_plan(J.plan_host, dtype, J.gmax, hx))
                                    ? (g_knob_h == 2 ? row_task_u(d.vpr) : hrow_static_u(d.vpr, xdom)) : 0u;      // (knob 9 = 2: every row, A/B)
            if (hu != 0u) {
                // 16-bit rows in their own domain (antq_k_hrow.h): 4 KiB of the row per wavefront unless 3 or 2 leave fewer
                // idle lanes; 24 workgroups per CU (kHRowLdsPad)
                d.kind = 13;
                d.u = hu;
                if (g_knob_u >= 2 && g_knob_u <= 4) d.u = (uint32_t)g_knob_u;     // knob 0 (A/B): vectors per lane and task
                d.tpr = (d.vpr + 64u * d.u - 1u) / (64u * d.u);
                const size_t total = (J.alpha_per_row ? J.rows : (size_t)1) * (size_t)d.tpr;   // per tensor: ONE row
                if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
                d.total_tasks = (uint32_t)total;
                blocks = (total + 3) / 4;
                d.rot = (d.vpr % (64u * d.u) != 0u && (d.tpr % 2u) == 0u) ? 1u : 0u;
                d.tlist = plan_tlist_dev(J.plan_host, J.plan_dev);
                d.h_n = hx.n_thr | (hx.n_neg << 16);
                d.hshift = hx.hshift;
            }
            if (d.kind == 0 && xdom) {
                // x-domain rows: the task size that leaves the fewest idle lanes for this row length
                d.kind = 2;
                d.u = row_task_u_small(d.vpr);
                if (g_knob_u >= 1 && g_knob_u <= 4) d.u = (uint32_t)g_knob_u;     // knob 0 (A/B): vectors per lane and task
                d.tpr = (d.vpr + 64u * d.u - 1u) / (64u * d.u);
                const size_t total = (J.alpha_per_row ? J.rows : (size_t)1) * (size_t)d.tpr;   // per tensor: ONE row
                if (total > 0xfffffff0ull) return ANTQ_ERR_UNSUPPORTED;
                d.total_tasks = (uint32_t)total;
                blocks = (total + 3) / 4;
                // a partial last task + tasks per row sharing a factor with the 8 XCDs: rotate this job's workgroup -> task map
                d.rot = (d.vpr % (64u * d.u) != 0u && (d.tpr % 2u) == 0u) ? 1u : 0u;
            }
            // (groups of 16 / 32 / 64 vectors: a per-group x-domain table was round 1's answer for bf16 group-128 ... 512; the
            //  lane kernel with the exact per-element decision matches it for 16-bit data -- 75.2-76.7 vs 76.5-77.5 % -- and
            //  beats it for fp32 with 2-vector tasks -- 79.7-80.4 vs 77-79 %: profiles/r02_lane_task_ab.log -- so it is gone)
            f = d.kind == 13 ? 5 : d.kind == 2 ? 0 : (d.kind == 3 ? -1 : (d.pa.adom ? 1 : 2));
        } else {
            // alpha computed in the kernel: the group / row has to live in the registers of a few lanes, one wavefront
            // or one workgroup
            if (!J.alpha_per_row || d.kind == 3 || J.rows > 0x3ffffff0ull) return ANTQ_ERR_UNSUPPORTED;
            HArgs hprobe;
            const bool h_dyn = d.kind == 0 && g_knob_h != 0 && d.vpr <= 8192u && hargs_from_plan(J.plan_host, dtype, J.gmax, hprobe);
            if (!h_dyn && d.kind == 0 && d.pa.adom && dtype != ANTQ_F32 && d.vpr == 128u && g_knob_u != 1) {
                // 16-bit rows of 128 vectors as lane jobs whose groups span 2 wavefronts of a workgroup (LDS exchange of
                // the wavefront maxima): 4 vectors in flight per lane instead of a wavefront per row: 71 -> 75 %; at 256
                // vectors the wavefront-per-row kernel stays ahead (79 vs 75 %)
                d.kind = 1; d.total_tasks = 0; d.tpr = 1; d.vshift = 7;
                blocks = (size_t)((d.n_vec + 256u * kBatchU - 1u) / (256u * kBatchU));
            }
            const HArgs &hx = hprobe;
            if (h_dyn) {
                // 16-bit rows of 128 .. 8192 vectors in their own domain: the row in one wavefront (<= 512 vectors), one
                // workgroup of 4 (<= 2048) or of 16 wavefronts
                const HDynShape sh = hrow_dyn_shape(d.vpr);
                d.kind = sh.wpr == 1 ? 14 : (sh.wpr == 4 ? 15 : 16);
                d.u = (uint32_t)sh.vpt;
                d.tpr = (uint32_t)sh.wpr;
                d.total_tasks = (uint32_t)(J.rows * (size_t)sh.wpr);
                blocks = sh.wpr == 1 ? (J.rows + 3) / 4 : J.rows;
                d.tlist = plan_tlist_dev(J.plan_host, J.plan_dev);
                d.h_n = hx.n_thr | (hx.n_neg << 16);
                d.hshift = hx.hshift;
                f = sh.wpr == 1 ? 6 : (sh.wpr == 4 ? 7 : 8);
            } else
            if (d.kind == 1) {
                if (d.vshift < 0 || d.vpr > 256u) return ANTQ_ERR_UNSUPPORTED;  // butterfly over a power-of-two group
                f = d.pa.adom ? 1 : 2;       // (per-group tables with the abs-max in front measured slower: 66 vs 71 %)
            } else if (xdom && !(d.pa.adom && d.vpr <= (dtype == ANTQ_F32 ? 128u : 256u) && g_knob_u != 1)) {
                // (rows of <= 256 vectors: bf16 / f16 measured faster through the exact per-element decision below -- 70 vs
                //  61 % at 128 vectors, 79 vs 72 % at 256 -- fp32 only at 128; profiles/r02_group_sweep.log)
                if (d.vpr > 8192u) return ANTQ_ERR_UNSUPPORTED;
                // one wavefront per row up to 512 vectors (4 or 8 per lane), one workgroup per row beyond: 4 wavefronts up
                // to 2048 vectors, 16 (a 1024-thread workgroup) up to 8192
                f = 3;
                if (d.vpr <= 256u) {
                    d.kind = 4; d.tpr = 1; d.total_tasks = (uint32_t)J.rows; blocks = (J.rows + 3) / 4;
                    d.u = d.vpr <= 128u ? 2u : d.vpr <= 192u ? 3u : 4u;
                }
                else if (d.vpr <= 512u && dtype == ANTQ_F32 && g_knob_u != 8) {
                    // fp32 rows of 257..512 vectors over the 4 wavefronts of a workgroup (80 vs 78 %); 16-bit rows of that many
                    // vectors are twice the elements and stay in one wavefront (74 vs 63 %)
                    d.kind = 12; d.tpr = 4; d.total_tasks = (uint32_t)(J.rows * 4); blocks = J.rows;
                }
                else if (d.vpr <= 512u) { d.kind = 6; d.tpr = 1; d.total_tasks = (uint32_t)J.rows; blocks = (J.rows + 3) / 4; }
                else if (d.vpr <= 2048u) { d.kind = d.vpr <= 1024u ? 5 : 7; d.tpr = 4; d.total_tasks = (uint32_t)(J.rows * 4); blocks = J.rows; }
                else { d.kind = d.vpr <= 4096u ? 9 : 10; d.tpr = 16; d.total_tasks = (uint32_t)(J.rows * 16); blocks = J.rows; f = 4; }
            } else {
                if (d.vpr > 64u * kBatchU) return ANTQ_ERR_UNSUPPORTED;         // the row in one wavefront's registers
                d.tpr = 1; d.total_tasks = (uint32_t)J.rows; blocks = (J.rows + 3) / 4;
                f = d.pa.adom ? 1 : 2;
            }
        }
        any_da = any_da || f == 1;
        d.x = static_cast<const uint4 *>(J.x_dev);
        d.out = static_cast<uint4 *>(J.out_dev);
        d.alpha = J.alpha_dev;
        d.plan_tab = plan_tab_ptr(J.plan_dev);
        d.per_row = J.alpha_per_row ? 1 : 0;
        d.gmax = J.gmax;
        d.inv_gmax = 1.0 / (double)J.gmax;
        descs[i] = d;
        fam[(size_t)i] = (uint8_t)(f < 0 ? 255 : f);
        nblk[(size_t)i] = blocks;
        if (f == 1 || f == 2 || f < 0) lds = std::max(lds, lds_table(d.pa, false));
    }
    // element-granular jobs (exact arithmetic, no table path) ride along with whichever d-domain launch exists
    for (int i = 0; i < n; i++)
        if (fam[(size_t)i] == 255) fam[(size_t)i] = any_da ? 1 : 2;
    {
        // Lane jobs (kind 1, adom) take 2 instead of 4 vectors per lane when the batch is only a few rounds of workgroups
        // (256 CUs x 8 workgroups = 2048 per round): smaller workgroups shorten the ramp and the tail of a short pass
        // (ResNet-50 group-16, 3 rounds: 72 -> ~75 %).  Knob 0: 2 forces it, 4 forbids it (A/B).
        size_t all_blocks = 0;
        for (int i = 0; i < n; i++) all_blocks += nblk[(size_t)i];
        // fp32 lane jobs always: 79.7-80.4 % with 2 against 75.7-76.3 % with 4 vectors per lane on 16 x 4096^2
        const bool small = g_knob_u == 2 || (g_knob_u != 4 && (all_blocks < 4u * 2048u || dtype == ANTQ_F32));
        for (int i = 0; i < n && small; i++) {
            BatchDesc &d = descs[i];
            if (d.kind == 1 && d.pa.adom && !(dyn && d.vpr > 64u)) {    // (groups of 2 wavefronts keep 4 vectors per lane)
                d.u = 2u;
                nblk[(size_t)i] = (size_t)((d.n_vec + 511u) / 512u);
            }
        }
    }
    if (!dyn) {
        // jobs of more than one static family: ONE launch of the all-in-one kernel instead of a launch per family
        // The 16-bit-domain rows (family 5) keep a launch of their own next to the others when they are a launch's worth of
        // work (>= 128 MiB moved: ~20 us at full speed, so that its ramp and tail are small change); a smaller share rides
        // in the all-in-one kernel as fp32-domain rows like round 3.
        double h_bytes = 0.0;
        for (int i = 0; i < n; i++)
            if (fam[(size_t)i] == 5) h_bytes += 2.0 * (double)jobs[i].rows * (double)jobs[i].row_len * (dtype == ANTQ_F32 ? 4.0 : 2.0);
        const bool h_apart = h_bytes >= 128.0 * 1048576.0;
        if (!h_apart) {
            // (every plan with hdom that came here as kind 0 has xdom or keeps the d-domain row kernel)
            for (int i = 0; i < n; i++) {
                BatchDesc &d = descs[i];
                if (d.kind != 13) continue;
                bool other = false;
                for (int k = 0; k < n && !other; k++) other = fam[(size_t)k] != 5;
                if (!other) break;                                  // the whole (small) batch is family 5: its own launch
                const PlanHeader *ph = static_cast<const PlanHeader *>(jobs[i].plan_host);
                if (g_knob_x && d.pa.kind == kPlanLut && ph->xdom) {
                    d.kind = 2;
                    d.u = row_task_u_small(d.vpr);
                    d.tpr = (d.vpr + 64u * d.u - 1u) / (64u * d.u);
                    d.total_tasks = (uint32_t)((jobs[i].alpha_per_row ? jobs[i].rows : (size_t)1) * (size_t)d.tpr);
                    d.rot = (d.vpr % (64u * d.u) != 0u && (d.tpr % 2u) == 0u) ? 1u : 0u;
                    fam[(size_t)i] = 0;
                } else {
                    const size_t tpr = (d.vpr + 64 * kBatchU - 1) / (64 * kBatchU);
                    d.kind = 0; d.u = (uint32_t)kBatchU; d.tpr = (uint32_t)tpr; d.rot = 0;
                    d.total_tasks = (uint32_t)((jobs[i].alpha_per_row ? jobs[i].rows : (size_t)1) * tpr);
                    lds = std::max(lds, lds_table(d.pa, false));
                    fam[(size_t)i] = (uint8_t)(d.pa.adom ? 1 : 2);
                }
                nblk[(size_t)i] = ((size_t)d.total_tasks + 3) / 4;
            }
        }
        bool seen[kBatchFamilies] = {false, false, false, false, false, false, false, false, false};
        int nf_rest = 0;
        for (int i = 0; i < n; i++)
            if (fam[(size_t)i] != 5 && !seen[fam[(size_t)i]]) { seen[fam[(size_t)i]] = true; nf_rest++; }
        if (nf_rest > 1) {
            // several families besides (a big) family 5: they share the all-in-one launch
            h.pad = 1u;
            for (int i = 0; i < n; i++)
                if (fam[(size_t)i] != 5) fam[(size_t)i] = 0;
        }
    }
    {
        // Wavefronts per workgroup of the row-table launch (family 0): 1 when most of its bytes sit in rows of >= 256 vectors
        // (same-box A/B, bf16: 512 ... 3584-vector rows 80.3-81.0 -> 81.4-82.5 %; rows of 128 vectors 79.3-80.7 -> 78.6 %;
        // profiles/r03_batch_rotation.log), 4 otherwise.  Stored in bits 8.. of `pad` (bit 0: mixed batch).
        double long_b = 0.0, short_b = 0.0;
        for (int i = 0; i < n; i++)
            if (descs[i].kind == 2) (descs[i].vpr >= 256u ? long_b : short_b) += (double)descs[i].total_tasks * descs[i].u;
        h.pad |= ((long_b >= short_b && !big_footprint) ? 1u : 4u) << 8;          // (big footprints: see batch_bytes above)
    }
    {
        uint32_t umax = 0;
        for (int i = 0; i < n; i++)
            if (descs[i].kind == 14) umax = std::max(umax, descs[i].u);
        h.pad2[0] = hrow_dyn_lds_pad(HDynShape{1, (int)umax});
    }
    size_t total_blocks = 0;
    for (int i = 0; i < n; i++) {
        descs[i].first_block = (uint32_t)fam_blocks[fam[(size_t)i]];
        fam_blocks[fam[(size_t)i]] += nblk[(size_t)i];
        total_blocks += nblk[(size_t)i];
    }
    if (total_blocks > 0x1fffffffull) return ANTQ_ERR_UNSUPPORTED;      // (x 4: one-wavefront workgroups)
    if (cap < h.map_offset + 4 * total_blocks) return ANTQ_ERR_PLAN;
    size_t off[kBatchFamilies], acc = 0;
    for (int f = 0; f < kBatchFamilies; f++) { off[f] = acc; acc += fam_blocks[f]; h.fam_blocks[f] = (uint32_t)fam_blocks[f]; }
    for (int i = 0; i < n; i++) {
        uint32_t *m = map + off[fam[(size_t)i]] + descs[i].first_block;
        for (size_t b = 0; b < nblk[(size_t)i]; b++) m[b] = (uint32_t)i;
    }
    h.total_blocks = (uint32_t)total_blocks;
    h.lds_bytes = (uint32_t)lds;
    h.bytes = (uint32_t)(h.map_offset + 4 * total_blocks);
    memcpy(p, &h, sizeof(h));
    return (int)h.bytes;
}

// family 5 (16-bit dtypes only)
template <typename T>
static void launch_hbatch(uint32_t map_entries, const BatchDesc *descs, const uint32_t *map, bool ovp, hipStream_t st)
{
    const dim3 g(map_entries * 4u), b(64u);
    // (the occupancy cap pays once the launch is several rounds of workgroups; a small batch wants every slot -- as a single
    //  tensor does, antq_fq.hip)
    const unsigned pad = g_knob_hlds >= 0 ? (unsigned)g_knob_hlds : (map_entries >= 8192u ? kHRowLdsPad : 0u);
    if (ovp) hipLaunchKernelGGL((k_fq_hbatch<T, true>), g, b, pad, st, descs, map, (uint32_t)g_knob_rot);
    else hipLaunchKernelGGL((k_fq_hbatch<T, false>), g, b, pad, st, descs, map, (uint32_t)g_knob_rot);
}
template <>
void launch_hbatch<float>(uint32_t, const BatchDesc *, const uint32_t *, bool, hipStream_t) {}

// families 6 / 7 / 8 (16-bit dtypes only).  The occupancy cap of family 6 follows the widest job of the batch.
template <typename T>
static void launch_hbatch_dyn(const BatchHeader *h, const BatchDesc *descs, const uint32_t *const *fmap, bool ovp, hipStream_t st)
{
    if (h->fam_blocks[6]) {
        const dim3 g(h->fam_blocks[6] * 4u), b(64u);
        const unsigned pad = g_knob_hlds >= 0 ? (unsigned)g_knob_hlds : h->pad2[0];
        if (ovp) hipLaunchKernelGGL((k_fq_hbatch_dyn<T, true, 1>), g, b, pad, st, descs, fmap[6]);
        else hipLaunchKernelGGL((k_fq_hbatch_dyn<T, false, 1>), g, b, pad, st, descs, fmap[6]);
    }
    if (h->fam_blocks[7]) {
        const dim3 g(h->fam_blocks[7]), b(256u);
        if (ovp) hipLaunchKernelGGL((k_fq_hbatch_dyn<T, true, 4>), g, b, 0, st, descs, fmap[7]);
        else hipLaunchKernelGGL((k_fq_hbatch_dyn<T, false, 4>), g, b, 0, st, descs, fmap[7]);
    }
    if (h->fam_blocks[8]) {
        const dim3 g(h->fam_blocks[8]), b(1024u);
        if (ovp) hipLaunchKernelGGL((k_fq_hbatch_dyn<T, true, 16>), g, b, 0, st, descs, fmap[8]);
        else hipLaunchKernelGGL((k_fq_hbatch_dyn<T, false, 16>), g, b, 0, st, descs, fmap[8]);
    }
}
template <>
void launch_hbatch_dyn<float>(const BatchHeader *, const BatchDesc *, const uint32_t *const *, bool, hipStream_t) {}

extern "C" int antq_fakequant_batch(const void *batch_host, const void *batch_dev, void *stream)
{
    if (!batch_host || !batch_dev) return ANTQ_ERR_ARG;
    const BatchHeader *h = static_cast<const BatchHeader *>(batch_host);
    if (h->magic != kBatchMagic) return ANTQ_ERR_PLAN;
    if (h->total_blocks == 0) return ANTQ_OK;
    const char *pd = static_cast<const char *>(batch_dev);
    const BatchDesc *descs = reinterpret_cast<const BatchDesc *>(pd + sizeof(BatchHeader));
    const uint32_t *map = reinterpret_cast<const uint32_t *>(pd + h->map_offset);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 block(256);
    const bool ovp = (h->flags & ANTQ_FLAG_OVP) != 0;
    const bool dyn = (h->flags & ANTQ_FLAG_DYNAMIC) != 0;
    const uint32_t *fmap[kBatchFamilies];
    {
        const uint32_t *m = map;
        for (int f = 0; f < kBatchFamilies; f++) { fmap[f] = m; m += h->fam_blocks[f]; }
    }
#define ANTQ_LAUNCH_D(TT, OO, AA)                                                                                   \
    do {                                                                                                            \
        const int f_ = (AA) ? 1 : 2;                                                                                \
        /* (fp32 static lane jobs, launches of many rounds: 6 workgroups = 24 wavefronts per CU, +1.5..2 points; r04_fp32_occupancy.log) */ \
        const unsigned cap_ = (sizeof(TT) == 4 && !dyn && h->fam_blocks[f_] >= 32768u && h->lds_bytes < 24576u) ? 24576u - h->lds_bytes : 0u;  \
        const unsigned lds_ = h->lds_bytes + (g_knob_dlds >= 0 ? (unsigned)g_knob_dlds : cap_);                    \
        if (dyn) hipLaunchKernelGGL((k_fq_batch_d<TT, OO, AA, true>), dim3(h->fam_blocks[f_]), block, lds_, st, descs, fmap[f_]);  \
        else hipLaunchKernelGGL((k_fq_batch_d<TT, OO, AA, false>), dim3(h->fam_blocks[f_]), block, lds_, st, descs, fmap[f_]);     \
    } while (0)
#define ANTQ_LAUNCH_B(TT)                                                                                         \
    do {                                                                                                          \
        if (h->pad & 1u) {      /* mixed static batch: the all-in-one kernel */                                       \
            if (ovp) hipLaunchKernelGGL((k_fq_batch_all<TT, true>), dim3(h->fam_blocks[0]), block, h->lds_bytes, st, descs, fmap[0]);  \
            else hipLaunchKernelGGL((k_fq_batch_all<TT, false>), dim3(h->fam_blocks[0]), block, h->lds_bytes, st, descs, fmap[0]);     \
            if (h->fam_blocks[5]) launch_hbatch<TT>(h->fam_blocks[5], descs, fmap[5], ovp, st);   /* (a big 16-bit-domain share keeps its own launch) */ \
            break;                                                                                                \
        }                                                                                                         \
        if (h->fam_blocks[0]) {                                                                                   \
            const int hw_ = (int)((h->pad >> 8) & 7u);                                                           \
            const int w_ = (g_knob_waves == 4 || g_knob_waves == 2 || g_knob_waves == 1) ? g_knob_waves : (hw_ == 1 ? 1 : 4);   \
            const dim3 g_(h->fam_blocks[0] * (4u / w_)), b_(64u * w_);                                           \
            const uint32_t rot_ = (uint32_t)g_knob_rot;                                                          \
            if (w_ == 1) { if (ovp) hipLaunchKernelGGL((k_fq_batch<TT, true, 1>), g_, b_, 0, st, descs, fmap[0], rot_);        \
                           else hipLaunchKernelGGL((k_fq_batch<TT, false, 1>), g_, b_, 0, st, descs, fmap[0], rot_); }         \
            else if (w_ == 2) { if (ovp) hipLaunchKernelGGL((k_fq_batch<TT, true, 2>), g_, b_, 0, st, descs, fmap[0], rot_);   \
                           else hipLaunchKernelGGL((k_fq_batch<TT, false, 2>), g_, b_, 0, st, descs, fmap[0], rot_); }         \
            else { if (ovp) hipLaunchKernelGGL((k_fq_batch<TT, true, 4>), g_, b_, 0, st, descs, fmap[0], rot_);                \
                   else hipLaunchKernelGGL((k_fq_batch<TT, false, 4>), g_, b_, 0, st, descs, fmap[0], rot_); }                 \
        }                                                                                                         \
        if (h->fam_blocks[5]) launch_hbatch<TT>(h->fam_blocks[5], descs, fmap[5], ovp, st);                       \
        if (h->fam_blocks[6] || h->fam_blocks[7] || h->fam_blocks[8]) launch_hbatch_dyn<TT>(h, descs, fmap, ovp, st);      \
        if (h->fam_blocks[1]) { if (ovp) ANTQ_LAUNCH_D(TT, true, true); else ANTQ_LAUNCH_D(TT, false, true); }    \
        if (h->fam_blocks[2]) { if (ovp) ANTQ_LAUNCH_D(TT, true, false); else ANTQ_LAUNCH_D(TT, false, false); }  \
        if (h->fam_blocks[3]) {                                                                                   \
            if (ovp) hipLaunchKernelGGL((k_fq_batch_dyn<TT, true>), dim3(h->fam_blocks[3]), block, 0, st, descs, fmap[3]);   \
            else hipLaunchKernelGGL((k_fq_batch_dyn<TT, false>), dim3(h->fam_blocks[3]), block, 0, st, descs, fmap[3]);      \
        }                                                                                                         \
        if (h->fam_blocks[4]) {                                                                                   \
            if (ovp) hipLaunchKernelGGL((k_fq_batch_dyn16<TT, true>), dim3(h->fam_blocks[4]), dim3(1024), 0, st, descs, fmap[4]);   \
            else hipLaunchKernelGGL((k_fq_batch_dyn16<TT, false>), dim3(h->fam_blocks[4]), dim3(1024), 0, st, descs, fmap[4]);      \
        }                                                                                                         \
    } while (0)
    switch (h->dtype) {
    case ANTQ_F32: ANTQ_LAUNCH_B(float); break;
    case ANTQ_BF16: ANTQ_LAUNCH_B(bf16_tag); break;
    case ANTQ_F16: ANTQ_LAUNCH_B(f16_tag); break;
    default: return ANTQ_ERR_UNSUPPORTED;
    }
#undef ANTQ_LAUNCH_B
#undef ANTQ_LAUNCH_D
    return hipGetLastError() == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}

namespace antq {
int prefetch_unit_batch()        // antq_prefetch_kernels (antq_kernels.hip): load this unit's code object now
{
    hipFuncAttributes at;
    return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_fq_hbatch<bf16_tag, false>)) == hipSuccess ? ANTQ_OK : ANTQ_ERR_LAUNCH;
}
}  // namespace antq
