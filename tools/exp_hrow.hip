// exp_hrow.hip -- round-4 dev experiment (no torch): the 16-bit-domain row kernel (csrc/antq_k_hrow.h) in several launch
// shapes on the headline workload (32 x 4096 x 4096 bf16, flint-4, per-row alpha), against the shipped batched launch:
//   * bit-exactness against the library's own output (antq_fakequant_batch, oracle-verified) on random rows AND on a
//     tensor of awkward rows (NaN / Inf / far-clipped elements, zero / negative / tiny / huge alphas, exact thresholds);
//   * per-launch durations of the first 40 launches after the GPU idled 1.5 s, in steady state, and in the bench's situation
//     (idle, 0.45 s of a slower kernel, 5 warm-up launches, 20 timed: what the driver's --steps 20 --warmup 5 sees).
// Findings (profiles/r04_exp_hrow_*.log): bytes in flight per CU decide the steady state (64-96 KiB best; 128 KiB: -3 points);
// whole-row tasks (8 KiB per wavefront, streamed or not) and persistent workgroups are slower than 2-4 KiB one-shot tasks;
// the start-up dip after idle grows with the VALU work per byte and with the number of wavefronts launched.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/exp_hrow.hip -o tools/exp_hrow -Lant_quantization_amd -lantq -Wl,-rpath,'$ORIGIN/../ant_quantization_amd'
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <functional>
#include <vector>

#include "../ant_quantization_amd/csrc/antq_host.h"
#include "../ant_quantization_amd/csrc/antq_k_hrow.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

using namespace antq;

struct Job { const uint4 *x; uint4 *out; const float *alpha; };

// one task = up to 64 * VPT vectors of one row; one wavefront per workgroup; jobs of equal shape
template <typename T, bool OVP, int VPT, bool WORK>
__global__ void __launch_bounds__(64)
k_hrow(const Job *__restrict__ jobs, uint32_t tasks_per_job, uint32_t vpr, uint32_t tpr, float gmax, HArgs ha,
       const uint4 *__restrict__ tlist, const float *__restrict__ grid)
{
    __shared__ __attribute__((aligned(16))) uint2 tab[kHSlots * 2];
    const uint32_t lane = threadIdx.x;
    const uint32_t j = blockIdx.x / tasks_per_job, task = blockIdx.x - j * tasks_per_job;
    const Job J = jobs[j];
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint4 thr = ld_global(tlist + min(lane, ha.n_thr - 1u));
    const float a = ld_global(J.alpha + row);
    const uint32_t v0 = g * (64u * VPT) + lane;
    const uint4 *p = J.x + (size_t)row * vpr;
    uint4 v[VPT];
#pragma unroll
    for (int u = 0; u < VPT; u++) v[u] = ld_stream(p + min(v0 + 64u * u, vpr - 1u));
    __builtin_amdgcn_sched_barrier(0);
    uint4 *q = J.out + (size_t)row * vpr + v0;
    if (!WORK) {
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            if (v0 + 64u * u < vpr) { uint4 o = v[u]; o.x ^= f2u(a) & 1u; st_stream(q + 64u * u, o); }
        }
        return;
    }
    const Scale sc = row_scale(a, gmax, ha.inv_gmax);
    const HRow R = hrow_build<T>(ha, thr, sc, tab, lane);
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)tab;
    const uint32_t othr = OVP ? H16<T>::out(ha.vout * sc.s) & 0x7fffu : 0u;
    const HFar far = {sc.rs, ha.flim, ha.vmin, ha.vmax};
    hrow_task<T, OVP, VPT>(v, q, v0, vpr, R, tab_addr, ha.hshift, othr, sc.s, far, grid, ha.m);
}

// WAVES wavefronts per workgroup share ONE table: the workgroup covers 64 * VPT * WAVES consecutive vectors of one row (the
// window of addresses in flight stays as compact as with 2 KiB tasks), wavefront 0 builds the row's table, one barrier.
template <typename T, bool OVP, int VPT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
k_hrow_w(const Job *__restrict__ jobs, uint32_t tasks_per_job, uint32_t vpr, uint32_t tpr, float gmax, HArgs ha,
         const uint4 *__restrict__ tlist, const float *__restrict__ grid)
{
    __shared__ __attribute__((aligned(16))) uint2 tab[kHSlots * 2];
    __shared__ uint32_t rowinfo[4];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t j = blockIdx.x / tasks_per_job, task = blockIdx.x - j * tasks_per_job;
    const Job J = jobs[j];
    uint32_t row = task, g = 0;
    if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
    const uint4 thr = ld_global(tlist + min(lane, ha.n_thr - 1u));
    const float a = ld_global(J.alpha + row);
    const uint32_t v0 = (g * WAVES + wv) * (64u * VPT) + lane;
    const uint4 *p = J.x + (size_t)row * vpr;
    uint4 v[VPT];
#pragma unroll
    for (int u = 0; u < VPT; u++) v[u] = ld_stream(p + min(v0 + 64u * u, vpr - 1u));
    __builtin_amdgcn_sched_barrier(0);
    uint4 *q = J.out + (size_t)row * vpr + v0;
    const Scale sc = row_scale(a, gmax, ha.inv_gmax);
    HRow R;
    if (wv == 0) {
        R = hrow_build<T>(ha, thr, sc, tab, lane);
        if (lane == 0) { rowinfo[0] = R.kmin; rowinfo[1] = R.klim; rowinfo[2] = R.fast ? 1u : 0u; }
    }
    __syncthreads();
    R.kmin = rowinfo[0]; R.klim = rowinfo[1]; R.fast = rowinfo[2] != 0u;
    R.kmin = __builtin_amdgcn_readfirstlane(R.kmin); R.klim = __builtin_amdgcn_readfirstlane(R.klim);
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)tab;
    const uint32_t othr = OVP ? H16<T>::out(ha.vout * sc.s) & 0x7fffu : 0u;
    const HFar far = {sc.rs, ha.flim, ha.vmin, ha.vmax};
    hrow_task<T, OVP, VPT>(v, q, v0, vpr, R, tab_addr, ha.hshift, othr, sc.s, far, grid, ha.m);
}

// persistent: G one-wavefront workgroups walk the tasks grid-stride, the next task's loads in flight during the current one
template <typename T, bool OVP, int VPT, bool PREFETCH>
__global__ void __launch_bounds__(64)
k_hrow_p(const Job *__restrict__ jobs, uint32_t tasks_per_job, uint32_t total, uint32_t vpr, uint32_t tpr, float gmax, HArgs ha,
         const uint4 *__restrict__ tlist, const float *__restrict__ grid)
{
    __shared__ __attribute__((aligned(16))) uint2 tab[kHSlots * 2];
    const uint32_t lane = threadIdx.x;
    const uint4 thr = ld_global(tlist + min(lane, ha.n_thr - 1u));
    const uint32_t tab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)tab;
    uint32_t t = blockIdx.x;
    uint4 v[VPT], nx[VPT];
    float a, na;
    auto locate = [&](uint32_t tt, const uint4 *&px, uint4 *&po, const float *&pa, uint32_t &v0) {
        const uint32_t j = tt / tasks_per_job, task = tt - j * tasks_per_job;
        const Job J = jobs[j];
        uint32_t row = task, g = 0;
        if (tpr != 1) { row = task / tpr; g = task - row * tpr; }
        v0 = g * (64u * VPT) + lane;
        px = J.x + (size_t)row * vpr; po = J.out + (size_t)row * vpr + v0; pa = J.alpha + row;
    };
    const uint4 *px; uint4 *po, *npo = nullptr; const float *pa; uint32_t v0, nv0 = 0;
    if (t >= total) return;
    locate(t, px, po, pa, v0);
    a = ld_global(pa);
#pragma unroll
    for (int u = 0; u < VPT; u++) v[u] = ld_stream(px + min(v0 + 64u * u, vpr - 1u));
    while (true) {
        const uint32_t tn = t + gridDim.x;
        const bool more = tn < total;
        if (PREFETCH && more) {
            const uint4 *npx; const float *npa;
            locate(tn, npx, npo, npa, nv0);
            na = ld_global(npa);
#pragma unroll
            for (int u = 0; u < VPT; u++) nx[u] = ld_stream(npx + min(nv0 + 64u * u, vpr - 1u));
        }
        __builtin_amdgcn_sched_barrier(0);
        const Scale sc = row_scale(a, gmax, ha.inv_gmax);
        const HRow R = hrow_build<T>(ha, thr, sc, tab, lane);
        const uint32_t othr = OVP ? H16<T>::out(ha.vout * sc.s) & 0x7fffu : 0u;
        const HFar far = {sc.rs, ha.flim, ha.vmin, ha.vmax};
        hrow_task<T, OVP, VPT>(v, po, v0, vpr, R, tab_addr, ha.hshift, othr, sc.s, far, grid, ha.m);
        if (!more) break;
        __builtin_amdgcn_wave_barrier();
        if (PREFETCH) {
#pragma unroll
            for (int u = 0; u < VPT; u++) v[u] = nx[u];
            a = na; po = npo; v0 = nv0;
        } else {
            locate(tn, px, po, pa, v0);
            a = ld_global(pa);
#pragma unroll
            for (int u = 0; u < VPT; u++) v[u] = ld_stream(px + min(v0 + 64u * u, vpr - 1u));
        }
        t = tn;
    }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char **argv)
{
    const bool ovp = argc > 1 && !strcmp(argv[1], "olive");
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int NT = 32;
    const size_t rows = 4096, cols = 4096, tb = rows * cols * 2;
    const uint32_t vpr = cols / 8;
    std::vector<void *> in(NT), out(NT), ref(NT);
    std::vector<uint16_t> h(rows * cols);
    std::vector<float> alpha(rows);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto gauss = [&]() { uint64_t r = rnd(); return ((float)(r & 0xffff) / 65536.0f + (float)((r >> 16) & 0xffff) / 65536.0f + (float)((r >> 32) & 0xffff) / 65536.0f - 1.5f) * 2.0f; };
    float *alpha_dev[2];
    for (int i = 0; i < NT; i++) { CK(hipMalloc(&in[i], tb)); CK(hipMalloc(&out[i], tb)); CK(hipMalloc(&ref[i], tb)); }
    // tensor 1..: gaussian rows, alpha = row abs-max (weights); with `olive`: 0.1 % of the entries multiplied by U(8, 64), alpha = 3 sigma
    for (size_t r = 0; r < rows; r++) {
        float amax = 0.0f;
        for (size_t c = 0; c < cols; c++) {
            float f = gauss() * 0.02f;
            if (ovp && (rnd() % 1000u) == 0u) f *= 8.0f + (float)(rnd() % 5600u) * 0.01f;
            const uint16_t b = f2bf(f);
            h[r * cols + c] = b;
            uint32_t u = (uint32_t)b << 16; float back; memcpy(&back, &u, 4);
            amax = fmaxf(amax, fabsf(back));
        }
        alpha[r] = ovp ? 0.06f : amax;
    }
    CK(hipMalloc(&alpha_dev[0], rows * 4)); CK(hipMemcpy(alpha_dev[0], alpha.data(), rows * 4, hipMemcpyHostToDevice));
    for (int i = 1; i < NT; i++) CK(hipMemcpy(in[i], h.data(), tb, hipMemcpyHostToDevice));
    // tensor 0: awkward rows
    {
        std::vector<float> al(rows);
        const uint16_t specials[] = {0x7fc0, 0xffc0, 0x7f80, 0xff80, 0x0000, 0x8000, 0x0001, 0x8001, 0x007f, 0x0080, 0x7f7f, 0xff7f, 0x3f80, 0xbf80, 0x4b00, 0xcb00};
        for (size_t r = 0; r < rows; r++) {
            float a;
            switch (r % 16) {
            case 0: a = 0.0f; break;
            case 1: a = -0.05f; break;
            case 2: a = 1e-30f; break;
            case 3: a = 1e30f; break;
            case 4: a = NAN; break;
            case 5: a = INFINITY; break;
            case 6: a = 1e-12f; break;
            case 7: a = 3e11f; break;
            default: a = 0.01f * (float)(1 + rnd() % 1000u) * (1.0f + (float)(rnd() % 4096u) / 4096.0f); break;
            }
            al[r] = a;
            const float sc = (isfinite(a) && a > 0.0f) ? a : 0.05f;
            for (size_t c = 0; c < cols; c++) {
                float f = gauss() * sc * 0.4f;
                const uint64_t k = rnd() % 64u;
                if (k == 0) f *= 16.0f + (float)(rnd() % 1000u);                 // far-clipped
                uint16_t b = f2bf(f);
                if (k == 1) b = specials[rnd() % 16u];
                if (k == 2) b = (uint16_t)rnd();                                    // any pattern at all
                if (k == 3) {                                                       // near a threshold: (i + 0.5) * s-ish patterns
                    const float t = (0.3125f + 0.625f * (float)(rnd() % 16u)) * sc / 10.0f * ((rnd() & 1u) ? 1.0f : -1.0f);
                    b = (uint16_t)(f2bf(t) + (int)(rnd() % 5u) - 2);
                }
                h[r * cols + c] = b;
            }
        }
        CK(hipMemcpy(in[0], h.data(), tb, hipMemcpyHostToDevice));
        CK(hipMalloc(&alpha_dev[1], rows * 4)); CK(hipMemcpy(alpha_dev[1], al.data(), rows * 4, hipMemcpyHostToDevice));
    }
    // codebook + plan
    static const float flint4[16] = {-10.f, -5.f, -3.75f, -2.5f, -1.875f, -1.25f, -0.625f, 0.f, 0.f, 0.625f, 1.25f, 1.875f, 2.5f, 3.75f, 5.f, 10.f};
    std::vector<float> gridv(flint4, flint4 + 16);
    float gmax = 10.0f;
    if (ovp) {        // OliVe flint-4 signed (15 values, max 32... scaled x 32 / 2^exp) + abfloat outliers (14)
        static const float on[15] = {-24.f, -16.f, -12.f, -8.f, -6.f, -4.f, -2.f, 0.f, 2.f, 4.f, 6.f, 8.f, 12.f, 16.f, 24.f};
        static const float oo[14] = {-384.f, -256.f, -192.f, -128.f, -96.f, -64.f, -48.f, 48.f, 64.f, 96.f, 128.f, 192.f, 256.f, 384.f};
        gridv.assign(on, on + 15); gridv.insert(gridv.end(), oo, oo + 14);
        gmax = 24.0f;
    }
    std::vector<unsigned char> plan(ANTQ_PLAN_MAX_BYTES);
    const int pb = antq_plan_build(gridv.data(), (int)gridv.size(), plan.data(), plan.size());
    void *plan_dev; CK(hipMalloc(&plan_dev, pb)); CK(hipMemcpy(plan_dev, plan.data(), pb, hipMemcpyHostToDevice));
    const PlanHeader *ph = reinterpret_cast<const PlanHeader *>(plan.data());
    PlanArgs pa;
    if (!plan_args_from_host(plan.data(), pa) || !ph->xdom) { printf("no x-domain plan\n"); return 1; }
    // threshold list from the plan's entries
    std::vector<HThr> tl;
    {
        const LutEntry *ent = plan_entries(plan.data());
        for (uint32_t i = 0; i < ph->n_entries; i++)
            if (ent[i].T < INFINITY) tl.push_back({ent[i].T, ent[i].v_lo, ent[i].v_hi, ((ent[i].idx >> 15) & 1u) | ((ent[i].idx >> 30) & 2u)});
        std::sort(tl.begin(), tl.end(), [](const HThr &a, const HThr &b) { return a.T < b.T; });
        tl.erase(std::unique(tl.begin(), tl.end(), [](const HThr &a, const HThr &b) { return a.T == b.T; }), tl.end());
    }
    HArgs ha;
    ha.n_thr = (uint32_t)tl.size();
    ha.n_neg = 0;
    for (auto &t : tl) if (t.T < 0.0f) ha.n_neg++;
    double rmin = 1e30;
    for (size_t i = 0; i + 1 < tl.size(); i++)
        if ((tl[i].T < 0) == (tl[i + 1].T < 0)) { const double a = fabs(tl[i].T), b = fabs(tl[i + 1].T); rmin = std::min(rmin, std::max(a, b) / std::min(a, b)); }
    int mb = 0;
    while (mb <= 7 && (1.0 + ldexp(1.0, -mb)) * (1.0 + ldexp(1.0, -7)) > rmin * (1.0 - 1e-6)) mb++;
    if (mb > 7 || tl.size() > kHMaxThr) { printf("grid not eligible (mb %d, %zu thresholds)\n", mb, tl.size()); return 1; }
    ha.hshift = 7 - mb;
    ha.m = ph->m;
    ha.flim = ph->fastlim * 0.99999f;
    ha.lim = fminf(ha.flim, ph->xlim);
    { const XArgs xe = xargs_from_plan(plan.data(), pa); ha.vmin = xe.vmin; ha.vmax = xe.vmax; }
    ha.vout = ph->vout;
    ha.inv_gmax = 1.0 / (double)gmax;
    printf("%s: %u thresholds (%u negative), min ratio %.4f -> %d mantissa bits in the key (hshift %u), lim %.3f, vout %.1f\n",
           ovp ? "olive flint-4 + outliers" : "flint-4", ha.n_thr, ha.n_neg, rmin, mb, ha.hshift, ha.lim, ha.vout);
    HThr *tl_dev; CK(hipMalloc(&tl_dev, 16 * 64)); CK(hipMemset(tl_dev, 0, 16 * 64)); CK(hipMemcpy(tl_dev, tl.data(), 16 * tl.size(), hipMemcpyHostToDevice));
    const float *grid_dev = reinterpret_cast<const float *>(plan_tab_ptr(plan_dev));

    // reference: the shipped batched launch
    std::vector<antq_job> jobs(NT);
    std::vector<Job> hj(NT);
    for (int i = 0; i < NT; i++) {
        jobs[i] = {in[i], ref[i], alpha_dev[i == 0 ? 1 : 0], rows, cols, 1, gmax, plan.data(), plan_dev};
        hj[i] = {(const uint4 *)in[i], (uint4 *)out[i], alpha_dev[i == 0 ? 1 : 0]};
    }
    const size_t cap = antq_batch_capacity(jobs.data(), NT, ANTQ_BF16);
    std::vector<unsigned char> blob(cap);
    const int bb = antq_batch_build(jobs.data(), NT, ANTQ_BF16, ovp ? ANTQ_FLAG_OVP : 0, blob.data(), cap);
    if (bb <= 0) { printf("batch build failed %d\n", bb); return 1; }
    void *blob_dev; CK(hipMalloc(&blob_dev, bb)); CK(hipMemcpy(blob_dev, blob.data(), bb, hipMemcpyHostToDevice));
    Job *jobs_all; CK(hipMalloc(&jobs_all, sizeof(Job) * NT)); CK(hipMemcpy(jobs_all, hj.data(), sizeof(Job) * NT, hipMemcpyHostToDevice));
    if (antq_fakequant_batch(blob.data(), blob_dev, st) != ANTQ_OK) { printf("lib launch failed\n"); exit(1); }
    CK(hipStreamSynchronize(st));
    // the timed launches leave the awkward tensor out (its exact-path rows are a tail of their own): a 32nd clean tensor instead
    void *in0c, *out0c; CK(hipMalloc(&in0c, tb)); CK(hipMalloc(&out0c, tb)); CK(hipMemcpy(in0c, in[1], tb, hipMemcpyDeviceToDevice));
    jobs[0] = {in0c, out0c, alpha_dev[0], rows, cols, 1, gmax, plan.data(), plan_dev};
    hj[0] = {(const uint4 *)in0c, (uint4 *)out0c, alpha_dev[0]};
    for (int i = 1; i < NT; i++) jobs[i].out_dev = out[i];
    std::vector<unsigned char> blob2(cap);
    const int bb2 = antq_batch_build(jobs.data(), NT, ANTQ_BF16, ovp ? ANTQ_FLAG_OVP : 0, blob2.data(), cap);
    void *blob2_dev; CK(hipMalloc(&blob2_dev, bb2)); CK(hipMemcpy(blob2_dev, blob2.data(), bb2, hipMemcpyHostToDevice));
    Job *jobs_clean; CK(hipMalloc(&jobs_clean, sizeof(Job) * NT)); CK(hipMemcpy(jobs_clean, hj.data(), sizeof(Job) * NT, hipMemcpyHostToDevice));
    const Job *jobs_dev = jobs_clean;
    auto lib = [&]() { if (antq_fakequant_batch(blob2.data(), blob2_dev, st) != ANTQ_OK) { printf("lib launch failed\n"); exit(1); } };
    std::vector<uint16_t> href(rows * cols), hout(rows * cols);

    auto check = [&](const char *name) {
        size_t bad = 0, first = 0;
        for (int i : {0, 1, NT - 1}) {
            CK(hipMemcpy(href.data(), ref[i], tb, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hout.data(), out[i], tb, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < rows * cols; k++)
                if (href[k] != hout[k]) { if (!bad) first = k + (size_t)i * rows * cols; bad++; }
        }
        if (bad) {
            const size_t i = first / (rows * cols), k = first % (rows * cols);
            std::vector<uint16_t> hx(rows * cols);
            CK(hipMemcpy(hx.data(), in[i], tb, hipMemcpyDeviceToHost));
            CK(hipMemcpy(href.data(), ref[i], tb, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hout.data(), out[i], tb, hipMemcpyDeviceToHost));
            std::vector<float> al(rows); CK(hipMemcpy(al.data(), alpha_dev[i == 0 ? 1 : 0], rows * 4, hipMemcpyDeviceToHost));
            printf("  !! %s: %zu mismatches; first: tensor %zu row %zu col %zu x=%04x alpha=%g ref=%04x got=%04x\n", name, bad, i, k / cols,
                   k % cols, hx[k], al[k / cols], href[k], hout[k]);
        } else printf("  ok %s: bit-identical to the library on 3 tensors (incl. the awkward one)\n", name);
        for (int i = 0; i < NT; i++) CK(hipMemsetAsync(out[i], 0xA5, tb, st));
        CK(hipStreamSynchronize(st));
    };

    const double BYTES = (double)NT * tb * 2;
    std::function<void()> prelude;
    auto measure = [&](const char *name, const std::function<void()> &launch, bool verify) {
        if (verify) { jobs_dev = jobs_all; launch(); CK(hipStreamSynchronize(st)); check(name); jobs_dev = jobs_clean; }
        launch(); CK(hipStreamSynchronize(st));
        const int N = 40;
        std::vector<hipEvent_t> ev(N + 1);
        for (auto &e : ev) CK(hipEventCreate(&e));
        usleep(1500000);
        CK(hipEventRecord(ev[0], st));
        for (int i = 0; i < N; i++) { launch(); CK(hipEventRecord(ev[i + 1], st)); }
        CK(hipStreamSynchronize(st));
        std::vector<float> d(N);
        for (int i = 0; i < N; i++) { CK(hipEventElapsedTime(&d[i], ev[i], ev[i + 1])); d[i] *= 1e3f; }
        double idle = 0; for (int i = 5; i < 25; i++) idle += d[i]; idle /= 20;
        // steady: 0.4 s of itself, then 60 launches
        const double t0 = (double)clock() / CLOCKS_PER_SEC;
        while ((double)clock() / CLOCKS_PER_SEC - t0 < 0.4) { for (int i = 0; i < 20; i++) launch(); CK(hipStreamSynchronize(st)); }
        CK(hipEventRecord(ev[0], st));
        for (int i = 0; i < 60; i++) launch();
        CK(hipEventRecord(ev[1], st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        const double steady = ms * 1e3 / 60;
        double mx = 0; for (int i = 0; i < N; i++) mx = std::max(mx, (double)d[i]);
        // the bench's situation: 1.5 s idle, 0.45 s of a slower kernel (the one-launch-per-tensor passes), 5 warm-up launches, a sync, 20 timed
        usleep(1500000);
        {
            const double t1 = (double)clock() / CLOCKS_PER_SEC;
            while ((double)clock() / CLOCKS_PER_SEC - t1 < 0.45) { for (int i = 0; i < 5; i++) prelude(); CK(hipStreamSynchronize(st)); }
        }
        for (int i = 0; i < 5; i++) launch();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(ev[0], st));
        for (int i = 0; i < 20; i++) launch();
        CK(hipEventRecord(ev[1], st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        const double bench = ms * 1e3 / 20;
        printf("%-44s from idle 6..25: %6.1f us = %5.2f %% (max %5.1f) | steady %6.1f us = %5.2f %% | bench-like %6.1f us = %5.2f %%\n", name, idle,
               BYTES / idle / 8e6 * 100, mx, steady, BYTES / steady / 8e6 * 100, bench, BYTES / bench / 8e6 * 100);
        printf("    ");
        for (int i = 0; i < 30; i++) printf("%.0f ", d[i]);
        printf("\n");
        fflush(stdout);
        for (auto &e : ev) CK(hipEventDestroy(e));
    };

#define ONE(VPT, WORK) ONEL(VPT, WORK, 0)
#define ONEL(VPT, WORK, LDSPAD) [&]() {                                                                                              \
        const uint32_t tpr = (vpr + 64u * VPT - 1) / (64u * VPT), tpj = (uint32_t)rows * tpr;                                \
        if (ovp) hipLaunchKernelGGL((k_hrow<bf16_tag, true, VPT, WORK>), dim3(tpj * NT), dim3(64), LDSPAD, st, (const Job *)jobs_dev, tpj, vpr, tpr, gmax, ha, (const uint4 *)tl_dev, grid_dev); \
        else hipLaunchKernelGGL((k_hrow<bf16_tag, false, VPT, WORK>), dim3(tpj * NT), dim3(64), LDSPAD, st, (const Job *)jobs_dev, tpj, vpr, tpr, gmax, ha, (const uint4 *)tl_dev, grid_dev); }
#define WG(VPT, W) WGL(VPT, W, 0)
#define WGL(VPT, W, LDSPAD) [&]() {                                                                                        \
        const uint32_t tpr = (vpr + 64u * VPT * W - 1) / (64u * VPT * W), tpj = (uint32_t)rows * tpr;                        \
        if (ovp) hipLaunchKernelGGL((k_hrow_w<bf16_tag, true, VPT, W>), dim3(tpj * NT), dim3(64 * W), LDSPAD, st, (const Job *)jobs_dev, tpj, vpr, tpr, gmax, ha, (const uint4 *)tl_dev, grid_dev); \
        else hipLaunchKernelGGL((k_hrow_w<bf16_tag, false, VPT, W>), dim3(tpj * NT), dim3(64 * W), LDSPAD, st, (const Job *)jobs_dev, tpj, vpr, tpr, gmax, ha, (const uint4 *)tl_dev, grid_dev); }
#define PER(VPT, G, PF) [&]() {                                                                                                 \
        const uint32_t tpr = (vpr + 64u * VPT - 1) / (64u * VPT), tpj = (uint32_t)rows * tpr;                                \
        if (ovp) hipLaunchKernelGGL((k_hrow_p<bf16_tag, true, VPT, PF>), dim3(G), dim3(64), 0, st, (const Job *)jobs_dev, tpj, tpj * NT, vpr, tpr, gmax, ha, (const uint4 *)tl_dev, grid_dev); \
        else hipLaunchKernelGGL((k_hrow_p<bf16_tag, false, VPT, PF>), dim3(G), dim3(64), 0, st, (const Job *)jobs_dev, tpj, tpj * NT, vpr, tpr, gmax, ha, (const uint4 *)tl_dev, grid_dev); }

    prelude = ONEL(8, true, 39 * 1024);          // ~58 % of peak: stands in for the per-tensor passes
    measure("library k_fq_batch (shipped)", lib, false);
    measure("hrow one-shot VPT=2", ONE(2, true), true);
    measure("hrow one-shot VPT=4", ONE(4, true), true);
    measure("hrow one-shot VPT=4 28 waves/CU", ONEL(4, true, 4 * 1024 + 512), true);
    measure("hrow one-shot VPT=4 24 waves/CU", ONEL(4, true, 5 * 1024 + 512), true);
    measure("hrow one-shot VPT=4 20 waves/CU", ONEL(4, true, 7 * 1024), true);
    measure("hrow one-shot VPT=4 16 waves/CU", ONEL(4, true, 9 * 1024), true);
    measure("hrow shared W=2 VPT=2", WG(2, 2), true);
    measure("hrow shared W=2 VPT=2 12 wg/CU", WGL(2, 2, 12 * 1024), true);
    measure("hrow shared W=2 VPT=4 (row) 16 wg/CU", WG(4, 2), true);
    measure("hrow shared W=2 VPT=4 (row) 12 wg/CU", WGL(4, 2, 12 * 1024), true);
    measure("hrow shared W=2 VPT=4 (row) 10 wg/CU", WGL(4, 2, 15 * 1024), true);
    measure("hrow shared W=2 VPT=4 (row) 8 wg/CU", WGL(4, 2, 19 * 1024), true);
    measure("hrow shared W=4 VPT=2 (row) 6 wg/CU", WGL(2, 4, 25 * 1024), true);
    measure("hrow shared W=4 VPT=2 (row) 4 wg/CU", WGL(2, 4, 39 * 1024), true);
    measure("library k_fq_batch (again)", lib, false);
    return 0;
}
