// exp_lane.hip -- round-3 dev experiment (no torch): anatomy of the one-launch-per-tensor lane kernel on a 4096 x 4096 bf16
// tensor (flint-4, per-row alpha), ordered and unordered launches.  Variants are built from the library's own device
// building blocks (csrc/antq_k_fakequant.h), so what is timed is the shipped arithmetic in other launch shapes:
//   base      k_fq_lane as shipped (4 wavefronts per workgroup, U vectors per lane)
//   nostage   the same without the table staging + barrier (table read from a stale LDS image: WRONG RESULTS, timing only)
//   noalpha   the same with a constant alpha instead of the per-vector gather
//   nowork    loads, staging, alpha gather, stores -- the element arithmetic replaced by a pass-through
//   persist   G workgroups, table staged once per workgroup, tasks walked grid-stride, next task's loads in flight
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/exp_lane.hip -o tools/exp_lane -Lant_quantization_amd -lantq -Wl,-rpath,'$ORIGIN/../ant_quantization_amd'
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <functional>

#include "../ant_quantization_amd/csrc/antq_host.h"
#include "../ant_quantization_amd/csrc/antq_k_fakequant.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

using namespace antq;
typedef bf16_tag T;

// MODE: 0 = shipped arithmetic, 1 = no staging / barrier, 2 = constant alpha, 3 = pass-through arithmetic
template <int U, int WAVES, int MODE>
__global__ void __launch_bounds__(64 * WAVES)
k_var(const uint4 *__restrict__ x, uint4 *__restrict__ out, size_t n_vec, uint32_t vpr, int vshift,
      const float *__restrict__ alpha, float gmax, PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    constexpr int EPL = 8;
    constexpr uint32_t TPB = 64u * WAVES;
    const size_t first = ((size_t)blockIdx.x * U) * TPB + threadIdx.x;
    uint4 tab0 = make_uint4(0, 0, 0, 0);
    if (MODE != 1) tab0 = atab_prefetch<false>(pa, plan_tab);
    uint4 v[U];
    float a[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * TPB;
        v[u] = make_uint4(0, 0, 0, 0);
        a[u] = 0.08f;
        if (vi < n_vec) {
            v[u] = ld_stream(x + vi);
            if (MODE != 2) a[u] = alpha[vi >> vshift];
        }
    }
    ATab A;
    if (MODE != 1) { A = stage_atab<false>(pa, plan_tab, smem, tab0); __syncthreads(); }
    else { A.tab = smem; A.grid = reinterpret_cast<const float *>(smem + pa.atab_slots); A.idx = nullptr; }
    const double inv_gmax = 1.0 / (double)gmax;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t vi = first + (size_t)u * TPB;
        if (vi < n_vec) {
            if (MODE == 3) {
                uint4 o = v[u];
                o.x ^= __float_as_uint(a[u]) & 1u;
                st_stream(out + vi, o);
            } else {
                float xf[EPL], of[EPL];
                int j[EPL];
                IO<T>::unpack(v[u], xf);
                const ScaleA sc = make_scale_a(a[u], gmax, inv_gmax);
                quant_vec_a<EPL, false, false>(pa, A, sc, xf, of, j);
                st_stream(out + vi, IO<T>::pack(of));
            }
        }
    }
}

// persistent: table staged once; wavefront-granular tasks of U vectors per lane (64 * U consecutive vectors), walked with a
// stride of all wavefronts of the launch; the next task's loads are issued before this task's arithmetic
template <int U, int WAVES, bool PREFETCH>
__global__ void __launch_bounds__(64 * WAVES)
k_persist(const uint4 *__restrict__ x, uint4 *__restrict__ out, size_t n_vec, uint32_t vpr, int vshift,
          const float *__restrict__ alpha, float gmax, PlanArgs pa, const uint4 *__restrict__ plan_tab)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    constexpr int EPL = 8;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t W = (size_t)gridDim.x * WAVES, w = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const size_t tasks = (n_vec + 64u * U - 1) / (64u * U);
    uint4 tab0 = atab_prefetch<false>(pa, plan_tab);
    uint4 v[U], nx[U];
    float a[U], na[U];
    size_t t = w;
    auto load = [&](size_t task, uint4 (&vv)[U], float (&aa)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t vi = task * (64u * U) + lane + 64u * u;
            vv[u] = make_uint4(0, 0, 0, 0);
            aa[u] = 1.0f;
            if (vi < n_vec) { vv[u] = ld_stream(x + vi); aa[u] = alpha[vi >> vshift]; }
        }
    };
    if (t < tasks) load(t, v, a);
    ATab A = stage_atab<false>(pa, plan_tab, smem, tab0);
    __syncthreads();
    const double inv_gmax = 1.0 / (double)gmax;
    while (t < tasks) {
        const size_t tn = t + W;
        if (PREFETCH && tn < tasks) load(tn, nx, na);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t vi = t * (64u * U) + lane + 64u * u;
            if (vi < n_vec) {
                float xf[EPL], of[EPL];
                int j[EPL];
                IO<T>::unpack(v[u], xf);
                const ScaleA sc = make_scale_a(a[u], gmax, inv_gmax);
                quant_vec_a<EPL, false, false>(pa, A, sc, xf, of, j);
                st_stream(out + vi, IO<T>::pack(of));
            }
        }
        if (PREFETCH) {
#pragma unroll
            for (int u = 0; u < U; u++) { v[u] = nx[u]; a[u] = na[u]; }
        } else if (tn < tasks) load(tn, v, a);
        t = tn;
    }
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NT = 16;
    const size_t rows = 4096, cols = 4096, tb = rows * cols * 2;
    std::vector<void *> in(NT), out(NT);
    std::vector<uint16_t> h(rows * cols);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        float f = ((float)(s & 0xffff) / 65536.0f + (float)((s >> 16) & 0xffff) / 65536.0f + (float)((s >> 32) & 0xffff) / 65536.0f - 1.5f) * 0.04f;
        uint32_t u; memcpy(&u, &f, 4); h[i] = (uint16_t)(u >> 16);
    }
    for (int i = 0; i < NT; i++) {
        CK(hipMalloc(&in[i], tb)); CK(hipMalloc(&out[i], tb));
        CK(hipMemcpy(in[i], h.data(), tb, hipMemcpyHostToDevice));
    }
    static const float flint4[16] = {-10.f, -5.f, -3.75f, -2.5f, -1.875f, -1.25f, -0.625f, 0.f, 0.f, 0.625f, 1.25f, 1.875f, 2.5f, 3.75f, 5.f, 10.f};
    std::vector<unsigned char> plan(ANTQ_PLAN_MAX_BYTES);
    const int pb = antq_plan_build(flint4, 16, plan.data(), plan.size());
    void *plan_dev; CK(hipMalloc(&plan_dev, pb)); CK(hipMemcpy(plan_dev, plan.data(), pb, hipMemcpyHostToDevice));
    std::vector<float> alpha(rows, 0.08f);
    float *alpha_dev; CK(hipMalloc(&alpha_dev, rows * 4)); CK(hipMemcpy(alpha_dev, alpha.data(), rows * 4, hipMemcpyHostToDevice));
    PlanArgs pa;
    if (!plan_args_from_host(plan.data(), pa) || !pa.adom) { printf("no adom plan\n"); return 1; }
    const size_t lds = lds_table(pa, false);
    const uint4 *tab = plan_tab_ptr(plan_dev);
    const size_t n_vec = rows * cols / 8;
    const uint32_t vpr = cols / 8;
    const int vshift = 9;
    printf("plan: %u a-table slots, %zu bytes of LDS per workgroup\n", pa.atab_slots, lds);

    auto time_pass = [&](const char *name, const std::function<void(int)> &launch) {
        for (int r = 0; r < 3; r++) for (int i = 0; i < NT; i++) launch(i);
        CK(hipStreamSynchronize(st));
        // warm clocks: ~60 ms
        for (int r = 0; r < 300; r++) for (int i = 0; i < NT; i++) launch(i);
        CK(hipStreamSynchronize(st));
        const int reps = 60;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++) for (int i = 0; i < NT; i++) launch(i);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * NT);
        printf("%-64s %7.2f us  %5.1f %%\n", name, us, 2.0 * tb / us / 1e6 / 8 * 100);
        fflush(stdout);
    };
#define VAR(U, W, M, ANY) [&](int i) {                                                                                          \
        const dim3 g((unsigned)((n_vec + 64u * W * U - 1) / (64u * W * U))), b(64 * W);                                         \
        if (ANY) hipExtLaunchKernelGGL((k_var<U, W, M>), g, b, lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, (const uint4 *)in[i], (uint4 *)out[i], n_vec, vpr, vshift, (const float *)alpha_dev, 10.0f, pa, tab); \
        else hipLaunchKernelGGL((k_var<U, W, M>), g, b, lds, st, (const uint4 *)in[i], (uint4 *)out[i], n_vec, vpr, vshift, (const float *)alpha_dev, 10.0f, pa, tab); }
#define PER(U, W, P, G, ANY) [&](int i) {                                                                                       \
        const dim3 g(G), b(64 * W);                                                                                             \
        if (ANY) hipExtLaunchKernelGGL((k_persist<U, W, P>), g, b, lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, (const uint4 *)in[i], (uint4 *)out[i], n_vec, vpr, vshift, (const float *)alpha_dev, 10.0f, pa, tab); \
        else hipLaunchKernelGGL((k_persist<U, W, P>), g, b, lds, st, (const uint4 *)in[i], (uint4 *)out[i], n_vec, vpr, vshift, (const float *)alpha_dev, 10.0f, pa, tab); }
    const PlanHeader *ph = reinterpret_cast<const PlanHeader *>(plan.data());
    const XArgs xa = xargs_from_plan(plan.data(), pa);
    const uint4 *entries = tab + (pa.m_pad >> 2);
    const float *grid_dev = reinterpret_cast<const float *>(tab);
#define XROW(U, W, ANY) [&](int i) {                                                                                            \
        const uint32_t tpr = (vpr + 64u * U - 1) / (64u * U), total = (uint32_t)rows * tpr;                                      \
        const dim3 g((total + W - 1) / W), b(64 * W);                                                                           \
        if (ANY) hipExtLaunchKernelGGL((k_fq_xrow<T, false, false, U, false, 1, W>), g, b, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, (const uint4 *)in[i], (uint4 *)out[i], (int16_t *)nullptr, total, vpr, tpr, (const float *)alpha_dev, 1, 10.0f, 1.0f, (float *)nullptr, xa, entries, grid_dev); \
        else hipLaunchKernelGGL((k_fq_xrow<T, false, false, U, false, 1, W>), g, b, 0, st, (const uint4 *)in[i], (uint4 *)out[i], (int16_t *)nullptr, total, vpr, tpr, (const float *)alpha_dev, 1, 10.0f, 1.0f, (float *)nullptr, xa, entries, grid_dev); }
    for (int any = 0; any < 2; any++) {
        printf("== %s launches ==\n", any ? "UNORDERED (hipExtAnyOrderLaunch)" : "ordered");
#define BOTH(NAME, L0, L1) time_pass(NAME, any ? std::function<void(int)>(L1) : std::function<void(int)>(L0))
        BOTH("base W=4 U=2 (shipped)", VAR(2, 4, 0, false), VAR(2, 4, 0, true));
        BOTH("base W=4 U=4", VAR(4, 4, 0, false), VAR(4, 4, 0, true));
        BOTH("base W=4 U=1", VAR(1, 4, 0, false), VAR(1, 4, 0, true));
        BOTH("base W=2 U=2", VAR(2, 2, 0, false), VAR(2, 2, 0, true));
        BOTH("base W=1 U=2", VAR(2, 1, 0, false), VAR(2, 1, 0, true));
        BOTH("base W=1 U=4", VAR(4, 1, 0, false), VAR(4, 1, 0, true));
        BOTH("noalpha W=4 U=2", VAR(2, 4, 2, false), VAR(2, 4, 2, true));
        BOTH("noalpha W=1 U=4", VAR(4, 1, 2, false), VAR(4, 1, 2, true));
        BOTH("nowork W=4 U=2", VAR(2, 4, 3, false), VAR(2, 4, 3, true));
        BOTH("nowork W=1 U=4", VAR(4, 1, 3, false), VAR(4, 1, 3, true));
        BOTH("persist W=4 U=2 g=2048 prefetch", PER(2, 4, true, 2048, false), PER(2, 4, true, 2048, true));
        BOTH("persist W=4 U=2 g=2048 no prefetch", PER(2, 4, false, 2048, false), PER(2, 4, false, 2048, true));
        BOTH("persist W=4 U=2 g=1024 prefetch", PER(2, 4, true, 1024, false), PER(2, 4, true, 1024, true));
        BOTH("persist W=4 U=1 g=2048 prefetch", PER(1, 4, true, 2048, false), PER(1, 4, true, 2048, true));
        BOTH("persist W=4 U=4 g=1024 prefetch", PER(4, 4, true, 1024, false), PER(4, 4, true, 1024, true));
        BOTH("persist W=1 U=2 g=8192 prefetch", PER(2, 1, true, 8192, false), PER(2, 1, true, 8192, true));
        BOTH("persist W=1 U=4 g=4096 prefetch", PER(4, 1, true, 4096, false), PER(4, 1, true, 4096, true));
        BOTH("persist W=2 U=2 g=4096 prefetch", PER(2, 2, true, 4096, false), PER(2, 2, true, 4096, true));
        BOTH("row-table kernel W=4 U=4", XROW(4, 4, false), XROW(4, 4, true));
        BOTH("row-table kernel W=4 U=2", XROW(2, 4, false), XROW(2, 4, true));
        BOTH("row-table kernel W=1 U=4", XROW(4, 1, false), XROW(4, 1, true));
        BOTH("row-table kernel W=1 U=2", XROW(2, 1, false), XROW(2, 1, true));
        BOTH("row-table kernel W=2 U=4", XROW(4, 2, false), XROW(4, 2, true));
        BOTH("row-table kernel W=1 U=8", XROW(8, 1, false), XROW(8, 1, true));
        BOTH("base W=4 U=2 (again)", VAR(2, 4, 0, false), VAR(2, 4, 0, true));
    }
    return 0;
}
