// launch_anatomy.hip -- where does the time of ONE streaming launch over a 33.5 MB tensor go?  (dev tool)
//
// A launch of the fused kernel over one 4096 x 4096 bf16 tensor takes ~14.5 us where the same work inside a batched launch
// takes ~10.4 us and a plain nontemporal copy ~12.3 us.  This probe runs copy-shaped kernels (16 B per lane, U vectors per
// lane, WORK dependent-free VALU ops per vector between the load and the store) over rotating 33.5 MB buffers and records,
// per wavefront, s_memtime at entry, when its first vector has arrived, and at exit.  Printed per variant: launch time
// (HIP events, back-to-back launches), the spread of wavefront start times (dispatch ramp), the time to first data, the
// time the last wavefront ends, and the percentiles of wavefront lifetimes -- enough to tell a dispatch-bound launch from a
// latency-bound one from a "read phase, then write phase" one.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_anatomy.hip -o tools/launch_anatomy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// s_memtime counters are per XCD and not synchronised with each other: every wavefront also records which XCD it ran on
__device__ __forceinline__ uint64_t xcc_id() { return (uint64_t)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf); }   // HW_REG_XCC_ID

__device__ __forceinline__ u4 work(u4 v, int n, float c1, float c2)
{
    float a = __uint_as_float(v.x), b = __uint_as_float(v.y), c = __uint_as_float(v.z), d = __uint_as_float(v.w);
    for (int i = 0; i < n; i += 4) {
        a = __builtin_fmaf(a, c1, c2); b = __builtin_fmaf(b, c1, c2); c = __builtin_fmaf(c, c1, c2); d = __builtin_fmaf(d, c1, c2);
    }
    v.x = __float_as_uint(a); v.y = __float_as_uint(b); v.z = __float_as_uint(c); v.w = __float_as_uint(d);
    return v;
}

// one-shot: wave handles U x 1 KiB contiguous; all U loads in flight, then per vector: WORK ops, store
template <int U, int THREADS>
__global__ void __launch_bounds__(THREADS) probe(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n, int nwork, float c1, float c2,
                                                 uint64_t *__restrict__ ts)
{
    const uint64_t t0 = __builtin_readcyclecounter();
    const size_t wave = (size_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    const size_t first = wave * (64 * U) + (threadIdx.x & 63);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(s + std::min(first + 64 * u, n - 1));
    uint64_t t1 = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        u4 w = work(v[u], nwork, c1, c2);
        if (u == 0) t1 = __builtin_readcyclecounter();
        if (first + 64 * u < n) __builtin_nontemporal_store(w, d + first + 64 * u);
    }
    const uint64_t t2 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { ts[4 * wave] = t0; ts[4 * wave + 1] = t1; ts[4 * wave + 2] = t2; ts[4 * wave + 3] = xcc_id(); }
}

// staged: the wave's U vectors in two halves; the second half is LOADED only after the first half has been stored
template <int U>
__global__ void __launch_bounds__(256) probe_staged(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n, int nwork, float c1, float c2,
                                                    uint64_t *__restrict__ ts)
{
    const uint64_t t0 = __builtin_readcyclecounter();
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t first = wave * (64 * U) + (threadIdx.x & 63);
    uint64_t t1 = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        u4 v[U / 2];
#pragma unroll
        for (int u = 0; u < U / 2; u++) v[u] = __builtin_nontemporal_load(s + std::min(first + 64 * (h * (U / 2) + u), n - 1));
#pragma unroll
        for (int u = 0; u < U / 2; u++) {
            u4 w = work(v[u], nwork, c1, c2);
            if (u == 0 && h == 0) t1 = __builtin_readcyclecounter();
            if (first + 64 * (h * (U / 2) + u) < n) __builtin_nontemporal_store(w, d + first + 64 * (h * (U / 2) + u));
        }
    }
    const uint64_t t2 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { ts[4 * wave] = t0; ts[4 * wave + 1] = t1; ts[4 * wave + 2] = t2; ts[4 * wave + 3] = xcc_id(); }
}

// persistent: G workgroups, each wave walks tasks of U x 1 KiB with stride; the next task's loads are issued before the
// current task's work (register double buffer)
template <int U>
__global__ void __launch_bounds__(256) probe_persist(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n, int nwork, float c1, float c2,
                                                     uint64_t *__restrict__ ts)
{
    const uint64_t t0 = __builtin_readcyclecounter();
    const size_t nw = (size_t)gridDim.x * 4, w0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t ntask = (n + 64 * U - 1) / (64 * U);
    const size_t lane = threadIdx.x & 63;
    u4 v[U], nx[U];
    uint64_t t1 = 0;
    size_t task = w0;
    if (task < ntask) {
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(s + std::min(task * (64 * U) + lane + 64 * u, n - 1));
    }
    while (task < ntask) {
        const size_t nt = task + nw;
        if (nt < ntask) {
#pragma unroll
            for (int u = 0; u < U; u++) nx[u] = __builtin_nontemporal_load(s + std::min(nt * (64 * U) + lane + 64 * u, n - 1));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            u4 w = work(v[u], nwork, c1, c2);
            if (t1 == 0) t1 = __builtin_readcyclecounter();
            const size_t i = task * (64 * U) + lane + 64 * u;
            if (i < n) __builtin_nontemporal_store(w, d + i);
        }
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = nx[u];
        task = nt;
    }
    const uint64_t t2 = __builtin_readcyclecounter();
    if (lane == 0) { ts[4 * w0] = t0; ts[4 * w0 + 1] = t1; ts[4 * w0 + 2] = t2; ts[4 * w0 + 3] = xcc_id(); }
}

struct Ctx {
    std::vector<void *> src, dst;
    size_t bytes, n;
    uint64_t *ts;
    hipStream_t st;
    hipEvent_t e0, e1;
};

template <typename L>
static void run(Ctx &C, const char *name, size_t waves, L launch)
{
    const int nb = (int)C.src.size();
    for (int i = 0; i < 2 * nb; i++) launch(C.src[i % nb], C.dst[i % nb]);       // warm-up (clocks)
    CK(hipStreamSynchronize(C.st));
    const int reps = 6 * nb;
    CK(hipEventRecord(C.e0, C.st));
    for (int i = 0; i < reps; i++) launch(C.src[i % nb], C.dst[i % nb]);
    CK(hipEventRecord(C.e1, C.st));
    CK(hipEventSynchronize(C.e1));
    float ms;
    CK(hipEventElapsedTime(&ms, C.e0, C.e1));
    const double us = ms * 1e3 / reps;
    // anatomy of the LAST launch, per XCD (each XCD has its own s_memtime); ticks -> us by equating the median XCD's
    // first-start-to-last-end span with the launch time measured by the events
    std::vector<uint64_t> h(4 * waves);
    CK(hipMemcpy(h.data(), C.ts, 4 * waves * 8, hipMemcpyDeviceToHost));
    uint64_t t_min[16], t_end[16];
    for (int x = 0; x < 16; x++) { t_min[x] = ~0ull; t_end[x] = 0; }
    for (size_t w = 0; w < waves; w++) {
        const int x = (int)(h[4 * w + 3] & 15);
        t_min[x] = std::min(t_min[x], h[4 * w]);
        t_end[x] = std::max(t_end[x], h[4 * w + 2]);
    }
    std::vector<double> spans;
    for (int x = 0; x < 16; x++) if (t_end[x]) spans.push_back((double)(t_end[x] - t_min[x]));
    std::sort(spans.begin(), spans.end());
    const double k = us / spans[spans.size() / 2];
    std::vector<double> st(waves), fd(waves), life(waves), en(waves);
    for (size_t w = 0; w < waves; w++) {
        const int x = (int)(h[4 * w + 3] & 15);
        st[w] = (double)(h[4 * w] - t_min[x]);
        fd[w] = (double)(h[4 * w + 1] - h[4 * w]);
        life[w] = (double)(h[4 * w + 2] - h[4 * w]);
        en[w] = (double)(h[4 * w + 2] - t_min[x]);
    }
    auto pct = [&](std::vector<double> &v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    printf("%-36s %6.2f us/launch %5.1f%% of 8 TB/s | wave start p50 %5.2f p99 %5.2f | load -> first use p50 %5.2f p99 %5.2f | "
           "wave life p50 %5.2f p99 %5.2f | wave end p50 %5.2f p99 %5.2f us  (%zu XCDs, %.0f ticks/us)\n",
           name, us, 2.0 * C.bytes / (us * 1e-6) / 8e12 * 100.0, pct(st, .5) * k, pct(st, .99) * k,
           pct(fd, .5) * k, pct(fd, .99) * k, pct(life, .5) * k, pct(life, .99) * k, pct(en, .5) * k, pct(en, .99) * k,
           spans.size(), 1.0 / k);
}

int main(int argc, char **argv)
{
    Ctx C;
    C.bytes = (size_t)4096 * 4096 * 2;
    if (argc > 1) C.bytes = (size_t)atol(argv[1]);
    C.n = C.bytes / 16;
    const int nb = 24;
    for (int i = 0; i < nb; i++) {
        void *a, *b;
        CK(hipMalloc(&a, C.bytes));
        CK(hipMalloc(&b, C.bytes));
        CK(hipMemset(a, 1, C.bytes));
        CK(hipMemset(b, 0, C.bytes));
        C.src.push_back(a);
        C.dst.push_back(b);
    }
    CK(hipMalloc(&C.ts, 4 * 8 * (C.n / 64 + 4096)));
    CK(hipStreamCreate(&C.st));
    CK(hipEventCreate(&C.e0));
    CK(hipEventCreate(&C.e1));
    const size_t n = C.n;
    printf("tensor %zu bytes, %d rotating buffer pairs\n", C.bytes, nb);
    char nm[96];
    for (int nwork : {4, 48, 96, 160}) {   // (4: one fma per float, so that the first-use stamp waits for the data)
#define ONE(U, THREADS)                                                                                               \
    snprintf(nm, 96, "one-shot U=%d wg=%d work=%d/vec", U, THREADS, nwork);                                           \
    run(C, nm, (n + 64 * U - 1) / (64 * U), [&](void *s, void *d) {                                                    \
        hipLaunchKernelGGL((probe<U, THREADS>), dim3((unsigned)((n + (size_t)THREADS * U - 1) / ((size_t)THREADS * U))), dim3(THREADS), 0, C.st, \
                           (const u4 *)s, (u4 *)d, n, nwork, 1.0001f, 1e-9f, C.ts); });
        ONE(1, 256) ONE(2, 256) ONE(4, 256) ONE(8, 256) ONE(4, 1024) ONE(2, 1024)
#undef ONE
        snprintf(nm, 96, "staged 2+2 U=4 work=%d/vec", nwork);
        run(C, nm, (n + 255) / 256, [&](void *s, void *d) {
            hipLaunchKernelGGL((probe_staged<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, C.st, (const u4 *)s, (u4 *)d, n, nwork, 1.0001f, 1e-9f, C.ts); });
        snprintf(nm, 96, "staged 4+4 U=8 work=%d/vec", nwork);
        run(C, nm, (n + 511) / 512, [&](void *s, void *d) {
            hipLaunchKernelGGL((probe_staged<8>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, C.st, (const u4 *)s, (u4 *)d, n, nwork, 1.0001f, 1e-9f, C.ts); });
        for (int g : {512, 1024, 2048}) {
            snprintf(nm, 96, "persistent U=2 grid=%d work=%d/vec", g, nwork);
            run(C, nm, (size_t)g * 4, [&](void *s, void *d) {
                hipLaunchKernelGGL((probe_persist<2>), dim3(g), dim3(256), 0, C.st, (const u4 *)s, (u4 *)d, n, nwork, 1.0001f, 1e-9f, C.ts); });
            snprintf(nm, 96, "persistent U=1 grid=%d work=%d/vec", g, nwork);
            run(C, nm, (size_t)g * 4, [&](void *s, void *d) {
                hipLaunchKernelGGL((probe_persist<1>), dim3(g), dim3(256), 0, C.st, (const u4 *)s, (u4 *)d, n, nwork, 1.0001f, 1e-9f, C.ts); });
        }
    }
    return 0;
}
