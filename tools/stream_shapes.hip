// stream_shapes.hip -- round-3 dev micro-benchmark (no torch, no libantq): what a read-N + write-N element-wise stream
// can reach on MI355X (a) in steady state (one launch over 1 GiB in + 1 GiB out: the batched entry's regime) as a
// function of launch shape, cache policy bits, workgroup -> address mapping and VALU work per 16-byte vector, and
// (b) as ONE LAUNCH PER 33.5 MB TENSOR (the reference's granularity) as a function of how the launches are issued:
// one stream, 2 / 4 streams, hipExtAnyOrderLaunch, a hipGraph of independent kernel nodes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_shapes.hip -o tools/stream_shapes
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0xffffffff, 0x00020000);
}

// N dependent fma per dword: stands for the quantiser's arithmetic (N * 4 VALU ops per 16-byte vector)
template <int N>
__device__ __forceinline__ u4 work(u4 v, float c1, float c2)
{
    if (N == 0) return v;
    float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int e = 0; e < 4; e++) f[e] = __builtin_fmaf(f[e], c1, c2);
    }
    u4 o = {__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    return o;
}

// MAP: 0 = block b takes chunk b; 1 = XCD-contiguous (block b runs on XCD b % 8: XCD x takes chunks [x * nb / 8, (x + 1) * nb / 8))
__device__ __forceinline__ unsigned remap(unsigned b, unsigned nb, int MAP)
{
    if (MAP == 1) return (b & 7u) * (nb >> 3) + (b >> 3);
    return b;
}

// one-shot: a wavefront takes U consecutive KiB (WAVES wavefronts per workgroup)
template <int U, int LA, int SA, int N, int MAP, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_oneshot(const u4 *__restrict__ s, u4 *__restrict__ d, unsigned n_vec, float c1, float c2)
{
    const auto rs = rsrc(s), rd = rsrc(d);
    const unsigned blk = remap(blockIdx.x, gridDim.x, MAP);
    const unsigned wave = blk * WAVES + (threadIdx.x >> 6);
    const unsigned first = wave * (64u * U) + (threadIdx.x & 63u);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (first + 64u * u) << 4, 0, LA);
#pragma unroll
    for (int u = 0; u < U; u++) {
        u4 o = work<N>(v[u], c1, c2);
        if (first + 64u * u < n_vec) __builtin_amdgcn_raw_buffer_store_b128(o, rd, (first + 64u * u) << 4, 0, SA);
    }
}

// persistent: G workgroups; wavefront w of the launch walks tasks w, w + W, ... (W = all wavefronts) with the next task's
// loads issued before this task's arithmetic and stores (CONTIG = 1: a wavefront walks its own contiguous region instead)
template <int U, int LA, int SA, int N, int CONTIG, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES) k_persist(const u4 *__restrict__ s, u4 *__restrict__ d, unsigned n_vec, float c1, float c2)
{
    const auto rs = rsrc(s), rd = rsrc(d);
    const unsigned W = gridDim.x * WAVES, w = blockIdx.x * WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const unsigned tasks = (n_vec + 64u * U - 1u) / (64u * U);
    unsigned t, t_end, step;
    if (CONTIG) { const unsigned per = (tasks + W - 1u) / W; t = w * per; t_end = min(tasks, t + per); step = 1u; }
    else { t = w; t_end = tasks; step = W; }
    if (t >= t_end) return;
    u4 v[U], nx[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t * (64u * U) + lane + 64u * u) << 4, 0, LA);
    while (t < t_end) {
        const unsigned tn = t + step;
        if (tn < t_end) {
#pragma unroll
            for (int u = 0; u < U; u++) nx[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tn * (64u * U) + lane + 64u * u) << 4, 0, LA);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            u4 o = work<N>(v[u], c1, c2);
            const unsigned i = t * (64u * U) + lane + 64u * u;
            if (i < n_vec) __builtin_amdgcn_raw_buffer_store_b128(o, rd, i << 4, 0, SA);
        }
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = nx[u];
        t = tn;
    }
}

struct Ctx {
    hipStream_t st[4];
    hipEvent_t e0, e1, ej[4];
};

static double time_us(Ctx &C, int reps, const std::function<void()> &body)
{
    body();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(C.e0, C.st[0]));
    for (int r = 0; r < reps; r++) body();
    CK(hipEventRecord(C.e1, C.st[0]));
    CK(hipEventSynchronize(C.e1));
    float ms;
    CK(hipEventElapsedTime(&ms, C.e0, C.e1));
    return ms * 1e3 / reps;
}

static void warm(Ctx &C, const u4 *s, u4 *d, unsigned n_vec, double seconds)
{
    // an idle MI355X ramps its clocks for ~50 ms: keep it busy first
    const int n = (int)(seconds / 350e-6) + 1;
    for (int i = 0; i < n; i++)
        hipLaunchKernelGGL((k_oneshot<4, 2, 2, 0, 0, 4>), dim3((n_vec + 1023) / 1024), dim3(256), 0, C.st[0], s, d, n_vec, 1.0f, 0.0f);
    CK(hipDeviceSynchronize());
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "all";
    Ctx C;
    for (int i = 0; i < 4; i++) { CK(hipStreamCreateWithFlags(&C.st[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&C.ej[i], hipEventDisableTiming)); }
    CK(hipEventCreate(&C.e0)); CK(hipEventCreate(&C.e1));
    const size_t big = 1ull << 30;                      // 1 GiB in, 1 GiB out
    void *in, *out;
    CK(hipMalloc(&in, big)); CK(hipMalloc(&out, big));
    CK(hipMemset(in, 0x3c, big)); CK(hipMemset(out, 0, big));
    CK(hipDeviceSynchronize());
    const u4 *s = (const u4 *)in;
    u4 *d = (u4 *)out;
    const float c1 = 1.0001f, c2 = 0.5f;

    if (!strcmp(mode, "all") || !strcmp(mode, "steady")) {
        const unsigned n_vec = (unsigned)(big / 16);
        const double moved = 2.0 * big;
        warm(C, s, d, n_vec, 0.2);
        printf("== steady state: ONE launch over 1 GiB in + 1 GiB out ==\n");
        auto line = [&](const char *name, const std::function<void()> &f) {
            const double us = time_us(C, 30, f);
            printf("%-58s %9.2f us  %6.3f TB/s  %5.1f %%\n", name, us, moved / us / 1e6, moved / us / 1e6 / 8 * 100);
            fflush(stdout);
        };
#define ONES(U, LA, SA, N, MAP, WV) [&]() { hipLaunchKernelGGL((k_oneshot<U, LA, SA, N, MAP, WV>), dim3((n_vec + 64u * U * WV - 1) / (64u * U * WV)), dim3(64 * WV), 0, C.st[0], s, d, n_vec, c1, c2); }
#define PERS(U, LA, SA, N, CT, G) [&]() { hipLaunchKernelGGL((k_persist<U, LA, SA, N, CT>), dim3(G), dim3(256), 0, C.st[0], s, d, n_vec, c1, c2); }
        for (int rnd = 0; rnd < 2; rnd++) {
            printf("-- round %d\n", rnd);
            line("oneshot wg=256 U=4 nt/nt (shipped shape)", ONES(4, 2, 2, 0, 0, 4));
            line("oneshot wg=64  U=1 nt/nt", ONES(1, 2, 2, 0, 0, 1));
            line("oneshot wg=64  U=2 nt/nt", ONES(2, 2, 2, 0, 0, 1));
            line("oneshot wg=64  U=4 nt/nt", ONES(4, 2, 2, 0, 0, 1));
            line("oneshot wg=64  U=8 nt/nt", ONES(8, 2, 2, 0, 0, 1));
            line("oneshot wg=64  U=2 ld nt sc0 sc1 st nt", ONES(2, 19, 2, 0, 0, 1));
            line("oneshot wg=64  U=4 ld nt sc0 sc1 st nt", ONES(4, 19, 2, 0, 0, 1));
            line("oneshot wg=64  U=4 ld nt sc0 sc1 st nt sc0 sc1", ONES(4, 19, 19, 0, 0, 1));
            line("oneshot wg=64  U=4 nt/nt XCD-contiguous map", ONES(4, 2, 2, 0, 1, 1));
            line("oneshot wg=64  U=4 nt/nt  48 ops/vec", ONES(4, 2, 2, 12, 0, 1));
            line("oneshot wg=64  U=4 nt/nt  96 ops/vec", ONES(4, 2, 2, 24, 0, 1));
            line("oneshot wg=64  U=4 nt/nt 128 ops/vec", ONES(4, 2, 2, 32, 0, 1));
            line("oneshot wg=64  U=2 nt/nt  96 ops/vec", ONES(2, 2, 2, 24, 0, 1));
            line("oneshot wg=128 U=4 nt/nt", ONES(4, 2, 2, 0, 0, 2));
            line("oneshot wg=128 U=2 nt/nt", ONES(2, 2, 2, 0, 0, 2));
            line("oneshot wg=128 U=4 nt/nt  96 ops/vec", ONES(4, 2, 2, 24, 0, 2));
            line("oneshot wg=256 U=4 nt/nt  96 ops/vec", ONES(4, 2, 2, 24, 0, 4));
            line("persist wg=64 U=4 nt/nt g=8192 interleaved", [&]() { hipLaunchKernelGGL((k_persist<4, 2, 2, 0, 0, 1>), dim3(8192), dim3(64), 0, C.st[0], s, d, n_vec, c1, c2); });
            line("persist wg=64 U=2 nt/nt g=8192 interleaved", [&]() { hipLaunchKernelGGL((k_persist<2, 2, 2, 0, 0, 1>), dim3(8192), dim3(64), 0, C.st[0], s, d, n_vec, c1, c2); });
            line("oneshot wg=256 U=4 nt/nt (again)", ONES(4, 2, 2, 0, 0, 4));
        }
    }

    if (!strcmp(mode, "all") || !strcmp(mode, "tensor")) {
        // one launch per 33.5 MB tensor, 16 rotating (in, out) pairs inside the 1 GiB buffers (1.07 GB touched per pass)
        const size_t tb = 4096ull * 4096ull * 2ull;
        const int NT = 16;
        const unsigned n_vec = (unsigned)(tb / 16);
        const double moved = 2.0 * tb;
        warm(C, s, d, (unsigned)(big / 16), 0.2);
        printf("== one launch per 33.5 MB tensor, %d rotating tensors; us per tensor ==\n", NT);
        auto src = [&](int i) { return (const u4 *)((const char *)in + (size_t)i * tb); };
        auto dst = [&](int i) { return (u4 *)((char *)out + (size_t)i * tb); };
        auto line = [&](const char *name, const std::function<void()> &pass) {
            const double us = time_us(C, 40, pass) / NT;
            printf("%-66s %7.2f us  %5.1f %%\n", name, us, moved / us / 1e6 / 8 * 100);
            fflush(stdout);
        };
        // fork / join helpers: streams 1..k-1 start after stream 0's position and stream 0 continues after them
        auto fork = [&](int k) { CK(hipEventRecord(C.ej[0], C.st[0])); for (int i = 1; i < k; i++) CK(hipStreamWaitEvent(C.st[i], C.ej[0], 0)); };
        auto join = [&](int k) { for (int i = 1; i < k; i++) { CK(hipEventRecord(C.ej[i], C.st[i])); CK(hipStreamWaitEvent(C.st[0], C.ej[i], 0)); } };
#define T_ONES(U, N, STREAM, I) hipLaunchKernelGGL((k_oneshot<U, 2, 2, N, 0, 4>), dim3((n_vec + 256u * U - 1) / (256u * U)), dim3(256), 0, STREAM, src(I), dst(I), n_vec, c1, c2)
#define T_PERS(U, N, G, STREAM, I) hipLaunchKernelGGL((k_persist<U, 2, 2, N, 0>), dim3(G), dim3(256), 0, STREAM, src(I), dst(I), n_vec, c1, c2)
#define T_ONEW(U, N, WV, STREAM, I) hipLaunchKernelGGL((k_oneshot<U, 2, 2, N, 0, WV>), dim3((n_vec + 64u * WV * U - 1) / (64u * WV * U)), dim3(64 * WV), 0, STREAM, src(I), dst(I), n_vec, c1, c2)
#define T_ANY(U, N, WV, I) hipExtLaunchKernelGGL((k_oneshot<U, 2, 2, N, 0, WV>), dim3((n_vec + 64u * WV * U - 1) / (64u * WV * U)), dim3(64 * WV), 0, C.st[0], nullptr, nullptr, hipExtAnyOrderLaunch, src(I), dst(I), n_vec, c1, c2)
        for (int rnd = 0; rnd < 2; rnd++) {
            printf("-- round %d\n", rnd);
            line("1 stream, oneshot wg=256 U=2, 96 ops/vec (shipped shape)", [&]() { for (int i = 0; i < NT; i++) T_ONEW(2, 24, 4, C.st[0], i); });
            line("1 stream, oneshot wg=64 U=1, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ONEW(1, 24, 1, C.st[0], i); });
            line("1 stream, oneshot wg=64 U=2, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ONEW(2, 24, 1, C.st[0], i); });
            line("1 stream, oneshot wg=64 U=4, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ONEW(4, 24, 1, C.st[0], i); });
            line("1 stream, oneshot wg=64 U=2, 0 ops", [&]() { for (int i = 0; i < NT; i++) T_ONEW(2, 0, 1, C.st[0], i); });
            line("1 stream, oneshot wg=128 U=2, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ONEW(2, 24, 2, C.st[0], i); });
            line("1 stream, persist wg=256 U=2 g=2048, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_PERS(2, 24, 2048, C.st[0], i); });
            line("1 stream, persist wg=64 U=2 g=8192, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) hipLaunchKernelGGL((k_persist<2, 2, 2, 24, 0, 1>), dim3(8192), dim3(64), 0, C.st[0], src(i), dst(i), n_vec, c1, c2); });
            line("1 stream, persist wg=64 U=4 g=4096, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) hipLaunchKernelGGL((k_persist<4, 2, 2, 24, 0, 1>), dim3(4096), dim3(64), 0, C.st[0], src(i), dst(i), n_vec, c1, c2); });
            line("2 streams, oneshot wg=256 U=2, 96 ops/vec", [&]() { fork(2); for (int i = 0; i < NT; i++) T_ONEW(2, 24, 4, C.st[i % 2], i); join(2); });
            line("2 streams, oneshot wg=64 U=2, 96 ops/vec", [&]() { fork(2); for (int i = 0; i < NT; i++) T_ONEW(2, 24, 1, C.st[i % 2], i); join(2); });
            line("2 streams, oneshot wg=64 U=4, 96 ops/vec", [&]() { fork(2); for (int i = 0; i < NT; i++) T_ONEW(4, 24, 1, C.st[i % 2], i); join(2); });
            line("any-order, oneshot wg=256 U=2, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ANY(2, 24, 4, i); });
            line("any-order, oneshot wg=256 U=4, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ANY(4, 24, 4, i); });
            line("any-order, oneshot wg=64 U=2, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ANY(2, 24, 1, i); });
            line("any-order, oneshot wg=64 U=4, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ANY(4, 24, 1, i); });
            line("any-order, oneshot wg=128 U=2, 96 ops/vec", [&]() { for (int i = 0; i < NT; i++) T_ANY(2, 24, 2, i); });
            line("any-order, oneshot wg=64 U=2, 0 ops", [&]() { for (int i = 0; i < NT; i++) T_ANY(2, 0, 1, i); });
#define T_ANYS(U, N, WV, STREAM, I) hipExtLaunchKernelGGL((k_oneshot<U, 2, 2, N, 0, WV>), dim3((n_vec + 64u * WV * U - 1) / (64u * WV * U)), dim3(64 * WV), 0, STREAM, nullptr, nullptr, hipExtAnyOrderLaunch, src(I), dst(I), n_vec, c1, c2)
            line("any-order on 2 streams, oneshot wg=64 U=4, 96 ops/vec", [&]() { fork(2); for (int i = 0; i < NT; i++) T_ANYS(4, 24, 1, C.st[i % 2], i); join(2); });
            line("any-order on 2 streams, oneshot wg=64 U=2, 96 ops/vec", [&]() { fork(2); for (int i = 0; i < NT; i++) T_ANYS(2, 24, 1, C.st[i % 2], i); join(2); });
            line("any-order on 4 streams, oneshot wg=64 U=4, 96 ops/vec", [&]() { fork(4); for (int i = 0; i < NT; i++) T_ANYS(4, 24, 1, C.st[i % 4], i); join(4); });
            line("any-order then ONE ordered launch per pass (wg=64 U=2, 96 ops)", [&]() { for (int i = 0; i < NT - 1; i++) T_ANY(2, 24, 1, i); T_ONEW(2, 24, 1, C.st[0], NT - 1); });
        }
        // a hipGraph whose NT kernel nodes have no edges between them (captured from forked streams)
        for (int k : {1, 2, 4}) {
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(C.st[0], hipStreamCaptureModeGlobal));
            if (k > 1) fork(k);
            for (int i = 0; i < NT; i++) T_ONES(2, 24, C.st[i % k], i);
            if (k > 1) join(k);
            CK(hipStreamEndCapture(C.st[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            char nm[96];
            snprintf(nm, 96, "hipGraph, %d branch(es), oneshot U=2, 96 ops/vec", k);
            line(nm, [&]() { CK(hipGraphLaunch(ge, C.st[0])); });
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
