// ubench.hip -- standalone HBM streaming micro-benchmarks (no torch): finds the access
// pattern / launch shape that gets closest to the MI355X HBM roofline for a
// read-N-bytes + write-N-bytes element-wise kernel, and times libantq's entry points
// natively with hipEvents.  Build: see tools/build_ubench.sh.  Dev tool, not product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <string>

#include "../include/antq.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// ---- copy variants --------------------------------------------------------------
// A: block handles 4 KiB*U contiguous; thread's U loads strided by 4 KiB (256 thr x 16 B)
template <int U>
__global__ void __launch_bounds__(256) copy_blk(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t first = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 256 * u < n) v[u] = s[first + 256 * u];
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 256 * u < n) d[first + 256 * u] = v[u];
}
// B: wave handles 1 KiB*U contiguous (lane loads strided by 1 KiB)
template <int U>
__global__ void __launch_bounds__(256) copy_wave(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t first = wave * (64 * U) + (threadIdx.x & 63);
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 64 * u < n) v[u] = s[first + 64 * u];
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 64 * u < n) d[first + 64 * u] = v[u];
}
// C: nontemporal variant of B
template <int U>
__global__ void __launch_bounds__(256) copy_wave_nt(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t first = wave * (64 * U) + (threadIdx.x & 63);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *sp = reinterpret_cast<const u4 *>(s);
    u4 *dp = reinterpret_cast<u4 *>(d);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 64 * u < n) v[u] = __builtin_nontemporal_load(sp + first + 64 * u);
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 64 * u < n) __builtin_nontemporal_store(v[u], dp + first + 64 * u);
}
// C2: nontemporal variant of A (block-contiguous rounds)
template <int U>
__global__ void __launch_bounds__(256) copy_blk_nt(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t first = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *sp = reinterpret_cast<const u4 *>(s);
    u4 *dp = reinterpret_cast<u4 *>(d);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 256 * u < n) v[u] = __builtin_nontemporal_load(sp + first + 256 * u);
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 256 * u < n) __builtin_nontemporal_store(v[u], dp + first + 256 * u);
}
// C3: 512-thread blocks, one vector per lane (8 KiB per block)
__global__ void __launch_bounds__(512) copy_nt_512(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const u4 *>(s) + i), reinterpret_cast<u4 *>(d) + i);
}
__global__ void __launch_bounds__(1024) copy_nt_1024(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const u4 *>(s) + i), reinterpret_cast<u4 *>(d) + i);
}
// C4: U far-apart streams: wave w copies chunk w of each of U equal slices of the buffer
template <int U>
__global__ void __launch_bounds__(256) copy_streams_nt(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *sp = reinterpret_cast<const u4 *>(s);
    u4 *dp = reinterpret_cast<u4 *>(d);
    const size_t slice = n / U;                       // n % (U*64) == 0 assumed
    const size_t i = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i < slice) v[u] = __builtin_nontemporal_load(sp + u * slice + i);
#pragma unroll
    for (int u = 0; u < U; u++) if (i < slice) __builtin_nontemporal_store(v[u], dp + u * slice + i);
}
// C5: U streams 1 row (8 KiB) apart: wave w takes the w-th KiB of U consecutive 8-KiB rows
template <int U>
__global__ void __launch_bounds__(256) copy_rows_nt(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *sp = reinterpret_cast<const u4 *>(s);
    u4 *dp = reinterpret_cast<u4 *>(d);
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t grp = wave / 8, k = wave % 8;        // group of U rows (512 vectors each), KiB k of each row
    const size_t base = grp * (U * 512) + k * 64 + (threadIdx.x & 63);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 512 < n) v[u] = __builtin_nontemporal_load(sp + base + u * 512);
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 512 < n) __builtin_nontemporal_store(v[u], dp + base + u * 512);
}
// C6: mixed cache policies for the U=4 wave-contiguous copy
template <int U, bool LNT, bool SNT>
__global__ void __launch_bounds__(256) copy_mix(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t first = wave * (64 * U) + (threadIdx.x & 63);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *sp = reinterpret_cast<const u4 *>(s);
    u4 *dp = reinterpret_cast<u4 *>(d);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 64 * u < n) v[u] = LNT ? __builtin_nontemporal_load(sp + first + 64 * u) : sp[first + 64 * u];
#pragma unroll
    for (int u = 0; u < U; u++) if (first + 64 * u < n) { if (SNT) __builtin_nontemporal_store(v[u], dp + first + 64 * u); else dp[first + 64 * u] = v[u]; }
}
// D: persistent grid-stride, G blocks, each iteration block copies 4 KiB*U, prefetch depth 1
template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_persist(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *sp = reinterpret_cast<const u4 *>(s);
    u4 *dp = reinterpret_cast<u4 *>(d);
    const size_t stride = (size_t)gridDim.x * (256 * U);
    size_t i = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    u4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i + 256 * u < n) v[u] = NT ? __builtin_nontemporal_load(sp + i + 256 * u) : sp[i + 256 * u];
    while (i < n) {
        size_t j = i + stride;
#pragma unroll
        for (int u = 0; u < U; u++) if (j + 256 * u < n) w[u] = NT ? __builtin_nontemporal_load(sp + j + 256 * u) : sp[j + 256 * u];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + 256 * u < n) { if (NT) __builtin_nontemporal_store(v[u], dp + i + 256 * u); else dp[i + 256 * u] = v[u]; }
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = w[u];
        i = j;
    }
}
// E: copy + N dependent FMAs per dword (N/2 VALU ops per bf16 element): how much arithmetic
// hides under the stream in the one-shot shape
template <int U, int N>
__global__ void __launch_bounds__(256) copy_work(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n, float c1, float c2)
{
    size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t first = wave * (64 * U) + (threadIdx.x & 63);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 *sp = reinterpret_cast<const u4 *>(s);
    u4 *dp = reinterpret_cast<u4 *>(d);
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(sp + (first + 64 * u < n ? first + 64 * u : n - 1));
#pragma unroll
    for (int u = 0; u < U; u++) {
        float f[4] = {__uint_as_float(v[u].x), __uint_as_float(v[u].y), __uint_as_float(v[u].z), __uint_as_float(v[u].w)};
#pragma unroll
        for (int k = 0; k < N; k++) {
#pragma unroll
            for (int e = 0; e < 4; e++) f[e] = __builtin_fmaf(f[e], c1, c2);
        }
        u4 o = {__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
        if (first + 64 * u < n) __builtin_nontemporal_store(o, dp + first + 64 * u);
    }
}
// read-only / write-only probes
__global__ void __launch_bounds__(256) read_only(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t first = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; u++) if (first + 256 * u < n) { uint4 v = s[first + 256 * u]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678 && acc.y == 0x9abcdef0) d[first] = acc;
}
__global__ void __launch_bounds__(256) write_only(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    size_t first = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
#pragma unroll
    for (int u = 0; u < 4; u++) if (first + 256 * u < n) d[first + 256 * u] = v;
}

struct Bench {
    std::vector<void *> in, out;
    size_t bytes;
    int nbuf;
    hipStream_t st;
    hipEvent_t e0, e1;
};

template <typename F>
static void run(Bench &B, const char *name, int reps, double bytes_moved, F launch)
{
    for (int i = 0; i < B.nbuf; i++) launch(B.in[i], B.out[i]);  // warm
    CK(hipStreamSynchronize(B.st));
    // whole-loop timing
    CK(hipEventRecord(B.e0, B.st));
    for (int r = 0; r < reps; r++)
        for (int i = 0; i < B.nbuf; i++) launch(B.in[i], B.out[i]);
    CK(hipEventRecord(B.e1, B.st));
    CK(hipEventSynchronize(B.e1));
    float ms;
    CK(hipEventElapsedTime(&ms, B.e0, B.e1));
    double us = ms * 1e3 / (reps * B.nbuf);
    printf("%-34s %8.2f us/launch  %7.3f TB/s (%.1f%% of 8)\n", name, us, bytes_moved / us / 1e6, bytes_moved / us / 1e6 / 8 * 100);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    size_t bytes = 4096ull * 4096ull * 2ull;  // 4096^2 bf16
    int nbuf = 32;
    if (argc > 1) bytes = strtoull(argv[1], 0, 10);
    if (argc > 2) nbuf = atoi(argv[2]);
    Bench B;
    B.bytes = bytes; B.nbuf = nbuf;
    CK(hipStreamCreate(&B.st));
    CK(hipEventCreate(&B.e0)); CK(hipEventCreate(&B.e1));
    for (int i = 0; i < nbuf; i++) {
        void *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 0x3c, bytes)); CK(hipMemset(b, 0, bytes));
        B.in.push_back(a); B.out.push_back(b);
    }
    CK(hipDeviceSynchronize());
    const size_t n = bytes / 16;
    const double moved = 2.0 * bytes;
    printf("buffer %zu bytes x %d (in) + %d (out)\n", bytes, nbuf, nbuf);
    const int reps = 20;
#define L(kern, blocks) [&](void *s, void *d) { hipLaunchKernelGGL(kern, dim3((unsigned)(blocks)), dim3(256), 0, B.st, (const uint4 *)s, (uint4 *)d, n); }
    run(B, "read_only", reps, (double)bytes, L(read_only, (n + 1023) / 1024));
    run(B, "write_only", reps, (double)bytes, L(write_only, (n + 1023) / 1024));
    run(B, "copy_blk<1>", reps, moved, L(copy_blk<1>, (n + 255) / 256));
    run(B, "copy_blk<2>", reps, moved, L(copy_blk<2>, (n + 511) / 512));
    run(B, "copy_blk<4>", reps, moved, L(copy_blk<4>, (n + 1023) / 1024));
    run(B, "copy_blk<8>", reps, moved, L(copy_blk<8>, (n + 2047) / 2048));
    run(B, "copy_wave<1>", reps, moved, L(copy_wave<1>, (n + 255) / 256));
    run(B, "copy_wave<2>", reps, moved, L(copy_wave<2>, (n + 511) / 512));
    run(B, "copy_wave<4>", reps, moved, L(copy_wave<4>, (n + 1023) / 1024));
    run(B, "copy_wave<8>", reps, moved, L(copy_wave<8>, (n + 2047) / 2048));
    run(B, "copy_wave_nt<1>", reps, moved, L(copy_wave_nt<1>, (n + 255) / 256));
    run(B, "copy_wave_nt<2>", reps, moved, L(copy_wave_nt<2>, (n + 511) / 512));
    run(B, "copy_wave_nt<4>", reps, moved, L(copy_wave_nt<4>, (n + 1023) / 1024));
    run(B, "copy_mix<4> load nt, store plain", reps, moved, L((copy_mix<4, true, false>), (n + 1023) / 1024));
    run(B, "copy_mix<4> load plain, store nt", reps, moved, L((copy_mix<4, false, true>), (n + 1023) / 1024));
    run(B, "copy_mix<1> load nt, store plain", reps, moved, L((copy_mix<1, true, false>), (n + 255) / 256));
    run(B, "copy_mix<1> load plain, store nt", reps, moved, L((copy_mix<1, false, true>), (n + 255) / 256));
    run(B, "copy_streams_nt<2>", reps, moved, L(copy_streams_nt<2>, (n / 2 + 255) / 256));
    run(B, "copy_streams_nt<4>", reps, moved, L(copy_streams_nt<4>, (n / 4 + 255) / 256));
    run(B, "copy_rows_nt<2>", reps, moved, L(copy_rows_nt<2>, (n / 2 + 255) / 256));
    run(B, "copy_rows_nt<4>", reps, moved, L(copy_rows_nt<4>, (n / 4 + 255) / 256));
    run(B, "copy_blk_nt<2>", reps, moved, L(copy_blk_nt<2>, (n + 511) / 512));
    run(B, "copy_blk_nt<4>", reps, moved, L(copy_blk_nt<4>, (n + 1023) / 1024));
    run(B, "copy_blk_nt<8>", reps, moved, L(copy_blk_nt<8>, (n + 2047) / 2048));
    run(B, "copy_nt_512", reps, moved, [&](void *s, void *d) { hipLaunchKernelGGL(copy_nt_512, dim3((unsigned)((n + 511) / 512)), dim3(512), 0, B.st, (const uint4 *)s, (uint4 *)d, n); });
    run(B, "copy_nt_1024", reps, moved, [&](void *s, void *d) { hipLaunchKernelGGL(copy_nt_1024, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, B.st, (const uint4 *)s, (uint4 *)d, n); });
    for (int g : {256, 512, 1024, 2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "copy_persist<1,0> g=%d", g); run(B, nm, reps, moved, L((copy_persist<1, false>), g));
        snprintf(nm, 64, "copy_persist<2,0> g=%d", g); run(B, nm, reps, moved, L((copy_persist<2, false>), g));
        snprintf(nm, 64, "copy_persist<4,0> g=%d", g); run(B, nm, reps, moved, L((copy_persist<4, false>), g));
        snprintf(nm, 64, "copy_persist<2,1> g=%d", g); run(B, nm, reps, moved, L((copy_persist<2, true>), g));
    }
#define LW(kern, blocks) [&](void *s, void *d) { hipLaunchKernelGGL(kern, dim3((unsigned)(blocks)), dim3(256), 0, B.st, (const uint4 *)s, (uint4 *)d, n, 1.0001f, 0.5f); }
    run(B, "copy_work<4,0>", reps, moved, LW((copy_work<4, 0>), (n + 1023) / 1024));
    run(B, "copy_work<4,8>  (4 ops/elem)", reps, moved, LW((copy_work<4, 8>), (n + 1023) / 1024));
    run(B, "copy_work<4,16> (8 ops/elem)", reps, moved, LW((copy_work<4, 16>), (n + 1023) / 1024));
    run(B, "copy_work<4,32> (16 ops/elem)", reps, moved, LW((copy_work<4, 32>), (n + 1023) / 1024));
    run(B, "copy_work<4,64> (32 ops/elem)", reps, moved, LW((copy_work<4, 64>), (n + 1023) / 1024));
    run(B, "copy_work<1,32> (16 ops/elem)", reps, moved, LW((copy_work<1, 32>), (n + 255) / 256));
    run(B, "copy_work<2,32> (16 ops/elem)", reps, moved, LW((copy_work<2, 32>), (n + 511) / 512));
    run(B, "copy_work<8,32> (16 ops/elem)", reps, moved, LW((copy_work<8, 32>), (n + 2047) / 2048));
    run(B, "hipMemcpyAsync D2D", reps, moved, [&](void *s, void *d) { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, B.st)); });

    // libantq entry points (bf16 4096x4096 flint 4-bit per-row), natively timed
    const size_t fq_rows = bytes / (4096ull * 2ull);
    if (bytes % (4096ull * 2ull) == 0) {
        static const float flint4[16] = {-10.f, -5.f, -3.75f, -2.5f, -1.875f, -1.25f, -0.625f, 0.f, 0.f,
                                         0.625f, 1.25f, 1.875f, 2.5f, 3.75f, 5.f, 10.f};
        std::vector<unsigned char> plan(ANTQ_PLAN_MAX_BYTES);
        int pb = antq_plan_build(flint4, 16, plan.data(), plan.size());
        printf("plan bytes %d kind %d\n", pb, antq_plan_kind(plan.data()));
        void *plan_dev; CK(hipMalloc(&plan_dev, pb)); CK(hipMemcpy(plan_dev, plan.data(), pb, hipMemcpyHostToDevice));
        std::vector<float> alpha(fq_rows, 0.08f);
        float *alpha_dev; CK(hipMalloc(&alpha_dev, fq_rows * 4)); CK(hipMemcpy(alpha_dev, alpha.data(), fq_rows * 4, hipMemcpyHostToDevice));
        // fill inputs with bf16 gaussian-ish data
        std::vector<uint16_t> h(bytes / 2);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < h.size(); i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            float f = ((float)(s & 0xffff) / 65536.0f + (float)((s >> 16) & 0xffff) / 65536.0f + (float)((s >> 32) & 0xffff) / 65536.0f - 1.5f) * 0.04f;
            uint32_t u; memcpy(&u, &f, 4); h[i] = (uint16_t)(u >> 16);
        }
        for (int i = 0; i < nbuf; i++) CK(hipMemcpy(B.in[i], h.data(), bytes, hipMemcpyHostToDevice));
        run(B, "antq_fakequant bf16 flint4 per-row", reps, moved, [&](void *x, void *o) {
            int rc = antq_fakequant(x, o, nullptr, fq_rows, 4096, alpha_dev, 1, 10.0f, plan.data(), plan_dev, 0, ANTQ_BF16, B.st);
            if (rc) { printf("antq_fakequant rc=%d\n", rc); exit(1); }
        });
        run(B, "antq_copy", reps, moved, [&](void *x, void *o) { antq_copy(x, o, bytes, B.st); });
        for (int U : {1, 2, 4, 8}) for (int blocks : {0}) {
            antq_debug_set(0, U); antq_debug_set(1, blocks);
            char nm[64]; snprintf(nm, 64, "fakequant U=%d blocks=%d", U, blocks);
            run(B, nm, reps, moved, [&](void *x, void *o) {
                antq_fakequant(x, o, nullptr, fq_rows, 4096, alpha_dev, 1, 10.0f, plan.data(), plan_dev, 0, ANTQ_BF16, B.st);
            });
        }
        antq_debug_set(0, 0); antq_debug_set(1, 0);
        for (int rep = 0; rep < 2; rep++) for (int xk : {0, 1}) for (int U : {1, 2, 4, 8}) {
            antq_debug_set(2, xk); antq_debug_set(0, U);
            char nm[64]; snprintf(nm, 64, "fakequant xkernel=%d U=%d", xk, U);
            run(B, nm, reps, moved, [&](void *x, void *o) {
                antq_fakequant(x, o, nullptr, fq_rows, 4096, alpha_dev, 1, 10.0f, plan.data(), plan_dev, 0, ANTQ_BF16, B.st);
            });
        }
        antq_debug_set(0, 0); antq_debug_set(2, 1);
        float *aout; CK(hipMalloc(&aout, fq_rows * 4));
        run(B, "antq_fakequant_dynamic bf16", reps, moved, [&](void *x, void *o) {
            int rc = antq_fakequant_dynamic(x, o, nullptr, aout, fq_rows, 4096, 1.0f, 10.0f, plan.data(), plan_dev, 0, ANTQ_BF16, B.st);
            if (rc) { printf("dyn rc=%d\n", rc); exit(1); }
        });
    }
    return 0;
}
