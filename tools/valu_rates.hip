// valu_rates.hip -- issue rate of a few VALU instructions on gfx950 (dev tool: sizes the element loop of antq_k_approx.h).
// Every wave runs ITER x 8 independent copies of one instruction; 256 CUs x 8 waves / SIMD.  Prints lane-ops per clock
// per CU (v_fma_f32 = 64 is the full rate).   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 4096;

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

#define KERNEL(NAME, DECL, INS, SINK)                                                  \
    __global__ void __launch_bounds__(256) NAME(float *out, float seed)                \
    {                                                                                  \
        DECL                                                                           \
        for (int i = 0; i < ITER; i++) { BODY8(INS) }                                  \
        SINK                                                                           \
    }

// f32 fma
#define D_F32 float a[8]; float b = seed, c = seed * 0.5f; for (int k = 0; k < 8; k++) a[k] = seed + k;
#define I_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_F32 float s = 0; for (int k = 0; k < 8; k++) s += a[k]; if (s == 12345.f) out[threadIdx.x] = s;
KERNEL(k_fma32, D_F32, I_FMA32, S_F32)
#define I_MUL32(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
KERNEL(k_mul32, D_F32, I_MUL32, S_F32)
#define I_RCP32(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
KERNEL(k_rcp32, D_F32, I_RCP32, S_F32)
#define I_MED3(k) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
KERNEL(k_med3, D_F32, I_MED3, S_F32)
#define I_CMP32(k) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
KERNEL(k_cmpsel32, D_F32, I_CMP32, S_F32)

// packed f32
#define D_PK float2 a[8]; float2 b = make_float2(seed, seed), c = make_float2(seed * .5f, seed); for (int k = 0; k < 8; k++) a[k] = make_float2(seed + k, seed);
#define I_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_PK float s = 0; for (int k = 0; k < 8; k++) s += a[k].x + a[k].y; if (s == 12345.f) out[threadIdx.x] = s;
KERNEL(k_pkfma32, D_PK, I_PKFMA, S_PK)
#define I_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
KERNEL(k_pkmul32, D_PK, I_PKMUL, S_PK)

// f64
#define D_F64 double a[8]; double b = seed, c = seed * 0.5; for (int k = 0; k < 8; k++) a[k] = seed + k;
#define I_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_F64 double s = 0; for (int k = 0; k < 8; k++) s += a[k]; if (s == 12345.) out[threadIdx.x] = (float)s;
KERNEL(k_fma64, D_F64, I_FMA64, S_F64)
#define I_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
KERNEL(k_mul64, D_F64, I_MUL64, S_F64)
#define I_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
KERNEL(k_add64, D_F64, I_ADD64, S_F64)

// conversions / compare in double
#define D_CVT float x[8]; double a[8]; float c = seed; for (int k = 0; k < 8; k++) { x[k] = seed + k; a[k] = 0; }
#define I_CVT(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(x[k]));
KERNEL(k_cvt64, D_CVT, I_CVT, S_F64)
#define D_CMP64 double a[8]; double b = seed; float r[8]; float c = seed; for (int k = 0; k < 8; k++) { a[k] = seed + k; r[k] = k; }
#define I_CMP64(k) asm volatile("v_cmp_ge_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(r[k]) : "v"(a[k]), "v"(b), "v"(c) : "vcc");
#define S_R float s = 0; for (int k = 0; k < 8; k++) s += r[k]; if (s == 12345.f) out[threadIdx.x] = s;
KERNEL(k_cmpsel64, D_CMP64, I_CMP64, S_R)

// integer helpers of the key computation
#define I_BFE(k) asm volatile("v_bfe_u32 %0, %0, 5, 10" : "+v"(a[k]));
KERNEL(k_bfe, D_F32, I_BFE, S_F32)
#define I_ALIGN(k) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[k]) : "v"(b));
KERNEL(k_alignbit, D_F32, I_ALIGN, S_F32)
#define I_CVTBF(k) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
KERNEL(k_cvtpkbf16, D_F32, I_CVTBF, S_F32)

template <typename K>
static void run(const char *name, K kern, int insts_per_body)
{
    float *out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;          // 8 workgroups of 4 waves per CU = 8 waves / SIMD
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_ops = (double)reps * blocks * 256.0 * ITER * 8.0 * insts_per_body;
    const double per_s = lane_ops / (ms * 1e-3);
    printf("%-14s %8.1f G lane-ops/s   = %6.1f lane-ops / clk / CU at 2.4 GHz (256 CUs)\n", name, per_s / 1e9, per_s / 256.0 / 2.4e9);
    CK(hipFree(out));
}

int main()
{
    run("v_fma_f32", k_fma32, 1);
    run("v_mul_f32", k_mul32, 1);
    run("v_pk_fma_f32", k_pkfma32, 1);
    run("v_pk_mul_f32", k_pkmul32, 1);
    run("v_rcp_f32", k_rcp32, 1);
    run("v_med3_i32", k_med3, 1);
    run("v_bfe_u32", k_bfe, 1);
    run("v_alignbit", k_alignbit, 1);
    run("v_cvt_pk_bf16", k_cvtpkbf16, 1);
    run("cmp+cndmask32", k_cmpsel32, 2);
    run("v_fma_f64", k_fma64, 1);
    run("v_mul_f64", k_mul64, 1);
    run("v_add_f64", k_add64, 1);
    run("v_cvt_f64_f32", k_cvt64, 1);
    run("cmp64+cndmask", k_cmpsel64, 2);
    return 0;
}
