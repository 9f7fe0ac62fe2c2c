// A torch-free consumer of the C ABI (include/antq.h): plain HIP runtime + libantq.so, checked against the CPU oracle
// (oracle/libantq_oracle.so -- test infrastructure, linked here as the checker only).
//
// This is what a C/C++ host of the reference's path would do: build a plan on the host,
// upload it, call the fused entry points on its own stream and buffers.  Covers antq_nearest (the quant_cuda.quant
// replacement, KQ/quant_kernel.cu:11-62), antq_fakequant (AQ:535-551), the OliVe victim rule (OQ:311-320),
// antq_fakequant_dynamic + antq_absmax, antq_fakequant_batch, the packed 4-bit codec, antq_nearest_hinted, group-16 and the
// calibration entry points (antq_search_sse with its workspace, antq_search_pick, and antq_calibrate: all of it in one call),
// and -- section 9 -- every ABI 4 / 5 entry: antq_nearest_plan, the host models antq_plan_eval_host[_a|_h], the 16-bit-domain row
// kernels on bf16, antq_absmax_into, antq_search_sse_multi, antq_moments + antq_xmax_3sigma, antq_affine, antq_alpha_grad,
// antq_calibrate_batch, antq_fakequant_f64, antq_copy, antq_prefetch_kernels, antq_debug_set, and ABI 7's antq_absmax_t /
// antq_alpha_grad_t / antq_calibrate_install: every prototype of include/antq.h is called here.
// Exit code 0 = every comparison bit-exact; prints one line per check.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/antq.h"

extern "C" {
void antq_oracle_nearest_f32(const float *x, float *z, int32_t *idx, size_t n, const float *grid, int m);
void antq_oracle_forward_f32(const float *x, float *out, int32_t *idx, size_t rows, size_t row_len, const float *alpha,
                             int alpha_per_row, const float *grid, int m, float gmax, int ovp);
void antq_oracle_absmax_f32(const float *x, float *alpha, size_t rows, size_t row_len, int per_row, float ratio);
int antq_oracle_search_mse_f32(const float *x, size_t rows, size_t row_len, int per_row, const float *x_max, int lb, int ub,
                               int step, const float *grid, int m, float gmax, int ovp, float *best_score, float *best_alpha,
                               float *trace);
void antq_oracle_nearest_f64(const double *x, double *z, int32_t *idx, size_t n, const double *grid, int m);
void antq_oracle_f32_to_bf16(const float *in, uint16_t *out, size_t n);
void antq_oracle_forward_bf16(const uint16_t *x, uint16_t *out, int32_t *idx, size_t rows, size_t row_len, const float *alpha,
                              int alpha_per_row, const float *grid, int m, float gmax, int ovp);
void antq_oracle_affine_f32(const float *x, float *out, int32_t *qout, size_t rows, size_t row_len, int k, const float *x_min,
                            const float *x_max, int per_row);
}

#define HIP_OK(e)                                                                                          \
    do {                                                                                                   \
        hipError_t err_ = (e);                                                                             \
        if (err_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(err_), __LINE__); return 2; } \
    } while (0)
#define ANTQ_OK_(e)                                                                                        \
    do {                                                                                                   \
        int rc_ = (e);                                                                                     \
        if (rc_ != ANTQ_OK) { printf("antq error %s (%d) at line %d\n", antq_strerror(rc_), rc_, __LINE__); return 3; } \
    } while (0)

static int failures = 0;

static void same_bits(const char *what, const std::vector<float> &got, const std::vector<float> &ref)
{
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); i++) {
        uint32_t a, b;
        memcpy(&a, &got[i], 4);
        memcpy(&b, &ref[i], 4);
        if (a != b && !(std::isnan(got[i]) && std::isnan(ref[i]))) bad++;
    }
    printf("%-58s %s (%zu / %zu differ)\n", what, bad ? "FAIL" : "ok", bad, ref.size());
    if (bad) failures++;
}

static void same_idx(const char *what, const std::vector<int16_t> &got, const std::vector<int32_t> &ref)
{
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); i++) bad += ((int32_t)got[i] != ref[i]);
    printf("%-58s %s (%zu / %zu differ)\n", what, bad ? "FAIL" : "ok", bad, ref.size());
    if (bad) failures++;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    explicit DevBuf(size_t n_) : n(n_) { if (hipMalloc(&p, n * sizeof(T) + 16) != hipSuccess) p = nullptr; }
    ~DevBuf() { if (p) (void)hipFree(p); }
    void up(const std::vector<T> &h, hipStream_t s) { (void)hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s); }
    std::vector<T> down(hipStream_t s) const
    {
        std::vector<T> h(n);
        (void)hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        return h;
    }
};

int main()
{
    if (antq_abi_version() != ANTQ_ABI_VERSION) { printf("ABI version mismatch\n"); return 4; }
    hipStream_t st;
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipStreamCreate(&st));

    // ANT 4-bit signed flint (AQ:223-278, values scaled to max 10) and OliVe flint + outliers (OQ:93-179)
    const std::vector<float> flint = {-10.f, -5.f, -3.75f, -2.5f, -1.875f, -1.25f, -0.625f, 0.f, 0.f,
                                      0.625f, 1.25f, 1.875f, 2.5f, 3.75f, 5.f, 10.f};
    std::vector<float> olive = {-32, -16, -12, -8, -6, -4, -2, 0, 2, 4, 6, 8, 12, 16, 32};
    const int n_normal = (int)olive.size();
    for (float o : {-384.f, -256.f, -192.f, -128.f, -96.f, -64.f, -48.f, 48.f, 64.f, 96.f, 128.f, 192.f, 256.f, 384.f}) olive.push_back(o);

    const size_t rows = 96, K = 4096, n = rows * K;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 0.02f);
    std::vector<float> x(n), alpha(rows), alpha3(rows);
    for (auto &v : x) v = nd(rng);
    for (size_t i = 0; i < n; i += 211) x[i] *= 25.f;                       // planted outliers
    x[5] = NAN; x[7] = INFINITY; x[9] = -3e30f; x[11] = 0.f; x[13] = -0.f; x[15] = 1e-41f;
    antq_oracle_absmax_f32(x.data(), alpha.data(), rows, K, 1, 1.0f);
    for (size_t r = 0; r < rows; r++) {                                      // rows with NaN/Inf: pick a sane alpha
        if (!(alpha[r] < 1e10f)) alpha[r] = 0.08f;
        alpha[r] *= 0.9f;
        alpha3[r] = 0.06f + 0.001f * (float)r;
    }

    DevBuf<float> dx(n), dout(n), dalpha(rows), dalpha3(rows), dgrid(64), damax(rows);
    DevBuf<int16_t> didx(n);
    DevBuf<uint8_t> dplan(ANTQ_PLAN_MAX_BYTES), dplan2(ANTQ_PLAN_MAX_BYTES), dcodes(n / 2);
    dx.up(x, st); dalpha.up(alpha, st); dalpha3.up(alpha3, st);

    std::vector<uint8_t> plan(ANTQ_PLAN_MAX_BYTES), plan2(ANTQ_PLAN_MAX_BYTES);
    int pb = antq_plan_build(flint.data(), (int)flint.size(), plan.data(), plan.size());
    int pb2 = antq_plan_build(olive.data(), (int)olive.size(), plan2.data(), plan2.size());
    if (pb <= 0 || pb2 <= 0 || antq_plan_kind(plan.data()) != 1 || antq_plan_kind(plan2.data()) != 1) { printf("plan build failed\n"); return 5; }
    HIP_OK(hipMemcpyAsync(dplan.p, plan.data(), pb, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(dplan2.p, plan2.data(), pb2, hipMemcpyHostToDevice, st));

    std::vector<float> ref(n);
    std::vector<int32_t> ridx(n);

    // 1. quant_cuda.quant replacement
    HIP_OK(hipMemcpyAsync(dgrid.p, flint.data(), flint.size() * 4, hipMemcpyHostToDevice, st));
    ANTQ_OK_(antq_nearest(dx.p, dout.p, didx.p, n, dgrid.p, (int)flint.size(), ANTQ_F32, st));
    antq_oracle_nearest_f32(x.data(), ref.data(), ridx.data(), n, flint.data(), (int)flint.size());
    same_bits("antq_nearest values (flint-4)", dout.down(st), ref);
    same_idx("antq_nearest indices", didx.down(st), ridx);

    // 2. fused Quantizer._forward, per-row alpha
    ANTQ_OK_(antq_fakequant(dx.p, dout.p, didx.p, rows, K, dalpha.p, 1, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st));
    antq_oracle_forward_f32(x.data(), ref.data(), ridx.data(), rows, K, alpha.data(), 1, flint.data(), (int)flint.size(), 10.0f, 0);
    same_bits("antq_fakequant ANT flint-4 per-row", dout.down(st), ref);
    same_idx("antq_fakequant indices", didx.down(st), ridx);

    // 3. the whole buffer as one quant group: one alpha per tensor (activations, AQ:477)
    std::vector<float> a1 = {0.07f};
    DevBuf<float> da1(1);
    da1.up(a1, st);
    ANTQ_OK_(antq_fakequant(dx.p, dout.p, nullptr, 1, n, da1.p, 0, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st));
    antq_oracle_forward_f32(x.data(), ref.data(), ridx.data(), 1, n, a1.data(), 0, flint.data(), (int)flint.size(), 10.0f, 0);
    same_bits("antq_fakequant per-tensor alpha", dout.down(st), ref);

    // 4. OliVe outlier-victim pairs
    ANTQ_OK_(antq_fakequant(dx.p, dout.p, didx.p, rows, K, dalpha3.p, 1, 32.0f, plan2.data(), dplan2.p, ANTQ_FLAG_OVP, ANTQ_F32, st));
    antq_oracle_forward_f32(x.data(), ref.data(), ridx.data(), rows, K, alpha3.data(), 1, olive.data(), (int)olive.size(), 32.0f, 1);
    same_bits("antq_fakequant OliVe flint-4 + outlier-victim pairs", dout.down(st), ref);
    same_idx("antq_fakequant OliVe indices (victims = -2)", didx.down(st), ridx);
    size_t victims = 0;
    for (int32_t v : ridx) victims += (v == ANTQ_IDX_VICTIM);
    if (!victims) { printf("no victims in the OliVe case\n"); failures++; }

    // 5. dynamic alpha (finite data only: the row abs-max must be meaningful)
    std::vector<float> xf = x;
    xf[5] = 0.3f; xf[7] = -0.2f; xf[9] = 0.1f;
    DevBuf<float> dxf(n);
    dxf.up(xf, st);
    std::vector<float> adyn(rows);
    antq_oracle_absmax_f32(xf.data(), adyn.data(), rows, K, 1, 0.85f);
    ANTQ_OK_(antq_fakequant_dynamic(dxf.p, dout.p, nullptr, damax.p, rows, K, 0.85f, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st));
    antq_oracle_forward_f32(xf.data(), ref.data(), ridx.data(), rows, K, adyn.data(), 1, flint.data(), (int)flint.size(), 10.0f, 0);
    same_bits("antq_fakequant_dynamic alpha", damax.down(st), adyn);
    same_bits("antq_fakequant_dynamic values", dout.down(st), ref);
    HIP_OK(hipMemsetAsync(damax.p, 0xff, 4, st));                            // poisoned: the entry point initialises it
    ANTQ_OK_(antq_absmax(dxf.p, damax.p, rows, K, 0, ANTQ_F32, st));
    std::vector<float> amax_t(1);
    antq_oracle_absmax_f32(xf.data(), amax_t.data(), rows, K, 0, 1.0f);
    {
        std::vector<float> got = damax.down(st);
        got.resize(1);
        same_bits("antq_absmax per tensor", got, amax_t);
    }

    // 6. two jobs (ANT per-row + the per-tensor view) in one batched launch
    DevBuf<float> dout2(n);
    antq_job jobs[2] = {{dxf.p, dout.p, dalpha.p, rows, K, 1, 10.0f, plan.data(), dplan.p},
                        {dxf.p, dout2.p, da1.p, 1, n, 0, 10.0f, plan.data(), dplan.p}};
    size_t cap = antq_batch_capacity(jobs, 2, ANTQ_F32);
    std::vector<uint8_t> batch(cap);
    int bb = antq_batch_build(jobs, 2, ANTQ_F32, 0, batch.data(), cap);
    if (bb <= 0) { printf("antq_batch_build failed: %d\n", bb); return 6; }
    DevBuf<uint8_t> dbatch((size_t)bb);
    HIP_OK(hipMemcpyAsync(dbatch.p, batch.data(), bb, hipMemcpyHostToDevice, st));
    ANTQ_OK_(antq_fakequant_batch(batch.data(), dbatch.p, st));
    antq_oracle_forward_f32(xf.data(), ref.data(), ridx.data(), rows, K, alpha.data(), 1, flint.data(), (int)flint.size(), 10.0f, 0);
    same_bits("antq_fakequant_batch job 0 (per-row)", dout.down(st), ref);
    antq_oracle_forward_f32(xf.data(), ref.data(), ridx.data(), 1, n, a1.data(), 0, flint.data(), (int)flint.size(), 10.0f, 0);
    same_bits("antq_fakequant_batch job 1 (per-tensor)", dout2.down(st), ref);

    // 7. packed 4-bit codec: decode(encode(x)) == fake-quant(x), OliVe pairs with the outlier identifier
    ANTQ_OK_(antq_encode4(dxf.p, dcodes.p, rows, K, dalpha3.p, 1, 32.0f, plan2.data(), dplan2.p, n_normal, ANTQ_FLAG_OVP, ANTQ_F32, st));
    ANTQ_OK_(antq_decode4(dcodes.p, dout.p, rows, K, dalpha3.p, 1, 32.0f, plan2.data(), dplan2.p, n_normal, ANTQ_FLAG_OVP, ANTQ_F32, st));
    antq_oracle_forward_f32(xf.data(), ref.data(), ridx.data(), rows, K, alpha3.data(), 1, olive.data(), (int)olive.size(), 32.0f, 1);
    same_bits("antq_decode4(antq_encode4(x)) OliVe, 0.5 B/elem", dout.down(st), ref);

    // 7b. the operator with a plan as a hint: right plan -> table path; a plan of ANOTHER grid -> the kernel notices, scans the
    //     device grid literally and raises the stale flag (antq_nearest_hinted)
    {
        DevBuf<int> dstale(1);
        std::vector<int> zero(1, 0);
        dstale.up(zero, st);
        ANTQ_OK_(antq_nearest_hinted(dxf.p, dout.p, didx.p, n, dgrid.p, (int)flint.size(), plan.data(), dplan.p, dstale.p, ANTQ_F32, st));
        antq_oracle_nearest_f32(xf.data(), ref.data(), ridx.data(), n, flint.data(), (int)flint.size());
        same_bits("antq_nearest_hinted, right hint", dout.down(st), ref);
        same_idx("antq_nearest_hinted indices", didx.down(st), ridx);
        if (dstale.down(st)[0] != 0) { printf("stale flag raised on a right hint\n"); failures++; }
        std::vector<float> other(flint);                       // same size, other values at the same address
        for (auto &v : other) v *= 0.7f;
        other[3] = 1.0f;
        HIP_OK(hipMemcpyAsync(dgrid.p, other.data(), other.size() * 4, hipMemcpyHostToDevice, st));
        ANTQ_OK_(antq_nearest_hinted(dxf.p, dout.p, didx.p, n, dgrid.p, (int)other.size(), plan.data(), dplan.p, dstale.p, ANTQ_F32, st));
        antq_oracle_nearest_f32(xf.data(), ref.data(), ridx.data(), n, other.data(), (int)other.size());
        same_bits("antq_nearest_hinted, stale hint (scans the device grid)", dout.down(st), ref);
        same_idx("antq_nearest_hinted indices, stale hint", didx.down(st), ridx);
        if (dstale.down(st)[0] != 1) { printf("stale flag not raised\n"); failures++; }
    }

    // 7c. group-16 (rows := n / 16, row_len := 16): calibrated alpha, and the abs-max computed in the kernel
    {
        const size_t g_rows = n / 16;
        std::vector<float> ag(g_rows);
        antq_oracle_absmax_f32(xf.data(), ag.data(), g_rows, 16, 1, 1.0f);
        DevBuf<float> dag(g_rows), dag2(g_rows);
        dag.up(ag, st);
        ANTQ_OK_(antq_fakequant(dxf.p, dout.p, nullptr, g_rows, 16, dag.p, 1, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st));
        antq_oracle_forward_f32(xf.data(), ref.data(), ridx.data(), g_rows, 16, ag.data(), 1, flint.data(), (int)flint.size(), 10.0f, 0);
        same_bits("antq_fakequant group-16 (exact decision on x)", dout.down(st), ref);
        ANTQ_OK_(antq_fakequant_dynamic(dxf.p, dout.p, nullptr, dag2.p, g_rows, 16, 1.0f, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st));
        same_bits("antq_fakequant_dynamic group-16 alpha", dag2.down(st), ag);
        same_bits("antq_fakequant_dynamic group-16 values", dout.down(st), ref);
    }

    // 7d. calibration (search_mse, AQ:287-326): abs-max, every clip candidate's squared error on one read, selection on
    //     the device -- per row and per tensor (whole-tensor sums need the caller's workspace).  The scores are sums in another
    //     order than the oracle's, so a pick may differ only where the ORACLE's own scores tie within 2e-5.
    for (int per_row = 1; per_row >= 0; per_row--) {
        const size_t na = per_row ? rows : 1;
        const int lb = 75, ub = 150, ncand = ub - lb;
        std::vector<float> ratios((size_t)ncand), xmax(na), o_score(na), o_alpha(na), trace((size_t)ncand * na);
        for (int i = 0; i < ncand; i++) ratios[(size_t)i] = (float)((double)(lb + i) * 0.01);
        antq_oracle_absmax_f32(xf.data(), xmax.data(), rows, K, per_row, 1.0f);
        antq_oracle_search_mse_f32(xf.data(), rows, K, per_row, xmax.data(), lb, ub, 1, flint.data(), (int)flint.size(), 10.0f, 0,
                                   o_score.data(), o_alpha.data(), trace.data());
        DevBuf<float> dratios((size_t)ncand), dxmax(na), dscore(na), dalpha_best(na);
        DevBuf<double> dsse((size_t)ncand * na);
        DevBuf<uint8_t> dws(antq_search_workspace_bytes());
        dratios.up(ratios, st);
        ANTQ_OK_(antq_absmax(dxf.p, dxmax.p, rows, K, per_row, ANTQ_F32, st));
        same_bits(per_row ? "antq_absmax per row" : "antq_absmax per tensor", dxmax.down(st), xmax);
        if (!per_row && antq_search_sse(dxf.p, rows, K, dxmax.p, 0, dratios.p, ncand, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32,
                                        dsse.p, nullptr, st) != ANTQ_ERR_ARG) { printf("missing workspace not rejected\n"); failures++; }
        ANTQ_OK_(antq_search_sse(dxf.p, rows, K, dxmax.p, per_row, dratios.p, ncand, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32,
                                 dsse.p, dws.p, st));
        ANTQ_OK_(antq_search_pick(dsse.p, dxmax.p, dratios.p, ncand, na, per_row ? K : n, dscore.p, dalpha_best.p, st));
        const std::vector<double> sse = dsse.down(st);
        const std::vector<float> g_alpha = dalpha_best.down(st), g_score = dscore.down(st);
        int bad = 0, flips = 0;
        for (size_t r = 0; r < na; r++) {
            const double len = per_row ? (double)K : (double)n;
            int c_gpu = -1, c_ora = -1;
            for (int c = 0; c < ncand; c++) {
                const double want = (double)trace[(size_t)c * na + r];
                if (std::fabs(sse[(size_t)c * na + r] / len - want) > 2e-6 * want) bad++;          // fp32 terms, other order
                if (xmax[r] * ratios[(size_t)c] == g_alpha[r] && c_gpu < 0) c_gpu = c;
                if (xmax[r] * ratios[(size_t)c] == o_alpha[r] && c_ora < 0) c_ora = c;
            }
            if (c_gpu < 0 || c_ora < 0) { bad++; continue; }
            if (c_gpu != c_ora) {
                flips++;
                const double a = trace[(size_t)c_gpu * na + r], b = trace[(size_t)c_ora * na + r];
                if (std::fabs(a - b) > 2e-5 * b) bad++;                                           // not a tie by the oracle's scores
            }
        }
        printf("%-62s %s (%d of %zu picks differ, all ties)\n", per_row ? "antq_search_sse + antq_search_pick, per row"
                                                                       : "antq_search_sse + antq_search_pick, per tensor",
               bad ? "MISMATCH" : "ok", flips, na);
        if (bad) failures++;
    }

    // 7e. the same calibration as ONE call (antq_calibrate): abs-max, two candidate codebooks (flint-4 and a 4-bit int
    //     grid), per-row picks and the type pick, no host step in between.  Checked against the step-by-step entry points
    //     (bit for bit: same kernels) and against the oracle's per-type sums for the type.
    {
        std::vector<float> int4;
        for (int k = -8; k <= 7; k++) int4.push_back((float)k * (10.0f / 7.0f));
        std::vector<uint8_t> plan_i(ANTQ_PLAN_MAX_BYTES);
        const int pbi = antq_plan_build(int4.data(), (int)int4.size(), plan_i.data(), plan_i.size());
        if (pbi <= 0) { printf("int-4 plan build failed\n"); return 5; }
        DevBuf<uint8_t> dplan_i(ANTQ_PLAN_MAX_BYTES);
        HIP_OK(hipMemcpyAsync(dplan_i.p, plan_i.data(), pbi, hipMemcpyHostToDevice, st));
        const int lb = 75, ub = 150, ncand = ub - lb;
        const void *ph[2] = {plan_i.data(), plan.data()}, *pd[2] = {dplan_i.p, dplan.p};
        const float gm[2] = {10.0f, 10.0f};
        const std::vector<float> *grids2[2] = {&int4, &flint};
        for (int per_row = 1; per_row >= 0; per_row--) {
            const size_t na = per_row ? rows : 1;
            const size_t wsb = antq_calibrate_workspace_bytes(rows, per_row, lb, ub, 1, 2);
            DevBuf<uint8_t> dws(wsb), dws2(antq_search_workspace_bytes());
            DevBuf<float> dxm(na), dal(2 * na), dsc(2), dratios((size_t)ncand), dscore(na), dbest(na);
            DevBuf<int32_t> dty(1);
            DevBuf<double> dsse((size_t)ncand * na);
            ANTQ_OK_(antq_calibrate(dxf.p, rows, K, per_row, ANTQ_F32, ANTQ_XMAX_ABSMAX, dxm.p, lb, ub, 1, 2, gm, ph, pd, 0, dal.p,
                                    dsc.p, dty.p, dws.p, wsb, st));
            const std::vector<float> al = dal.down(st), sc = dsc.down(st), xm = dxm.down(st);
            const int ty = dty.down(st)[0];
            std::vector<float> ratios((size_t)ncand);
            for (int i = 0; i < ncand; i++) ratios[(size_t)i] = (float)((double)(lb + i) * 0.01);
            dratios.up(ratios, st);
            int bad = 0;
            double osum[2];
            for (int t = 0; t < 2; t++) {
                ANTQ_OK_(antq_search_sse(dxf.p, rows, K, dxm.p, per_row, dratios.p, ncand, 10.0f, ph[t], pd[t], 0, ANTQ_F32, dsse.p,
                                         dws2.p, st));
                ANTQ_OK_(antq_search_pick(dsse.p, dxm.p, dratios.p, ncand, na, per_row ? K : n, dscore.p, dbest.p, st));
                const std::vector<float> step_alpha = dbest.down(st), step_score = dscore.down(st);
                double ssum = 0.0;
                for (size_t r = 0; r < na; r++) {
                    if (std::memcmp(&step_alpha[r], &al[(size_t)t * na + r], 4) != 0) bad++;
                    ssum += (double)step_score[r];
                }
                if (std::fabs(ssum - (double)sc[t]) > 1e-6 * ssum) bad++;
                std::vector<float> o_score(na), o_alpha(na), trace((size_t)ncand * na);
                antq_oracle_search_mse_f32(xf.data(), rows, K, per_row, xm.data(), lb, ub, 1, grids2[t]->data(),
                                           (int)grids2[t]->size(), 10.0f, 0, o_score.data(), o_alpha.data(), trace.data());
                osum[t] = 0.0;
                for (size_t r = 0; r < na; r++) osum[t] += (double)o_score[r];
            }
            const int want = osum[1] < osum[0] ? 1 : 0;
            const bool tie = std::fabs(osum[0] - osum[1]) <= 1e-4 * std::min(osum[0], osum[1]);
            if (ty != want && !tie) bad++;
            if (ty != (sc[1] < sc[0] ? 1 : 0)) bad++;
            printf("%-62s %s (type %d, oracle sums %.6g / %.6g)\n", per_row ? "antq_calibrate (abs-max, 2 types, picks, type), per row"
                                                                             : "antq_calibrate (abs-max, 2 types, picks, type), per tensor",
                   bad ? "MISMATCH" : "ok", ty, osum[0], osum[1]);
            if (bad) failures++;
            if (!per_row) {
                // ABI 7: antq_calibrate_install -- from the pick that never left the device: alpha, score and codebook of the
                // winner, and the calibrating call's own output == antq_fakequant with the winner's plan at the winner's alpha
                std::vector<float> stack(32);
                for (int k = 0; k < 16; k++) { stack[(size_t)k] = int4[(size_t)k]; stack[16 + (size_t)k] = flint[(size_t)k]; }
                DevBuf<float> dstack(32), dgo(16), dao(1), dmo(1), dinst(n), dwant(n);
                dstack.up(stack, st);
                ANTQ_OK_(antq_calibrate_install(dxf.p, dinst.p, n, ANTQ_F32, 2, gm, ph, pd, 0, dty.p, dal.p, dsc.p, dstack.p, 16, dgo.p,
                                                nullptr, 0, nullptr, dao.p, dmo.p, st));
                ANTQ_OK_(antq_fakequant(dxf.p, dwant.p, nullptr, 1, n, dal.p + ty, 0, 10.0f, ph[ty], pd[ty], 0, ANTQ_F32, st));
                same_bits("antq_calibrate_install: output == the winner's antq_fakequant", dinst.down(st), dwant.down(st));
                const std::vector<float> go = dgo.down(st), ao = dao.down(st), mo = dmo.down(st);
                const bool oki = std::memcmp(go.data(), stack.data() + 16 * (size_t)ty, 64) == 0 && std::memcmp(&ao[0], &al[(size_t)ty], 4) == 0 &&
                                 std::memcmp(&mo[0], &sc[(size_t)ty], 4) == 0;
                printf("%-62s %s\n", "antq_calibrate_install: alpha / score / codebook of the winner", oki ? "ok" : "FAIL");
                if (!oki) failures++;
            }
        }
    }

    // 9. ABI 4 / 5 entry points from the torch-free host ------------------------------------------------------------------
    // 9a. antq_nearest_plan: the operator through a host-known plan == the literal scan (KQ/quant_kernel.cu:20-38)
    {
        ANTQ_OK_(antq_nearest_plan(dxf.p, dout.p, didx.p, n, plan.data(), dplan.p, ANTQ_F32, st));
        antq_oracle_nearest_f32(xf.data(), ref.data(), ridx.data(), n, flint.data(), (int)flint.size());
        same_bits("antq_nearest_plan values", dout.down(st), ref);
        same_idx("antq_nearest_plan indices", didx.down(st), ridx);
        if (antq_plan_bytes(plan.data()) != pb) { printf("antq_plan_bytes != antq_plan_build's size\n"); failures++; }
    }

    // 9b. host models of the device element paths (pure CPU): antq_plan_eval_host == the literal scan on grid-domain inputs;
    //     antq_plan_eval_host_a == the oracle's forward for one scale, whatever the reciprocal's last bit
    {
        const size_t ne = 4096;
        std::vector<float> d(ne), q(ne), qref(ne);
        std::vector<int16_t> qi(ne);
        std::vector<int32_t> qiref(ne);
        for (size_t i = 0; i < ne; i++) d[i] = x[i] * 400.0f;
        d[5] = NAN; d[7] = INFINITY;
        ANTQ_OK_(antq_plan_eval_host(plan.data(), d.data(), q.data(), qi.data(), ne));
        antq_oracle_nearest_f32(d.data(), qref.data(), qiref.data(), ne, flint.data(), (int)flint.size());
        same_bits("antq_plan_eval_host == literal scan", q, qref);
        same_idx("antq_plan_eval_host indices", qi, qiref);
        const float a_one = 0.071f;
        std::vector<float> xs(xf.begin(), xf.begin() + ne), oa(ne), oref(ne);
        antq_oracle_forward_f32(xs.data(), oref.data(), qiref.data(), 1, ne, &a_one, 0, flint.data(), (int)flint.size(), 10.0f, 0);
        for (int ulps = -1; ulps <= 1; ulps++) {
            const int rc = antq_plan_eval_host_a(plan.data(), xs.data(), ne, a_one, 10.0f, ulps, oa.data(), qi.data(), nullptr);
            if (rc == ANTQ_ERR_UNSUPPORTED) { printf("antq_plan_eval_host_a: plan without that path (skipped)\n"); break; }
            ANTQ_OK_(rc);
            same_bits(ulps == -1 ? "antq_plan_eval_host_a, reciprocal -1 ulp" : ulps == 0 ? "antq_plan_eval_host_a, exact reciprocal"
                                                                                          : "antq_plan_eval_host_a, reciprocal +1 ulp", oa, oref);
            same_idx("antq_plan_eval_host_a indices", qi, qiref);
        }
    }

    // 9c. the 16-bit-domain row path: its host model (antq_plan_eval_host_h) and the kernel (antq_fakequant on bf16 rows of
    //     4096) against the oracle's bf16 forward, ANT and OliVe pairs
    {
        std::vector<uint16_t> xb(n), refb(n), hostb(K);
        antq_oracle_f32_to_bf16(xf.data(), xb.data(), n);
        DevBuf<uint16_t> dxb(n), doutb(n);
        dxb.up(xb, st);
        for (int ovp = 0; ovp < 2; ovp++) {
            const std::vector<float> &gr = ovp ? olive : flint;
            const std::vector<float> &al = ovp ? alpha3 : alpha;
            const float gm = ovp ? 32.0f : 10.0f;
            antq_oracle_forward_bf16(xb.data(), refb.data(), ridx.data(), rows, K, al.data(), 1, gr.data(), (int)gr.size(), gm, ovp);
            ANTQ_OK_(antq_fakequant(dxb.p, doutb.p, nullptr, rows, K, ovp ? dalpha3.p : dalpha.p, 1, gm, ovp ? plan2.data() : plan.data(),
                                    ovp ? dplan2.p : dplan.p, ovp ? ANTQ_FLAG_OVP : 0u, ANTQ_BF16, st));
            const std::vector<uint16_t> got = doutb.down(st);
            size_t bad = 0, badh = 0;
            for (size_t i = 0; i < n; i++) bad += got[i] != refb[i] && !((got[i] & 0x7fff) > 0x7f80 && (refb[i] & 0x7fff) > 0x7f80);
            for (size_t r = 0; r < rows; r += 7) {
                const int rc = antq_plan_eval_host_h(ovp ? plan2.data() : plan.data(), xb.data() + r * K, K, al[r], gm, ANTQ_BF16,
                                                     ovp ? ANTQ_FLAG_OVP : 0u, hostb.data(), nullptr);
                if (rc != ANTQ_OK) { badh += K; continue; }
                for (size_t c = 0; c < K; c++) {
                    const uint16_t a = hostb[c], b = refb[r * K + c];
                    badh += a != b && !((a & 0x7fff) > 0x7f80 && (b & 0x7fff) > 0x7f80);
                }
            }
            printf("%-58s %s (%zu kernel / %zu host-model elements differ)\n", ovp ? "bf16 rows of 4096, 16-bit domain, OliVe pairs"
                                                                                   : "bf16 rows of 4096, 16-bit domain, ANT flint-4",
                   (bad || badh) ? "FAIL" : "ok", bad, badh);
            if (bad || badh) failures++;
        }
    }

    // 9d. antq_absmax_into: a running maximum over a tensor that arrives in two pieces, from a zeroed slot
    {
        DevBuf<float> dslot(1);
        HIP_OK(hipMemsetAsync(dslot.p, 0, 4, st));
        const size_t half = (n / 2) & ~(size_t)7;
        ANTQ_OK_(antq_absmax_into(dxf.p, dslot.p, half, ANTQ_F32, st));
        ANTQ_OK_(antq_absmax_into(dxf.p + half, dslot.p, n - half, ANTQ_F32, st));
        std::vector<float> want(1);
        antq_oracle_absmax_f32(xf.data(), want.data(), 1, n, 0, 1.0f);
        same_bits("antq_absmax_into, two pieces into one slot", dslot.down(st), want);
    }

    // 9e. antq_search_sse_multi: the sums of two codebooks on ONE read == one antq_search_sse per codebook, bit for bit
    //     (same kernels' fixed summation order), per row and per tensor
    {
        std::vector<float> int4;
        for (int k = -8; k <= 7; k++) int4.push_back((float)k * (10.0f / 7.0f));
        std::vector<uint8_t> plan_i(ANTQ_PLAN_MAX_BYTES);
        const int pbi = antq_plan_build(int4.data(), (int)int4.size(), plan_i.data(), plan_i.size());
        DevBuf<uint8_t> dplan_i(ANTQ_PLAN_MAX_BYTES);
        HIP_OK(hipMemcpyAsync(dplan_i.p, plan_i.data(), pbi, hipMemcpyHostToDevice, st));
        const void *ph[2] = {plan_i.data(), plan.data()}, *pd[2] = {dplan_i.p, dplan.p};
        const float gm[2] = {10.0f, 10.0f};
        const int lb = 80, ub = 120, ncand = ub - lb;
        std::vector<float> ratios((size_t)ncand);
        for (int i = 0; i < ncand; i++) ratios[(size_t)i] = (float)((double)(lb + i) * 0.01);
        DevBuf<float> dratios((size_t)ncand);
        dratios.up(ratios, st);
        DevBuf<uint8_t> dws(antq_search_workspace_bytes());
        for (int per_row = 1; per_row >= 0; per_row--) {
            const size_t na = per_row ? rows : 1;
            DevBuf<float> dxm(na);
            DevBuf<double> dm(2 * (size_t)ncand * na), ds((size_t)ncand * na);
            ANTQ_OK_(antq_absmax(dxf.p, dxm.p, rows, K, per_row, ANTQ_F32, st));
            const int rc = antq_search_sse_multi(dxf.p, rows, K, dxm.p, per_row, dratios.p, ncand, 2, gm, ph, pd, 0, ANTQ_F32, dm.p, dws.p, st);
            ANTQ_OK_(rc);
            const std::vector<double> multi = dm.down(st);
            size_t bad = 0;
            for (int t = 0; t < 2; t++) {
                ANTQ_OK_(antq_search_sse(dxf.p, rows, K, dxm.p, per_row, dratios.p, ncand, 10.0f, ph[t], pd[t], 0, ANTQ_F32, ds.p, dws.p, st));
                const std::vector<double> one = ds.down(st);
                for (size_t i = 0; i < one.size(); i++) bad += std::memcmp(&one[i], &multi[(size_t)t * one.size() + i], 8) != 0;
            }
            printf("%-58s %s (%zu sums differ)\n", per_row ? "antq_search_sse_multi == per-type sums, per row" : "antq_search_sse_multi == per-type sums, per tensor",
                   bad ? "FAIL" : "ok", bad);
            if (bad) failures++;
        }
    }

    // 9f. antq_moments + antq_xmax_3sigma (OliVe's clip statistic, OQ:193-197 / :213-218) against a double-precision host
    //     computation of mean / unbiased std (the reference's torch reductions agree to their own summation noise)
    {
        DevBuf<double> dsums(2 * rows);
        DevBuf<float> dx3(rows);
        DevBuf<uint8_t> dws(antq_search_workspace_bytes());
        for (int per_row = 1; per_row >= 0; per_row--) {
            const size_t na = per_row ? rows : 1, per = per_row ? K : n;
            ANTQ_OK_(antq_moments(dxf.p, rows, K, per_row, ANTQ_F32, dsums.p, dws.p, st));
            ANTQ_OK_(antq_xmax_3sigma(dsums.p, na, per, ANTQ_F32, dx3.p, st));
            const std::vector<double> sums = dsums.down(st);
            const std::vector<float> x3 = dx3.down(st);
            size_t bad = 0;
            for (size_t r = 0; r < na; r++) {
                double s1 = 0.0, s2 = 0.0;
                for (size_t c = 0; c < per; c++) { const double v = xf[r * per + c]; s1 += v; s2 += v * v; }
                if (std::fabs(sums[2 * r] - s1) > 1e-9 * (std::fabs(s1) + 1.0) || std::fabs(sums[2 * r + 1] - s2) > 1e-9 * s2) bad++;
                const double mean = s1 / (double)per, var = (s2 - s1 * s1 / (double)per) / (double)(per - 1), sd = std::sqrt(var);
                const double want = std::max(std::fabs(mean + 3.0 * sd), std::fabs(mean - 3.0 * sd));
                if (std::fabs((double)x3[r] - want) > 2e-6 * want) bad++;
            }
            printf("%-58s %s (%zu of %zu off)\n", per_row ? "antq_moments + antq_xmax_3sigma, per row" : "antq_moments + antq_xmax_3sigma, per tensor",
                   bad ? "FAIL" : "ok", bad, na);
            if (bad) failures++;
        }
    }

    // 9g. antq_affine (AsymmetricQuantFunction.forward, quant_affine.py:95-115) against the oracle, per row and per tensor
    {
        std::vector<float> mn(rows), mx(rows);
        for (size_t r = 0; r < rows; r++) {
            mn[r] = mx[r] = xf[r * K];
            for (size_t c = 1; c < K; c++) { mn[r] = std::min(mn[r], xf[r * K + c]); mx[r] = std::max(mx[r], xf[r * K + c]); }
        }
        DevBuf<float> dmn(rows), dmx(rows);
        DevBuf<int32_t> dq(n);
        dmn.up(mn, st); dmx.up(mx, st);
        std::vector<int32_t> qref(n);
        for (int per_row = 1; per_row >= 0; per_row--) {
            for (int k : {8, 4}) {
                ANTQ_OK_(antq_affine(dxf.p, dout.p, dq.p, rows, K, k, dmn.p, dmx.p, per_row, st));
                antq_oracle_affine_f32(xf.data(), ref.data(), qref.data(), rows, K, k, mn.data(), mx.data(), per_row);
                same_bits(per_row ? (k == 8 ? "antq_affine 8-bit per row" : "antq_affine 4-bit per row")
                                  : (k == 8 ? "antq_affine 8-bit per tensor" : "antq_affine 4-bit per tensor"), dout.down(st), ref);
                const std::vector<int32_t> gq = dq.down(st);
                size_t bad = 0;
                for (size_t i = 0; i < n; i++) bad += gq[i] != qref[i];
                if (bad) { printf("antq_affine integer codes: %zu differ\n", bad); failures++; }
            }
        }
    }

    // 9h. antq_alpha_grad: gsum[r] = sum_c fl32(g * fl32(out - x)) with fp32 terms and fp64 accumulation (the backward of
    //     AQ:535-551 with respect to alpha, before the division by alpha) against the same sum formed on the host
    {
        std::vector<float> gout(n);
        for (size_t i = 0; i < n; i++) gout[i] = nd(rng) * 50.0f;
        DevBuf<float> dg(n);
        dg.up(gout, st);
        DevBuf<uint8_t> dws(antq_search_workspace_bytes());
        ANTQ_OK_(antq_fakequant(dxf.p, dout.p, nullptr, rows, K, dalpha.p, 1, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st));
        const std::vector<float> o = dout.down(st);
        for (int per_row = 1; per_row >= 0; per_row--) {
            const size_t na = per_row ? rows : 1, per = per_row ? K : n;
            DevBuf<double> dgs(na);
            ANTQ_OK_(antq_alpha_grad(dxf.p, dout.p, dg.p, rows, K, per_row, dgs.p, dws.p, ANTQ_F32, st));
            const std::vector<double> gs = dgs.down(st);
            size_t bad = 0;
            for (size_t r = 0; r < na; r++) {
                double want = 0.0, mag = 0.0;
                for (size_t c = 0; c < per; c++) {
                    const size_t i = r * per + c;
                    const float diff = o[i] - xf[i];
                    const float term = gout[i] * diff;
                    want += (double)term;
                    mag += std::fabs((double)term);
                }
                if (std::fabs(gs[r] - want) > 1e-6 * mag + 1e-30) bad++;      // (fp32 partial sums per lane: 1e-6 of the magnitude)
            }
            printf("%-58s %s (%zu of %zu off)\n", per_row ? "antq_alpha_grad per row" : "antq_alpha_grad per tensor", bad ? "FAIL" : "ok", bad, na);
            if (bad) failures++;
        }
    }

    // 9h'. ABI 7: the same two whole-tensor reductions in ONE launch each through a ticket block zeroed once
    //      (antq_absmax_t writes the maximum: the slot starts as garbage; antq_alpha_grad_t: one fixed-order tree), three
    //      calls back to back through the SAME block (every call leaves it zeroed), then the block is checked to be zero
    {
        DevBuf<uint8_t> dred(ANTQ_REDUCE_WS_BYTES);
        HIP_OK(hipMemsetAsync(dred.p, 0, ANTQ_REDUCE_WS_BYTES, st));
        std::vector<float> want(1);
        antq_oracle_absmax_f32(xf.data(), want.data(), 1, n, 0, 1.0f);
        for (int rep = 0; rep < 3; rep++) {
            DevBuf<float> dslot(1);
            HIP_OK(hipMemsetAsync(dslot.p, 0x7f, 4, st));          // garbage: the entry point WRITES its result
            ANTQ_OK_(antq_absmax_t(dxf.p, dslot.p, n - (size_t)rep, ANTQ_F32, dred.p, st));
            std::vector<float> w2(1);
            antq_oracle_absmax_f32(xf.data(), w2.data(), 1, n - (size_t)rep, 0, 1.0f);
            same_bits("antq_absmax_t (one launch, ticket block)", dslot.down(st), w2);
        }
        std::vector<float> gout(n);
        for (size_t i = 0; i < n; i++) gout[i] = nd(rng) * 50.0f;
        DevBuf<float> dg(n);
        dg.up(gout, st);
        ANTQ_OK_(antq_fakequant(dxf.p, dout.p, nullptr, rows, K, dalpha.p, 1, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st));
        const std::vector<float> o = dout.down(st);
        DevBuf<double> dgs(1);
        ANTQ_OK_(antq_alpha_grad_t(dxf.p, dout.p, dg.p, n, dgs.p, ANTQ_F32, dred.p, st));
        const std::vector<double> g1 = dgs.down(st);
        ANTQ_OK_(antq_alpha_grad_t(dxf.p, dout.p, dg.p, n, dgs.p, ANTQ_F32, dred.p, st));
        const std::vector<double> g2 = dgs.down(st);
        double wantg = 0.0, mag = 0.0;
        for (size_t i = 0; i < n; i++) {
            const float diff = o[i] - xf[i];
            const float term = gout[i] * diff;
            wantg += (double)term;
            mag += std::fabs((double)term);
        }
        const bool okg = std::fabs(g1[0] - wantg) <= 1e-6 * mag + 1e-30 && std::memcmp(&g1[0], &g2[0], 8) == 0;
        printf("%-58s %s\n", "antq_alpha_grad_t (one launch, bit-reproducible)", okg ? "ok" : "FAIL");
        if (!okg) failures++;
        const std::vector<uint8_t> red = dred.down(st);
        size_t nz = 0;
        for (size_t i = 0; i < 16384; i++) nz += red[i] != 0;     // (the counter region: what must be zero between calls)
        printf("%-58s %s\n", "ticket block left zeroed by every call", nz ? "FAIL" : "ok");
        if (nz) failures++;
    }

    // 9i. antq_calibrate_batch: three quantisers (per row, per tensor, per row on a sub-tensor) in ONE call == one
    //     antq_calibrate each, bit for bit (same kernels, same order), through one shared workspace
    {
        std::vector<float> int4;
        for (int k = -8; k <= 7; k++) int4.push_back((float)k * (10.0f / 7.0f));
        std::vector<uint8_t> plan_i(ANTQ_PLAN_MAX_BYTES);
        const int pbi = antq_plan_build(int4.data(), (int)int4.size(), plan_i.data(), plan_i.size());
        DevBuf<uint8_t> dplan_i(ANTQ_PLAN_MAX_BYTES);
        HIP_OK(hipMemcpyAsync(dplan_i.p, plan_i.data(), pbi, hipMemcpyHostToDevice, st));
        const void *ph[2] = {plan_i.data(), plan.data()}, *pd[2] = {dplan_i.p, dplan.p};
        const float gm[2] = {10.0f, 10.0f};
        struct Shape { size_t off, rows, K; int per_row, lb, ub; } shapes[3] = {{0, rows, K, 1, 75, 150}, {0, rows, K, 0, 90, 130},
                                                                               {16 * K, 32, 2 * K, 1, 95, 100}};
        antq_calib_job jobs3[3];
        std::vector<DevBuf<float> *> keep;
        DevBuf<int32_t> dty(3), dty1(1);
        size_t na_of[3];
        DevBuf<float> *bx[3], *ba[3], *bs[3];
        for (int j = 0; j < 3; j++) {
            const Shape &s = shapes[j];
            na_of[j] = s.per_row ? s.rows : 1;
            bx[j] = new DevBuf<float>(na_of[j]); ba[j] = new DevBuf<float>(2 * na_of[j]); bs[j] = new DevBuf<float>(2);
            jobs3[j] = {dxf.p + s.off, s.rows, s.K, s.per_row, ANTQ_XMAX_ABSMAX, bx[j]->p, s.lb, s.ub, 1, 2, gm, ph, pd, ba[j]->p, bs[j]->p,
                        dty.p + j};
        }
        const size_t wsb = antq_calibrate_batch_workspace_bytes(jobs3, 3);
        if (wsb == 0) { printf("antq_calibrate_batch_workspace_bytes returned 0\n"); failures++; }
        DevBuf<uint8_t> dws(wsb ? wsb : 16);
        ANTQ_OK_(antq_calibrate_batch(jobs3, 3, ANTQ_F32, 0, dws.p, wsb, st));
        const std::vector<int32_t> types = dty.down(st);
        size_t bad = 0;
        for (int j = 0; j < 3; j++) {
            const Shape &s = shapes[j];
            const std::vector<float> b_al = ba[j]->down(st), b_sc = bs[j]->down(st), b_xm = bx[j]->down(st);
            const size_t w1 = antq_calibrate_workspace_bytes(s.rows, s.per_row, s.lb, s.ub, 1, 2);
            if (w1 > wsb) { printf("batch workspace smaller than job %d's own\n", j); bad++; }
            DevBuf<uint8_t> dws1(w1);
            DevBuf<float> dxm(na_of[j]), dal(2 * na_of[j]), dsc(2);
            ANTQ_OK_(antq_calibrate(dxf.p + s.off, s.rows, s.K, s.per_row, ANTQ_F32, ANTQ_XMAX_ABSMAX, dxm.p, s.lb, s.ub, 1, 2, gm, ph, pd, 0,
                                    dal.p, dsc.p, dty1.p, dws1.p, w1, st));
            const std::vector<float> o_al = dal.down(st), o_sc = dsc.down(st), o_xm = dxm.down(st);
            bad += std::memcmp(o_al.data(), b_al.data(), 4 * o_al.size()) != 0;
            bad += std::memcmp(o_sc.data(), b_sc.data(), 8) != 0;
            bad += std::memcmp(o_xm.data(), b_xm.data(), 4 * o_xm.size()) != 0;
            bad += dty1.down(st)[0] != types[(size_t)j];
            delete bx[j]; delete ba[j]; delete bs[j];
        }
        printf("%-58s %s (types %d %d %d)\n", "antq_calibrate_batch == antq_calibrate per job", bad ? "FAIL" : "ok", types[0], types[1], types[2]);
        if (bad) failures++;
        jobs3[1].ub = jobs3[1].lb - 5;                                        // an empty range is legal (AQ:299: the loop never runs) ...
        jobs3[1].ntypes = 0;                                                  // ... no codebook is not
        if (antq_calibrate_batch_workspace_bytes(jobs3, 3) != 0) { printf("bad job not rejected by the workspace query\n"); failures++; }
    }

    // 9k. antq_fakequant_f64: a double tensor through the fused kernel == the reference's double op sequence around its
    //     float-narrowing operator (KQ/quant_kernel.cu:28, :51; AQ:535-551 / OQ:311-320), restated here on the oracle's scan
    {
        const size_t nd = rows * K - 1;                                       // odd element count: the pair rule's wrap
        std::vector<double> xd(nd), ad(1, 0.0731), od(nd), dd(nd), gd(olive.size()), qd(nd);
        for (size_t i = 0; i < nd; i++) xd[i] = (double)xf[i] * 1.000000123;
        for (size_t i = 0; i < olive.size(); i++) gd[i] = (double)olive[i];
        const double s = ad[0] / 32.0;
        for (size_t i = 0; i < nd; i++) dd[i] = xd[i] / s;
        std::vector<int32_t> qi(nd);
        antq_oracle_nearest_f64(dd.data(), qd.data(), qi.data(), nd, gd.data(), (int)gd.size());
        std::vector<uint8_t> mask(nd), vo(nd, 0), ve(nd, 0);
        for (size_t i = 0; i < nd; i++) mask[i] = std::fabs(qd[i]) > 32.0;
        for (size_t i = 1; i < nd; i += 2) vo[i] = mask[i - 1];
        for (size_t i = 0; i < nd; i += 2) { const size_t j = (i + 1) % nd; ve[i] = mask[j] && !vo[j]; }
        for (size_t i = 0; i < nd; i++) { const double q = qd[i] * ((ve[i] || vo[i]) ? 0.0 : 1.0); od[i] = ((q - dd[i]) + dd[i]) * s; }
        DevBuf<double> dxd(nd), dod(nd), dad(1);
        dxd.up(xd, st); dad.up(ad, st);
        ANTQ_OK_(antq_fakequant_f64(dxd.p, dod.p, 1, nd, dad.p, 0, 32.0, plan2.data(), dplan2.p, ANTQ_FLAG_OVP, st));
        const std::vector<double> got = dod.down(st);
        size_t bad = 0;
        for (size_t i = 0; i < nd; i++) bad += std::memcmp(&got[i], &od[i], 8) != 0 && !(std::isnan(got[i]) && std::isnan(od[i]));
        printf("%-58s %s (%zu / %zu differ)\n", "antq_fakequant_f64, OliVe pairs, odd element count", bad ? "FAIL" : "ok", bad, nd);
        if (bad) failures++;
    }

    // 9j. the rest of the surface: antq_copy, antq_prefetch_kernels, antq_debug_set (thread-local knob, result unchanged)
    {
        ANTQ_OK_(antq_prefetch_kernels());
        ANTQ_OK_(antq_copy(dxf.p, dout.p, n * sizeof(float), st));
        same_bits("antq_copy", dout.down(st), xf);
        ANTQ_OK_(antq_debug_set(9, 0));                                       // 16-bit-domain kernels off: fp32-domain row table
        ANTQ_OK_(antq_fakequant(dxf.p, dout.p, nullptr, rows, K, dalpha.p, 1, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st));
        ANTQ_OK_(antq_debug_set(9, 1));
        antq_oracle_forward_f32(xf.data(), ref.data(), ridx.data(), rows, K, alpha.data(), 1, flint.data(), (int)flint.size(), 10.0f, 0);
        same_bits("antq_fakequant under antq_debug_set(9, 0)", dout.down(st), ref);
        if (antq_debug_set(-1, 0) == ANTQ_OK) { printf("unknown debug key accepted\n"); failures++; }
    }

    // 8. error behaviour: codes, not exceptions
    if (antq_fakequant(nullptr, dout.p, nullptr, rows, K, dalpha.p, 1, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st) != ANTQ_ERR_ARG) { printf("null x not rejected\n"); failures++; }
    if (antq_fakequant(dx.p, dout.p, nullptr, rows, K, dalpha.p, 1, 10.0f, plan.data(), dplan.p, 0, 99, st) != ANTQ_ERR_UNSUPPORTED) { printf("bad dtype not rejected\n"); failures++; }
    plan[0] ^= 0xff;
    if (antq_fakequant(dx.p, dout.p, nullptr, rows, K, dalpha.p, 1, 10.0f, plan.data(), dplan.p, 0, ANTQ_F32, st) != ANTQ_ERR_PLAN) { printf("corrupt plan not rejected\n"); failures++; }
    printf("error codes: ok\n");

    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipStreamDestroy(st));
    printf(failures ? "CABI CHECK FAILED (%d)\n" : "CABI CHECK OK\n", failures);
    return failures ? 1 : 0;
}
