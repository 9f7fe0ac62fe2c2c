// antq_torch.cpp -- the compiled torch extension on top of the C ABI (include/antq.h).
//
// The reference's operator is a pybind11 extension module `quant_cuda` with one function,
//     quant(x, grid) -> (z, idx)                    ant_quantization/quant/quant.cpp:17-29, setup.py:6-17
// This file is its MI355X counterpart: the same function, compiled, running the hand-written gfx950 kernels of libantq.so on
// the CURRENT torch stream of x's device (the reference used the legacy default stream), plus thin compiled fast paths for
// the fused entry points so that small tensors are not bound by Python / ctypes overhead (a ctypes call of antq_fakequant
// costs ~6.5 us of host time; the launch itself ~2.5 us).
//
// Built in-tree by csrc/Makefile (target `ext`, g++ against the torch headers; no device code here) as
// ant_quantization_amd/_antq_ext*.so; ant_quantization_amd/quant_cuda.py re-exports `quant` from it and falls back to its
// ctypes implementation when the extension is absent.  No CPU fallback: a CPU tensor raises.
//
// "Speed without trust" (see quant_cuda.py): a plan remembered for a grid ADDRESS is only a hint -- the kernel
// (antq_nearest_hinted) compares the device grid with the plan's copy and scans the device values literally when they
// differ, raising a pinned flag the next call reads.  A wrong belief costs microseconds, never a wrong result.
#include <torch/extension.h>
#include <pybind11/numpy.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/antq.h"

namespace {

inline int dtype_code(const at::Tensor &t)
{
    switch (t.scalar_type()) {
    case at::kFloat: return ANTQ_F32;
    case at::kBFloat16: return ANTQ_BF16;
    case at::kHalf: return ANTQ_F16;
    case at::kDouble: return ANTQ_F64;
    default: return -1;
    }
}

inline void check_rc(int rc, const char *what)
{
    TORCH_CHECK(rc == ANTQ_OK, what, " failed: ", antq_strerror(rc), " (", rc, ")");
}

inline void require_gpu(const at::Tensor &t, const char *name)
{
    TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device (got ", t.device(), "); libantq has no CPU path");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

inline void *current_stream(const at::Tensor &t)
{
    return static_cast<void *>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

// ------------------------------------------------------------------------------------------------
// quant(x, grid): hints
// ------------------------------------------------------------------------------------------------
struct Hint {
    std::vector<unsigned char> plan_host;     // empty: no belief
    at::Tensor plan_dev;                      // the plan blob on the grid's device
    std::vector<float> grid;                  // the values the plan was built from (introspection / tests)
    int *stale = nullptr;                     // slot of the pinned pool
    int seen = 0, need = 2, strikes = 0;
};

struct Key {
    uintptr_t ptr;
    int64_t numel;
    int dev;
    bool operator==(const Key &o) const { return ptr == o.ptr && numel == o.numel && dev == o.dev; }
};
struct KeyHash {
    size_t operator()(const Key &k) const { return std::hash<uintptr_t>()(k.ptr) ^ (std::hash<int64_t>()(k.numel) << 1) ^ (size_t)k.dev; }
};

constexpr size_t kMaxHints = 64;
constexpr int kStaleSlots = 4096;

// nn.DataParallel calls quant() from one thread per GPU.  Every entry point of this module runs with the GIL held from
// start to end (nothing here releases it), so a Hint returned by hint_for is not modified by another thread while quant()
// uses it; the mutex keeps the cache itself consistent should a caller ever release the GIL around these calls.
std::mutex g_mu;
// The cache and the pinned pool hold device / pinned tensors: they are heap objects that are never destroyed (static
// destruction at process exit can run after the HIP runtime and the caching host allocator are gone).
using Lru = std::list<std::pair<Key, std::shared_ptr<Hint>>>;
Lru &g_lru = *new Lru;                                       // front = most recent
std::unordered_map<Key, Lru::iterator, KeyHash> &g_map = *new std::unordered_map<Key, Lru::iterator, KeyHash>;
at::Tensor &g_stale_pool = *new at::Tensor;                  // pinned int32[kStaleSlots]: never returned to the allocator, so a
int g_stale_next = 0;                                        // late write of a long-gone launch can only cost a re-plan

// (after kStaleSlots beliefs a slot is handed out again: two live beliefs may then share a flag, and one going stale
//  makes the other re-plan once -- slower, never wrong)
int *new_stale_flag()
{
    if (!g_stale_pool.defined())
        g_stale_pool = at::zeros({kStaleSlots}, at::TensorOptions().dtype(at::kInt).pinned_memory(true));
    int *p = g_stale_pool.data_ptr<int>() + (g_stale_next++ % kStaleSlots);
    *p = 0;
    return p;
}

std::shared_ptr<Hint> hint_for(const at::Tensor &grid)
{
    const Key key{reinterpret_cast<uintptr_t>(grid.data_ptr()), grid.numel(), (int)grid.device().index()};
    std::lock_guard<std::mutex> lock(g_mu);
    std::shared_ptr<Hint> h;
    auto it = g_map.find(key);
    if (it == g_map.end()) {
        h = std::make_shared<Hint>();
        g_lru.emplace_front(key, h);
        g_map[key] = g_lru.begin();
        if (g_map.size() > kMaxHints) {
            g_map.erase(g_lru.back().first);
            g_lru.pop_back();
        }
    } else {
        g_lru.splice(g_lru.begin(), g_lru, it->second);
        h = it->second->second;
    }
    if (!h->plan_host.empty() && *h->stale != 0) {
        // an earlier launch found other values at this address: forget, and be slower to believe again
        h->plan_host.clear();
        h->plan_dev = at::Tensor();
        h->seen = 0;
        h->strikes += 1;
        h->need = std::min(2 << h->strikes, 256);
    }
    if (h->plan_host.empty()) {
        h->seen += 1;
        if (h->seen >= h->need) {
            // the one read-back per long-lived buffer
            at::Tensor g = grid.detach().to(at::kFloat).cpu().contiguous();
            const int m = (int)g.numel();
            h->grid.assign(g.data_ptr<float>(), g.data_ptr<float>() + m);
            std::vector<unsigned char> blob(ANTQ_PLAN_MAX_BYTES);
            const int nb = antq_plan_build(h->grid.data(), m, blob.data(), blob.size());
            if (nb > 0) {
                blob.resize((size_t)nb);
                at::Tensor host = at::empty({nb}, at::TensorOptions().dtype(at::kByte));
                std::memcpy(host.data_ptr(), blob.data(), (size_t)nb);
                h->plan_dev = host.to(grid.device());
                h->plan_host = std::move(blob);
                h->stale = new_stale_flag();
            }
        }
    }
    return h;
}

// quant_cuda.quant (quant.cpp:17-29): z = nearest grid value of every element of the 1-D tensor x under the scan's rule
// (last minimum wins, 0 beyond 102400); idx = the reference's second output, a fresh all-zero tensor shaped like x that its
// kernel never writes (quant_kernel.cu:18,49).
std::tuple<at::Tensor, at::Tensor> quant(const at::Tensor &x_in, const at::Tensor &y_in)
{
    TORCH_CHECK(x_in.dim() == 1, "quant_cuda.quant: x must be 1-D (got ", x_in.dim(), "-D)");
    at::Tensor x = x_in.contiguous(), y = y_in.contiguous();
    require_gpu(x, "x");
    require_gpu(y, "grid");
    const int dt = dtype_code(x);
    TORCH_CHECK(dt >= 0, "unsupported dtype ", x.scalar_type());
    const at::ScalarType want = (dt == ANTQ_F32 || dt == ANTQ_F64) ? x.scalar_type() : at::kFloat;
    TORCH_CHECK(y.scalar_type() == want, "grid dtype ", y.scalar_type(), ", expected ", want);
    c10::hip::OptionalHIPGuard guard;
    if (x.device().index() != c10::hip::current_device()) guard.set_device(x.device());
    at::Tensor z = at::empty_like(x);
    void *st = current_stream(x);
    const size_t n = (size_t)x.numel();
    bool done = false;
    if (dt == ANTQ_F32 && y.numel() > 0 && y.numel() <= ANTQ_MAX_GRID && n % 4 == 0 &&
        reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0 && n > 0) {
        std::shared_ptr<Hint> h = hint_for(y);
        if (!h->plan_host.empty() && (int64_t)h->grid.size() == y.numel()) {
            check_rc(antq_nearest_hinted(x.data_ptr(), z.data_ptr(), nullptr, n, y.data_ptr<float>(), (int)y.numel(),
                                         h->plan_host.data(), h->plan_dev.data_ptr(), h->stale, dt, st),
                     "antq_nearest_hinted");
            done = true;
        }
    }
    if (!done)
        check_rc(antq_nearest(x.data_ptr(), z.data_ptr(), nullptr, n, y.data_ptr(), (int)y.numel(), dt, st), "antq_nearest");
    return std::make_tuple(z, at::zeros_like(x));
}

// introspection for the tests: None, or (has_plan, stale flag, grid values the belief was built from)
py::object hint_info(uintptr_t ptr, int64_t numel, int dev)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_map.find(Key{ptr, numel, dev});
    if (it == g_map.end()) return py::none();
    const Hint &h = *it->second->second;
    const bool has = !h.plan_host.empty();
    return py::make_tuple(has, has ? *h.stale : 0, py::array_t<float>((py::ssize_t)h.grid.size(), h.grid.data()));
}

void hints_clear()
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_map.clear();
    g_lru.clear();
}

// ------------------------------------------------------------------------------------------------
// fused fast paths (arguments already validated once by the Python layer that owns the plan objects)
// ------------------------------------------------------------------------------------------------
py::object fakequant(const at::Tensor &x, const at::Tensor &alpha, uintptr_t plan_host, uintptr_t plan_dev, double gmax,
                     int64_t rows, int64_t row_len, bool per_row, unsigned flags, const c10::optional<at::Tensor> &out_opt,
                     bool want_idx)
{
    require_gpu(x, "x");
    require_gpu(alpha, "alpha");
    const int dt = dtype_code(x);
    TORCH_CHECK(dt >= 0 && dt != ANTQ_F64, "unsupported dtype ", x.scalar_type());
    TORCH_CHECK(alpha.scalar_type() == at::kFloat, "alpha must be float32");
    TORCH_CHECK(rows * row_len == x.numel(), "rows*row_len != numel");
    TORCH_CHECK(alpha.numel() == (per_row ? rows : 1), "alpha has ", alpha.numel(), " entries, expected ", per_row ? rows : 1);
    TORCH_CHECK(alpha.device() == x.device(), "alpha lives on ", alpha.device(), ", x on ", x.device());
    if (out_opt.has_value()) {
        const at::Tensor &o = *out_opt;
        TORCH_CHECK(o.is_cuda() && o.device() == x.device() && o.is_contiguous() && o.scalar_type() == x.scalar_type() &&
                        o.numel() == x.numel(),
                    "out must be a contiguous tensor of x's dtype, element count and device");
    }
    // an unordered launch may overlap whatever the stream ran last: its output must be a buffer the CALLER owns (a block the
    // caching allocator just recycled can still be in use by a kernel in flight), and the index tensor would be such a block
    TORCH_CHECK(!(flags & ANTQ_FLAG_UNORDERED) || (out_opt.has_value() && !want_idx),
                "ANTQ_FLAG_UNORDERED needs a caller-owned `out` and no index output");
    c10::hip::OptionalHIPGuard guard;
    if (x.device().index() != c10::hip::current_device()) guard.set_device(x.device());
    at::Tensor out = out_opt.has_value() ? *out_opt : at::empty_like(x);
    at::Tensor idx;
    if (want_idx) idx = at::empty(x.sizes(), x.options().dtype(at::kShort));
    check_rc(antq_fakequant(x.data_ptr(), out.data_ptr(), want_idx ? idx.data_ptr<int16_t>() : nullptr, (size_t)rows,
                            (size_t)row_len, alpha.data_ptr<float>(), per_row ? 1 : 0, (float)gmax,
                            reinterpret_cast<const void *>(plan_host), reinterpret_cast<const void *>(plan_dev), flags, dt,
                            current_stream(x)),
             "antq_fakequant");
    if (want_idx) return py::make_tuple(out, idx);
    return py::cast(out);
}

void batch_run(uintptr_t batch_host, const at::Tensor &batch_dev)
{
    c10::hip::OptionalHIPGuard guard;
    if (batch_dev.device().index() != c10::hip::current_device()) guard.set_device(batch_dev.device());
    check_rc(antq_fakequant_batch(reinterpret_cast<const void *>(batch_host), batch_dev.data_ptr(), current_stream(batch_dev)),
             "antq_fakequant_batch");
}

// _lib.absmax(per_row=False): the whole-tensor abs-max accumulated into a slot that already holds 0 (antq_absmax_into) --
// one pybind call instead of a ctypes call behind a device guard written in Python (9.6 -> ~5 us of host time per call: a
// 33.5 MB bf16 tensor's kernel takes 6.5 us, the Python path was host-bound)
void absmax_into(const at::Tensor &x, const at::Tensor &slot)
{
    require_gpu(x, "x");
    const int dt = dtype_code(x);
    TORCH_CHECK(dt >= 0 && dt != ANTQ_F64, "unsupported dtype ", x.scalar_type());
    TORCH_CHECK(slot.is_cuda() && slot.device() == x.device() && slot.scalar_type() == at::kFloat && slot.numel() == 1,
                "slot must be one float32 on x's device");
    c10::hip::OptionalHIPGuard guard;
    if (x.device().index() != c10::hip::current_device()) guard.set_device(x.device());
    check_rc(antq_absmax_into(x.data_ptr(), slot.data_ptr<float>(), (size_t)x.numel(), dt, current_stream(x)), "antq_absmax_into");
}

// weight_bank.WeightBank.refresh, default schedule (one batched re-quantisation of every weight per no-grad forward): the
// descriptor tables hold raw addresses, so before launching make sure every weight / alpha tensor still lives where the
// tables say (a dtype / device move, load_state_dict(assign=True), a rebound `.data`) and that no codebook buffer was
// edited (versions[i] >= 0: tensor i must also still carry that version counter; -1: contents may change, that is the
// point).  Returns false -- nothing launched -- otherwise: the Python side rebuilds its tables.
// batches: (host blob address, device blob).
bool bank_refresh(const std::vector<at::Tensor> &tensors, const std::vector<uintptr_t> &addrs,
                  const std::vector<int64_t> &versions, const std::vector<std::pair<uintptr_t, at::Tensor>> &batches)
{
    if (tensors.size() != addrs.size() || tensors.size() != versions.size()) return false;
    for (size_t i = 0; i < tensors.size(); i++) {
        if (reinterpret_cast<uintptr_t>(tensors[i].data_ptr()) != addrs[i]) return false;
        if (versions[i] >= 0 && (tensors[i].is_inference() || (int64_t)tensors[i]._version() != versions[i])) return false;
    }
    for (const auto &b : batches) batch_run(b.first, b.second);
    return true;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "MI355X counterpart of the reference's compiled `quant_cuda` extension (quant.cpp:27-29) + fused fast paths";
    m.def("quant", &quant, "nearest grid value of every element: quant(x, grid) -> (z, idx)");
    m.def("_hint_info", &hint_info);
    m.def("_hints_clear", &hints_clear);
    m.def("fakequant", &fakequant, py::arg("x"), py::arg("alpha"), py::arg("plan_host"), py::arg("plan_dev"), py::arg("gmax"),
          py::arg("rows"), py::arg("row_len"), py::arg("per_row"), py::arg("flags"), py::arg("out") = py::none(),
          py::arg("want_idx") = false);
    m.def("batch_run", &batch_run);
    m.def("bank_refresh", &bank_refresh);
    m.def("absmax_into", &absmax_into);
    m.def("abi_version", []() { return antq_abi_version(); });
}
