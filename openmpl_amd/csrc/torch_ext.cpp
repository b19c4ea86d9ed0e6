// openmpl_amd's torch extension (TORCH_LIBRARY): the per-call host side of MultiView_MPL.forward as C++ operators over the C ABI of
// libmpl_hip.so (include/mpl_hip.h).  north_star: "hand-written HIP ... exposed as a torch extension"; SURVEY.md 8b suggests
// TORCH_LIBRARY ops; reference caller: MPL/lib/core/function_mpl.py:347-350 (`output = model(input, centers=, rays=)`, once per batch).
//
// Why it exists (round 6): the ctypes route costs ~90 us of Python per forward -- a ~300-element data_ptr() tuple to notice moved
// parameters, ~40 ctypes arguments, two torch.empty calls -- 16 % of a single-frame call.  Here the same work is: one dispatcher
// call, 300 pointer / version compares in C++, two caching-allocator calls, ONE C-ABI call (mpl_forward) on the current HIP stream.
//
//   openmpl_amd::bind(...)   -> int   registers the struct of addresses Python marshalled (mpl_config, mpl_weights, the HOST array of
//                                     FPT blocks) together with the parameter tensors they point into.  The tensors are the
//                                     module's own TensorImpls: `p.data = other`, `p.set_(...)`, optimizer steps and copy_() all
//                                     show up as a changed data_ptr() / _version() and make lift() answer "stale".
//   openmpl_amd::lift(h, poses, rays, centers, flags) -> Tensor   validates the V view tensors exactly as the Python route does
//                                     (RuntimeError, same messages), allocates workspace + output through torch's caching
//                                     allocator on the inputs' device, calls mpl_forward on c10::hip::getCurrentHIPStream.
//                                     An UNDEFINED-size answer (0-d tensor of -1) = the binding is stale: Python re-marshals.
//   openmpl_amd::unbind(h)            drops a binding (the parameter tensors and derived operands it kept alive).
//
// The entry points of libmpl_hip.so are handed over as addresses (openmpl_amd::set_entry_points, from the ctypes binding): this file
// links against torch only, the boundary to the kernels stays the C ABI, and a laboratory build of the library (tools/ab.sh) is
// picked up without rebuilding the extension.  PyTorch is plumbing here: device memory, streams, the dispatcher.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/mpl_hip.h"

namespace {

typedef int (*forward_fn)(const mpl_config*, const mpl_weights*, const mpl_inputs*, float*, void*, size_t, void*);
typedef size_t (*ws_bytes_fn)(const mpl_config*, int);
typedef const char* (*errstr_fn)(int);

struct EntryPoints {
    forward_fn forward = nullptr;
    ws_bytes_fn ws_bytes = nullptr;
    errstr_fn errstr = nullptr;
    int abi = 0;
} g_ep;

struct Binding {
    mpl_config cfg;
    mpl_weights w;
    std::vector<mpl_block_weights> fpt;        // HOST array mpl_weights.fpt_blocks points at (owned here)
    std::vector<at::Tensor> params;            // every parameter / buffer whose address is inside the structs
    std::vector<const void*> ptrs;             // their addresses when the structs were built
    std::vector<uint32_t> versions;            // their versions (derived operands fold the VALUES of the block tensors)
    std::vector<uint8_t> versioned;            // 1: a changed version makes the binding stale (tensors folded into packed operands)
    std::vector<at::Tensor> keep;              // struct blob on the device + derived (packed) operands
    int device = 0;
    bool needs_rays = false;
    hipEvent_t ready = nullptr;                // recorded behind the packing / blob copy on the stream that marshalled
    hipStream_t ready_stream = nullptr;
    std::mutex ws_mu;
    std::unordered_map<int, size_t> ws_cache;  // batch -> workspace bytes
    ~Binding() {
        if (ready) (void)hipEventDestroy(ready);
    }
};

std::mutex g_mu;
std::unordered_map<int64_t, std::shared_ptr<Binding>> g_bindings;
int64_t g_next = 1;

std::shared_ptr<Binding> lookup(int64_t h) {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_bindings.find(h);
    TORCH_CHECK(it != g_bindings.end(), "openmpl_amd::lift: binding ", h, " is not alive");
    return it->second;
}

void set_entry_points(int64_t forward_addr, int64_t ws_bytes_addr, int64_t errstr_addr, int64_t abi) {
    TORCH_CHECK(abi == MPL_HIP_ABI_VERSION, "openmpl_amd torch extension was built against C ABI ", MPL_HIP_ABI_VERSION,
                ", libmpl_hip.so reports ", abi, " (rebuild: python -m openmpl_amd.build --force)");
    TORCH_CHECK(forward_addr && ws_bytes_addr && errstr_addr, "openmpl_amd::set_entry_points: null entry point");
    std::lock_guard<std::mutex> g(g_mu);
    g_ep.forward = reinterpret_cast<forward_fn>(forward_addr);
    g_ep.ws_bytes = reinterpret_cast<ws_bytes_fn>(ws_bytes_addr);
    g_ep.errstr = reinterpret_cast<errstr_fn>(errstr_addr);
    g_ep.abi = (int)abi;
}

template <typename T>
T from_bytes(const at::Tensor& t, const char* what) {
    TORCH_CHECK(t.device().is_cpu() && t.scalar_type() == at::kByte && t.is_contiguous() && (size_t)t.numel() == sizeof(T),
                "openmpl_amd::bind: ", what, " must be ", sizeof(T), " bytes of a CPU uint8 tensor (got ", t.numel(), ")");
    T v;
    std::memcpy(&v, t.data_ptr(), sizeof(T));
    return v;
}

// cfg / weights / fpt_blocks: the ctypes structs of cabi.py as raw bytes (CPU uint8).  params: the tensors the structs point into;
// versioned[i] != 0: tensor i is folded into a derived operand.  keep: tensors that must outlive the binding (device blob, packs).
int64_t bind(const at::Tensor& cfg_b, const at::Tensor& w_b, const at::Tensor& fpt_b, at::TensorList params, at::IntArrayRef versioned,
             at::TensorList keep, int64_t device, bool needs_rays) {
    TORCH_CHECK(g_ep.forward, "openmpl_amd::bind before set_entry_points");
    auto b = std::make_shared<Binding>();
    b->cfg = from_bytes<mpl_config>(cfg_b, "cfg");
    b->w = from_bytes<mpl_weights>(w_b, "weights");
    TORCH_CHECK(fpt_b.device().is_cpu() && fpt_b.scalar_type() == at::kByte && fpt_b.is_contiguous() &&
                    fpt_b.numel() % (int64_t)sizeof(mpl_block_weights) == 0,
                "openmpl_amd::bind: fpt_blocks must be whole mpl_block_weights structs");
    b->fpt.resize((size_t)fpt_b.numel() / sizeof(mpl_block_weights));
    if (!b->fpt.empty()) std::memcpy(b->fpt.data(), fpt_b.data_ptr(), (size_t)fpt_b.numel());
    b->w.fpt_blocks = b->fpt.empty() ? nullptr : b->fpt.data();
    TORCH_CHECK(versioned.size() == params.size(), "openmpl_amd::bind: one `versioned` flag per parameter");
    b->params.assign(params.begin(), params.end());
    for (size_t i = 0; i < b->params.size(); ++i) {
        const at::Tensor& t = b->params[i];
        TORCH_CHECK(t.is_cuda() && t.device().index() == device && t.scalar_type() == at::kFloat && t.is_contiguous(),
                    "MultiView_MPL (HIP): every parameter must be a contiguous float32 tensor on cuda:", device);
        b->ptrs.push_back(t.data_ptr());
        b->versions.push_back(t.is_inference() ? 0u : (uint32_t)t._version());      // inference tensors track no version
        b->versioned.push_back(versioned[i] ? 1 : 0);
    }
    b->keep.assign(keep.begin(), keep.end());
    b->device = (int)device;
    b->needs_rays = needs_rays;
    {
        c10::hip::HIPGuard guard((c10::DeviceIndex)device);
        b->ready_stream = c10::hip::getCurrentHIPStream((c10::DeviceIndex)device).stream();
        TORCH_CHECK(hipEventCreateWithFlags(&b->ready, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
        TORCH_CHECK(hipEventRecord(b->ready, b->ready_stream) == hipSuccess, "hipEventRecord failed");
    }
    std::lock_guard<std::mutex> g(g_mu);
    const int64_t h = g_next++;
    g_bindings[h] = std::move(b);
    return h;
}

void unbind(int64_t h) {
    std::shared_ptr<Binding> dead;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_bindings.find(h);
        if (it == g_bindings.end()) return;
        dead = std::move(it->second);
        g_bindings.erase(it);
    }
}

int64_t live_bindings() {
    std::lock_guard<std::mutex> g(g_mu);
    return (int64_t)g_bindings.size();
}

at::Tensor stale_answer() { return at::full({}, -1, at::TensorOptions().dtype(at::kInt)); }

at::Tensor lift(int64_t h, at::TensorList poses, at::TensorList rays, at::TensorList centers, int64_t flags) {
    std::shared_ptr<Binding> b = lookup(h);
    // ---- is the struct of addresses still what the module holds?  (moved storage: any tensor; changed values: folded tensors)
    for (size_t i = 0; i < b->params.size(); ++i) {
        const at::Tensor& t = b->params[i];
        if (t.data_ptr() != b->ptrs[i] || (b->versioned[i] && !t.is_inference() && (uint32_t)t._version() != b->versions[i]))
            return stale_answer();
    }
    // ---- MultiView_MPL._check_inputs (same conditions, same messages: RuntimeError)
    const int V = b->cfg.num_views, J = b->cfg.num_joints;
    TORCH_CHECK((int)poses.size() == V, "expected ", V, " views (num_views is a constructor constant), got ", poses.size());
    TORCH_CHECK(poses[0].dim() == 3, "pose tensor has shape ", poses[0].sizes(), ", expected (B, ", J, ", 3)");
    const int64_t B = poses[0].size(0);
    TORCH_CHECK(B < (int64_t)1 << 31, "batch too large");
    mpl_inputs in;
    std::memset(&in, 0, sizeof(in));
    in.batch = (int32_t)B;
    std::vector<at::Tensor> hold;              // contiguous copies made here must outlive the launch enqueue
    auto prep = [&](at::TensorList lst, const float** dst, int64_t d1, const char* name, bool required) {
        if (lst.empty()) {
            TORCH_CHECK(!required, name, "s are required by this flag set");
            return;
        }
        TORCH_CHECK((int)lst.size() == V, "expected ", V, " ", name, " tensors, got ", lst.size());
        for (int v = 0; v < V; ++v) {
            const at::Tensor& t = lst[v];
            TORCH_CHECK(t.is_cuda() && t.device().index() == b->device, name, " tensor on ", t.device(), " but the model is on cuda:",
                        b->device);
            TORCH_CHECK(t.dim() == 3 && t.size(0) == B && t.size(1) == d1 && t.size(2) == 3, name, " tensor has shape ", t.sizes(),
                        ", expected (", B, ", ", d1, ", 3)");
            TORCH_CHECK(t.scalar_type() == at::kFloat, name, " tensor must be float32 (got ", t.scalar_type(), ")");
            if (t.is_contiguous()) {
                dst[v] = t.data_ptr<float>();
            } else {
                hold.push_back(t.contiguous());
                dst[v] = hold.back().data_ptr<float>();
            }
        }
    };
    prep(poses, in.poses, J, "pose", true);
    prep(rays, in.rays, J, "ray", b->needs_rays);
    prep(centers, in.centers, 1, "center", b->needs_rays);

    c10::hip::HIPGuard guard((c10::DeviceIndex)b->device);
    const hipStream_t stream = c10::hip::getCurrentHIPStream((c10::DeviceIndex)b->device).stream();
    if (stream != b->ready_stream)             // the blob copy / packing kernels were enqueued on another stream
        TORCH_CHECK(hipStreamWaitEvent(stream, b->ready, 0) == hipSuccess, "hipStreamWaitEvent failed");
    mpl_config cfg = b->cfg;
    cfg.flags = (cfg.flags & ~MPL_F_NO_SMALL_STACK) | ((uint32_t)flags & MPL_F_NO_SMALL_STACK);
    const auto opts = at::TensorOptions().device(at::kCUDA, (c10::DeviceIndex)b->device);
    at::Tensor out = at::empty({B, (int64_t)J, 3}, opts.dtype(at::kFloat));
    if (B == 0) return out;
    size_t ws_bytes;
    {
        std::lock_guard<std::mutex> g(b->ws_mu);
        auto it = b->ws_cache.find((int)B);
        if (it == b->ws_cache.end()) it = b->ws_cache.emplace((int)B, g_ep.ws_bytes(&cfg, (int)B)).first;
        ws_bytes = it->second;
    }
    // allocated and consumed on the current stream: the caching allocator's stream-ordered reuse keeps it alive for the kernels
    at::Tensor ws = at::empty({(int64_t)ws_bytes}, opts.dtype(at::kByte));
    const int rc = g_ep.forward(&cfg, &b->w, &in, out.data_ptr<float>(), ws.data_ptr(), ws_bytes, stream);
    TORCH_CHECK(rc == MPL_OK, "mpl_forward failed: ", g_ep.errstr(rc), " (code ", rc, ")");
    return out;
}

}  // namespace

// FRAGMENT: the Python side defines openmpl_amd::forward (the general operator, every flag set) in the same namespace
TORCH_LIBRARY_FRAGMENT(openmpl_amd, m) {
    m.def("set_entry_points(int forward, int ws_bytes, int errstr, int abi) -> ()", &set_entry_points);
    m.def("bind(Tensor cfg, Tensor weights, Tensor fpt_blocks, Tensor[] params, int[] versioned, Tensor[] keep, int device, bool needs_rays) -> int", &bind);
    m.def("unbind(int handle) -> ()", &unbind);
    m.def("live_bindings() -> int", &live_bindings);
    m.def("lift(int handle, Tensor[] poses, Tensor[] rays, Tensor[] centers, int flags) -> Tensor");
}

TORCH_LIBRARY_IMPL(openmpl_amd, CUDA, m) { m.impl("lift", &lift); }
