// C ABI of libmpl_hip.so (include/mpl_hip.h): argument validation + launch orchestration.
// One mpl_forward call enqueues the whole forward on the caller's stream; nothing synchronises.
#include "common.hpp"

#include <stdlib.h>

#include <mutex>
#include <vector>

using namespace mpl;

// ------------------------------------------------------------------ profiling aid
// The only process-global mutable state of the library besides the set-once function-attribute flags: a list of event
// pairs, guarded by a mutex (launches may come from one thread per GPU under DataParallel).  Off unless
// mpl_profile_start() was called; the fast path is one relaxed atomic load.
namespace {
struct ProfRec {
    hipEvent_t e0, e1;
    int kind;
};
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
}  // namespace

mpl::ProfScope::ProfScope(int kind, hipStream_t s) : slot(-1), stream(s) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    ProfRec r;
    r.kind = kind;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    hipEventRecord(r.e0, s);
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}
mpl::ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> g(g_prof_mu);
    if (slot < (int)g_prof.size()) hipEventRecord(g_prof[slot].e1, stream);
}

// ------------------------------------------------------------------ device-side failure word (common.hpp)
namespace {
std::mutex g_err_mu;
unsigned* g_err_host[64];      // pinned host words, one per device, allocated on first use, never freed
unsigned* g_err_dev[64];       // the same words as the device addresses them
}  // namespace

unsigned* mpl::device_error_word(int dev) {
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_err_mu);
    if (!g_err_host[dev]) {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) return nullptr;
        *reinterpret_cast<volatile unsigned*>(h) = 0u;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) d = h;
        g_err_host[dev] = reinterpret_cast<unsigned*>(h);
        g_err_dev[dev] = reinterpret_cast<unsigned*>(d);
    }
    return g_err_dev[dev];
}
int mpl::device_error_pending(int dev) {
    if (dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> g(g_err_mu);
    return g_err_host[dev] ? (int)*reinterpret_cast<volatile unsigned*>(g_err_host[dev]) : 0;
}
void mpl::device_error_clear(int dev) {
    if (dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> g(g_err_mu);
    if (g_err_host[dev]) *reinterpret_cast<volatile unsigned*>(g_err_host[dev]) = 0u;
}

namespace {
std::atomic<int> g_fault_phase{0};
}  // namespace
void mpl::set_fault_injection(int phase) { g_fault_phase.store(phase); }
int mpl::take_fault_injection() { return g_fault_phase.exchange(0); }
int mpl::refuse_stream_capture(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return MPL_OK;                       // the legacy default stream cannot be queried while another stream captures
    }
    return st == hipStreamCaptureStatusNone ? MPL_OK : MPL_E_UNSUPPORTED;
}

namespace {
std::mutex g_chain_mu[64];
hipEvent_t g_chain_ev[64];
std::mutex g_chain_create_mu;
}  // namespace
std::mutex& mpl::stack_chain_mutex(int dev) { return g_chain_mu[(dev >= 0 && dev < 64) ? dev : 0]; }
hipEvent_t mpl::stack_chain_event(int dev) {
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_chain_create_mu);
    if (!g_chain_ev[dev] && hipEventCreateWithFlags(&g_chain_ev[dev], hipEventDisableTiming) != hipSuccess) g_chain_ev[dev] = nullptr;
    return g_chain_ev[dev];
}

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// a failure a kernel of an earlier call reported on the current device fails every later call until it is cleared
inline int earlier_device_failure() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MPL_E_LAUNCH;
    return device_error_pending(dev) ? MPL_E_DEVICE : MPL_OK;
}

struct StackWs {
    float *qkv, *att, *hid, *stats;
    size_t bytes;
};

StackWs carve_stack_ws(void* base, size_t M, size_t D) {
    StackWs w;
    size_t off = 0;
    auto take = [&](size_t n_floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += align_up(n_floats * sizeof(float), 256);
        return p;
    };
    w.qkv = take(M * 3 * D);
    w.att = take(M * D);
    w.hid = take(M * 2 * D);
    w.stats = take(M * 2 * (size_t)ln_stat_slices((int)D));
    w.bytes = off;
    return w;
}

std::atomic<bool> g_x3_per_gemm{getenv("MPL_X3_LAUNCHES") != nullptr};
std::atomic<int> g_spin_log2{23};    // polls before a wait inside a persistent kernel counts as lost (mpl_x3_spin_limit)
std::atomic<int> g_x3_stop{0};   // diagnostics: stop a stack after this many GEMMs (0 = run everything)

// 2 = every block carries fp16x2 operands (h2_gemm.hip, the default fp32 engine), 1 = packed bf16 operands (b1_gemm.hip), 0 = none
// of them (or the shape has no packed layout): the stack then runs on the fp32 matrix instructions
int stack_packed_parts(const mpl_block_weights* blocks, const uint8_t* schedule, int n_apps, int n_tok, int D, int H) {
    if (n_apps <= 0 || !h2_attention_fusable(n_tok, D, H) || !h2_shape_ok(D, 2 * D)) return 0;
    int np = 0;
    for (int a = 0; a < n_apps; ++a) {
        const mpl_block_weights& b = blocks[schedule[a]];
        const int bp = (b.qkv_w16 && b.proj_w16 && b.fc1_w16 && b.fc2_w16) ? 1 : ((b.qkv_h2 && b.proj_h2 && b.fc1_h2 && b.fc2_h2) ? 2 : 0);
        if (bp == 0 || (np && bp != np)) return 0;
        np = bp;
    }
    return np;
}

// fp16x2 path (h2_gemm.hip): x stays fp32 in place and is the A operand of the LayerNorm GEMMs; the attention output and
// the GELU output travel as packed operands
struct H2Ws {
    unsigned short *att2, *hid2;
    float* stats;
    unsigned* counters;
    size_t bytes;
};
H2Ws carve_h2_ws(void* base, size_t M, size_t D, int rpt) {
    H2Ws w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? reinterpret_cast<char*>(base) + off : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    w.att2 = reinterpret_cast<unsigned short*>(take(h2_act_bytes((int)M, (int)D, rpt)));
    w.hid2 = reinterpret_cast<unsigned short*>(take(h2_act_bytes((int)M, (int)(2 * D), rpt)));
    w.stats = reinterpret_cast<float*>(take((M + 64) * 2 * (size_t)ln_stat_slices((int)D) * sizeof(float)));
    w.counters = reinterpret_cast<unsigned*>(take((h2_err_index((int)((M + rpt - 1) / rpt)) + 64) * sizeof(unsigned)));
    w.bytes = off;
    return w;
}
int block_stack_h2(float* x, int n_seq, int n_tok, int D, int H, const mpl_block_weights* blocks, const uint8_t* schedule,
                   int n_apps, void* ws, size_t ws_bytes, const unsigned** err_ws, hipStream_t s);

// bf16 path (b1_gemm.hip): x stays fp32 in place (residual stream, statistics) and travels to the LayerNorm GEMMs as the packed
// bf16 copy x16 that the residual epilogues rewrite; the attention output and the GELU output travel as packed bf16 operands
struct B1Ws {
    unsigned short *x16, *att1, *hid1;
    float* stats;
    unsigned* counters;
    size_t bytes;
};
B1Ws carve_b1_ws(void* base, size_t M, size_t D, int rpt) {
    B1Ws w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? reinterpret_cast<char*>(base) + off : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    w.x16 = reinterpret_cast<unsigned short*>(take(h2_act_bytes((int)M, (int)D, rpt, 1)));
    w.att1 = reinterpret_cast<unsigned short*>(take(h2_act_bytes((int)M, (int)D, rpt, 1)));
    w.hid1 = reinterpret_cast<unsigned short*>(take(h2_act_bytes((int)M, (int)(2 * D), rpt, 1)));
    w.stats = reinterpret_cast<float*>(take((M + 64) * 2 * (size_t)ln_stat_slices((int)D) * sizeof(float)));
    w.counters = reinterpret_cast<unsigned*>(take((h2_err_index((int)((M + rpt - 1) / rpt)) + 64) * sizeof(unsigned)));
    w.bytes = off;
    return w;
}
int block_stack_b1(float* x, int n_seq, int n_tok, int D, int H, const mpl_block_weights* blocks, const uint8_t* schedule,
                   int n_apps, void* ws, size_t ws_bytes, const unsigned** err_ws, hipStream_t s) {
    const int M = n_seq * n_tok, rpt = h2_rows_per_tile(n_tok);
    const float eps = 1e-6f;  // norm_layer = partial(nn.LayerNorm, eps=1e-6), multiview_mpl.py:139
    if (n_apps > MPL_MAX_APPS) return MPL_E_UNSUPPORTED;
    const B1Ws w = carve_b1_ws(ws, (size_t)M, (size_t)D, rpt);
    if (!ws || ws_bytes < w.bytes) return MPL_E_WORKSPACE;
    const int n_tiles = (M + rpt - 1) / rpt;
    if (err_ws) *err_ws = w.counters + h2_err_index(n_tiles);
    int rc;
    const unsigned short* ops[MPL_MAX_APPS * 4];
    for (int a = 0; a < n_apps; ++a) {
        const mpl_block_weights& b = blocks[schedule[a]];
        for (int i = 0; i < 4; ++i) ops[4 * a + i] = (&b.qkv_w16)[i];
    }
    // entry of the stack, one launch: the rows as packed bf16 operand, their LayerNorm slice partials, zeroed counters + error word
    if ((rc = launch_b1_entry(x, M, D, D, rpt, w.x16, w.stats, w.counters, h2_err_index(n_tiles) + 1, s))) return rc;
    if (!g_x3_per_gemm.load(std::memory_order_relaxed))
        return launch_b1_stack(x, w.x16, M, D, n_tok, H, ops, n_apps, w.att1, w.hid1, w.stats, w.counters, eps, g_x3_stop.load(), s);
    // A/B switch (mpl_x3_stack_mode): the same phases as one launch per GEMM
    const int stop = g_x3_stop.load();
    for (int a = 0; a < n_apps; ++a) {
        const mpl_block_weights& b = blocks[schedule[a]];
        if ((rc = launch_b1_qkv_attention(w.x16, b.qkv_w16, w.stats, eps, M, D, n_tok, H, w.att1, s))) return rc;
        if (stop && 4 * a + 1 >= stop) return MPL_OK;
        if ((rc = launch_b1_gemm(w.att1, b.proj_w16, false, nullptr, 0.f, x, D, x, D, w.x16, w.stats, M, D, D, rpt, MPL_EPI_BIAS_RESIDUAL, s)))
            return rc;
        if (stop && 4 * a + 2 >= stop) return MPL_OK;
        if ((rc = launch_b1_gemm(w.x16, b.fc1_w16, true, w.stats, eps, nullptr, 0, nullptr, 0, w.hid1, nullptr, M, 2 * D, D, rpt,
                                 MPL_EPI_BIAS_GELU, s)))
            return rc;
        if (stop && 4 * a + 3 >= stop) return MPL_OK;
        if ((rc = launch_b1_gemm(w.hid1, b.fc2_w16, false, nullptr, 0.f, x, D, x, D, w.x16, w.stats, M, D, 2 * D, rpt,
                                 MPL_EPI_BIAS_RESIDUAL, s)))
            return rc;
    }
    return MPL_OK;
}

int block_stack_h2(float* x, int n_seq, int n_tok, int D, int H, const mpl_block_weights* blocks, const uint8_t* schedule,
                   int n_apps, void* ws, size_t ws_bytes, const unsigned** err_ws, hipStream_t s) {
    const int M = n_seq * n_tok, rpt = h2_rows_per_tile(n_tok);
    const float eps = 1e-6f;  // norm_layer = partial(nn.LayerNorm, eps=1e-6), multiview_mpl.py:139
    if (n_apps > MPL_MAX_APPS) return MPL_E_UNSUPPORTED;
    const H2Ws w = carve_h2_ws(ws, (size_t)M, (size_t)D, rpt);
    if (!ws || ws_bytes < w.bytes) return MPL_E_WORKSPACE;
    const int n_tiles = (M + rpt - 1) / rpt;
    if (err_ws) *err_ws = w.counters + h2_err_index(n_tiles);
    int rc;
    const unsigned short* ops[MPL_MAX_APPS * 4];
    for (int a = 0; a < n_apps; ++a) {
        const mpl_block_weights& b = blocks[schedule[a]];
        for (int i = 0; i < 4; ++i) ops[4 * a + i] = (&b.qkv_h2)[i];
    }
    // entry of the stack, one launch: LayerNorm slice partials of the incoming rows, zeroed arrival counters + error word, and
    // the check that proj / fc2 were packed against the static scales of their producers (mpl_pack_h2_scaled)
    if ((rc = launch_h2_entry(x, M, D, D, w.stats, w.counters, h2_err_index(n_tiles) + 1, ops, n_apps, s))) return rc;
    if (!g_x3_per_gemm.load(std::memory_order_relaxed))
        return launch_h2_stack(x, M, D, n_tok, H, ops, n_apps, w.att2, w.hid2, w.stats, w.counters, eps, g_x3_stop.load(), s);
    // A/B switch (mpl_x3_stack_mode): the same phases as one launch per GEMM
    const int stop = g_x3_stop.load();
    for (int a = 0; a < n_apps; ++a) {
        const mpl_block_weights& b = blocks[schedule[a]];
        if ((rc = launch_h2_qkv_attention(x, b.qkv_h2, w.stats, eps, M, D, n_tok, H, w.att2, s))) return rc;
        if (stop && 4 * a + 1 >= stop) return MPL_OK;
        if ((rc = launch_h2_gemm(nullptr, w.att2, nullptr, b.proj_h2, false, nullptr, 0.f, x, D, x, D, nullptr, w.stats, M, D, D,
                                 rpt, MPL_EPI_BIAS_RESIDUAL, s)))
            return rc;
        if (stop && 4 * a + 2 >= stop) return MPL_OK;
        if ((rc = launch_h2_gemm(x, nullptr, nullptr, b.fc1_h2, true, w.stats, eps, nullptr, 0, nullptr, 0, w.hid2, nullptr, M, 2 * D,
                                 D, rpt, MPL_EPI_BIAS_GELU, s)))
            return rc;
        if (stop && 4 * a + 3 >= stop) return MPL_OK;
        if ((rc = launch_h2_gemm(nullptr, w.hid2, nullptr, b.fc2_h2, false, nullptr, 0.f, x, D, x, D, nullptr, w.stats, M, D,
                                 2 * D, rpt, MPL_EPI_BIAS_RESIDUAL, s)))
            return rc;
    }
    return MPL_OK;
}

thread_local int t_last_form = MPL_E_INVALID;  // MPL_FORM_* of this thread's most recent block-stack launch (mpl_block_stack_last_form)

inline int device_cu_count(int* cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || *cus < 1)
        return MPL_E_LAUNCH;
    return MPL_OK;
}

// THE rule for "the small-batch engine (sm_stack.hip) takes this stack": block_stack_impl launches by it, mpl_block_stack_form
// reports by it.  np = packed operand parts of the blocks (0 none, 2 fp16x2, 1 bf16: an explicit bf16 request keeps its engine);
// raw = every nn.Linear / LayerNorm tensor of every scheduled block is present (the engine reads them in place); cus = compute
// units of the device (every column tile of the widest GEMM needs a resident workgroup of its own: they poll each other's output).
inline bool small_engine_taken(int np, bool allow_small, int M, int D, int n_tok, int H, int n_apps, int n_blocks, bool raw, int cus) {
    return (np == 0 || np == 2) && allow_small && sm_stack_enabled() && n_apps <= MPL_MAX_APPS && raw &&
           !g_x3_per_gemm.load(std::memory_order_relaxed) && g_x3_stop.load() == 0 && sm_stack_ok(M, D, n_tok, H, n_apps, n_blocks, cus);
}

int block_stack_impl(float* x, int n_seq, int n_tok, int D, int H, const mpl_block_weights* blocks,
                     const uint8_t* schedule, int n_apps, void* ws, size_t ws_bytes, const unsigned** err_ws, bool allow_small,
                     hipStream_t s) {
    if (err_ws) *err_ws = nullptr;
    if (!x || n_seq <= 0 || n_tok <= 0 || D <= 0 || H <= 0 || n_apps < 0) return MPL_E_INVALID;
    // row counts are 32-bit in the kernels (byte offsets are 64-bit): refuse what would overflow instead of wrapping
    if ((long long)n_seq * n_tok > (1ll << 30)) return MPL_E_UNSUPPORTED;
    if (n_apps == 0) return MPL_OK;
    if (!blocks || !schedule) return MPL_E_INVALID;
    const int np0 = stack_packed_parts(blocks, schedule, n_apps, n_tok, D, H);
    // up to 80 token rows (a single frame, a few frames): the whole chip on every GEMM instead of one team of D / 136
    // workgroups (sm_stack.hip) -- for the fp32 engines; an explicit bf16 request keeps its engine
    // (not when the caller asked for batch-invariant bits -- MPL_F_NO_SMALL_STACK --, nor under the A/B switches of the team
    // kernels: one launch per GEMM, stop after n phases).  ONE predicate decides (small_engine_taken): this function launches
    // by it and mpl_block_stack_form reports by it.
    if (n_apps <= MPL_MAX_APPS && (long long)n_seq * n_tok <= sm_stack_max_rows()) {
        int n_blocks = 0;
        bool raw = true;        // the engine reads the nn.Linear tensors in place: a caller that hands over packed operands only
        for (int a = 0; a < n_apps; ++a) {      // (the C ABI allows it) gets the team kernels, not an error
            n_blocks = schedule[a] + 1 > n_blocks ? schedule[a] + 1 : n_blocks;
            const mpl_block_weights& b = blocks[schedule[a]];
            raw = raw && b.ln1_w && b.ln1_b && b.qkv_w && b.qkv_b && b.proj_w && b.proj_b && b.ln2_w && b.ln2_b && b.fc1_w && b.fc1_b &&
                  b.fc2_w && b.fc2_b;
        }
        int cus = 0;
        if (device_cu_count(&cus) != MPL_OK) return MPL_E_LAUNCH;
        if (small_engine_taken(np0, allow_small, n_seq * n_tok, D, n_tok, H, n_apps, n_blocks, raw, cus)) {
            const int rc = launch_sm_stack(x, n_seq, n_tok, D, H, blocks, schedule, n_apps, ws, ws_bytes, err_ws, g_spin_log2.load(), s);
            if (rc != MPL_E_UNSUPPORTED) {               // (UNSUPPORTED: the occupancy query refused one workgroup per CU -- the team kernels below)
                if (rc == MPL_OK) t_last_form = MPL_FORM_SMALL;
                return rc;
            }
            if (err_ws) *err_ws = nullptr;
        }
    }
    if (const int np = np0) {
        const int rc = np == 2 ? block_stack_h2(x, n_seq, n_tok, D, H, blocks, schedule, n_apps, ws, ws_bytes, err_ws, s)
                               : block_stack_b1(x, n_seq, n_tok, D, H, blocks, schedule, n_apps, ws, ws_bytes, err_ws, s);
        if (rc == MPL_OK) {
            int cus = 0;
            t_last_form = g_x3_per_gemm.load(std::memory_order_relaxed) ? (int)MPL_FORM_PER_GEMM
                          : (device_cu_count(&cus) == MPL_OK ? h2_stack_form_code(n_seq * n_tok, D, n_tok, np, cus) : (int)MPL_E_LAUNCH);
        }
        return rc;
    }
    t_last_form = MPL_FORM_UNPACKED;
    const int M = n_seq * n_tok;
    const StackWs w = carve_stack_ws(ws, (size_t)M, (size_t)D);
    if (!ws || ws_bytes < w.bytes) return MPL_E_WORKSPACE;
    const float eps = 1e-6f;  // norm_layer = partial(nn.LayerNorm, eps=1e-6), multiview_mpl.py:139
    int rc;
    // LayerNorm statistics are produced by whoever writes x: a stand-alone pass for the incoming x, then the
    // epilogues of proj (-> norm2) and fc2 (-> norm1 of the next application).
    float* st_out = (D % 136 == 0) ? w.stats : nullptr;
    bool have_stats = false;
    // qkv projection and attention run as one kernel when a 64-row tile holds whole sequences and a 136-column
    // slice whole heads (V in {1,2,4,8,16,32,64}; hd in {68,136}); otherwise as two kernels through `qkv`
    const bool fusable = qkv_attention_fusable(n_tok, D, H);
    // D = 32 blocks (keypoint-token FPT) that carry the operand of mpl_d32_pack in qkv_w3: everything but the attention is
    // row-local and runs as two launches from split fp16 operands (spt.hip: d32_qkv_kernel, d32_mlp_kernel)
    bool d32 = D == 32;
    for (int a = 0; a < n_apps && d32; ++a) {
        const mpl_block_weights& b = blocks[schedule[a]];
        d32 = b.qkv_w3 && !b.proj_w3 && !b.fc1_w3 && !b.fc2_w3 && !b.qkv_w16 && !b.qkv_h2;
    }
    if (d32) {
        for (int a = 0; a < n_apps; ++a) {
            const mpl_block_weights& b = blocks[schedule[a]];
            if ((rc = launch_d32_qkv(x, M, b.qkv_w3, w.qkv, s))) return rc;
            if ((rc = launch_token_attention(w.qkv, n_seq, n_tok, D, H, w.att, s))) return rc;
            if ((rc = launch_d32_mlp(x, w.att, M, b.qkv_w3, s))) return rc;
        }
        return MPL_OK;
    }
    for (int a = 0; a < n_apps; ++a) {
        const mpl_block_weights& b = blocks[schedule[a]];
            // packed operands (fp16x2 / bf16) take the routes above; here they cannot be used
        if (b.qkv_w3 || b.proj_w3 || b.fc1_w3 || b.fc2_w3 || b.qkv_w16 || b.proj_w16 || b.fc1_w16 || b.fc2_w16 || b.qkv_h2 ||
            b.proj_h2 || b.fc1_h2 || b.fc2_h2)
            return MPL_E_UNSUPPORTED;
        const bool fused_att = fusable;
        // x = x + proj(attn(qkv(norm1(x))))   (Block.forward :84-90)
        if (!have_stats && (rc = launch_row_stats(x, M, D, D, w.stats, s))) return rc;
        if (fused_att) {
            rc = launch_ln_qkv_attention(x, M, D, w.stats, b.ln1_w, b.ln1_b, eps, b.qkv_w, b.qkv_b, n_tok, H, w.att, s);
            if (rc) return rc;
        } else {
            rc = launch_ln_gemm(x, D, w.stats, b.ln1_w, b.ln1_b, eps, b.qkv_w, b.qkv_b, nullptr, 0, w.qkv, 3 * D, M,
                                3 * D, D, MPL_EPI_BIAS, nullptr, s);
            if (rc) return rc;
            if ((rc = launch_token_attention(w.qkv, n_seq, n_tok, D, H, w.att, s))) return rc;
        }
        rc = launch_ln_gemm(w.att, D, nullptr, nullptr, nullptr, 0.f, b.proj_w, b.proj_b, x, D, x, D, M, D, D,
                            MPL_EPI_BIAS_RESIDUAL, st_out, s);
        if (rc) return rc;
        // x = x + fc2(gelu(fc1(norm2(x))))    (Block.forward :91, Mlp.forward :31-37)
        if (!st_out && (rc = launch_row_stats(x, M, D, D, w.stats, s))) return rc;
        rc = launch_ln_gemm(x, D, w.stats, b.ln2_w, b.ln2_b, eps, b.fc1_w, b.fc1_b, nullptr, 0, w.hid, 2 * D, M,
                            2 * D, D, MPL_EPI_BIAS_GELU, nullptr, s);
        if (rc) return rc;
        rc = launch_ln_gemm(w.hid, 2 * D, nullptr, nullptr, nullptr, 0.f, b.fc2_w, b.fc2_b, x, D, x, D, M, D,
                            2 * D, MPL_EPI_BIAS_RESIDUAL, st_out, s);
        if (rc) return rc;
        have_stats = st_out != nullptr;
    }
    return MPL_OK;
}

// hipGetLastError() is per-thread state shared with the caller: a benign failure inside the caller's own HIP use (e.g.
// torch probing a host pointer) would otherwise be reported by the first launch check of this library.
inline void clear_stale_hip_error() { (void)hipGetLastError(); }

// either engine may run the stack (the binding decides by the operands it supplies): size for the larger layout
size_t stack_ws_bytes(size_t M, size_t D, int n_tok) {
    size_t b = carve_stack_ws(nullptr, M, D).bytes;
    if (M <= (size_t)sm_stack_max_rows()) {
        const size_t bs = sm_stack_ws_bytes((int)M, (int)D);
        b = bs > b ? bs : b;
    }
    const int rpt = h2_rows_per_tile(n_tok);
    if (rpt > 0 && n_tok <= 32 && h2_shape_ok((int)D, (int)(2 * D))) {
        const size_t b2 = carve_h2_ws(nullptr, M, D, rpt).bytes;
        b = b2 > b ? b2 : b;
        const size_t b1 = carve_b1_ws(nullptr, M, D, rpt).bytes;
        b = b1 > b ? b1 : b;
    }
    return b;
}

int check_cfg(const mpl_config* cfg) {
    if (!cfg) return MPL_E_INVALID;
    if (cfg->num_views < 1 || cfg->num_views > MPL_MAX_VIEWS || cfg->depth < 0 || cfg->depth > 60) return MPL_E_INVALID;
    return MPL_OK;
}

}  // namespace

extern "C" {

int mpl_profile_start(void) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof.clear();
    g_prof_on.store(true);
    return MPL_OK;
}

int mpl_profile_stop(float* kind_ms, int* kind_launches, int n_kinds) {
    g_prof_on.store(false);
    std::lock_guard<std::mutex> g(g_prof_mu);
    if (!kind_ms || !kind_launches || n_kinds < MPL_K_COUNT) return MPL_E_INVALID;
    for (int k = 0; k < n_kinds; ++k) {
        kind_ms[k] = 0.f;
        kind_launches[k] = 0;
    }
    int rc = MPL_OK;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) rc = MPL_E_LAUNCH;
        kind_ms[r.kind] += ms;
        kind_launches[r.kind] += 1;
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    g_prof.clear();
    return rc;
}

int mpl_hip_abi_version(void) { return MPL_HIP_ABI_VERSION; }

const char* mpl_hip_error_string(int code) {
    switch (code) {
        case MPL_OK: return "ok";
        case MPL_E_INVALID: return "invalid argument or shape";
        case MPL_E_UNSUPPORTED: return "configuration not supported by the HIP path";
        case MPL_E_WORKSPACE: return "workspace missing or too small";
        case MPL_E_LAUNCH: return "HIP runtime error at kernel launch";
        case MPL_E_DEVICE:
            return "an earlier forward on this device failed on the device and its poses are NaN (mpl_device_error() bits: 1 = a "
                   "workgroup of the persistent block-stack kernel lost a hand-off -- the GPU was shared with other work for "
                   "longer than the wait bound -- or proj / fc2 operands were not packed against their producers' scales "
                   "(mpl_pack_h2_scaled skipped); 2 = confidence_as_attention_uncertainty_weight with confidences so large that "
                   "the weighted attention output left the fp16 window of the split operands: run that model with "
                   "set_matmul_precision('fp32_mfma'), the native-fp32 engine has no window); clear with mpl_device_error_clear()";
        default: return "unknown error";
    }
}

int mpl_fpt_width(const mpl_config* cfg) {
    if (!cfg) return MPL_E_INVALID;
    return cfg->num_joints * cfg->dim * ((cfg->flags & MPL_F_RAYS_TOKEN) ? 2 : 1);
}

size_t mpl_block_stack_workspace_bytes(int n_seq, int n_tok, int dim) {
    return stack_ws_bytes((size_t)n_seq * n_tok, (size_t)dim, n_tok);
}

size_t mpl_forward_workspace_bytes(const mpl_config* cfg, int batch) {
    if (!cfg || batch <= 0) return 0;
    const size_t M = (size_t)batch * cfg->num_views, D = (size_t)mpl_fpt_width(cfg);
    // joints x views token grid (:496-497): the same xs memory seen as (B, V*J, d)
    const bool kp = (cfg->flags & MPL_F_KPTOK) != 0;
    const size_t Ms = kp ? M * cfg->num_joints : M, Ds = kp ? (size_t)cfg->dim : D;
    return align_up(M * D * sizeof(float), 256) + stack_ws_bytes(Ms, Ds, kp ? cfg->num_views * cfg->num_joints : cfg->num_views);
}

int mpl_spt_tokens(const mpl_config* cfg, const mpl_weights* w, const mpl_inputs* in, float* xs, void* stream) {
    clear_stale_hip_error();
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!w || !in || !xs) return MPL_E_INVALID;
    return launch_spt(cfg, w, in, xs, w->spt_packed != 0, (hipStream_t)stream);
}

int mpl_block_stack_ex(float* x, int n_seq, int n_tok, int dim, int heads, const mpl_block_weights* blocks,
                       const uint8_t* schedule, int n_apps, void* workspace, size_t workspace_bytes, unsigned flags, void* stream) {
    clear_stale_hip_error();
    if (int rc = earlier_device_failure()) return rc;
    return block_stack_impl(x, n_seq, n_tok, dim, heads, blocks, schedule, n_apps, workspace, workspace_bytes, nullptr,
                            (flags & MPL_F_NO_SMALL_STACK) == 0, (hipStream_t)stream);
}

int mpl_block_stack(float* x, int n_seq, int n_tok, int dim, int heads, const mpl_block_weights* blocks,
                    const uint8_t* schedule, int n_apps, void* workspace, size_t workspace_bytes, void* stream) {
    return mpl_block_stack_ex(x, n_seq, n_tok, dim, heads, blocks, schedule, n_apps, workspace, workspace_bytes, 0u, stream);
}

int mpl_ln_linear(const float* x, int M, int K, const float* ln_w, const float* ln_b, float eps, const float* W,
                  const float* bias, int N, int epilogue, const float* residual, float* y, float* stats,
                  void* stream) {
    clear_stale_hip_error();
    if (!x || !W || !bias || !y) return MPL_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (ln_w) {
        if (!stats) return MPL_E_INVALID;
        int rc = launch_row_stats(x, M, K, K, stats, s);
        if (rc) return rc;
    }
    // bench-only: with MPL_GEMM_ABL=4 (tools/gemm_phase.py) the scratch pointer receives per-wave phase timings
    static const bool timing = lab_getenv("MPL_GEMM_ABL") && atoi(lab_getenv("MPL_GEMM_ABL")) == 4;
    return launch_ln_gemm(x, K, stats, ln_w, ln_b, eps, W, bias, residual, N, y, N, M, N, K, epilogue,
                          (timing && !ln_w) ? stats : nullptr, s);
}

size_t mpl_spt_pack_bytes(void) { return spt_pack_bytes(); }

int mpl_spt_pack(const mpl_block_weights* block, uint16_t* dst, void* stream) {
    clear_stale_hip_error();
    return launch_spt_pack(block, dst, 1, (hipStream_t)stream);
}

int mpl_d32_pack(const mpl_block_weights* block, uint16_t* dst, void* stream) {
    clear_stale_hip_error();
    return launch_spt_pack(block, dst, 0, (hipStream_t)stream);
}

size_t mpl_pack_bf16_bytes(int N, int K) { return h2_operand_bytes(N, K, 1); }

int mpl_pack_bf16(const float* W, const float* bias, const float* ln_w, const float* ln_b, int N, int K, uint16_t* dst, void* stream) {
    clear_stale_hip_error();
    if (mpl_pack_bf16_bytes(N, K) == 0) return MPL_E_INVALID;
    return launch_pack_b1(W, N, K, ln_w, ln_b, bias, dst, (hipStream_t)stream);
}

size_t mpl_pack_h2_bytes(int N, int K) { return h2_operand_bytes(N, K); }

int mpl_pack_h2(const float* W, const float* bias, const float* ln_w, const float* ln_b, int N, int K, uint16_t* dst, void* stream) {
    clear_stale_hip_error();
    if (mpl_pack_h2_bytes(N, K) == 0) return MPL_E_INVALID;
    return launch_pack_h2(W, N, K, ln_w, ln_b, bias, nullptr, dst, (hipStream_t)stream);
}

int mpl_pack_h2_scaled(const float* W, const float* bias, const float* in_scale, int N, int K, uint16_t* dst, void* stream) {
    clear_stale_hip_error();
    if (mpl_pack_h2_bytes(N, K) == 0 || !in_scale) return MPL_E_INVALID;
    return launch_pack_h2(W, N, K, nullptr, nullptr, bias, in_scale, dst, (hipStream_t)stream);
}

const float* mpl_pack_h2_out_scale(const uint16_t* operand, int N, int K) { return h2_out_scale(operand, N, K); }

size_t mpl_ln_linear_h2_workspace_bytes(int M, int K) {
    const size_t a = h2_act_bytes(M, K, 64);
    return a ? a + 256 : 0;
}

int mpl_ln_linear_h2(const float* x, int M, int K, int has_ln, float eps, const uint16_t* W2, int N, int epilogue,
                     const float* residual, float* y, float* stats, void* workspace, size_t workspace_bytes, void* stream) {
    clear_stale_hip_error();
    if (!x || !W2 || !y || h2_operand_bytes(N, K) == 0 || M <= 0) return MPL_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (has_ln) {
        if (!stats) return MPL_E_INVALID;
        if ((rc = launch_row_stats(x, M, K, K, stats, s))) return rc;
        return launch_h2_gemm(x, nullptr, nullptr, W2, true, stats, eps, residual, N, y, N, nullptr, nullptr, M, N, K, 64, epilogue, s);
    }
    const size_t need = mpl_ln_linear_h2_workspace_bytes(M, K);
    if (!workspace || workspace_bytes < need) return MPL_E_WORKSPACE;
    float* sc = reinterpret_cast<float*>(workspace);
    unsigned short* a2 = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(workspace) + 256);
    if ((rc = launch_h2_pack_rows(x, M, K, K, 64, a2, sc, s))) return rc;
    return launch_h2_gemm(nullptr, a2, sc + 2, W2, false, nullptr, 0.f, residual, N, y, N, nullptr, nullptr, M, N, K, 64, epilogue, s);
}

int mpl_block_stack_form_ex(int n_seq, int n_tok, int D, int heads, int n_apps, int n_blocks, int raw_tensors, int operand_parts,
                            unsigned flags) {
    if (n_seq <= 0 || n_tok <= 0 || D <= 0 || heads <= 0 || n_apps <= 0 || n_apps > MPL_MAX_APPS || n_blocks <= 0 || n_blocks > n_apps ||
        operand_parts < 0 || operand_parts > 2)
        return MPL_E_INVALID;
    if ((long long)n_seq * n_tok > (1ll << 30)) return MPL_E_UNSUPPORTED;
    const int M = n_seq * n_tok;
    // the same questions, in the same order, as block_stack_impl asks -- through the same predicates
    const int np = (h2_attention_fusable(n_tok, D, heads) && h2_shape_ok(D, 2 * D)) ? operand_parts : 0;
    int cus = 0;
    if (device_cu_count(&cus) != MPL_OK) return MPL_E_LAUNCH;
    if (M <= sm_stack_max_rows() && small_engine_taken(np, !(flags & MPL_F_NO_SMALL_STACK), M, D, n_tok, heads, n_apps, n_blocks, raw_tensors != 0, cus))
        return MPL_FORM_SMALL;
    if (np == 0) return MPL_FORM_UNPACKED;
    if (g_x3_per_gemm.load(std::memory_order_relaxed)) return MPL_FORM_PER_GEMM;
    return h2_stack_form_code(M, D, n_tok, np, cus);
}

// the reference's schedule (every block once, the last one twice: multiview_mpl.py:420-423) with the nn.Linear tensors present
int mpl_block_stack_form(int n_seq, int n_tok, int D, int heads, int n_apps, int operand_parts, unsigned flags) {
    return mpl_block_stack_form_ex(n_seq, n_tok, D, heads, n_apps, n_apps > 1 ? n_apps - 1 : 1, 1, operand_parts, flags);
}

int mpl_block_stack_last_form(void) { return t_last_form; }

int mpl_x3_stack_mode(int one_launch_per_gemm) {
    g_x3_per_gemm.store((one_launch_per_gemm & 1) != 0);
    g_x3_stop.store(one_launch_per_gemm >> 8);
    h2_set_write_through((one_launch_per_gemm >> 7) & 1);  // bit 7: write-through hand-off stores also for teams that sit on one XCD
    h2_set_direct_w(((one_launch_per_gemm >> 4) & 1) ^ 1);  // bit 4: the 16-row teams in the ring form (A/B against the direct-W form)
    h2_set_narrow((one_launch_per_gemm >> 5) & 3);         // bits 5, 6: row-narrow teams: 0 = by shape, 1 = never, 2 / 3 = 32- / 16-row workgroups where legal
    h2_set_row_tiles((one_launch_per_gemm >> 1) & 3);      // bits 1, 2: 0 = by shape, 1 / 2 = force the one- / two-tile stage
    sm_stack_disable((one_launch_per_gemm >> 3) & 1);      // bit 3: no small-batch engine (the team kernels for every batch)
    return MPL_OK;
}

int mpl_device_error(int device) {
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return 0;
    return device_error_pending(device);
}

int mpl_device_error_clear(int device) {
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return MPL_E_LAUNCH;
    device_error_clear(device);
    return MPL_OK;
}

int mpl_x3_spin_limit(int log2_polls) {
    if ((log2_polls & 0xff) < 1 || (log2_polls & 0xff) > 30 || log2_polls < 0) return MPL_E_INVALID;
    // the fault injection (bits 8..) is compiled into the product library but inert unless the process opted in
    // (MPL_FAULT_INJECT=1 in the environment at the FIRST call of this function), and it is ONE-SHOT: the first stack launch that
    // consumes it clears it, so a test that dies between set and reset cannot leave the process deserting workgroups
    static const bool inject_ok = getenv("MPL_FAULT_INJECT") != nullptr && atoi(getenv("MPL_FAULT_INJECT")) != 0;
    if ((log2_polls >> 8) != 0 && !inject_ok) return MPL_E_UNSUPPORTED;
    g_spin_log2.store(log2_polls & 0xff);
    h2_set_spin_log2(log2_polls & 0xff);
    set_fault_injection(log2_polls >> 8);
    return MPL_OK;
}

int mpl_x3_debug_buffer(void* device_buffer) {
    h2_set_debug_buffer(reinterpret_cast<unsigned long long*>(device_buffer));
    return MPL_OK;
}

int mpl_token_attention(const float* qkv, int n_seq, int n_tok, int dim, int heads, float* out, void* stream) {
    clear_stale_hip_error();
    if (!qkv || !out) return MPL_E_INVALID;
    return launch_token_attention(qkv, n_seq, n_tok, dim, heads, out, (hipStream_t)stream);
}

int mpl_fuse_head(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* out, void* stream) {
    clear_stale_hip_error();
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!w || !x || !out) return MPL_E_INVALID;
    return launch_fuse_head(cfg, w, x, batch, out, nullptr, nullptr, (hipStream_t)stream);
}

int mpl_view_fuse(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* y, void* stream) {
    clear_stale_hip_error();
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!w || !x || !y) return MPL_E_INVALID;
    return launch_fuse_head(cfg, w, x, batch, nullptr, y, nullptr, (hipStream_t)stream);
}

int mpl_view_norm(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* xn, void* stream) {
    clear_stale_hip_error();
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!w || !x || !xn || batch <= 0) return MPL_E_INVALID;
    return launch_view_norm(cfg, w, x, batch, xn, (hipStream_t)stream);
}

int mpl_layernorm(const float* x, int M, int K, const float* gamma, const float* beta, float eps, float* y,
                  void* stream) {
    clear_stale_hip_error();
    if (!x || !gamma || !beta || !y) return MPL_E_INVALID;
    return launch_layernorm_rows(x, M, K, K, gamma, beta, eps, y, K, (hipStream_t)stream);
}

int mpl_linear(const float* xa, int Ka, const float* xb, int Kb, int M, const float* W, const float* bias, int N,
               const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var, float bn_eps, int relu,
               float* y, void* stream) {
    clear_stale_hip_error();
    return launch_linear_act(xa, Ka, Ka, xb, Kb, Kb, M, W, Ka + Kb, bias, N, bn_w, bn_b, bn_mean, bn_var, bn_eps, relu, y, N,
                             (hipStream_t)stream);
}

int mpl_prepare_inputs(const float* joints_px, const float* conf, const double* cams_dev, int batch, int views, int joints,
                       float img_w, float img_h, int normalize_inputs, int normalize_cameras, float* const* poses,
                       float* const* rays, float* const* centers, void* stream) {
    clear_stale_hip_error();
    return launch_prepare_inputs(joints_px, conf, cams_dev, batch, views, joints, img_w, img_h, normalize_inputs,
                                 normalize_cameras, poses, rays, centers, (hipStream_t)stream);
}

int mpl_pose_metrics_size(int joints) { return 4 + 2 * (joints + 1) + 3 * joints + 3; }

int mpl_pose_metrics(const float* output, const float* target, const float* weight, int batch, int joints,
                     const float* scale3, const float* offset3, float* result, void* stream) {
    clear_stale_hip_error();
    if (int rc = earlier_device_failure()) return rc;
    return launch_pose_metrics(output, target, weight, batch, joints, scale3, offset3, 0u, result, (hipStream_t)stream);
}

int mpl_pose_metrics_ex(const float* output, const float* target, const float* weight, int batch, int joints,
                        const float* scale3, const float* offset3, uint32_t not_consider_mask, float* result, void* stream) {
    clear_stale_hip_error();
    if (int rc = earlier_device_failure()) return rc;
    return launch_pose_metrics(output, target, weight, batch, joints, scale3, offset3, not_consider_mask, result, (hipStream_t)stream);
}

int mpl_forward(const mpl_config* cfg, const mpl_weights* w, const mpl_inputs* in, float* out, void* workspace,
                size_t workspace_bytes, void* stream) {
    clear_stale_hip_error();
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if ((rc = earlier_device_failure())) return rc;
    if (!w || !in || !out || in->batch <= 0) return MPL_E_INVALID;
    if ((long long)in->batch * cfg->num_views * cfg->num_joints > (1ll << 30)) return MPL_E_UNSUPPORTED;
    const size_t need = mpl_forward_workspace_bytes(cfg, in->batch);
    if (!workspace || workspace_bytes < need) return MPL_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int B = in->batch, V = cfg->num_views, D = mpl_fpt_width(cfg);
    float* xs = reinterpret_cast<float*>(workspace);
    char* rest = reinterpret_cast<char*>(workspace) + align_up((size_t)B * V * D * sizeof(float), 256);
    const size_t rest_bytes = workspace_bytes - (size_t)(rest - reinterpret_cast<char*>(workspace));

    if ((rc = launch_spt(cfg, w, in, xs, w->spt_packed != 0, s))) return rc;

    const unsigned* err_ws = nullptr;     // set by the block stack when it runs as the persistent launch
    if (!(cfg->flags & MPL_F_NO_FPT) && cfg->depth > 0) {
        // forward_features :420-423: the last block is applied twice
        uint8_t sched[MPL_MAX_APPS];
        int n = 0;
        for (int l = 0; l < cfg->depth; ++l) {
            if (n + 2 > MPL_MAX_APPS) return MPL_E_UNSUPPORTED;
            if (l == cfg->depth - 1) sched[n++] = (uint8_t)l;
            sched[n++] = (uint8_t)l;
        }
        const bool kp = (cfg->flags & MPL_F_KPTOK) != 0;
        if ((rc = block_stack_impl(xs, B, kp ? V * cfg->num_joints : V, kp ? cfg->dim : D, cfg->heads, w->fpt_blocks,
                                   sched, n, rest, rest_bytes, &err_ws, (cfg->flags & MPL_F_NO_SMALL_STACK) == 0, s)))
            return rc;
    }
    return launch_fuse_head(cfg, w, xs, B, out, nullptr, err_ws, s);
}

}  // extern "C"
