// Tail of the forward: strip ray features, View_norm, learned weighted mean over views, head.
//
// Reference (MPL/lib/models/multiview_mpl.py):
//   :425-434  ray-token strip  -- token variant keeps [2][J][d] half 0, feature variant keeps [J][2d][:d]
//   :439      View_norm = LayerNorm(J*d, eps 1e-6) per (b, v) row
//   :445      weighted_mean = Conv1d(V -> 1, k = 1):  y[f] = sum_v w_v * xn[v][f] + bias
//   :521-523  head = LayerNorm(J*d, eps 1e-5) -> Linear(J*d, 3J) -> view(B, J, 3)
// Four poses per 256-thread workgroup (one wave each), the head Linear batched over the four.
#include "fuse_head.hpp"

namespace mpl {

// FH_POSES poses (= waves) per workgroup, one WAVE per pose for everything that is per pose (View_norm statistics, weighted
// mean over views, head LayerNorm: wave reductions only, no block barrier: fuse_head.hpp fh_pose), then the head Linear for all
// of them at once (fh_linear).  Round 1 ran one workgroup per pose: a chain of ~10 dependent global-memory round trips and 51 x 4
// re-reads of the 111 kB head weight per workgroup made it 45-54 us for 8.9 MB of input; this form is ~4 round trips.
// FH_POSES = 4 (256 threads).  Eight poses per 512-thread workgroup (half the workgroups, half the copies of the 111-kB head weight
// through L2) measured 20.3 against 20.9 us at B = 1024 (profiles/r06_ab_fh8.txt): the kernel is a chain of memory round trips, not
// bytes -- not kept.
template <int FH_POSES>
__global__ __launch_bounds__(64 * FH_POSES) void fuse_head_kernel(const float* __restrict__ x, int B, const FhParams p, float* __restrict__ out,
                                                                  float* __restrict__ y_out, const unsigned* __restrict__ err_ws) {
    extern __shared__ __attribute__((aligned(1024))) float fh_smem[];
    float* hws = fh_smem;                                    // the head weight [n_out][E], staged by LDS-DMA
    float (*y)[kMaxE] = reinterpret_cast<float (*)[kMaxE]>(fh_smem + FH_W_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the block stack of this call reported a lost hand-off: its rows are not to be trusted -- the result is NaN, never a
    // plausible-looking pose (the host raises on the next call: MPL_E_DEVICE).  The word was written (if at all) by the
    // previous kernel of this stream; requested now, used at the very end.
    const unsigned poisoned = err_ws ? __hip_atomic_load(err_ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (!y_out) fh_stage_weight(p, hws, wave, lane, FH_POSES);       // lands while the waves normalise and fuse their poses
    const int b = blockIdx.x * FH_POSES + wave;
    if (b < B)
        fh_pose<false>(p, x + (size_t)b * p.V * p.Df, lane, y[wave], y_out ? y_out + (size_t)b * p.E : nullptr, poisoned != 0u);
    if (y_out) return;
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int np = (B - blockIdx.x * FH_POSES) < FH_POSES ? (B - blockIdx.x * FH_POSES) : FH_POSES;
    const int b0 = blockIdx.x * FH_POSES;
    fh_linear(p, hws, y, np, tid, 64 * FH_POSES, out, poisoned != 0u, [&](int ps) { return b0 + ps; });
}

int launch_fuse_head(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* out, float* y_out,
                     const unsigned* err_ws, hipStream_t s) {
    FhParams p;
    if (batch <= 0 || !fh_params(cfg, w, &p)) return MPL_E_UNSUPPORTED;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    constexpr int poses = 4;
    constexpr int LDS = (FH_W_FLOATS + poses * kMaxE) * 4;
    void (*kernel)(const float*, int, const FhParams, float*, float*, const unsigned*) = fuse_head_kernel<poses>;
    static std::atomic<bool> attr_set[64];   // set-once flags: a racing second hipFuncSetAttribute is harmless
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    ProfScope prof(MPL_K_FUSE_HEAD, s);
    hipLaunchKernelGGL(kernel, dim3((batch + poses - 1) / poses), dim3(64 * poses), LDS, s, x, batch, p, out, y_out, err_ws);
    return hip_check_launch();
}

}  // namespace mpl
