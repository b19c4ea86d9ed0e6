// Tail of the forward: strip ray features, View_norm, learned weighted mean over views, head.
//
// Reference (MPL/lib/models/multiview_mpl.py):
//   :425-434  ray-token strip  -- token variant keeps [2][J][d] half 0, feature variant keeps [J][2d][:d]
//   :439      View_norm = LayerNorm(J*d, eps 1e-6) per (b, v) row
//   :445      weighted_mean = Conv1d(V -> 1, k = 1):  y[f] = sum_v w_v * xn[v][f] + bias
//   :521-523  head = LayerNorm(J*d, eps 1e-5) -> Linear(J*d, 3J) -> view(B, J, 3)
// Four poses per 256-thread workgroup (one wave each), the head Linear batched over the four.
#include "gemm_common.hpp"

namespace mpl {

constexpr int kMaxE = 1024;  // J*d upper bound held in LDS

// Four poses per 256-thread workgroup, one WAVE per pose for everything that is per pose (View_norm statistics, weighted
// mean over views, head LayerNorm: wave reductions only, no block barrier), then the head Linear for the four poses at
// once: a wave owns outputs o = wave, wave + 4, ... and reads each weight row ONCE (coalesced) for all four poses.
// Round 1 ran one workgroup per pose: a chain of ~10 dependent global-memory round trips and 51 x 4 re-reads of the
// 111 kB head weight per workgroup made it 45-54 us for 8.9 MB of input; this form is ~4 round trips.
constexpr int FH_POSES = 4;
constexpr int FH_W_FLOATS = 28 * 1024;        // LDS floats reserved for the head weight (3J * J*d = 27 744 at J = 17, d = 32)

__global__ __launch_bounds__(256) void fuse_head_kernel(const float* __restrict__ x, int B, int V, int Df, int E, int d,
                                                         int strip_mode,  // 0 none, 1 feature concat [J][2d], 2 token concat
                                                         const float* __restrict__ vn_w, const float* __restrict__ vn_b,
                                                         const float* __restrict__ wm_w, const float* __restrict__ wm_b,
                                                         const float* __restrict__ hl_w, const float* __restrict__ hl_b,
                                                         const float* __restrict__ hw, const float* __restrict__ hb,
                                                         int n_out, float* __restrict__ out,
                                                         float* __restrict__ y_out,
                                                         const unsigned* __restrict__ err_ws) {
    extern __shared__ __attribute__((aligned(1024))) float fh_smem[];
    float* hws = fh_smem;                                    // the head weight [n_out][E], staged by LDS-DMA
    float (*y)[kMaxE] = reinterpret_cast<float (*)[kMaxE]>(fh_smem + FH_W_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the block stack of this call reported a lost hand-off (x3_stack_kernel): its rows are not to be trusted -- the
    // result is NaN, never a plausible-looking pose (the host raises on the next call: MPL_E_DEVICE).  The word was
    // written (if at all) by the previous kernel of this stream; requested now, used at the very end.
    const unsigned poisoned = err_ws ? __hip_atomic_load(err_ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (!y_out) {
        // request the whole head weight (111 kB at E = 544, n_out = 51) now: it lands while the waves normalise and fuse
        // their poses; 1-KiB pieces, round-robin over the four waves (the last piece may read past the weight: the
        // launcher checks that the parameter tensor is followed by the bias -- it is not needed: clamp instead)
        const int n_w = n_out * E;                           // floats
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)hws;
        for (int pc = wave; pc * 256 < n_w; pc += 4) {
            int off = pc * 256 + lane * 4;
            off = off + 4 <= n_w ? off : n_w - 4;            // n_w is a multiple of 4: the clamped lanes re-read the tail
            dma16(hw + off, lds0 + (unsigned)(pc * 1024));
        }
    }
    const int b = blockIdx.x * FH_POSES + wave;
    auto src = [&](int f) { return strip_mode == 1 ? (f / d) * 2 * d + (f % d) : f; };
    if (b < B) {
        const float* xb = x + (size_t)b * V * Df;
        // weighted mean over views of the View_norm-ed rows (:439, :445): per view two-pass statistics (the row is
        // L1 / L2 resident after the first pass), accumulated into this lane's features f = lane, lane + 64, ...
        // Rows are held in registers (NF = 9 features per lane at E = 544 -- E <= 64 NF is checked by the launcher) and read
        // ONCE, four views at a time with all their loads in flight together: one memory round trip per four views
        // instead of three per view.
        constexpr int NF = 9, VC = 4;
        float acc[NF], gam[NF], bet[NF], hlw[NF], hlb[NF];
        // EVERY parameter this wave needs is requested up front, together with the first rows: one memory round trip for the
        // whole per-pose part (requested where they are used, the head LayerNorm vectors and the view weights each cost their
        // own ~2 us round trip behind the wave reductions: 14 us of this 28-us kernel at any batch size)
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = lane + 64 * i;
            acc[i] = 0.f;
            gam[i] = f < E ? vn_w[f] : 0.f;
            bet[i] = f < E ? vn_b[f] : 0.f;
            hlw[i] = (!y_out && f < E) ? hl_w[f] : 0.f;
            hlb[i] = (!y_out && f < E) ? hl_b[f] : 0.f;
        }
        const float wv_l = lane < V ? wm_w[lane] : 0.f;      // view weight v in lane v (V <= 32), broadcast by readlane below
        const float wb = wm_b[0];
        for (int v0 = 0; v0 < V; v0 += VC) {
            float xv[VC][NF];
#pragma unroll
            for (int u = 0; u < VC; ++u) {
                const float* xr = xb + (size_t)(v0 + u < V ? v0 + u : V - 1) * Df;
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const int f = lane + 64 * i;
                    xv[u][i] = f < E ? xr[src(f)] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < VC; ++u) {
                if (v0 + u >= V) break;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NF; ++i) s += xv[u][i];
                const float mean = wave_sum(s) / (float)E;
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const float t = (lane + 64 * i < E) ? xv[u][i] - mean : 0.f;
                    ss += t * t;
                }
                const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)E + 1e-6f);
                const float wv = __shfl(wv_l, v0 + u, 64);
#pragma unroll
                for (int i = 0; i < NF; ++i) acc[i] = fmaf(wv, (xv[u][i] - mean) * rstd * gam[i] + bet[i], acc[i]);
            }
        }
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = lane + 64 * i;
            if (f < E) {
                acc[i] += wb;
                part += acc[i];
            }
        }
        if (y_out) {  // caller wants the fused (B, E) feature only (non-default heads): stop before head[0]
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = lane + 64 * i;
                if (f < E) y_out[(size_t)b * E + f] = poisoned ? __builtin_nanf("") : acc[i];
            }
        } else {
            // head LayerNorm (eps 1e-5), two-pass in registers
            const float mean = wave_sum(part) / (float)E;
            float p2 = 0.f;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = lane + 64 * i;
                if (f < E) {
                    const float t = acc[i] - mean;
                    p2 += t * t;
                }
            }
            const float rstd = 1.0f / sqrtf(wave_sum(p2) / (float)E + 1e-5f);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = lane + 64 * i;
                if (f < E) y[wave][f] = (acc[i] - mean) * rstd * hlw[i] + hlb[i];
            }
        }
    }
    if (y_out) return;
    __syncthreads();
    // Linear(E -> 3J) for the four poses out of LDS: thread = (pose, output), a complete dot product without any cross-lane
    // reduction, read as float4.  Thread o walks its weight row rotated by 4 o floats (f = (i + 4 o) mod E): the 16 lanes a
    // ds_read_b128 serves together then sit on 16 different 16-byte bank groups (E = 544: row stride 544 + 4 = 548 = 36 mod 64
    // dwords, and 36 o mod 64 is distinct for 16 consecutive o).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int np = (B - blockIdx.x * FH_POSES) < FH_POSES ? (B - blockIdx.x * FH_POSES) : FH_POSES;
    for (int t = tid; t < np * n_out; t += 256) {
        const int p = t / n_out, o = t - p * n_out;
        const float* wr = hws + o * E;
        const float* yr = y[p];
        float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
        int f = (4 * o) % E;
        if ((E & 3) == 0) {
#pragma unroll 4
            for (int i = 0; i < E; i += 4) {
                const float4 a4 = ld4(yr + f), w4 = ld4(wr + f);
                sx = fmaf(a4.x, w4.x, sx);
                sy = fmaf(a4.y, w4.y, sy);
                sz = fmaf(a4.z, w4.z, sz);
                sw = fmaf(a4.w, w4.w, sw);
                f = f + 4 == E ? 0 : f + 4;
            }
        } else {
            for (int i = 0; i < E; ++i) {
                sx = fmaf(yr[f], wr[f], sx);
                f = f + 1 == E ? 0 : f + 1;
            }
        }
        out[(size_t)(blockIdx.x * FH_POSES + p) * n_out + o] = poisoned ? __builtin_nanf("") : ((sx + sy) + (sz + sw)) + hb[o];
    }
}

int launch_fuse_head(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* out, float* y_out,
                     const unsigned* err_ws, hipStream_t s) {
    const int J = cfg->num_joints, d = cfg->dim, V = cfg->num_views;
    const int E = J * d;
    if (E > kMaxE || E > 64 * 9 || V > MPL_MAX_VIEWS || batch <= 0) return MPL_E_UNSUPPORTED;   // 9 features per lane (J*d = 544)
    int strip = 0;
    if (cfg->flags & MPL_F_POS3D_TO_RAYS) strip = 1;           // :430-434 (takes precedence, elif order)
    else if (cfg->flags & MPL_F_RAYS_TOKEN) strip = 2;         // :425-429
    const int Df = mpl_fpt_width(cfg);
    if (3 * J * E > FH_W_FLOATS || (E & 1)) return MPL_E_UNSUPPORTED;
    constexpr int LDS = (FH_W_FLOATS + FH_POSES * kMaxE) * 4;
    static std::atomic<bool> attr_set[64];   // set-once flags: a racing second hipFuncSetAttribute is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)fuse_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    ProfScope prof(MPL_K_FUSE_HEAD, s);
    hipLaunchKernelGGL(fuse_head_kernel, dim3((batch + FH_POSES - 1) / FH_POSES), dim3(256), LDS, s, x, batch, V, Df, E, d, strip, w->view_norm_w,
                       w->view_norm_b, w->wmean_w, w->wmean_b, w->head_ln_w, w->head_ln_b, w->head_w, w->head_b, 3 * J,
                       out, y_out, err_ws);
    return hip_check_launch();
}

}  // namespace mpl
