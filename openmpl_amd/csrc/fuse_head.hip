// Tail of the forward: strip ray features, View_norm, learned weighted mean over views, head.
//
// Reference (MPL/lib/models/multiview_mpl.py):
//   :425-434  ray-token strip  -- token variant keeps [2][J][d] half 0, feature variant keeps [J][2d][:d]
//   :439      View_norm = LayerNorm(J*d, eps 1e-6) per (b, v) row
//   :445      weighted_mean = Conv1d(V -> 1, k = 1):  y[f] = sum_v w_v * xn[v][f] + bias
//   :521-523  head = LayerNorm(J*d, eps 1e-5) -> Linear(J*d, 3J) -> view(B, J, 3)
// One 256-thread workgroup per pose; everything after the V row reads lives in LDS/registers.
#include "common.hpp"

namespace mpl {

constexpr int kMaxE = 1024;  // J*d upper bound held in LDS

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void fuse_head_kernel(const float* __restrict__ x, int V, int Df, int E, int d,
                                                         int strip_mode,  // 0 none, 1 feature concat [J][2d], 2 token concat
                                                         const float* __restrict__ vn_w, const float* __restrict__ vn_b,
                                                         const float* __restrict__ wm_w, const float* __restrict__ wm_b,
                                                         const float* __restrict__ hl_w, const float* __restrict__ hl_b,
                                                         const float* __restrict__ hw, const float* __restrict__ hb,
                                                         int n_out, float* __restrict__ out,
                                                         float* __restrict__ y_out) {
    __shared__ float y[kMaxE];
    __shared__ float vstat[MPL_MAX_VIEWS][2];
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (size_t)b * V * Df;
    auto src = [&](int f) { return strip_mode == 1 ? (f / d) * 2 * d + (f % d) : f; };

    // View_norm statistics: one wave per view row, two-pass
    for (int v = wave; v < V; v += 4) {
        const float* xr = xb + (size_t)v * Df;
        float s = 0.f;
        for (int f = lane; f < E; f += 64) s += xr[src(f)];
        s = wave_sum(s);
        const float mean = s / (float)E;
        float ss = 0.f;
        for (int f = lane; f < E; f += 64) {
            const float t = xr[src(f)] - mean;
            ss += t * t;
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            vstat[v][0] = mean;
            vstat[v][1] = 1.0f / sqrtf(ss / (float)E + 1e-6f);
        }
    }
    __syncthreads();
    // weighted mean over views of the normalised rows
    const float wb = wm_b[0];
    float part = 0.f;
    for (int f = tid; f < E; f += 256) {
        const float g = vn_w[f], be = vn_b[f];
        const int sf = src(f);
        float acc = 0.f;
        for (int v = 0; v < V; ++v) {
            const float xn = (xb[(size_t)v * Df + sf] - vstat[v][0]) * vstat[v][1] * g + be;
            acc = fmaf(wm_w[v], xn, acc);
        }
        acc += wb;
        y[f] = acc;
        part += acc;
    }
    if (y_out) {  // caller wants the fused (B, E) feature only (non-default heads): stop before head[0]
        __syncthreads();
        for (int f = tid; f < E; f += 256) y_out[(size_t)b * E + f] = y[f];
        return;
    }
    // head LayerNorm (eps 1e-5), two-pass over LDS
    const float mean = block_sum(part, red) / (float)E;
    float p2 = 0.f;
    for (int f = tid; f < E; f += 256) {
        const float t = y[f] - mean;
        p2 += t * t;
    }
    const float rstd = 1.0f / sqrtf(block_sum(p2, red) / (float)E + 1e-5f);
    for (int f = tid; f < E; f += 256) y[f] = (y[f] - mean) * rstd * hl_w[f] + hl_b[f];
    __syncthreads();
    // Linear(E -> 3J): one wave per output, lane-strided dot product
    for (int o = wave; o < n_out; o += 4) {
        const float* wr = hw + (size_t)o * E;
        float s = 0.f;
        for (int f = lane; f < E; f += 64) s = fmaf(y[f], wr[f], s);
        s = wave_sum(s);
        if (lane == 0) out[(size_t)b * n_out + o] = s + hb[o];
    }
}

int launch_fuse_head(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* out, float* y_out,
                     hipStream_t s) {
    const int J = cfg->num_joints, d = cfg->dim, V = cfg->num_views;
    const int E = J * d;
    if (E > kMaxE || V > MPL_MAX_VIEWS || batch <= 0) return MPL_E_UNSUPPORTED;
    int strip = 0;
    if (cfg->flags & MPL_F_POS3D_TO_RAYS) strip = 1;           // :430-434 (takes precedence, elif order)
    else if (cfg->flags & MPL_F_RAYS_TOKEN) strip = 2;         // :425-429
    const int Df = mpl_fpt_width(cfg);
    ProfScope prof(MPL_K_FUSE_HEAD, s);
    hipLaunchKernelGGL(fuse_head_kernel, dim3(batch), dim3(256), 0, s, x, V, Df, E, d, strip, w->view_norm_w,
                       w->view_norm_b, w->wmean_w, w->wmean_b, w->head_ln_w, w->head_ln_b, w->head_w, w->head_b, 3 * J,
                       out, y_out);
    return hip_check_launch();
}

}  // namespace mpl
