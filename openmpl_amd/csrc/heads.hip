// Output-side variants of the forward: everything after the FPT stack that is not the
// default "Conv1d weighted mean + LayerNorm + Linear" tail fused in fuse_head.hip.
//
// Reference (MPL/lib/models/multiview_mpl.py):
//   linear_weighted_mean   :277-279, :441-443   View_norm rows flattened to (B, V*E) -> Linear(V*E, E)
//   deep_head              :287-300, :517-519   LN -> 3 x (Linear -> BatchNorm1d(eval) -> ReLU) -> Linear
//   head_kadkhod           :301-317, :506-516   three cascaded MLPs, stages 2/3 fed cat([previous 3J outputs, x])
// These layers are tiny (M = B rows, K <= 1024+51): a plain LDS-tiled VALU GEMM with a fused
// bias / BatchNorm(eval) / ReLU epilogue is enough; the FPT GEMMs stay on the MFMA kernel.
#include "common.hpp"

namespace mpl {

// y[m][:] = LayerNorm(x[m][:]) : one wave per row, two-pass
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, int M, int K, int ldx,
                                                              const float* __restrict__ g, const float* __restrict__ be,
                                                              float eps, float* __restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int i = lane; i < K; i += 64) s += xr[i];
    s = wave_sum(s);
    const float mean = s / (float)K;
    float ss = 0.f;
    for (int i = lane; i < K; i += 64) {
        const float t = xr[i] - mean;
        ss += t * t;
    }
    ss = wave_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)K + eps);
    float* yr = y + (size_t)row * ldy;
    for (int i = lane; i < K; i += 64) yr[i] = (xr[i] - mean) * rstd * g[i] + be[i];
}

// strip ray features (:425-434) + View_norm (:439) per (b, v) row -> xn[(b*V+v)*E + f]
__global__ __launch_bounds__(256) void view_norm_kernel(const float* __restrict__ x, int rows, int Df, int E, int d,
                                                         int strip_mode, const float* __restrict__ g,
                                                         const float* __restrict__ be, float* __restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * Df;
    auto src = [&](int f) { return strip_mode == 1 ? (f / d) * 2 * d + (f % d) : f; };
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
    const float rstd = 1.0f / sqrtf(ss / (float)E + 1e-6f);
    float* yr = xn + (size_t)row * E;
    for (int f = lane; f < E; f += 64) yr[f] = (xr[src(f)] - mean) * rstd * g[f] + be[f];
}

// y[m][n] = act( bn( sum_k xa[m][k] W[n][k] + sum_k xb[m][k] W[n][Ka + k] + bias[n] ) ),  64 x 64 tile per workgroup,
// 4 x 4 outputs per thread, k in slabs of 16 through LDS.  bn (optional) = BatchNorm1d in eval mode.
constexpr int LT = 64, LK = 16;
__global__ __launch_bounds__(256) void linear_act_kernel(const float* __restrict__ xa, int Ka, int lda,
                                                          const float* __restrict__ xb, int Kb, int ldb, int M,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, int N,
                                                          const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                          const float* __restrict__ bn_mean,
                                                          const float* __restrict__ bn_var, float bn_eps, int relu,
                                                          float* __restrict__ y, int ldy) {
    __shared__ float Xs[LK][LT + 1];
    __shared__ float Ws[LK][LT + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * LT, n0 = blockIdx.x * LT;
    const int K = Ka + Kb;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += LK) {
        for (int i = tid; i < LT * LK; i += 256) {
            const int r = i / LK, kk = i % LK, k = k0 + kk;
            float xv = 0.f, wv = 0.f;
            if (k < K) {
                const int m = m0 + r, n = n0 + r;
                if (m < M) xv = (k < Ka) ? xa[(size_t)m * lda + k] : xb[(size_t)m * ldb + (k - Ka)];
                if (n < N) wv = W[(size_t)n * ldw + k];
            }
            Xs[kk][r] = xv;
            Ws[kk][r] = wv;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < LK; ++kk) {
            float xr[4], wr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xr[i] = Xs[kk][ty * 4 + i];
                wr[i] = Ws[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xr[i], wr[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + tx * 4 + j;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
        float sc = 1.f, sh = 0.f;
        if (bn_w) {  // (v - mean) / sqrt(var + eps) * w + b
            sc = bn_w[n] / sqrtf(bn_var[n] + bn_eps);
            sh = bn_b[n] - bn_mean[n] * sc;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ty * 4 + i;
            if (m >= M) continue;
            float v = (acc[i][j] + bv) * sc + sh;
            if (relu) v = fmaxf(v, 0.f);
            y[(size_t)m * ldy + n] = v;
        }
    }
}

int launch_layernorm_rows(const float* x, int M, int K, int ldx, const float* g, const float* b, float eps, float* y,
                          int ldy, hipStream_t s) {
    if (M <= 0 || K <= 0) return MPL_E_INVALID;
    ProfScope prof(MPL_K_FUSE_HEAD, s);
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, M, K, ldx, g, b, eps, y, ldy);
    return hip_check_launch();
}

int launch_view_norm(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* xn, hipStream_t s) {
    const int J = cfg->num_joints, d = cfg->dim, V = cfg->num_views, E = J * d;
    int strip = 0;
    if (cfg->flags & MPL_F_POS3D_TO_RAYS) strip = 1;
    else if (cfg->flags & MPL_F_RAYS_TOKEN) strip = 2;
    const int rows = batch * V;
    ProfScope prof(MPL_K_FUSE_HEAD, s);
    hipLaunchKernelGGL(view_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, mpl_fpt_width(cfg), E, d, strip,
                       w->view_norm_w, w->view_norm_b, xn);
    return hip_check_launch();
}

int launch_linear_act(const float* xa, int Ka, int lda, const float* xb, int Kb, int ldb, int M, const float* W, int ldw,
                      const float* bias, int N, const float* bn_w, const float* bn_b, const float* bn_mean,
                      const float* bn_var, float bn_eps, int relu, float* y, int ldy, hipStream_t s) {
    if (M <= 0 || N <= 0 || Ka <= 0 || Kb < 0 || !xa || !W || !y || (Kb > 0 && !xb)) return MPL_E_INVALID;
    if (bn_w && (!bn_b || !bn_mean || !bn_var)) return MPL_E_INVALID;
    ProfScope prof(MPL_K_FUSE_HEAD, s);
    hipLaunchKernelGGL(linear_act_kernel, dim3((N + LT - 1) / LT, (M + LT - 1) / LT), dim3(256), 0, s, xa, Ka, lda, xb, Kb,
                       ldb, M, W, ldw, bias, N, bn_w, bn_b, bn_mean, bn_var, bn_eps, relu, y, ldy);
    return hip_check_launch();
}

}  // namespace mpl
