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

// y[m][n] = act( bn( sum_k xa[m][k] W[n][k] + sum_k xb[m][k] W[n][Ka + k] + bias[n] ) ): the Linear (+ BatchNorm1d in eval
// mode + ReLU) layers of the deep / kadkhod heads and of linear_weighted_mean (multiview_mpl.py:287-317, :441-443, :506-519),
// whose concat inputs cat([prev_out, x]) are read from their two sources without being materialised.
//
// Exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain per output): at the reference's default
// TRANSFORMER_OUTPUT_HEAD_HIDDEN_DIM = 1024 (config.py:98) the heads are 5.5 (deep) / 16.6 (kadkhod) GFLOP per 1024 poses, which
// the 64 x 64 VALU tile of round 3 ran at a few TFLOP/s.  Workgroup = 4 waves = 64 rows x 64
// columns (wave = 32 x 32: 2 x 2 MFMA tiles), k in slabs of 32 through LDS: any K, N, leading dimension or alignment (the
// kadkhod stages read K = 51 + 544 with rows that are not 16-byte aligned), zero-filled tails.  The slab of the NEXT step is
// requested into registers before the MFMAs of the current one (one workgroup per CU at N = 1024: nothing else hides the loads).
constexpr int LT = 64, LK = 32, LKP = 36;        // LKP: LDS row stride (floats): 16-byte aligned rows, 144 B apart
// VEC: every row of xa, xb and W starts 16-byte aligned and the concat boundary Ka is a multiple of 4, so a thread's four
// consecutive k come with one 16-byte load; otherwise element by element (the 51 + 544 concat layers of the kadkhod head)
template <bool VEC>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ xa, int Ka, int lda,
                                                           const float* __restrict__ xb, int Kb, int ldb, int M,
                                                           const float* __restrict__ W, int ldw,
                                                           const float* __restrict__ bias, int N,
                                                           const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                           const float* __restrict__ bn_mean,
                                                           const float* __restrict__ bn_var, float bn_eps, int relu,
                                                           float* __restrict__ y, int ldy) {
    __shared__ __attribute__((aligned(16))) float Xs[2][LT][LKP];
    __shared__ __attribute__((aligned(16))) float Ws[2][LT][LKP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * LT, n0 = blockIdx.x * LT;
    const int K = Ka + Kb;
    const int wr = wave >> 1, wc = wave & 1;
    // staging: thread t brings row r = t / 4 of both tiles, k = 4 (t % 4) + 16 h .. + 3 (h = 0, 1) of the slab
    const int sr = tid >> 2, sk = 4 * (tid & 3);
    const int xm = m0 + sr, wn = n0 + sr;
    const float* xa_r = xa + (size_t)(xm < M ? xm : 0) * lda;
    const float* xb_r = xb ? xb + (size_t)(xm < M ? xm : 0) * ldb : nullptr;
    const float* w_r = W + (size_t)(wn < N ? wn : 0) * ldw;
    struct Slab {
        float4 xv[2], wv[2];
    };
    auto elem = [&](const float* r, int k) -> float { return r[k]; };
    auto fetch = [&](Slab& sl, int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + sk + 16 * h;
            float4 x4 = {0.f, 0.f, 0.f, 0.f}, w4 = {0.f, 0.f, 0.f, 0.f};
            if (VEC && k + 3 < K) {
                if (xm < M) x4 = (k < Ka) ? ld4(xa_r + k) : ld4(xb_r + (k - Ka));
                if (wn < N) w4 = ld4(w_r + k);
            } else {
                float xe[4] = {0.f, 0.f, 0.f, 0.f}, we[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j < K) {
                        if (xm < M) xe[j] = (k + j < Ka) ? elem(xa_r, k + j) : elem(xb_r, k + j - Ka);
                        if (wn < N) we[j] = elem(w_r, k + j);
                    }
                x4 = float4{xe[0], xe[1], xe[2], xe[3]};
                w4 = float4{we[0], we[1], we[2], we[3]};
            }
            sl.xv[h] = x4;
            sl.wv[h] = w4;
        }
    };
    auto stash = [&](const Slab& sl, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            st4(&Xs[buf][sr][sk + 16 * h], sl.xv[h]);
            st4(&Ws[buf][sr][sk + 16 * h], sl.wv[h]);
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // slabs are requested TWO steps ahead (two register sets that swap roles): one workgroup per CU at N = 1024 leaves a wave
    // alone on its SIMD, and one step of 32 MFMAs is shorter than a trip to L2
    Slab ra, rb;
    fetch(ra, 0);
    stash(ra, 0);
    fetch(ra, LK);
    fetch(rb, 2 * LK);
    __syncthreads();
    auto step = [&](int k0, int buf, Slab& nxt) {       // nxt holds slab k0 + LK; refilled with slab k0 + 3 LK
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = ld4(&Xs[buf][32 * wr + 16 * i + li][16 * h + 4 * kq]);
                b[i] = ld4(&Ws[buf][32 * wc + 16 * i + li][16 * h + 4 * kq]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_k16(a[i], b[j], acc[i][j]);
        }
        if (k0 + LK < K) stash(nxt, buf ^ 1);   // the other buffer: its readers finished before the barrier that ended the previous step
        if (k0 + 3 * LK < K) fetch(nxt, k0 + 3 * LK);
        __syncthreads();
    };
    for (int k0 = 0; k0 < K; k0 += 2 * LK) {
        step(k0, 0, ra);
        if (k0 + LK < K) step(k0 + LK, 1, rb);
    }
    // D[row = 4 kq + r][col = li] of every tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 32 * wc + 16 * j + li;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
        float sc = 1.f, sh = 0.f;
        if (bn_w) {  // (v - mean) / sqrt(var + eps) * w + b
            sc = bn_w[n] / sqrtf(bn_var[n] + bn_eps);
            sh = bn_b[n] - bn_mean[n] * sc;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wr + 16 * i + 4 * kq + r;
                if (m >= M) continue;
                float v = (acc[i][j][r] + bv) * sc + sh;
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
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = al16(xa) && al16(W) && (lda & 3) == 0 && (ldw & 3) == 0 && (Kb == 0 || (al16(xb) && (ldb & 3) == 0 && (Ka & 3) == 0));
    const dim3 grid((N + LT - 1) / LT, (M + LT - 1) / LT);
    if (vec)
        hipLaunchKernelGGL(linear_mfma_kernel<true>, grid, dim3(256), 0, s, xa, Ka, lda, xb, Kb, ldb, M, W, ldw, bias, N, bn_w, bn_b,
                           bn_mean, bn_var, bn_eps, relu, y, ldy);
    else
        hipLaunchKernelGGL(linear_mfma_kernel<false>, grid, dim3(256), 0, s, xa, Ka, lda, xb, Kb, ldb, M, W, ldw, bias, N, bn_w, bn_b,
                           bn_mean, bn_var, bn_eps, relu, y, ldy);
    return hip_check_launch();
}

}  // namespace mpl
