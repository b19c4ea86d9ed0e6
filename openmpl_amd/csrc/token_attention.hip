// Per-(sequence, head) softmax attention over a handful of tokens (FPT: n_tok = V <= 32).
//
// Reference: Attention.forward, MPL/lib/models/multiview_mpl.py:55-64 --
//   qkv column = s*D + h*hd + e (s = 0:q, 1:k, 2:v);  att = (q k^T) * hd^-0.5;  softmax(-1);  out = att v,
//   out channel = h*hd + e.
// Work per (sequence, head) is tiny (V^2 * hd MACs); the kernel is bound by reading the packed qkv
// rows, so one thread owns one (query row, head), keeps its V scores in registers (static unroll,
// VT = padded token count) and streams q/k/v as float4.
#include "common.hpp"

namespace mpl {

template <int VT>
__global__ __launch_bounds__(256) void token_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               int n_seq, int n_tok, int D, int H, float scale) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = n_seq * n_tok * H;
    if (idx >= total) return;
    const int h = idx % H;
    const int i = (idx / H) % n_tok;
    const int sq = idx / (H * n_tok);
    const int hd = D / H;
    const int hd4 = hd >> 2;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)sq * n_tok * ld + (size_t)h * hd;
    const float* q = base + (size_t)i * ld;

    float sc[VT];
#pragma unroll
    for (int j = 0; j < VT; ++j) {
        sc[j] = -INFINITY;
        if (j < n_tok) {
            const float* k = base + (size_t)j * ld + D;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int e = 0; e < hd4; ++e) {
                const float4 a = ld4(q + 4 * e), b = ld4(k + 4 * e);
                s0 = fmaf(a.x, b.x, s0);
                s1 = fmaf(a.y, b.y, s1);
                s2 = fmaf(a.z, b.z, s2);
                s3 = fmaf(a.w, b.w, s3);
            }
            sc[j] = ((s0 + s1) + (s2 + s3)) * scale;
        }
    }
    float mx = sc[0];
#pragma unroll
    for (int j = 1; j < VT; ++j) mx = fmaxf(mx, sc[j]);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < VT; ++j) {
        sc[j] = (j < n_tok) ? __expf(sc[j] - mx) : 0.f;
        l += sc[j];
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < VT; ++j) sc[j] *= inv;

    float* o = out + ((size_t)sq * n_tok + i) * D + (size_t)h * hd;
    const float* v0 = base + 2 * D;
    for (int e = 0; e < hd4; ++e) {
        float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            if (j < n_tok) {
                const float4 vv = ld4(v0 + (size_t)j * ld + 4 * e);
                acc.x = fmaf(sc[j], vv.x, acc.x);
                acc.y = fmaf(sc[j], vv.y, acc.y);
                acc.z = fmaf(sc[j], vv.z, acc.z);
                acc.w = fmaf(sc[j], vv.w, acc.w);
            }
        }
        st4(o + 4 * e, acc);
    }
}

int launch_token_attention(const float* qkv, int n_seq, int n_tok, int dim, int heads, float* out, hipStream_t s) {
    if (n_seq <= 0 || n_tok <= 0 || heads <= 0 || dim % heads) return MPL_E_INVALID;
    const int hd = dim / heads;
    if (hd & 3) return MPL_E_UNSUPPORTED;
    if (n_tok > 32) return MPL_E_UNSUPPORTED;
    const float scale = 1.0f / sqrtf((float)hd);
    const int total = n_seq * n_tok * heads;
    ProfScope prof(MPL_K_ATTENTION, s);
    const dim3 grid((total + 255) / 256), block(256);
#define MPL_ATT(VT) hipLaunchKernelGGL((token_attention_kernel<VT>), grid, block, 0, s, qkv, out, n_seq, n_tok, dim, heads, scale)
    if (n_tok <= 2) MPL_ATT(2);
    else if (n_tok <= 4) MPL_ATT(4);
    else if (n_tok <= 8) MPL_ATT(8);
    else if (n_tok <= 16) MPL_ATT(16);
    else MPL_ATT(32);
#undef MPL_ATT
    return hip_check_launch();
}

}  // namespace mpl
