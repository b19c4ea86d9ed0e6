// Per-(sequence, head) softmax attention over a handful of tokens (FPT: n_tok = V <= 32).
//
// Reference: Attention.forward, MPL/lib/models/multiview_mpl.py:55-64 --
//   qkv column = s*D + h*hd + e (s = 0:q, 1:k, 2:v);  att = (q k^T) * hd^-0.5;  softmax(-1);  out = att v,
//   out channel = h*hd + e.
// Work per (sequence, head) is tiny (V^2 * hd MACs); the kernel is bound by reading the packed qkv
// rows, so one thread owns one (query row, head), keeps its V scores in registers (static unroll,
// VT = padded token count) and streams q/k/v as float4.
#include <stdlib.h>

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


// ---------------------------------------------------------------------------------------------
// LDS-staged version (the one used whenever a sequence's packed qkv rows fit in LDS): a workgroup copies
// the qkv rows of SPW consecutive sequences into LDS with fully coalesced 16-byte loads (the kernel is
// bound by that read: 3*D floats in, D floats out per token), then
//   phase 1  one thread per (sequence, head, i, j): s_ij = scale * <q_i, k_j>        -> LDS
//   phase 2  one thread per (sequence, head, i):    softmax over j, in place
//   phase 3  one thread per output float4:          o_i = sum_j p_ij v_j             -> global, coalesced
// Rows are padded by 4 floats (3*D is a multiple of 32 banks, so unpadded rows of different tokens alias).
__global__ __launch_bounds__(256) void token_attention_lds_kernel(const float* __restrict__ qkv,
                                                                   float* __restrict__ out, int n_seq, int n_tok,
                                                                   int D, int H, float scale, int spw) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int seq0 = blockIdx.x * spw;
    const int ns = (n_seq - seq0 < spw) ? (n_seq - seq0) : spw;
    const int ld = 3 * D + 4;
    const int hd = D / H, hd4 = hd >> 2;
    const int rows = ns * n_tok;
    float* sc = sm + (size_t)spw * n_tok * ld;  // [spw][H][n_tok][n_tok]

    const int row4 = (3 * D) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(qkv + (size_t)seq0 * n_tok * 3 * D);
    for (int i = tid; i < rows * row4; i += 256) {
        const int r = i / row4, c = i - r * row4;
        st4(sm + r * ld + 4 * c, g4[i]);
    }
    __syncthreads();

    const int nn = n_tok * n_tok;
    for (int t = tid; t < ns * H * nn; t += 256) {
        const int j = t % n_tok, i = (t / n_tok) % n_tok, h = (t / nn) % H, s = t / (nn * H);
        const float* q = sm + (s * n_tok + i) * ld + h * hd;
        const float* k = sm + (s * n_tok + j) * ld + D + h * hd;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int e = 0; e < hd4; ++e) {
            const float4 a = ld4(q + 4 * e), b = ld4(k + 4 * e);
            s0 = fmaf(a.x, b.x, s0);
            s1 = fmaf(a.y, b.y, s1);
            s2 = fmaf(a.z, b.z, s2);
            s3 = fmaf(a.w, b.w, s3);
        }
        sc[t] = ((s0 + s1) + (s2 + s3)) * scale;
    }
    __syncthreads();
    for (int t = tid; t < ns * H * n_tok; t += 256) {
        float* p = sc + t * n_tok;
        float mx = p[0];
        for (int j = 1; j < n_tok; ++j) mx = fmaxf(mx, p[j]);
        float l = 0.f;
        for (int j = 0; j < n_tok; ++j) {
            const float e = __expf(p[j] - mx);
            p[j] = e;
            l += e;
        }
        const float inv = 1.0f / l;
        for (int j = 0; j < n_tok; ++j) p[j] *= inv;
    }
    __syncthreads();
    const int d4 = D >> 2;
    float4* o4 = reinterpret_cast<float4*>(out + (size_t)seq0 * n_tok * D);
    for (int t = tid; t < rows * d4; t += 256) {
        const int c = t % d4, r = t / d4;          // r = s*n_tok + i
        const int h = (4 * c) / hd;
        const int s = r / n_tok, i = r - s * n_tok;
        const float* p = sc + ((s * H + h) * n_tok + i) * n_tok;
        const float* v = sm + (s * n_tok) * ld + 2 * D + 4 * c;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < n_tok; ++j) {
            const float4 vv = ld4(v + j * ld);
            const float pj = p[j];
            acc.x = fmaf(pj, vv.x, acc.x);
            acc.y = fmaf(pj, vv.y, acc.y);
            acc.z = fmaf(pj, vv.z, acc.z);
            acc.w = fmaf(pj, vv.w, acc.w);
        }
        o4[t] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Long sequences with a tiny head (the joints x views token grid, FPT_blocks_view_keypoint_tokens :261-266:
// n_tok = 17 V up to 527 tokens, D = 32, hd = 4): one workgroup per (sequence, head) keeps that head's K and V
// rows in LDS (n_tok x hd floats each).  A thread owns up to R = 3 query rows AT ONCE (t, t + 256, t + 512), so a K / V
// row is read from LDS once for all of them (uniform address: a broadcast), and the softmax is the streaming form over
// chunks of 16 keys: chunk scores in registers, running maximum m and normaliser l per row, the accumulators rescaled
// only when a chunk raises the maximum -- one exp per score, one pass over K and V.  (Round 1: one query row at a time,
// two passes, three LDS reads per (row, key): LDS-instruction bound, 504 us per launch at V = 31, B = 256.)
// Same result as the reference's softmax(-1) up to the rounding of the (mathematically exact) rescaling.
// NR query rows of one thread (i0, i0 + 256, ..) against all keys of the head
// One chunk of CH keys for the NR rows of a thread.  FULL (compile time): every key of the chunk exists -- no masking at all;
// else the chunk is the ragged last one.  The scores are kept in the exp2 domain: q was scaled by hd^-0.5 log2(e), so that
// the softmax numerators are exp2(s - m) -- one v_exp_f32 per score, no multiply in front of it.
template <int HD4, int NR, bool FULL>
__device__ __forceinline__ void attend_chunk(const float4 (&q)[NR][HD4], float4 (&o)[NR][HD4], float (&m)[NR], float (&l)[NR],
                                             const float* Ks, const float* Vs, int n_tok, int j0) {
    constexpr int HD = 4 * HD4, CH = 16;
    float sc[NR][CH];
#pragma unroll
    for (int jj = 0; jj < CH; ++jj) {
        const int j = (FULL || j0 + jj < n_tok) ? j0 + jj : n_tok - 1;
        float4 k[HD4];
#pragma unroll
        for (int c = 0; c < HD4; ++c) k[c] = ld4(Ks + j * HD + 4 * c);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float s = q[r][0].x * k[0].x;
            s = fmaf(q[r][0].y, k[0].y, s);
            s = fmaf(q[r][0].z, k[0].z, s);
            s = fmaf(q[r][0].w, k[0].w, s);
#pragma unroll
            for (int c = 1; c < HD4; ++c) {
                s = fmaf(q[r][c].x, k[c].x, s);
                s = fmaf(q[r][c].y, k[c].y, s);
                s = fmaf(q[r][c].z, k[c].z, s);
                s = fmaf(q[r][c].w, k[c].w, s);
            }
            sc[r][jj] = (FULL || j0 + jj < n_tok) ? s : -INFINITY;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float mc = sc[r][0];
#pragma unroll
        for (int jj = 1; jj < CH; ++jj) mc = fmaxf(mc, sc[r][jj]);
        // rescale what has been accumulated under the old maximum (factor 1 when the chunk does not raise it)
        const float mn = fmaxf(m[r], mc);
        const float f = __builtin_amdgcn_exp2f(m[r] - mn);
        l[r] *= f;
#pragma unroll
        for (int c = 0; c < HD4; ++c) { o[r][c].x *= f; o[r][c].y *= f; o[r][c].z *= f; o[r][c].w *= f; }
        m[r] = mn;
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) sc[r][jj] = __builtin_amdgcn_exp2f(sc[r][jj] - mn);
    }
#pragma unroll
    for (int jj = 0; jj < CH; ++jj) {
        const int j = (FULL || j0 + jj < n_tok) ? j0 + jj : n_tok - 1;
        float4 v[HD4];
#pragma unroll
        for (int c = 0; c < HD4; ++c) v[c] = ld4(Vs + j * HD + 4 * c);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float e = sc[r][jj];
            l[r] += e;
#pragma unroll
            for (int c = 0; c < HD4; ++c) {
                o[r][c].x = fmaf(e, v[c].x, o[r][c].x);
                o[r][c].y = fmaf(e, v[c].y, o[r][c].y);
                o[r][c].z = fmaf(e, v[c].z, o[r][c].z);
                o[r][c].w = fmaf(e, v[c].w, o[r][c].w);
            }
        }
    }
}

template <int HD4, int NR>
__device__ __forceinline__ void attend_rows(const float* __restrict__ base, size_t ld, const float* Ks, const float* Vs,
                                            int n_tok, int i0, int stride, float scale, float* __restrict__ out_row0,
                                            size_t out_ld) {
    constexpr int CH = 16;
    float4 q[NR][HD4], o[NR][HD4];
    float m[NR], l[NR];
    const float qs = scale * 1.4426950408889634f;        // scores in the exp2 domain
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = i0 + stride * r;
        const int ic = i < n_tok ? i : n_tok - 1;
#pragma unroll
        for (int c = 0; c < HD4; ++c) {
            q[r][c] = ld4(base + ic * ld + 4 * c);
            q[r][c].x *= qs; q[r][c].y *= qs; q[r][c].z *= qs; q[r][c].w *= qs;
            o[r][c] = float4{0.f, 0.f, 0.f, 0.f};
        }
        m[r] = -INFINITY;
        l[r] = 0.f;
    }
    int j0 = 0;
    for (; j0 + CH <= n_tok; j0 += CH) attend_chunk<HD4, NR, true>(q, o, m, l, Ks, Vs, n_tok, j0);
    if (j0 < n_tok) attend_chunk<HD4, NR, false>(q, o, m, l, Ks, Vs, n_tok, j0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = i0 + stride * r;
        if (i >= n_tok) continue;
        const float inv = 1.0f / l[r];
        float* op = out_row0 + (size_t)i * out_ld;
#pragma unroll
        for (int c = 0; c < HD4; ++c) st4(op + 4 * c, float4{o[r][c].x * inv, o[r][c].y * inv, o[r][c].z * inv, o[r][c].w * inv});
    }
}

template <int HD4>
__global__ __launch_bounds__(256) void token_attention_long_kernel(const float* __restrict__ qkv,
                                                                    float* __restrict__ out, int n_tok, int D, int H,
                                                                    float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int HD = 4 * HD4, R = 3;
    const int nthr = blockDim.x;                     // 64 * ceil(n_tok / 192): every wave gets the same number of row slots
    const int sq = blockIdx.x / H, h = blockIdx.x % H;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)sq * n_tok * ld + (size_t)h * HD;
    float* Ks = sm;
    float* Vs = sm + (size_t)n_tok * HD;
    for (int i = threadIdx.x; i < n_tok * HD4; i += nthr) {
        const int r = i / HD4, c = i % HD4;
        st4(Ks + r * HD + 4 * c, ld4(base + r * ld + D + 4 * c));
        st4(Vs + r * HD + 4 * c, ld4(base + r * ld + 2 * D + 4 * c));
    }
    __syncthreads();
    float* orow = out + (size_t)sq * n_tok * D + (size_t)h * HD;
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));     // first query row of this wave
    for (int w0 = wbase; w0 < n_tok; w0 += nthr * R) {
        // row slots of this wave that hold at least one query row (wave-uniform): w0, w0 + nthr, w0 + 2 nthr
        const int ns = (n_tok - w0 + nthr - 1) / nthr;
        const int i0 = w0 + (int)(threadIdx.x & 63u);
        if (ns >= 3) attend_rows<HD4, 3>(base, ld, Ks, Vs, n_tok, i0, nthr, scale, orow, (size_t)D);
        else if (ns == 2) attend_rows<HD4, 2>(base, ld, Ks, Vs, n_tok, i0, nthr, scale, orow, (size_t)D);
        else attend_rows<HD4, 1>(base, ld, Ks, Vs, n_tok, i0, nthr, scale, orow, (size_t)D);
    }
}

// ---- head dim 4, keys in PAIRS: the kernel above spends 12 VALU instructions per score (4 for q.k, subtract, exp, max,
// sum, 4 for p.v) at one instruction per 4 cycles and wave -- it is VALU-issue bound.  Here two keys share every multiply-add:
// K and V live in LDS as [pair]{x_j, x_j+1, y_j, y_j+1, z_j, z_j+1, w_j, w_j+1}, the scores, the exponent arguments, the row
// sum and the output accumulate as 2-vectors (packed fp32 instructions); the two halves of the output / row sum (even and odd
// keys) are added once at the end.
// Round 4 put this attention on the fp32 matrix cores three ways (exact fp32: v_mfma_f32_16x16x4_f32 is bitwise an fmaf chain
// and its K = 4 is the head width) and measured all of them SLOWER than this kernel (V = 31, B = 256, one launch, 240 us here):
// scores by 16x16x4 tiles with P V on the VALU 280 us (a lane then owns 16 keys of a row instead of a row: every score needs its
// own 16-byte V read and the LDS pipe becomes the limit); scores and P V on v_mfma_f32_4x4x1_16B_f32 with lane = query row
// (layout: tools/mfma4_probe.hip) 285 us, bitwise this kernel's result (that instruction runs at a quarter of the 16x16x4 rate);
// scores and P V (O^T += V^T P^T, 12 of 16 rows padding) on 16x16x4 369 us.  P V from fp16 pairs on the 16-bit pipe was priced
// and not built: splitting p costs as many VALU instructions per score (two conversions, a subtraction, a conversion back) as
// the two packed multiply-adds it would replace.  The kernel stays VALU-issue bound.
typedef float taf2 __attribute__((ext_vector_type(2)));
template <int NR, bool FULL>
__device__ __forceinline__ void attend_chunk_p4(const float4 (&q)[NR], taf2 (&o)[NR][4], float (&m)[NR], taf2 (&l)[NR],
                                                const float* Kp, const float* Vp, int n_tok, int j0) {
    constexpr int CP = 8;                       // pairs per chunk (16 keys)
    taf2 sc[NR][CP];
#pragma unroll
    for (int pp = 0; pp < CP; ++pp) {
        const int jp = (FULL || j0 + 2 * pp < n_tok) ? (j0 >> 1) + pp : (n_tok - 1) >> 1;
        const float4 k01 = ld4(Kp + jp * 8), k23 = ld4(Kp + jp * 8 + 4);
        const taf2 kx = {k01.x, k01.y}, ky = {k01.z, k01.w}, kz = {k23.x, k23.y}, kw = {k23.z, k23.w};
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const taf2 qx = {q[r].x, q[r].x}, qy = {q[r].y, q[r].y}, qz = {q[r].z, q[r].z}, qw = {q[r].w, q[r].w};
            taf2 s2 = qx * kx;
            s2 = __builtin_elementwise_fma(qy, ky, s2);
            s2 = __builtin_elementwise_fma(qz, kz, s2);
            s2 = __builtin_elementwise_fma(qw, kw, s2);
            if (!FULL) {
                if (j0 + 2 * pp >= n_tok) s2[0] = -INFINITY;
                if (j0 + 2 * pp + 1 >= n_tok) s2[1] = -INFINITY;
            }
            sc[r][pp] = s2;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float mc = fmaxf(sc[r][0][0], sc[r][0][1]);
#pragma unroll
        for (int pp = 1; pp < CP; ++pp) mc = fmaxf(mc, fmaxf(sc[r][pp][0], sc[r][pp][1]));
        const float mn = fmaxf(m[r], mc);
        const float f = __builtin_amdgcn_exp2f(m[r] - mn);
        const taf2 f2 = {f, f};
        l[r] *= f2;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[r][c] *= f2;
        m[r] = mn;
        const taf2 mn2 = {mn, mn};
#pragma unroll
        for (int pp = 0; pp < CP; ++pp) {
            const taf2 d = sc[r][pp] - mn2;
            sc[r][pp] = taf2{__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
        }
    }
#pragma unroll
    for (int pp = 0; pp < CP; ++pp) {
        const int jp = (FULL || j0 + 2 * pp < n_tok) ? (j0 >> 1) + pp : (n_tok - 1) >> 1;
        const float4 v01 = ld4(Vp + jp * 8), v23 = ld4(Vp + jp * 8 + 4);
        const taf2 vx = {v01.x, v01.y}, vy = {v01.z, v01.w}, vz = {v23.x, v23.y}, vw = {v23.z, v23.w};
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const taf2 e = sc[r][pp];
            l[r] += e;
            o[r][0] = __builtin_elementwise_fma(e, vx, o[r][0]);
            o[r][1] = __builtin_elementwise_fma(e, vy, o[r][1]);
            o[r][2] = __builtin_elementwise_fma(e, vz, o[r][2]);
            o[r][3] = __builtin_elementwise_fma(e, vw, o[r][3]);
        }
    }
}

template <int NR>
__device__ __forceinline__ void attend_rows_p4(const float* __restrict__ base, size_t ld, const float* Kp, const float* Vp,
                                               int n_tok, int i0, int stride, float scale, float* __restrict__ out_row0,
                                               size_t out_ld) {
    constexpr int CH = 16;
    float4 q[NR];
    taf2 o[NR][4], l[NR];
    float m[NR];
    const float qs = scale * 1.4426950408889634f;        // scores in the exp2 domain
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = i0 + stride * r;
        const int ic = i < n_tok ? i : n_tok - 1;
        q[r] = ld4(base + ic * ld);
        q[r].x *= qs; q[r].y *= qs; q[r].z *= qs; q[r].w *= qs;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[r][c] = taf2{0.f, 0.f};
        m[r] = -INFINITY;
        l[r] = taf2{0.f, 0.f};
    }
    int j0 = 0;
    for (; j0 + CH <= n_tok; j0 += CH) attend_chunk_p4<NR, true>(q, o, m, l, Kp, Vp, n_tok, j0);
    if (j0 < n_tok) attend_chunk_p4<NR, false>(q, o, m, l, Kp, Vp, n_tok, j0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = i0 + stride * r;
        if (i >= n_tok) continue;
        const float inv = 1.0f / (l[r][0] + l[r][1]);
        st4(out_row0 + (size_t)i * out_ld, float4{(o[r][0][0] + o[r][0][1]) * inv, (o[r][1][0] + o[r][1][1]) * inv,
                                                  (o[r][2][0] + o[r][2][1]) * inv, (o[r][3][0] + o[r][3][1]) * inv});
    }
}

__global__ __launch_bounds__(256) void token_attention_long_p4_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                       int n_tok, int D, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int R = 3;
    const int nthr = blockDim.x;
    const int sq = blockIdx.x / H, h = blockIdx.x % H;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)sq * n_tok * ld + (size_t)h * 4;
    const int n_pair = (n_tok + 1) >> 1;
    float* Kp = sm;
    float* Vp = sm + (size_t)n_pair * 8;
    for (int j = threadIdx.x; j < 2 * n_pair; j += nthr) {
        float4 k = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};      // the missing partner of an odd last key
        if (j < n_tok) {
            k = ld4(base + j * ld + D);
            v = ld4(base + j * ld + 2 * D);
        }
        float* kp = Kp + (j >> 1) * 8 + (j & 1);
        float* vp = Vp + (j >> 1) * 8 + (j & 1);
        kp[0] = k.x; kp[2] = k.y; kp[4] = k.z; kp[6] = k.w;
        vp[0] = v.x; vp[2] = v.y; vp[4] = v.z; vp[6] = v.w;
    }
    __syncthreads();
    float* orow = out + (size_t)sq * n_tok * D + (size_t)h * 4;
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));     // first query row of this wave
    for (int w0 = wbase; w0 < n_tok; w0 += nthr * R) {
        const int ns = (n_tok - w0 + nthr - 1) / nthr;
        const int i0 = w0 + (int)(threadIdx.x & 63u);
        if (ns >= 3) attend_rows_p4<3>(base, ld, Kp, Vp, n_tok, i0, nthr, scale, orow, (size_t)D);
        else if (ns == 2) attend_rows_p4<2>(base, ld, Kp, Vp, n_tok, i0, nthr, scale, orow, (size_t)D);
        else attend_rows_p4<1>(base, ld, Kp, Vp, n_tok, i0, nthr, scale, orow, (size_t)D);
    }
}

int launch_token_attention(const float* qkv, int n_seq, int n_tok, int dim, int heads, float* out, hipStream_t s) {
    if (n_seq <= 0 || n_tok <= 0 || heads <= 0 || dim % heads) return MPL_E_INVALID;
    const int hd = dim / heads;
    if (hd & 3) return MPL_E_UNSUPPORTED;
    if (n_tok > 32) {
        if ((hd != 4 && hd != 8) || (size_t)n_tok * hd * 8 > 64 * 1024) return MPL_E_UNSUPPORTED;
        const float sc = 1.0f / sqrtf((float)hd);
        ProfScope prof(MPL_K_ATTENTION, s);
        const size_t lds = (size_t)n_tok * hd * 8;
        // Waves per (sequence, head) block: a lane takes up to three query rows (the kernel walks the remaining rows in further
        // rounds).  Rounds 3-5 used as few waves as hold three rows per lane (527 tokens: 3).  Round 6 measured the block size
        // (profiles/r06_kptok_waves_*.txt, attention us per launch at depth 2): 85 tokens 172 (1 wave) / 153 (2); 136: 279 / 267 /
        // 249 (1 / 2 / 3); 204: 155 / 174 / 135 (2 / 3 / 4); 272: 217 / 230 / 223; 408: 407 / 391 (3 / 4); 527: 592 / 549 (3 / 4) --
        // more waves with fewer rows each win almost everywhere.  Kernels specialised for at most two / one rows per lane (110 / 60
        // instead of 158 registers: 4 / 8 waves per SIMD instead of 3) do NOT: 527 tokens 587 (2 rows, 4 waves), 607 (1 row, 9
        // waves) against 549 (profiles/r06_kptok_rows_x_waves.txt) -- the kernel is bound by issue slots, and a row that shares
        // its K / V reads with two others is cheaper than a resident wave more.  The result of a row does not depend on any of this.
        int waves = (n_tok + 47) / 48;
        waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
        static const bool no_pairs = lab_getenv("MPL_ATT_NOPAIRS") != nullptr;      // bench-only A/B switch
        if (hd == 4 && !no_pairs)
            hipLaunchKernelGGL(token_attention_long_p4_kernel, dim3(n_seq * heads), dim3(64 * waves), (size_t)((n_tok + 1) / 2) * 64, s,
                               qkv, out, n_tok, dim, heads, sc);
        else if (hd == 4)
            hipLaunchKernelGGL((token_attention_long_kernel<1>), dim3(n_seq * heads), dim3(64 * waves), lds, s, qkv, out, n_tok,
                               dim, heads, sc);
        else
            hipLaunchKernelGGL((token_attention_long_kernel<2>), dim3(n_seq * heads), dim3(64 * waves), lds, s, qkv, out, n_tok,
                               dim, heads, sc);
        return hip_check_launch();
    }
    const float scale = 1.0f / sqrtf((float)hd);
    const int total = n_seq * n_tok * heads;
    ProfScope prof(MPL_K_ATTENTION, s);
    {
        const size_t seq_bytes = (size_t)n_tok * (3 * dim + 4) * 4 + (size_t)heads * n_tok * n_tok * 4;
        static const bool force_v1 = lab_getenv("MPL_ATT_V1") != nullptr;   // bench-only A/B switch
        if (seq_bytes <= 150 * 1024 && !force_v1) {
            int spw = (int)((56 * 1024) / seq_bytes);
            if (spw < 1) spw = 1;
            if (spw > 8) spw = 8;
            const size_t lds = spw * seq_bytes;
            if (lds > 64 * 1024) {
                static std::atomic<bool> attr_set[64];   // set-once flags: a racing second hipFuncSetAttribute is harmless
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
                if (!attr_set[dev].load(std::memory_order_acquire)) {
                    if (hipFuncSetAttribute((const void*)token_attention_lds_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                        return MPL_E_LAUNCH;
                    attr_set[dev].store(true, std::memory_order_release);
                }
            }
            hipLaunchKernelGGL(token_attention_lds_kernel, dim3((n_seq + spw - 1) / spw), dim3(256), lds, s, qkv, out,
                               n_seq, n_tok, dim, heads, scale, spw);
            return hip_check_launch();
        }
    }
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
