// Device helpers shared by the FPT GEMM kernels (ln_gemm.hip: fp32 matrix cores; h2_phase.hpp: the packed-operand engines; round 2's fp32 via 3-way bf16
// operand split on the bf16 matrix cores): tile constants, LayerNorm partial statistics, epilogues, LDS-DMA.
#pragma once
#include "common.hpp"

namespace mpl {

constexpr int BM = 64;
constexpr int BN = 136;
constexpr int BNP = 144;  // 9 MFMA column tiles
constexpr int NT = 9;
constexpr int BK = 32;

// ------------------------------------------------------------------------------------------
// LayerNorm statistics travel as per-slice partials so that the GEMMs that PRODUCE a row tile by tile can emit
// them from their epilogue (no extra pass over x, no atomics):  stats[(m * NS + s) * 2 + {0,1}] = {mean_s, M2_s}
// of columns [s*SL, (s+1)*SL) of row m, SL = 136 when K is a multiple of 136 (the GEMM column-tile width), else
// K (one slice).  The consumer combines them with Chan's parallel formula:
//   mean = avg_s(mean_s),  M2 = sum_s M2_s + SL * sum_s (mean_s - mean)^2,  rstd = 1/sqrt(M2 / K + eps).
// Two-pass inside a slice + exact combination across slices: no E[x^2]-E[x]^2 cancellation anywhere.
inline int ln_slice_len(int K) { return (K % BN == 0) ? BN : K; }

__device__ __forceinline__ void ln_combine(const float* __restrict__ st, int ns, int sl, int K, float eps, float& mu,
                                           float& rs) {
    float msum = 0.f;
    for (int i = 0; i < ns; ++i) msum += st[2 * i];
    const float mean = msum / (float)ns;
    float m2 = 0.f;
    for (int i = 0; i < ns; ++i) {
        const float d = st[2 * i] - mean;
        m2 += st[2 * i + 1] + (float)sl * d * d;
    }
    mu = mean;
    rs = 1.0f / sqrtf(m2 / (float)K + eps);
}


// Epilogue shared by both GEMM kernels.  acc[n][r] = D[row0 + r][n0 + 16 n + li].  All loads (bias, residual)
// are issued before the first store: vmcnt counts stores too, so a load queued behind stores would wait for
// them to drain.
// Residual values of this lane's 36 outputs, loaded with clamped (always valid) addresses so that exactly
// NT * 4 load instructions are issued: the k loop prefetches them two stages before the epilogue and has to
// account for them in its counted vmcnt waits.
constexpr int RES_LOADS = NT * 4;
template <int NTW>
__device__ __forceinline__ void load_residual_w(float (&rv)[NTW][4], const float* R, int ldr, int M, int N, int row0,
                                                int c0, int li) {
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        int col = c0 + n * 16 + li;
        col = col < N ? col : N - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = row0 + r;
            row = row < M ? row : M - 1;
            rv[n][r] = R[(size_t)row * ldr + col];
        }
    }
}
__device__ __forceinline__ void load_residual(float (&rv)[NT][4], const float* R, int ldr, int M, int N, int row0,
                                              int n0, int li) {
    load_residual_w<NT>(rv, R, ldr, M, N, row0, n0, li);
}

// v[n][r] = epi(acc[n][r] + bias) for NTW column tiles starting at column c0 of the 136-column tile at n0; stored to C
template <int EPI, int NTW>
__device__ __forceinline__ void tile_values_store(const f32x4 (&acc)[NTW], const float* __restrict__ bias,
                                                  const float (&rv)[NTW][4], float* C, int ldc, int M, int N, int row0,
                                                  int n0, int c0, int li, float (&v)[NTW][4]) {
    const int n_end = (n0 + BN < N) ? (n0 + BN) : N;
    float bv[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int col = c0 + n * 16 + li;
        bv[n] = col < n_end ? bias[col] : 0.f;
    }
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int col = c0 + n * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = acc[n][r] + bv[n];
            if (EPI == MPL_EPI_BIAS_GELU) t = gelu_erf(t);
            if (EPI == MPL_EPI_BIAS_RESIDUAL) t += rv[n][r];
            v[n][r] = t;
            if (col < n_end && row0 + r < M) C[(size_t)(row0 + r) * ldc + col] = t;
        }
    }
}

// LayerNorm partials {mean, M2} of one full 136-column slice for the 4 rows of this lane's kq group.  A row's 136
// values sit in the 16 lanes of one kq group (8 full column tiles + lanes li < 8 of the 9th).
__device__ __forceinline__ void slice_stats_store(const float (&v)[NT][4], float* stats_out, int stats_ns, int M, int row0,
                                                  int n0, int li) {
    const bool tail = li < 8;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = tail ? v[8][r] : 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) s += v[n][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.0f / (float)BN);
        float q = 0.f;
        if (tail) {
            const float d = v[8][r] - mean;
            q = d * d;
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const float d = v[n][r] - mean;
            q = fmaf(d, d, q);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o, 64);
        if (li == 0 && row0 + r < M) {
            float* so = stats_out + ((size_t)(row0 + r) * stats_ns + n0 / BN) * 2;
            so[0] = mean;
            so[1] = q;
        }
    }
}

template <int EPI>
__device__ __forceinline__ void store_tile_epilogue(const f32x4 (&acc)[NT], const float* __restrict__ bias,
                                                    const float (&rv)[NT][4], float* C, int ldc, int M, int N,
                                                    int row0, int n0, int li, float* stats_out, int stats_ns) {
    float v[NT][4];
    tile_values_store<EPI, NT>(acc, bias, rv, C, ldc, M, N, row0, n0, n0, li, v);
    // the tile is a full slice: the host only passes stats_out when N is a multiple of 136
    if (EPI == MPL_EPI_BIAS_RESIDUAL && stats_out) slice_stats_store(v, stats_out, stats_ns, M, row0, n0, li);
}


constexpr int ATT_SCORE_FLOATS = 3584;  // LDS left for attention scores beside the 64 x 412 q|k|v tile in a 2 x 60 kB ring
constexpr int SUB_A = BM * BK * 4;   // 8192 bytes
constexpr int SUB_B = BN * BK * 4;   // 17408 bytes per column group
template <int NG> struct SubStage {
    static constexpr int GB = SUB_A + NG * SUB_B;       // gamma/beta piece offset
    static constexpr int BYTES = GB + 1024;
    static constexpr int PIECES = 8 + 17 * NG + 1;      // incl. the gamma/beta piece
};

// Fast form for the A / W pieces: address = 64-bit SGPR base (advanced by the k offset once per stage) + 32-bit
// per-lane VGPR offset that never changes, so a piece costs three instructions.  M0 is saved / restored once
// per group by the caller (dma_m0_save / dma_m0_restore).
__device__ __forceinline__ void dma16_fast(unsigned voff, const float* sbase, unsigned lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ unsigned dma_m0_save() {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
    return keep;
}
__device__ __forceinline__ void dma_m0_restore(unsigned keep) { asm volatile("s_mov_b32 m0, %0" ::"s"(keep)); }

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vm(int n) {
#define MPL_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        MPL_W(0) MPL_W(1) MPL_W(2) MPL_W(3) MPL_W(4) MPL_W(5) MPL_W(6) MPL_W(7) MPL_W(8) MPL_W(9) MPL_W(10) MPL_W(11) MPL_W(12) MPL_W(13) MPL_W(14) MPL_W(15) MPL_W(16) MPL_W(17) MPL_W(18) MPL_W(19) MPL_W(20)
        MPL_W(21) MPL_W(22) MPL_W(23) MPL_W(24) MPL_W(25) MPL_W(26) MPL_W(27) MPL_W(28) MPL_W(29) MPL_W(30) MPL_W(31) MPL_W(32) MPL_W(33) MPL_W(34) MPL_W(35) MPL_W(36) MPL_W(37) MPL_W(38) MPL_W(39) MPL_W(40) MPL_W(41)
        MPL_W(42) MPL_W(43) MPL_W(44) MPL_W(45) MPL_W(46) MPL_W(47) MPL_W(48) MPL_W(49) MPL_W(50) MPL_W(51) MPL_W(52) MPL_W(53) MPL_W(54) MPL_W(55) MPL_W(56) MPL_W(57) MPL_W(58) MPL_W(59) MPL_W(60) MPL_W(61) MPL_W(62)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef MPL_W
}


// Attention.forward :55-64 on the q | k | v tile T[64][ATT_TS] a workgroup has just produced (+bias) for its 136
// channels: scores into SC, softmax, P.V, output rows m0.. of att_out[M][Dq] at column n0.  Called by ALL threads of
// the workgroup (three barriers inside, one in front); threads tid < nth do the work.
constexpr int ATT_TS = 3 * BN + 4;                      // row stride (floats); +4 breaks the bank alignment
__device__ __forceinline__ void attention_on_tile(float* T, float* SC, int tid, int nth, int att_ntok, int att_hd,
                                                  float* att_out, int m0, int n0, int M, int Dq) {
    __syncthreads();
    const int nt = att_ntok, hd = att_hd, hd4 = hd >> 2;
    const int HP = BN / hd, S = BM / nt, nn = nt * nt;
    const float scale = 1.0f / sqrtf((float)hd);
    for (int t = tid; t < S * HP * nn; t += nth) {
        const int j = t % nt, i = (t / nt) % nt, hh = (t / nn) % HP, sq = t / (nn * HP);
        const float* q = T + (sq * nt + i) * ATT_TS + hh * hd;
        const float* k = T + (sq * nt + j) * ATT_TS + BN + hh * hd;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int e = 0; e < hd4; ++e) {
            const float4 a = ld4(q + 4 * e), b = ld4(k + 4 * e);
            s0 = fmaf(a.x, b.x, s0);
            s1 = fmaf(a.y, b.y, s1);
            s2 = fmaf(a.z, b.z, s2);
            s3 = fmaf(a.w, b.w, s3);
        }
        SC[t] = ((s0 + s1) + (s2 + s3)) * scale;
    }
    __syncthreads();
    for (int t = tid; t < S * HP * nt; t += nth) {
        float* pr = SC + t * nt;
        float mx = pr[0];
        for (int j = 1; j < nt; ++j) mx = fmaxf(mx, pr[j]);
        float l = 0.f;
        for (int j = 0; j < nt; ++j) {
            const float e = __expf(pr[j] - mx);
            pr[j] = e;
            l += e;
        }
        const float inv = 1.0f / l;
        for (int j = 0; j < nt; ++j) pr[j] *= inv;
    }
    __syncthreads();
    constexpr int C4 = BN / 4;                          // 34 float4 per output row slice
    for (int t = tid; t < BM * C4; t += nth) {
        const int c = t % C4, row = t / C4;
        const int sq = row / nt, i = row - sq * nt;
        const int hh = (4 * c) / hd;
        const float* pr = SC + ((sq * HP + hh) * nt + i) * nt;
        const float* v = T + (sq * nt) * ATT_TS + 2 * BN + 4 * c;
        float4 o = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < nt; ++j) {
            const float4 vv = ld4(v + j * ATT_TS);
            const float pj = pr[j];
            o.x = fmaf(pj, vv.x, o.x);
            o.y = fmaf(pj, vv.y, o.y);
            o.z = fmaf(pj, vv.z, o.z);
            o.w = fmaf(pj, vv.w, o.w);
        }
        if (m0 + row < M) st4(att_out + (size_t)(m0 + row) * Dq + n0 + 4 * c, o);
    }
}

}  // namespace mpl
