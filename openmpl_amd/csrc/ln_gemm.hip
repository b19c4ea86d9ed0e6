// LayerNorm-fused fp32 MFMA GEMM for the FPT blocks:  C = epi( LN(A) . W^T + bias ).
//
// Reference ops replaced (MPL/lib/models/multiview_mpl.py): Block.norm1 + Attention.qkv (:55),
// Attention.proj + residual (:65, :88-90), Block.norm2 + Mlp.fc1 + nn.GELU (:32-33),
// Mlp.fc2 + residual (:35, :91).
//
// Shapes: A [M][K] row-major (M = B*V rows of the fusion transformer), W [N][K] row-major
// (nn.Linear layout, consumed as stored), K, N in {544, 1088, 1632, 2176, 3264} = multiples of
// 136 = 17*8.  Tile: 64 x 136 outputs per 256-thread workgroup -- at M = 4096 that is
// 64 x {4, 8, 12} = 256 / 512 / 768 workgroups, an exact multiple of the 256 CUs for every GEMM of
// a block.  Each of the 4 waves owns 16 rows x 9 MFMA tiles of 16 columns (the 9th tile is half
// padding: 136 = 8.5 * 16).  BK = 32; LDS tiles are padded to a 36-float row so that the
// ds_read_b128 fragment reads and ds_write_b128 staging writes stay (nearly) conflict free.
// The k index inside a 16-deep step is permuted identically for A and B (common.hpp) so that each
// lane fetches its four k values with ONE 128-bit LDS read.
#include <stdlib.h>

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

// one wave per row; used for rows that no GEMM epilogue produced (the SPT output, stand-alone mpl_ln_linear)
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int M, int K, int ldx, int sl,
                                                         float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int ns = K / sl, n4 = sl >> 2;
    for (int sidx = 0; sidx < ns; ++sidx) {
        const float* xr = x + (size_t)row * ldx + sidx * sl;
        float s = 0.f;
        for (int i = lane; i < n4; i += 64) {
            const float4 v = ld4(xr + 4 * i);
            s += (v.x + v.y) + (v.z + v.w);
        }
        s = wave_sum(s);
        const float mean = s / (float)sl;
        float ss = 0.f;
        for (int i = lane; i < n4; i += 64) {
            const float4 v = ld4(xr + 4 * i);
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            stats[((size_t)row * ns + sidx) * 2] = mean;
            stats[((size_t)row * ns + sidx) * 2 + 1] = ss;
        }
    }
}

int launch_row_stats(const float* x, int M, int K, int ldx, float* stats, hipStream_t s) {
    if (M <= 0 || (K & 3)) return MPL_E_INVALID;
    ProfScope prof(MPL_K_ROW_STATS, s);
    hipLaunchKernelGGL(row_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, M, K, ldx, ln_slice_len(K), stats);
    return hip_check_launch();
}

// Epilogue shared by both GEMM kernels.  acc[n][r] = D[row0 + r][n0 + 16 n + li].  All loads (bias, residual)
// are issued before the first store: vmcnt counts stores too, so a load queued behind stores would wait for
// them to drain.
// Residual values of this lane's 36 outputs, loaded with clamped (always valid) addresses so that exactly
// NT * 4 load instructions are issued: the k loop prefetches them two stages before the epilogue and has to
// account for them in its counted vmcnt waits.
constexpr int RES_LOADS = NT * 4;
__device__ __forceinline__ void load_residual(float (&rv)[NT][4], const float* R, int ldr, int M, int N, int row0,
                                              int n0, int li) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int col = n0 + n * 16 + li;
        col = col < N ? col : N - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = row0 + r;
            row = row < M ? row : M - 1;
            rv[n][r] = R[(size_t)row * ldr + col];
        }
    }
}

template <int EPI>
__device__ __forceinline__ void store_tile_epilogue(const f32x4 (&acc)[NT], const float* __restrict__ bias,
                                                    const float (&rv)[NT][4], float* C, int ldc, int M, int N,
                                                    int row0, int n0, int li, float* stats_out, int stats_ns) {
    const int n_end = (n0 + BN < N) ? (n0 + BN) : N;
    float bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = n0 + n * 16 + li;
        bv[n] = col < n_end ? bias[col] : 0.f;
    }
    float v[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = n0 + n * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = acc[n][r] + bv[n];
            if (EPI == MPL_EPI_BIAS_GELU) t = gelu_erf(t);
            if (EPI == MPL_EPI_BIAS_RESIDUAL) t += rv[n][r];
            v[n][r] = t;
            if (col < n_end && row0 + r < M) C[(size_t)(row0 + r) * ldc + col] = t;
        }
    }
    if (EPI == MPL_EPI_BIAS_RESIDUAL && stats_out) {
        // LayerNorm partials of this 136-column slice for the rows of this wave (the tile is a full slice:
        // the host only passes stats_out when N is a multiple of 136).  A row's 136 values sit in the 16 lanes
        // of one kq group (8 full column tiles + lanes li < 8 of the 9th).
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
}

// ------------------------------------------------------------------------------------------
// DMA-staged, multi-group kernel (the product path).
//
// Workgroup = 64 rows x (NG x 136) columns, 4*NG waves: wave w owns row group w & 3 (16 rows) of column
// group w >> 2 (136 columns = 9 MFMA tiles, the 9th half padding).  The A k-tile is staged once and shared by
// the NG column groups, so ONE workgroup per CU has NG waves per SIMD and every barrier interval carries
// NG x KS x 72 MFMAs per SIMD: the per-k-tile bubble (barrier, DMA issue, first fragment read) is amortised
// over 2-3x more matrix work than with one 4-wave workgroup per tile, and at M = 4096 every GEMM of a block is
// exactly 256 workgroups: QKV (N = 3 D) NG = 3, fc1 (N = 2 D) NG = 2, proj / fc2 (N = D) NG = 1 with KS = 2
// k-tiles per stage.
//
// Staging: k-tiles go L2/HBM -> LDS by global_load_lds_dwordx4 (no staging VGPRs, no ds_write) into a ring of
// NST stages; one raw s_barrier and one counted s_waitcnt vmcnt per stage.  Sub-stage layout (per k-tile of 32):
//   A 64 rows x 128 B | B group 0: 136 rows x 128 B | ... | B group NG-1 | 1 KiB: gamma[32] beta[32] of the tile
// (the 8 padding rows 136..143 of a B group alias the next region: finite data, results never stored).
// Rows are unpadded; the 16-B column c of row r sits at c ^ ((r >> 1) & 7) -- applied to the per-lane DMA
// SOURCE address (the LDS image of a DMA piece is lane-linear) and to the fragment read (bank conflicts: 0).
// The DMA is issued from inline asm: with the builtin hipcc assumes the LDS write aliases every later ds_read
// and drains vmcnt(0) in front of it; the asm DMA is invisible to the compiler's counters and is tracked by the
// explicit counted waits.  LayerNorm is applied when the A fragment is read.
constexpr int ATT_SCORE_FLOATS = 3584;  // LDS left for attention scores beside the 64 x 412 q|k|v tile in a 2 x 60 kB ring
constexpr int SUB_A = BM * BK * 4;   // 8192 bytes
constexpr int SUB_B = BN * BK * 4;   // 17408 bytes per column group
template <int NG> struct SubStage {
    static constexpr int GB = SUB_A + NG * SUB_B;       // gamma/beta piece offset
    static constexpr int BYTES = GB + 1024;
    static constexpr int PIECES = 8 + 17 * NG + 1;      // incl. the gamma/beta piece
};

// One 1-KiB DMA piece: lane l's 16 bytes at g land at LDS byte address lds_dst + 16 l (lds_dst wave-uniform).
// M0 is saved/restored inside the statement (the compiler does not preserve it around asm).
__device__ __forceinline__ void dma16(const float* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_dst)
        : "memory");
}

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

struct Frag {
    float4 a, g, be;
    float4 b[NT];
};

// fragments of 16-deep k step `step` (sub-tile step >> 1, half step & 1) of the stage at `st`
template <bool LN, int NG>
__device__ __forceinline__ void load_frag(Frag& f, const char* st, int step, int rg, int cg, int li, int kq, int swz) {
    typedef SubStage<NG> SS;
    const char* sb = st + (step >> 1) * SS::BYTES;
    const int cl = ((step & 1) << 2) + kq;          // logical 16-B column of this lane
    const int cc = (cl ^ swz) << 2;
    const float* as = reinterpret_cast<const float*>(sb) + (rg * 16 + li) * BK;
    const float* bs = reinterpret_cast<const float*>(sb + SUB_A + cg * SUB_B) + li * BK;
    f.a = ld4(as + cc);
#pragma unroll
    for (int n = 0; n < NT; ++n) f.b[n] = ld4(bs + n * 16 * BK + cc);
    if (LN) {
        const float* gb = reinterpret_cast<const float*>(sb + SS::GB);
        f.g = ld4(gb + 4 * cl);
        f.be = ld4(gb + 32 + 4 * cl);
    }
}

// ATT (QKV projection only, NG = 3): the three column groups of a workgroup are the q, k and v slices of the SAME
// 136 attention channels (column base cg * D + tn * 136 instead of three adjacent tiles), so the workgroup owns
// q, k, v of 136 / hd heads for its 64 rows = 64 / n_tok whole sequences and finishes Attention.forward :55-64
// in its epilogue (scores, softmax, P.V through LDS): the packed qkv tensor never goes to memory.
// LDW: one extra "loader" wave per workgroup issues every DMA piece and owns the counted vmcnt waits; the compute
// waves then execute nothing but barrier -> fragment reads -> MFMA (a DMA piece costs ~40 issue cycles, 7 of them
// per k-tile were 10 % of a compute wave's critical path).
template <int EPI, bool LN, int NG, int KG, int KS, int NST, int ABL = 0, bool ATT = false, bool LDW = false>
__global__ __launch_bounds__(256 * NG * KG + (LDW ? 64 : 0), 1) void ln_gemm_ng_kernel(const float* __restrict__ A, int lda,
                                                                       const float* __restrict__ stats,
                                                                       const float* __restrict__ ln_w,
                                                                       const float* __restrict__ ln_b,
                                                                       const float* __restrict__ W,
                                                                       const float* __restrict__ bias, const float* R,
                                                                       int ldr, float* C, int ldc, int M, int N, int K,
                                                                       int grid_m, int grid_n, float eps,
                                                                       float* stats_out, int att_ntok, int att_hd,
                                                                       float* att_out) {
    static_assert(!ATT || (NG == 3 && KG == 1 && EPI == MPL_EPI_BIAS), "fused attention needs the q|k|v geometry");
    typedef SubStage<NG> SS;
    constexpr int NW = 4 * NG * KG;                  // compute waves: 4 row groups x NG column groups x KG k groups
    constexpr int NTHREADS = 64 * NW + (LDW ? 64 : 0);
    constexpr int STAGE = KS * SS::BYTES;
    constexpr int SPW = (2 * KS) / KG;               // 16-deep k steps per wave per full stage
    static_assert(SPW * KG == 2 * KS, "k groups must divide the steps of a stage");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const unsigned long long t_entry = (ABL & 4) ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long t_real = (ABL & 4) ? __builtin_amdgcn_s_memrealtime() : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & 3, cg = (wave >> 2) % NG, kg = (wave >> 2) / NG;
    const int li = lane & 15;
    const int kq = lane >> 4;

    int tm, tn;
    {
        const int b = blockIdx.x;
        if ((grid_m & 7) == 0) {  // XCD-aware: blocks b, b+8, .. share an XCD/L2 -> give each XCD a band of m tiles
            const int per = grid_m >> 3;
            const int xcd = b & 7, i = b >> 3;
            tm = xcd * per + (i % per);
            tn = i / per;
        } else {
            tm = b % grid_m;
            tn = b / grid_m;
        }
    }
    const int m0 = tm * BM;
    const int n0 = ATT ? tn * BN : tn * (BN * NG);
    const int Dq = N / 3;                                    // ATT: width of each of q, k, v
    auto colbase = [&](int g) -> int { return ATT ? g * Dq + n0 : n0 + g * BN; };   // first column of group g

    float mu = 0.f, rs = 1.f;
    if (LN) {
        int m = m0 + rg * 16 + li;
        m = m < M ? m : M - 1;
        const int sl = (K % BN == 0) ? BN : K, ns = K / sl;
        ln_combine(stats + (size_t)m * ns * 2, ns, sl, K, eps, mu, rs);
        // consume the loads here: otherwise hipcc parks their s_waitcnt vmcnt(0) at the first use INSIDE the
        // k loop, where it would drain the (compiler-invisible) DMA ring every iteration
        asm volatile("" : "+v"(mu), "+v"(rs));
    }
    if (!LN) {  // the gamma/beta KiB doubles as padding rows of the last B group: keep it finite
        for (int i = tid; i < NST * KS * 256; i += NTHREADS) {
            const int sub = i >> 8;
            reinterpret_cast<float*>(smem + sub * SS::BYTES + SS::GB)[i & 255] = 0.f;
        }
    }

    // ---- DMA piece table of this wave.  Per k-tile: 8 A pieces, 17*NG W pieces, (LN) one gamma/beta piece, dealt
    // so that the slot TYPE is a compile-time property (no per-piece control flow):
    //   A slot a   : piece a*NW + wave                      (a < A_PER; valid when < 8)
    //   W slot b   : piece b*NW + wave of the W slab        (b < W_FULL, always valid)
    //   W extra    : piece W_FULL*NW + wave                 (only waves < W_REM)
    //   gamma/beta : last wave only (generic 64-bit-address form: two unrelated base pointers in one wave)
    // A / W pieces use the fast form: wave-uniform 64-bit base in SGPRs + constant per-lane 32-bit byte offset.
    constexpr int A_PER = (8 + NW - 1) / NW;
    constexpr int W_FULL = (17 * NG) / NW;
    constexpr int W_REM = (17 * NG) % NW;
    unsigned voA[A_PER], voW[W_FULL + 1];
#pragma unroll
    for (int a = 0; a < A_PER; ++a) {
        const int r = (a * NW + wave) * 8 + (lane >> 3);
        int m = m0 + r;
        m = m < M ? m : M - 1;
        voA[a] = (unsigned)(((size_t)m * lda + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * sizeof(float));
        asm volatile("" : "+v"(voA[a]));   // opaque: keep it in a register instead of re-deriving it every stage
    }
#pragma unroll
    for (int b = 0; b < W_FULL + 1; ++b) {
        int pw = b * NW + wave;
        pw = pw < 17 * NG ? pw : 17 * NG - 1;
        const int r = pw * 8 + (lane >> 3);                 // row inside the NG*136-row B slab
        const int rr = r % BN;                              // row inside its column group (swizzle key)
        int n = colbase(r / BN) + rr;
        n = n < N ? n : N - 1;
        voW[b] = (unsigned)(((size_t)n * K + 4 * ((lane & 7) ^ ((rr >> 1) & 7))) * sizeof(float));
        asm volatile("" : "+v"(voW[b]));
    }
    const float* gb_src = ((lane & 8) ? ln_b : ln_w) + 4 * (lane & 7);   // lanes >= 16 re-load the same 256 B
    const bool a_on = A_PER * NW <= 8 || wave < 8 - (A_PER - 1) * NW;     // last A slot valid for this wave?
    const bool w_extra = wave < W_REM;
    const bool gb_on = LN && wave == NW - 1;
    const int per_sub = (A_PER - 1) + (a_on ? 1 : 0) + W_FULL + (w_extra ? 1 : 0) + (gb_on ? 1 : 0);
    const int KT = K / BK;                      // k-tiles
    const int T = (KT + KS - 1) / KS;           // stages
    const int last_sub = KT - (T - 1) * KS;     // valid sub-tiles of the last stage
    auto pieces_of = [&](int t) -> int {        // pieces this wave issues for stage t (wave-uniform)
        return (t == T - 1 ? last_sub : KS) * per_sub;
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue_stage = [&](int t) {
        const unsigned keep = dma_m0_save();
        const int nsub = (t == T - 1) ? last_sub : KS;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks < nsub) {
                const unsigned st = lds0 + (unsigned)((t % NST) * STAGE + ks * SS::BYTES);
                const int k0 = (t * KS + ks) * BK;
#pragma unroll
                for (int a = 0; a < A_PER; ++a)
                    if (a < A_PER - 1 || a_on) dma16_fast(voA[a], A + k0, st + (unsigned)((a * NW + wave) * 1024));
#pragma unroll
                for (int b = 0; b < W_FULL; ++b) dma16_fast(voW[b], W + k0, st + (unsigned)((8 + b * NW + wave) * 1024));
                if (w_extra) dma16_fast(voW[W_FULL], W + k0, st + (unsigned)((8 + W_FULL * NW + wave) * 1024));
                if (gb_on) dma16(gb_src + k0, st + (unsigned)SS::GB);
            }
        }
        dma_m0_restore(keep);
    };

    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (!LN) __syncthreads();                   // zero fill above is ordinary LDS traffic: order it first
    if (LDW && wave == NW) {
        // ---------------- loader wave: all DMA pieces of every stage, nothing else
        constexpr int NPW = 17 * NG;
        unsigned lA[8], lW[NPW];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = p * 8 + (lane >> 3);
            int m = m0 + r;
            m = m < M ? m : M - 1;
            lA[p] = (unsigned)(((size_t)m * lda + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * sizeof(float));
            asm volatile("" : "+v"(lA[p]));
        }
#pragma unroll
        for (int p = 0; p < NPW; ++p) {
            const int r = p * 8 + (lane >> 3);
            const int rr = r % BN;
            int n = colbase(r / BN) + rr;
            n = n < N ? n : N - 1;
            lW[p] = (unsigned)(((size_t)n * K + 4 * ((lane & 7) ^ ((rr >> 1) & 7))) * sizeof(float));
            asm volatile("" : "+v"(lW[p]));
        }
        constexpr int PSUB = 8 + NPW + (LN ? 1 : 0);
        auto l_issue = [&](int t) {
            const unsigned keep = dma_m0_save();
            const int nsub = (t == T - 1) ? last_sub : KS;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < nsub) {
                    const unsigned st = lds0 + (unsigned)((t % NST) * STAGE + ks * SS::BYTES);
                    const int k0 = (t * KS + ks) * BK;
#pragma unroll
                    for (int p = 0; p < 8; ++p) dma16_fast(lA[p], A + k0, st + (unsigned)(p * 1024));
#pragma unroll
                    for (int p = 0; p < NPW; ++p) dma16_fast(lW[p], W + k0, st + (unsigned)((8 + p) * 1024));
                    if (LN) dma16(gb_src + k0, st + (unsigned)SS::GB);
                }
            }
            dma_m0_restore(keep);
        };
        auto l_pieces = [&](int t) -> int { return (t == T - 1 ? last_sub : KS) * PSUB; };
#pragma unroll
        for (int t = 0; t < NST - 1; ++t)
            if (t < T) l_issue(t);
        for (int t = 0; t < T; ++t) {
            int allow = 0;
#pragma unroll
            for (int j = 1; j <= NST - 2; ++j)
                if (t + j < T) allow += l_pieces(t + j);
            wait_vm(allow);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + NST - 1 < T) l_issue(t + NST - 1);
        }
        if (KG > 1) { __syncthreads(); __syncthreads(); }
        if (ATT) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
        return;
    }
    if (!LDW) {
#pragma unroll
        for (int t = 0; t < NST - 1; ++t)
            if (t < T) issue_stage(t);
    }

    float rv[NT][4];                            // residual (EPI_BIAS_RESIDUAL), prefetched inside the k loop
    const int t_res = T >= 2 ? T - 2 : 0;      // stage during which the residual loads are issued
    const int swz = (li >> 1) & 7;
    // bench-only phase timing (ABL & 4): shader-clock cycles summed over the stages of this wave
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, acc_wait = 0, acc_bar = 0, acc_issue = 0, acc_comp = 0;
    const unsigned long long t_begin = (ABL & 4) ? __builtin_amdgcn_s_memtime() : 0;
    for (int t = 0; t < T; ++t) {
        if (ABL & 4) tk0 = __builtin_amdgcn_s_memtime();
        // stage t has landed for this wave once only the pieces of stages t+1 .. t+NST-2 are outstanding
        int allow = 0;
#pragma unroll
        for (int j = 1; j <= NST - 2; ++j)
            if (t + j < T) allow += pieces_of(t + j);
        if (EPI == MPL_EPI_BIAS_RESIDUAL && t > t_res) allow += RES_LOADS;   // younger than every DMA piece
        if (ABL & 1) allow = 0;
        if (!LDW) wait_vm(allow);
        if (ABL & 4) tk1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();           // everyone's pieces landed; everyone is done reading stage t-1
        asm volatile("" ::: "memory");
        if (ABL & 4) tk2 = __builtin_amdgcn_s_memtime();
        // refill the stage the barrier just freed, at once: measured on MI355X, issuing the DMA here beats hiding
        // its address arithmetic behind the first fragment reads or the first MFMA block for every ring depth
        if (!LDW && t + NST - 1 < T && !(ABL & 1)) issue_stage(t + NST - 1);
        // the epilogue's residual operand: issue its loads now, two stages of MFMA work ahead of their use
        if (EPI == MPL_EPI_BIAS_RESIDUAL && t == t_res)
            load_residual(rv, R, ldr, M, N, m0 + rg * 16 + 4 * kq, colbase(cg), li);
        if (ABL & 4) tk3 = __builtin_amdgcn_s_memtime();

        const char* st = smem + (t % NST) * STAGE;
        const int nstep = 2 * ((t == T - 1) ? last_sub : KS);   // 16-deep k steps in this stage
        // This wave owns steps kg, kg + KG, ...  Fragment registers are double buffered by hand (reads of the
        // next own step are issued before the 36 MFMAs of the current one); sched_barrier(0) keeps hipcc from
        // re-serialising them into read->wait->4 dependent MFMAs per column tile (its minimum-register schedule).
        Frag f[2];
        if (kg < nstep) load_frag<LN, NG>(f[0], st, kg, rg, cg, li, kq, swz);
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int i = kg + j * KG;
            if (i < nstep) {
                if (j + 1 < SPW && i + KG < nstep) load_frag<LN, NG>(f[(j + 1) & 1], st, i + KG, rg, cg, li, kq, swz);
                __builtin_amdgcn_sched_barrier(0);
                Frag& c = f[j & 1];
                if (LN) {
                    c.a.x = (c.a.x - mu) * rs * c.g.x + c.be.x;
                    c.a.y = (c.a.y - mu) * rs * c.g.y + c.be.y;
                    c.a.z = (c.a.z - mu) * rs * c.g.z + c.be.z;
                    c.a.w = (c.a.w - mu) * rs * c.g.w + c.be.w;
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = mfma16(c.a.x, c.b[n].x, acc[n]);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = mfma16(c.a.y, c.b[n].y, acc[n]);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = mfma16(c.a.z, c.b[n].z, acc[n]);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = mfma16(c.a.w, c.b[n].w, acc[n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ABL & 4) {
            const unsigned long long tk4 = __builtin_amdgcn_s_memtime();
            acc_wait += tk1 - tk0; acc_bar += tk2 - tk1; acc_issue += tk3 - tk2; acc_comp += tk4 - tk3;
        }
    }
    unsigned long long t_loop_end = 0;
    if (ABL & 4) t_loop_end = __builtin_amdgcn_s_memtime();
    if (KG > 1) {
        // k groups hold partial sums of the same output tile: fold groups 1..KG-1 into group 0 through LDS in a
        // fixed order (deterministic).  All DMA has been waited for, so the ring memory is free.
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (kg > 0) {
            float* dstp = red + (size_t)(((kg - 1) * NG + cg) * 4 + rg) * (NT * 4 * 64) + lane;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) dstp[(n * 4 + r) * 64] = acc[n][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int k = 1; k < KG; ++k) {
            const float* sp = red + (size_t)(((k - 1) * NG + cg) * 4 + rg) * (NT * 4 * 64) + lane;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[n][r] += sp[(n * 4 + r) * 64];
        }
    }
    if (ABL & 2) {  // bench-only: keep acc live, store (almost) nothing
        float sacc = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) sacc += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
        if (sacc == 12345.678f) C[tid] = sacc;
        return;
    }
    if (ATT) {
        // ---- fused attention epilogue.  T[64][412]: q | k | v (+bias) of this workgroup's 136 channels.
        constexpr int TS = 3 * BN + 4;                      // row stride (floats); +4 breaks the bank alignment
        float* T = reinterpret_cast<float*>(smem);
        float* SC = T + BM * TS;                            // scores [S][HP][n_tok][n_tok]
        __syncthreads();                                    // every wave is done reading the last stage
        {
            const int cb = colbase(cg);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c = n * 16 + li;
                if (c < BN) {
                    const float bv = bias[cb + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[(rg * 16 + 4 * kq + r) * TS + cg * BN + c] = acc[n][r] + bv;
                }
            }
        }
        __syncthreads();
        const int nt = att_ntok, hd = att_hd, hd4 = hd >> 2;
        const int HP = BN / hd, S = BM / nt, nn = nt * nt;
        const float scale = 1.0f / sqrtf((float)hd);
        constexpr int NTH = 64 * NW;   // compute waves only (the loader wave just matches the barriers)
        for (int t = tid; t < S * HP * nn; t += NTH) {
            const int j = t % nt, i = (t / nt) % nt, hh = (t / nn) % HP, sq = t / (nn * HP);
            const float* q = T + (sq * nt + i) * TS + hh * hd;
            const float* k = T + (sq * nt + j) * TS + BN + hh * hd;
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
        for (int t = tid; t < S * HP * nt; t += NTH) {
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
        for (int t = tid; t < BM * C4; t += NTH) {
            const int c = t % C4, row = t / C4;
            const int sq = row / nt, i = row - sq * nt;
            const int hh = (4 * c) / hd;
            const float* pr = SC + ((sq * HP + hh) * nt + i) * nt;
            const float* v = T + (sq * nt) * TS + 2 * BN + 4 * c;
            float4 o = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < nt; ++j) {
                const float4 vv = ld4(v + j * TS);
                const float pj = pr[j];
                o.x = fmaf(pj, vv.x, o.x);
                o.y = fmaf(pj, vv.y, o.y);
                o.z = fmaf(pj, vv.z, o.z);
                o.w = fmaf(pj, vv.w, o.w);
            }
            if (m0 + row < M) st4(att_out + (size_t)(m0 + row) * Dq + n0 + 4 * c, o);
        }
        return;
    }
    store_tile_epilogue<EPI>(acc, bias, rv, C, ldc, M, N, m0 + rg * 16 + 4 * kq, colbase(cg), li,
                             (ABL & 4) ? nullptr : stats_out, N / BN);
    if (ABL & 4) {   // bench-only: per-wave phase cycles into the (otherwise unused) stats_out buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            float* o = stats_out + (size_t)(blockIdx.x * NW + wave) * 12;
            o[0] = (float)acc_wait; o[1] = (float)acc_bar; o[2] = (float)acc_issue; o[3] = (float)acc_comp;
            o[4] = (float)(t_loop_end - t_begin); o[5] = (float)T; o[6] = (float)(t_begin - t_entry);
            o[7] = (float)(t_end - t_loop_end); o[8] = (float)(t_real & 0xFFFFFF); o[9] = (float)(t_end - t_entry);
            o[10] = 0.f; o[11] = 0.f;
        }
    }
}

template <int EPI, bool LN, int NG, int KG, int KS, int NST, int ABL = 0, bool ATT = false, bool LDW = false>
static int launch_ng(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b, const float* W,
                     const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N, int K, float eps,
                     float* stats_out, hipStream_t s, int att_ntok = 0, int att_hd = 0, float* att_out = nullptr) {
    constexpr int LDS = NST * KS * SubStage<NG>::BYTES;
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    static_assert((KG - 1) * NG * 4 * NT * 4 * 64 * 4 <= LDS, "k-group reduction does not fit in the ring");
    const int gm = (M + BM - 1) / BM, gn = (N + BN * NG - 1) / (BN * NG);
    static_assert(!ATT || (BM * (3 * BN + 4) + ATT_SCORE_FLOATS) * 4 <= LDS, "attention epilogue does not fit in the ring");
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)ln_gemm_ng_kernel<EPI, LN, NG, KG, KS, NST, ABL, ATT, LDW>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev] = true;
    }
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((ln_gemm_ng_kernel<EPI, LN, NG, KG, KS, NST, ABL, ATT, LDW>), dim3(gm * gn),
                       dim3(256 * NG * KG + (LDW ? 64 : 0)), LDS, s,
                       A, lda, stats, ln_w, ln_b, W, bias, R, ldr, C, ldc, M, N, K, gm, gn, eps, stats_out, att_ntok,
                       att_hd, att_out);
    return hip_check_launch();
}

template <int EPI, bool LN>
static int launch_ng_auto(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b,
                          const float* W, const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N,
                          int K, float eps, float* stats_out, hipStream_t s) {
    // widest column grouping that still yields >= 256 workgroups (one per CU); small problems stay at NG = 1
    const int gm = (M + BM - 1) / BM;
    static const int force = getenv("MPL_GEMM_NG") ? atoi(getenv("MPL_GEMM_NG")) : 0;   // bench-only
    static const int kgsel = getenv("MPL_GEMM_KG") ? atoi(getenv("MPL_GEMM_KG")) : 0;    // bench-only
    int ng = 1;
    if (N % (BN * 3) == 0 && gm * (N / (BN * 3)) >= 256) ng = 3;
    if (force) ng = force;
#define MPL_ARGS2 A, lda, stats, ln_w, ln_b, W, bias, R, ldr, C, ldc, M, N, K, eps, stats_out, s
    static const int abl = getenv("MPL_GEMM_ABL") ? atoi(getenv("MPL_GEMM_ABL")) : 0;   // bench-only ablations
    if (abl == 4 && !LN) {
        if (ng == 3) return launch_ng<EPI, false, 3, 1, 1, 2, 4>(MPL_ARGS2);
        return launch_ng<EPI, false, 1, 1, 1, 2, 4>(MPL_ARGS2);
    }
    if (abl && EPI == 0 && !LN) {
        if (ng == 3) return abl == 1 ? launch_ng<0, false, 3, 1, 1, 2, 1>(MPL_ARGS2) : abl == 2 ? launch_ng<0, false, 3, 1, 1, 2, 2>(MPL_ARGS2) : launch_ng<0, false, 3, 1, 1, 2, 3>(MPL_ARGS2);
        if (ng == 1) return abl == 1 ? launch_ng<0, false, 1, 1, 1, 3, 1>(MPL_ARGS2) : abl == 2 ? launch_ng<0, false, 1, 1, 1, 3, 2>(MPL_ARGS2) : launch_ng<0, false, 1, 1, 1, 3, 3>(MPL_ARGS2);
    }
    // Every configuration keeps KG = 1: the k order of each output element is then independent of the launch
    // geometry, so results do not depend on the batch size (tests: batch split / permutation are bitwise equal).
    // Measured on MI355X (tools/gemm_ab.py, M = 4096, D = 544): QKV 80 us with either geometry; the 4-wave
    // workgroup (two per CU, 3-stage ring) wins for N = D and N = 2 D, k-group splitting never paid.
    // bench-only geometry override: MPL_GEMM_CFG = ng*100 + ks*10 + nst
    static const int cfg = getenv("MPL_GEMM_CFG") ? atoi(getenv("MPL_GEMM_CFG")) : 0;
    switch (cfg) {
        case 116: return launch_ng<EPI, LN, 1, 1, 1, 6>(MPL_ARGS2);
        case 114: return launch_ng<EPI, LN, 1, 1, 1, 4>(MPL_ARGS2);
        case 113: return launch_ng<EPI, LN, 1, 1, 1, 3>(MPL_ARGS2);
        case 112: return launch_ng<EPI, LN, 1, 1, 1, 2>(MPL_ARGS2);
        case 123: return launch_ng<EPI, LN, 1, 1, 2, 3>(MPL_ARGS2);
        case 213: return launch_ng<EPI, LN, 2, 1, 1, 3>(MPL_ARGS2);
        case 312: return launch_ng<EPI, LN, 3, 1, 1, 2>(MPL_ARGS2);
        case 9112: return launch_ng<EPI, LN, 1, 1, 1, 2, 0, false, true>(MPL_ARGS2);   // loader-wave variants
        case 9113: return launch_ng<EPI, LN, 1, 1, 1, 3, 0, false, true>(MPL_ARGS2);
        case 9312: return launch_ng<EPI, LN, 3, 1, 1, 2, 0, false, true>(MPL_ARGS2);
        default: break;
    }
    if (ng == 3) return launch_ng<EPI, LN, 3, 1, 1, 2>(MPL_ARGS2);
    // 4-wave workgroups: a 2-stage ring (53 kB) lets 3 workgroups share a CU, a 3-stage ring (80 kB) 2.  More
    // independent workgroups per CU hide each other's barrier bubbles better than a deeper ring does (measured:
    // proj 32 vs 35 us, fc2 57 vs 61 us at D = 544), unless 3 per CU quantises badly (1024 workgroups = 1.33
    // rounds of 768 slots): pick the occupancy with the fewer CU-time units, ties go to 3 per CU.
    const int wgs = gm * ((N + BN - 1) / BN);
    auto cost = [&](int k) {
        const int rounds = (wgs + 256 * k - 1) / (256 * k);
        const int per_cu = (wgs + 255) / 256;
        return rounds * (per_cu < k ? per_cu : k);
    };
    // a lone workgroup per CU has nobody to hide its DMA issue behind: give it a loader wave (measured: proj
    // 31.4 -> 29.9 us, fc2 56.2 -> 53.1 us; neutral or slightly negative once two workgroups share the CU)
    if (wgs <= 256) return launch_ng<EPI, LN, 1, 1, 1, 2, 0, false, true>(MPL_ARGS2);
    if (cost(3) <= cost(2)) return launch_ng<EPI, LN, 1, 1, 1, 2>(MPL_ARGS2);
    return launch_ng<EPI, LN, 1, 1, 1, 3>(MPL_ARGS2);
#undef MPL_ARGS2
}

// LN1 + qkv projection + softmax attention in one launch: att[M, D] from x[M, D].  Requirements (else the caller
// uses the separate kernels): 136 % hd == 0, 64 % n_tok == 0, D % 136 == 0.
bool qkv_attention_fusable(int n_tok, int dim, int heads) {
    static const bool off = getenv("MPL_NO_ATT_FUSION") != nullptr;   // bench-only A/B switch
    if (off || heads <= 0 || dim % heads) return false;
    const int hd = dim / heads;
    return dim % BN == 0 && BN % hd == 0 && (hd & 3) == 0 && n_tok >= 1 && BM % n_tok == 0 && n_tok * n_tok * (BN / hd) * (BM / n_tok) <= ATT_SCORE_FLOATS;
}

int launch_ln_qkv_attention(const float* x, int M, int D, const float* stats, const float* ln_w, const float* ln_b,
                            float eps, const float* W, const float* bias, int n_tok, int heads, float* att,
                            hipStream_t s) {
    if (!qkv_attention_fusable(n_tok, D, heads) || !stats || !ln_w || !ln_b) return MPL_E_INVALID;
    return launch_ng<MPL_EPI_BIAS, true, 3, 1, 1, 2, 0, true>(x, D, stats, ln_w, ln_b, W, bias, nullptr, 0, nullptr, 0, M,
                                                               3 * D, D, eps, nullptr, s, n_tok, D / heads, att);
}


// ------------------------------------------------------------------------------------------
// bf16 matrix-core variant (BASELINE.json configs[2], "CMU Panoptic ... bf16"): C = epi(LN(A) . W16^T + b) with
// v_mfma_f32_16x16x32_bf16 -- bf16 operands, fp32 accumulate; LayerNorm, bias, GELU, residual and the stored
// activations stay fp32.  A (fp32) is staged exactly as in the fp32 kernel and rounded to bf16 when the fragment is
// built (after LayerNorm); W16 is a bf16 copy of the nn.Linear weight ([N][K], derived data owned by the binding).
// One MFMA covers a whole 32-deep k-tile, so a k-tile costs a wave 9 MFMAs instead of 72: the kernel is bound by
// staging, not by the matrix pipe.  Stage (18 KiB): A 64 x 128 B (swizzled) | W 9 column tiles x 1 KiB
// ([16 rows][32 bf16], one DMA piece each, lane-linear = fragment order) | 1 KiB gamma, beta.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BF_SUB_B = NT * 1024;                 // 9216
constexpr int BF_GB = SUB_A + BF_SUB_B;             // 17408
constexpr int BF_STAGE = BF_GB + 1024;              // 18432
constexpr int BF_NST = 2;

template <int EPI, bool LN>
__global__ __launch_bounds__(256, 2) void ln_gemm_bf16_kernel(const float* __restrict__ A, int lda,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ ln_w,
                                                               const float* __restrict__ ln_b,
                                                               const unsigned short* __restrict__ W16,
                                                               const float* __restrict__ bias, const float* R, int ldr,
                                                               float* C, int ldc, int M, int N, int K, int grid_m,
                                                               int grid_n, float eps, float* stats_out) {
    __shared__ __attribute__((aligned(1024))) char smem[BF_NST * BF_STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    int tm, tn;
    {
        const int b = blockIdx.x;
        if ((grid_m & 7) == 0) {
            const int per = grid_m >> 3;
            const int xcd = b & 7, i = b >> 3;
            tm = xcd * per + (i % per);
            tn = i / per;
        } else {
            tm = b % grid_m;
            tn = b / grid_m;
        }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    float mu = 0.f, rs = 1.f;
    if (LN) {
        int m = m0 + wave * 16 + li;
        m = m < M ? m : M - 1;
        const int sl = (K % BN == 0) ? BN : K, ns = K / sl;
        ln_combine(stats + (size_t)m * ns * 2, ns, sl, K, eps, mu, rs);
        asm volatile("" : "+v"(mu), "+v"(rs));
    }
    if (!LN) {
        for (int i = tid; i < BF_NST * 256; i += 256) reinterpret_cast<float*>(smem + (i >> 8) * BF_STAGE + BF_GB)[i & 255] = 0.f;
        __syncthreads();
    }
    // DMA pieces per k-tile: 8 (A) + 9 (W16 column tiles) + 1 (gamma/beta).  wave w: A pieces w, w+4; W pieces w, w+4
    // (+ piece 8 on wave 0); gamma/beta on wave 3.
    unsigned voA[2], voW[3];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int r = (a * 4 + wave) * 8 + (lane >> 3);
        int m = m0 + r;
        m = m < M ? m : M - 1;
        voA[a] = (unsigned)(((size_t)m * lda + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * sizeof(float));
        asm volatile("" : "+v"(voA[a]));
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        int q = b * 4 + wave;
        q = q < NT ? q : NT - 1;
        int n = n0 + q * 16 + (lane >> 2);
        n = n < N ? n : N - 1;
        voW[b] = (unsigned)(((size_t)n * K + 8 * (lane & 3)) * sizeof(unsigned short));
        asm volatile("" : "+v"(voW[b]));
    }
    const float* gb_src = ((lane & 8) ? ln_b : ln_w) + 4 * (lane & 7);
    const bool w_extra = wave == 0, gb_on = LN && wave == 3;
    const int per = 2 + 2 + (w_extra ? 1 : 0) + (gb_on ? 1 : 0);   // pieces of this wave per k-tile
    const int T = K / BK;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const float* W16f = reinterpret_cast<const float*>(W16);   // dma16_fast only needs a byte address
    auto issue = [&](int t) {
        const unsigned keep = dma_m0_save();
        const unsigned st = lds0 + (unsigned)((t % BF_NST) * BF_STAGE);
        const int k0 = t * BK;
#pragma unroll
        for (int a = 0; a < 2; ++a) dma16_fast(voA[a], A + k0, st + (unsigned)((a * 4 + wave) * 1024));
#pragma unroll
        for (int b = 0; b < 2; ++b) dma16_fast(voW[b], W16f + k0 / 2, st + (unsigned)(SUB_A + (b * 4 + wave) * 1024));
        if (w_extra) dma16_fast(voW[2], W16f + k0 / 2, st + (unsigned)(SUB_A + 8 * 1024));
        if (gb_on) dma16(gb_src + k0, st + (unsigned)BF_GB);
        dma_m0_restore(keep);
    };
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rv[NT][4];
    const int t_res = T >= 2 ? T - 2 : 0;
    issue(0);
    const int swz = (li >> 1) & 7;
    for (int t = 0; t < T; ++t) {
        int allow = 0;                                      // 2-stage ring: only this stage's pieces are in flight
        if (EPI == MPL_EPI_BIAS_RESIDUAL && t > t_res) allow += RES_LOADS;
        wait_vm(allow);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < T) issue(t + 1);
        if (EPI == MPL_EPI_BIAS_RESIDUAL && t == t_res) load_residual(rv, R, ldr, M, N, m0 + wave * 16 + 4 * kq, n0, li);
        const char* st = smem + (t % BF_NST) * BF_STAGE;
        // A fragment of 16x16x32: lane (i, kq) holds A[i][8 kq .. 8 kq + 7] = logical 16-B columns 2kq, 2kq+1
        const float* as = reinterpret_cast<const float*>(st) + (wave * 16 + li) * BK;
        float4 a0 = ld4(as + (((2 * kq) ^ swz) << 2)), a1 = ld4(as + (((2 * kq + 1) ^ swz) << 2));
        bf16x8 bfrag[NT];
        const bf16x8* bs = reinterpret_cast<const bf16x8*>(st + SUB_A) + (li * 4 + kq);
#pragma unroll
        for (int n = 0; n < NT; ++n) bfrag[n] = bs[n * 64];
        if (LN) {
            const float* gb = reinterpret_cast<const float*>(st + BF_GB);
            const float4 g0 = ld4(gb + 8 * kq), g1 = ld4(gb + 8 * kq + 4), e0 = ld4(gb + 32 + 8 * kq), e1 = ld4(gb + 36 + 8 * kq);
            a0.x = (a0.x - mu) * rs * g0.x + e0.x; a0.y = (a0.y - mu) * rs * g0.y + e0.y;
            a0.z = (a0.z - mu) * rs * g0.z + e0.z; a0.w = (a0.w - mu) * rs * g0.w + e0.w;
            a1.x = (a1.x - mu) * rs * g1.x + e1.x; a1.y = (a1.y - mu) * rs * g1.y + e1.y;
            a1.z = (a1.z - mu) * rs * g1.z + e1.z; a1.w = (a1.w - mu) * rs * g1.w + e1.w;
        }
        bf16x8 af;
        af[0] = (__bf16)a0.x; af[1] = (__bf16)a0.y; af[2] = (__bf16)a0.z; af[3] = (__bf16)a0.w;
        af[4] = (__bf16)a1.x; af[5] = (__bf16)a1.y; af[6] = (__bf16)a1.z; af[7] = (__bf16)a1.w;
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[n], acc[n], 0, 0, 0);
    }
    store_tile_epilogue<EPI>(acc, bias, rv, C, ldc, M, N, m0 + wave * 16 + 4 * kq, n0, li, stats_out, N / BN);
}

template <int EPI, bool LN>
static int launch_bf16(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b,
                       const unsigned short* W16, const float* bias, const float* R, int ldr, float* C, int ldc, int M,
                       int N, int K, float eps, float* stats_out, hipStream_t s) {
    const int gm = (M + BM - 1) / BM, gn = (N + BN - 1) / BN;
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((ln_gemm_bf16_kernel<EPI, LN>), dim3(gm * gn), dim3(256), 0, s, A, lda, stats, ln_w, ln_b, W16, bias,
                       R, ldr, C, ldc, M, N, K, gm, gn, eps, stats_out);
    return hip_check_launch();
}

// fp32 -> bf16 (round to nearest even) copy of a weight tensor: the derived operand of the bf16 kernels
__global__ __launch_bounds__(256) void convert_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (__bf16)src[i];
}

int launch_convert_bf16(const float* src, unsigned short* dst, size_t n, hipStream_t s) {
    if (!src || !dst || n == 0) return MPL_E_INVALID;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(convert_bf16_kernel, dim3(grid), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(dst), n);
    return hip_check_launch();
}

int launch_ln_gemm(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b, float eps,
                   const float* W, const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N, int K,
                   int epi, float* stats_out, hipStream_t s, const unsigned short* W16) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0 || (lda & 3)) return MPL_E_INVALID;
    const bool ln = ln_w != nullptr;
    if (ln && (!stats || !ln_b)) return MPL_E_INVALID;
    if (epi == MPL_EPI_BIAS_RESIDUAL && !R) return MPL_E_INVALID;
    if (W16) {   // bf16 matrix cores (operands rounded to bf16, fp32 accumulate)
        if (stats_out && (epi != MPL_EPI_BIAS_RESIDUAL || N % BN != 0)) return MPL_E_INVALID;
#define MPL_BF(E)                                                                                                     \
    case E:                                                                                                           \
        return ln ? launch_bf16<E, true>(A, lda, stats, ln_w, ln_b, W16, bias, R, ldr, C, ldc, M, N, K, eps, stats_out, s) \
                  : launch_bf16<E, false>(A, lda, stats, ln_w, ln_b, W16, bias, R, ldr, C, ldc, M, N, K, eps, stats_out, s);
        switch (epi) {
            MPL_BF(MPL_EPI_BIAS)
            MPL_BF(MPL_EPI_BIAS_GELU)
            MPL_BF(MPL_EPI_BIAS_RESIDUAL)
            default:
                return MPL_E_INVALID;
        }
#undef MPL_BF
    }
    static const bool timing = getenv("MPL_GEMM_ABL") && atoi(getenv("MPL_GEMM_ABL")) == 4;
    if (stats_out && !timing && (epi != MPL_EPI_BIAS_RESIDUAL || N % BN != 0)) return MPL_E_INVALID;
#define MPL_ARGS A, lda, stats, ln_w, ln_b, W, bias, R, ldr, C, ldc, M, N, K, eps, stats_out, s
#define MPL_GEMM_CASE(E)                                                                              \
    case E:                                                                                           \
        return ln ? launch_ng_auto<E, true>(MPL_ARGS) : launch_ng_auto<E, false>(MPL_ARGS);
    switch (epi) {
        MPL_GEMM_CASE(MPL_EPI_BIAS)
        MPL_GEMM_CASE(MPL_EPI_BIAS_GELU)
        MPL_GEMM_CASE(MPL_EPI_BIAS_RESIDUAL)
        default:
            return MPL_E_INVALID;
    }
#undef MPL_GEMM_CASE
#undef MPL_ARGS
}

}  // namespace mpl
