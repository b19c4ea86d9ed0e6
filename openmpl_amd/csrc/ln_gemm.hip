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

#include "gemm_common.hpp"


namespace mpl {

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

// narrow rows (K = 32: the joints x views grid, 135 k rows at V = 31, B = 256): 8 lanes per row, one float4 each, so a
// wave covers 8 rows with one coalesced 1-KiB read instead of one row with half its lanes idle; same two-pass
// arithmetic, reductions over the 8 lanes of a row
__global__ __launch_bounds__(256) void row_stats32_kernel(const float* __restrict__ x, int M, int ldx, float* __restrict__ stats) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int row = t >> 3, c = t & 7;
    const bool ok = row < M;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = ld4(x + (size_t)row * ldx + 4 * c);
    float s = (v.x + v.y) + (v.z + v.w);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    const float mean = s / 32.0f;
    const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
    float ss = (a * a + b * b) + (cc * cc + d * d);
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    if (ok && c == 0) {
        stats[(size_t)row * 2] = mean;
        stats[(size_t)row * 2 + 1] = ss;
    }
}

int launch_row_stats(const float* x, int M, int K, int ldx, float* stats, hipStream_t s) {
    if (M <= 0 || (K & 3)) return MPL_E_INVALID;
    ProfScope prof(MPL_K_ROW_STATS, s);
    if (K == 32 && (ldx & 3) == 0)
        hipLaunchKernelGGL(row_stats32_kernel, dim3((int)(((size_t)M * 8 + 255) / 256)), dim3(256), 0, s, x, M, ldx, stats);
    else
        hipLaunchKernelGGL(row_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, M, K, ldx, ln_slice_len(K), stats);
    return hip_check_launch();
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
// PF (with LDW, NST = 3, KS = KG = 1): the loader keeps TWO stages landed at every barrier, so a compute wave reads
// the first fragment of stage t+1 under the last MFMA block of stage t: no ds_read latency behind the barrier.
template <int EPI, bool LN, int NG, int KG, int KS, int NST, int ABL = 0, bool ATT = false, bool LDW = false, bool PF = false>
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
    static_assert(!PF || (LDW && NST == 3 && KS == 1 && KG == 1), "fragment prefetch needs the loader wave and a 3-stage ring");
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
    // row base of the tile in the wave-uniform 64-bit DMA base: the per-lane 32-bit offsets are tile relative, so
    // operands beyond 4 GiB are addressed correctly
    const float* At = A + (size_t)m0 * lda;
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
        voA[a] = (unsigned)(((size_t)(m - m0) * lda + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * sizeof(float));
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
                    if (a < A_PER - 1 || a_on) dma16_fast(voA[a], At + k0, st + (unsigned)((a * NW + wave) * 1024));
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
            lA[p] = (unsigned)(((size_t)(m - m0) * lda + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * sizeof(float));
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
                    for (int p = 0; p < 8; ++p) dma16_fast(lA[p], At + k0, st + (unsigned)(p * 1024));
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
            if (PF) allow = 0;                              // stages t and t+1 are both landed at barrier t
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
    auto mma_step = [&](Frag& c) {              // LayerNorm the A fragment in registers, then 4 x NT MFMAs
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
    };
    Frag pf[2];                                  // PF: fragment registers that live across the stage barrier
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
        if (PF) {
            // two 16-deep k steps per stage; pf[0] of stage t was read under the last MFMA block of stage t-1
            if (t == 0) load_frag<LN, NG>(pf[0], st, 0, rg, cg, li, kq, swz);
            load_frag<LN, NG>(pf[1], st, 1, rg, cg, li, kq, swz);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(pf[0]);
            __builtin_amdgcn_sched_barrier(0);
            // unconditional (the last one reads a stale slot and is dropped): a branch here makes hipcc merge the
            // two lgkmcnt states conservatively and wait for THESE reads before the MFMAs of pf[1] below
            load_frag<LN, NG>(pf[0], smem + ((t + 1) % NST) * STAGE, 0, rg, cg, li, kq, swz);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(pf[1]);
            __builtin_amdgcn_sched_barrier(0);
        } else {
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
                mma_step(f[j & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
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
        constexpr int TS = ATT_TS;
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
        attention_on_tile(T, SC, tid, 64 * NW, att_ntok, att_hd, att_out, m0, n0, M, Dq);
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

template <int EPI, bool LN, int NG, int KG, int KS, int NST, int ABL = 0, bool ATT = false, bool LDW = false, bool PF = false>
static int launch_ng(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b, const float* W,
                     const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N, int K, float eps,
                     float* stats_out, hipStream_t s, int att_ntok = 0, int att_hd = 0, float* att_out = nullptr) {
    constexpr int LDS = NST * KS * SubStage<NG>::BYTES;
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    static_assert((KG - 1) * NG * 4 * NT * 4 * 64 * 4 <= LDS, "k-group reduction does not fit in the ring");
    const int gm = (M + BM - 1) / BM, gn = (N + BN * NG - 1) / (BN * NG);
    static_assert(!ATT || (BM * (3 * BN + 4) + ATT_SCORE_FLOATS) * 4 <= LDS, "attention epilogue does not fit in the ring");
    static std::atomic<bool> attr_set[64];   // set-once flags: a racing second hipFuncSetAttribute is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)ln_gemm_ng_kernel<EPI, LN, NG, KG, KS, NST, ABL, ATT, LDW, PF>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((ln_gemm_ng_kernel<EPI, LN, NG, KG, KS, NST, ABL, ATT, LDW, PF>), dim3(gm * gn),
                       dim3(256 * NG * KG + (LDW ? 64 : 0)), LDS, s,
                       A, lda, stats, ln_w, ln_b, W, bias, R, ldr, C, ldc, M, N, K, gm, gn, eps, stats_out, att_ntok,
                       att_hd, att_out);
    return hip_check_launch();
}

// ------------------------------------------------------------------------------------------
// Column-split variant for launches with at most ONE workgroup per CU (N = D at B V <= 4096 rows: proj, fc2).
// A 4-wave workgroup alone on a CU runs one wave per SIMD, and a single wave cannot overlap its own fragment reads
// with its MFMAs (tools/loop_probe.hip: 105 TFLOP/s with one wave per SIMD against 125-139 with two or three).
// Here the same 64 x 136 tile is computed by EIGHT waves: wave w and wave w + 4 (same SIMD) share row group w & 3
// and take column tiles 0..4 and 5..8 of the 9, so every SIMD keeps the same 72 MFMAs per k-tile but from two
// independent instruction streams.  A ninth wave issues all DMA (see LDW above).  The k order of every output
// element is unchanged, and the epilogue applies the 4-wave kernel's own operations in its order (the second half hands
// its final values over through LDS for the LayerNorm partials), so results are bitwise identical to ln_gemm_ng_kernel -- batch-size invariance is preserved.
template <int NTW>
struct FragW {
    float4 a, g, be;
    float4 b[NTW];
};
template <bool LN, int NTW>
__device__ __forceinline__ void load_frag_w(FragW<NTW>& f, const char* sb, int half, int rg, int tile0, int li, int kq, int swz) {
    typedef SubStage<1> SS;
    const int cl = (half << 2) + kq;
    const int cc = (cl ^ swz) << 2;
    const float* as = reinterpret_cast<const float*>(sb) + (rg * 16 + li) * BK;
    const float* bs = reinterpret_cast<const float*>(sb + SUB_A) + (tile0 * 16 + li) * BK;
    f.a = ld4(as + cc);
#pragma unroll
    for (int n = 0; n < NTW; ++n) f.b[n] = ld4(bs + n * 16 * BK + cc);
    if (LN) {
        const float* gb = reinterpret_cast<const float*>(sb + SS::GB);
        f.g = ld4(gb + 4 * cl);
        f.be = ld4(gb + 32 + 4 * cl);
    }
}
template <bool LN, int NTW>
__device__ __forceinline__ void mma_step_w(FragW<NTW>& c, f32x4 (&acc)[NTW], float mu, float rs) {
    if (LN) {
        c.a.x = (c.a.x - mu) * rs * c.g.x + c.be.x;
        c.a.y = (c.a.y - mu) * rs * c.g.y + c.be.y;
        c.a.z = (c.a.z - mu) * rs * c.g.z + c.be.z;
        c.a.w = (c.a.w - mu) * rs * c.g.w + c.be.w;
    }
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = mfma16(c.a.x, c.b[n].x, acc[n]);
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = mfma16(c.a.y, c.b[n].y, acc[n]);
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = mfma16(c.a.z, c.b[n].z, acc[n]);
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = mfma16(c.a.w, c.b[n].w, acc[n]);
}
// k loop of one compute wave over its NTW column tiles starting at tile0; the residual loads of those tiles are
// issued two stages before the epilogue
template <int EPI, bool LN, int NST, int NTW>
__device__ __forceinline__ void cs_k_loop(f32x4 (&acc)[NTW], float (&rv)[NTW][4], const char* smem, int T, int rg, int tile0,
                                          int li, int kq, float mu, float rs, const float* R, int ldr, int M, int N,
                                          int row0, int n0) {
    constexpr int STAGE = SubStage<1>::BYTES;
    const int swz = (li >> 1) & 7;
    const int t_res = T >= 2 ? T - 2 : 0;
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_barrier();           // the loader saw stage t land; everyone is done reading stage t-1
        asm volatile("" ::: "memory");
        if (EPI == MPL_EPI_BIAS_RESIDUAL && t == t_res) load_residual_w<NTW>(rv, R, ldr, M, N, row0, n0 + tile0 * 16, li);
        const char* st = smem + (t % NST) * STAGE;
        FragW<NTW> f0, f1;
        load_frag_w<LN, NTW>(f0, st, 0, rg, tile0, li, kq, swz);
        load_frag_w<LN, NTW>(f1, st, 1, rg, tile0, li, kq, swz);
        __builtin_amdgcn_sched_barrier(0);
        mma_step_w<LN, NTW>(f0, acc, mu, rs);
        __builtin_amdgcn_sched_barrier(0);
        mma_step_w<LN, NTW>(f1, acc, mu, rs);
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr int CS_T0 = 5;                         // column tiles of the first half (waves 0..3); the rest go to waves 4..7
constexpr int CS_XFER = 4 * (NT - CS_T0) * 64 * 16;   // accumulator hand-over area behind the ring (16 KiB)

template <int EPI, bool LN, int NST>
__global__ __launch_bounds__(576, 1) void ln_gemm_cs_kernel(const float* __restrict__ A, int lda,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ ln_w,
                                                             const float* __restrict__ ln_b,
                                                             const float* __restrict__ W, const float* __restrict__ bias,
                                                             const float* R, int ldr, float* C, int ldc, int M, int N,
                                                             int K, int grid_m, int grid_n, float eps, float* stats_out) {
    typedef SubStage<1> SS;
    constexpr int STAGE = SS::BYTES;
    constexpr int NTHREADS = 576;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7 compute, 8 loader
    const int rg = wave & 3, half = (wave >> 2) & 1;
    const int li = lane & 15, kq = lane >> 4;
    int tm, tn;
    {
        const int b = blockIdx.x;
        if ((grid_m & 7) == 0) {  // XCD-aware, as in ln_gemm_ng_kernel
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
    const float* At = A + (size_t)m0 * lda;   // tile row base in the 64-bit DMA base (per-lane offsets tile relative)
    float mu = 0.f, rs = 1.f;
    if (LN && wave < 8) {
        int m = m0 + rg * 16 + li;
        m = m < M ? m : M - 1;
        const int sl = (K % BN == 0) ? BN : K, ns = K / sl;
        ln_combine(stats + (size_t)m * ns * 2, ns, sl, K, eps, mu, rs);
        asm volatile("" : "+v"(mu), "+v"(rs));
    }
    if (!LN) {
        for (int i = tid; i < NST * 256; i += NTHREADS)
            reinterpret_cast<float*>(smem + (i >> 8) * STAGE + SS::GB)[i & 255] = 0.f;
        __syncthreads();
    }
    const int T = K / BK;
    if (wave == 8) {
        // ---------------- loader wave: 8 A pieces + 17 W pieces (+ gamma/beta) per k-tile
        unsigned lA[8], lW[17];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = p * 8 + (lane >> 3);
            int m = m0 + r;
            m = m < M ? m : M - 1;
            lA[p] = (unsigned)(((size_t)(m - m0) * lda + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * sizeof(float));
            asm volatile("" : "+v"(lA[p]));
        }
#pragma unroll
        for (int p = 0; p < 17; ++p) {
            const int r = p * 8 + (lane >> 3);
            int n = n0 + r;
            n = n < N ? n : N - 1;
            lW[p] = (unsigned)(((size_t)n * K + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * sizeof(float));
            asm volatile("" : "+v"(lW[p]));
        }
        const float* gb_src = ((lane & 8) ? ln_b : ln_w) + 4 * (lane & 7);
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        constexpr int PIECES = 8 + 17 + (LN ? 1 : 0);
        auto l_issue = [&](int t) {
            const unsigned keep = dma_m0_save();
            const unsigned st = lds0 + (unsigned)((t % NST) * STAGE);
            const int k0 = t * BK;
#pragma unroll
            for (int p = 0; p < 8; ++p) dma16_fast(lA[p], At + k0, st + (unsigned)(p * 1024));
#pragma unroll
            for (int p = 0; p < 17; ++p) dma16_fast(lW[p], W + k0, st + (unsigned)((8 + p) * 1024));
            if (LN) dma16(gb_src + k0, st + (unsigned)SS::GB);
            dma_m0_restore(keep);
        };
#pragma unroll
        for (int t = 0; t < NST - 1; ++t)
            if (t < T) l_issue(t);
        for (int t = 0; t < T; ++t) {
            int allow = 0;
#pragma unroll
            for (int j = 1; j <= NST - 2; ++j)
                if (t + j < T) allow += PIECES;
            wait_vm(allow);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + NST - 1 < T) l_issue(t + NST - 1);
        }
        __builtin_amdgcn_s_barrier();            // matches the accumulator hand-over barrier below
        return;
    }

    const int row0 = m0 + rg * 16 + 4 * kq;
    float4* xfer = reinterpret_cast<float4*>(smem + NST * STAGE);
    constexpr int NT1 = NT - CS_T0;
    const bool want_stats = EPI == MPL_EPI_BIAS_RESIDUAL && stats_out != nullptr;
    if (half) {
        f32x4 acc[NT1];
#pragma unroll
        for (int n = 0; n < NT1; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        float rv[NT1][4], v[NT1][4];
        cs_k_loop<EPI, LN, NST, NT1>(acc, rv, smem, T, rg, CS_T0, li, kq, mu, rs, R, ldr, M, N, row0, n0);
        tile_values_store<EPI, NT1>(acc, bias, rv, C, ldc, M, N, row0, n0, n0 + CS_T0 * 16, li, v);
        if (want_stats) {
#pragma unroll
            for (int n = 0; n < NT1; ++n) xfer[(rg * NT1 + n) * 64 + lane] = float4{v[n][0], v[n][1], v[n][2], v[n][3]};
        }
        __syncthreads();
        return;
    }
    f32x4 acc[CS_T0];
#pragma unroll
    for (int n = 0; n < CS_T0; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rv[CS_T0][4], v0[CS_T0][4];
    cs_k_loop<EPI, LN, NST, CS_T0>(acc, rv, smem, T, rg, 0, li, kq, mu, rs, R, ldr, M, N, row0, n0);
    tile_values_store<EPI, CS_T0>(acc, bias, rv, C, ldc, M, N, row0, n0, n0, li, v0);
    __syncthreads();
    if (want_stats) {   // the other half's final values: the slice statistics are reduced in the 4-wave kernel's order
        float v[NT][4];
#pragma unroll
        for (int n = 0; n < CS_T0; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[n][r] = v0[n][r];
#pragma unroll
        for (int n = 0; n < NT1; ++n) {
            const float4 x = xfer[(rg * NT1 + n) * 64 + lane];
            v[CS_T0 + n][0] = x.x; v[CS_T0 + n][1] = x.y; v[CS_T0 + n][2] = x.z; v[CS_T0 + n][3] = x.w;
        }
        slice_stats_store(v, stats_out, N / BN, M, row0, n0, li);
    }
}

template <int EPI, bool LN, int NST>
static int launch_cs(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b, const float* W,
                     const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N, int K, float eps,
                     float* stats_out, hipStream_t s) {
    constexpr int LDS = NST * SubStage<1>::BYTES + CS_XFER;
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    const int gm = (M + BM - 1) / BM, gn = (N + BN - 1) / BN;
    static std::atomic<bool> attr_set[64];   // set-once flags: a racing second hipFuncSetAttribute is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)ln_gemm_cs_kernel<EPI, LN, NST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((ln_gemm_cs_kernel<EPI, LN, NST>), dim3(gm * gn), dim3(576), LDS, s, A, lda, stats, ln_w, ln_b, W,
                       bias, R, ldr, C, ldc, M, N, K, gm, gn, eps, stats_out);
    return hip_check_launch();
}

template <int EPI, bool LN>
static int launch_ng_auto(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b,
                          const float* W, const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N,
                          int K, float eps, float* stats_out, hipStream_t s) {
    // widest column grouping that still yields >= 256 workgroups (one per CU); small problems stay at NG = 1
    const int gm = (M + BM - 1) / BM;
    static const int force = lab_getenv("MPL_GEMM_NG") ? atoi(lab_getenv("MPL_GEMM_NG")) : 0;   // bench-only
    static const int kgsel = lab_getenv("MPL_GEMM_KG") ? atoi(lab_getenv("MPL_GEMM_KG")) : 0;    // bench-only
    int ng = 1;
    if (N % (BN * 3) == 0 && gm * (N / (BN * 3)) >= 256) ng = 3;
    if (force) ng = force;
#define MPL_ARGS2 A, lda, stats, ln_w, ln_b, W, bias, R, ldr, C, ldc, M, N, K, eps, stats_out, s
    static const int abl = lab_getenv("MPL_GEMM_ABL") ? atoi(lab_getenv("MPL_GEMM_ABL")) : 0;   // bench-only ablations
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
    static const int cfg = lab_getenv("MPL_GEMM_CFG") ? atoi(lab_getenv("MPL_GEMM_CFG")) : 0;
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
        case 7112: return launch_cs<EPI, LN, 2>(MPL_ARGS2);                              // column-split variants
        case 7113: return launch_cs<EPI, LN, 3>(MPL_ARGS2);
        case 8113: return launch_ng<EPI, LN, 1, 1, 1, 3, 0, false, true, true>(MPL_ARGS2);   // + fragment prefetch
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
    // One workgroup per CU (or two of the register-light non-residual variants): the 8-wave column-split kernel
    // with its loader wave.  Measured on MI355X at M = 4096, D = 544 against the best 4-wave configuration:
    // proj 29.1 -> 27.1 us, fc2 52.2 -> 48.7 us, fc1 (512 workgroups, two per CU) 56.0 -> 52.0 us; with more
    // workgroups per CU the 4-wave kernels below win (D = 1088: fc1 186 vs 210 us).  Bitwise identical results.
    static const int nocs = lab_getenv("MPL_GEMM_NOCS") ? atoi(lab_getenv("MPL_GEMM_NOCS")) : 0;   // bench-only A/B switch
    if (!(nocs & 1) && wgs <= 256) return launch_cs<EPI, LN, 2>(MPL_ARGS2);
    if (!(nocs & 2) && EPI != MPL_EPI_BIAS_RESIDUAL && wgs <= 512) return launch_cs<EPI, LN, 2>(MPL_ARGS2);
    if (wgs <= 256) return launch_ng<EPI, LN, 1, 1, 1, 2, 0, false, true>(MPL_ARGS2);
    if (cost(3) <= cost(2)) return launch_ng<EPI, LN, 1, 1, 1, 2>(MPL_ARGS2);
    return launch_ng<EPI, LN, 1, 1, 1, 3>(MPL_ARGS2);
#undef MPL_ARGS2
}

// LN1 + qkv projection + softmax attention in one launch: att[M, D] from x[M, D].  Requirements (else the caller
// uses the separate kernels): 136 % hd == 0, 64 % n_tok == 0, D % 136 == 0.
bool qkv_attention_fusable(int n_tok, int dim, int heads) {
    static const bool off = lab_getenv("MPL_NO_ATT_FUSION") != nullptr;   // bench-only A/B switch
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


int launch_ln_gemm(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b, float eps,
                   const float* W, const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N, int K,
                   int epi, float* stats_out, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0 || (lda & 3)) return MPL_E_INVALID;
    const bool ln = ln_w != nullptr;
    if (ln && (!stats || !ln_b)) return MPL_E_INVALID;
    if (epi == MPL_EPI_BIAS_RESIDUAL && !R) return MPL_E_INVALID;
    static const bool timing = lab_getenv("MPL_GEMM_ABL") && atoi(lab_getenv("MPL_GEMM_ABL")) == 4;
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
