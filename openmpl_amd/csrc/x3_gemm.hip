// fp32 GEMM on the bf16 matrix cores by exact operand splitting ("fp32x3"):  C = epi( LN(A) . W^T + bias ).
//
// Same reference ops and tile geometry as ln_gemm.hip (Block.norm1 + Attention.qkv :55, Attention.proj :65,
// Block.norm2 + Mlp.fc1 + GELU :32-33, Mlp.fc2 :35 of MPL/lib/models/multiview_mpl.py), different arithmetic:
// on gfx950 the bf16 matrix pipe is 16x faster than the fp32 one (2.5 PFLOP/s vs 157 TFLOP/s dense), so each
// fp32 operand is written as the exact sum of three bf16 numbers
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)     (3 x 8 = 24 mantissa bits)
// and a product a.b is accumulated (fp32 accumulators, v_mfma_f32_16x16x32_bf16) from the six partial products
// whose weight is >= 2^-16:  lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi.  The three dropped ones (mid.lo,
// lo.mid, lo.lo) are <= 2^-24 relative -- below the rounding of an fp32 multiply; measured against an fp64
// product the 6-term sum is 70x MORE accurate than an fp32 GEMM (7e-9 vs 5e-7 max-scaled at K = 544), so the
// result is limited by the fp32 accumulation exactly like the native fp32 MFMA path.  6 bf16 MFMAs of 16x16x32
// replace 8 fp32 MFMAs of 16x16x4 at 1/16 of the cycles each: 2.7x less matrix-pipe time per k-tile.
//
// Operands: A stays fp32 in memory and LDS (LayerNorm is applied in fp32, then the fragment is split in registers:
// ~6 VALU ops per element with v_cvt_pk_bf16_f32); W is split ONCE by the binding (launch_split_bf16x3) into
// fragment order  W3[N/136 groups][K/32 k-tiles][9 column tiles][3 parts][64 lanes][8 bf16]  so that a k-tile of a
// 136-column group is 27 contiguous 1-KiB DMA pieces and a B fragment is one conflict-free ds_read_b128 at
// lane * 16.
//
// Workgroup = 64 rows x 136 columns, 8 waves: wave w owns row group w & 3 (16 rows) and column tiles 0..4
// (w < 4) or 5..8 (w >= 4) -- waves w and w + 4 share a SIMD, which therefore sees 9 tiles = 54 MFMAs per k-tile
// whatever the wave.  All 8 waves issue the LDS-DMA (4-6 pieces each per stage, counted vmcnt waits).  Stage =
// one k-tile of 32: A 64 x 128 B (16-B columns XOR-swizzled with key (row >> 1) & 5: conflict free for the
// two-b128-per-lane A fragment) | W 27 KiB | gamma, beta 1 KiB = 36 KiB; ring of NST stages (2: two workgroups
// per CU; 3: one; 4: the attention variant, whose epilogue needs 117 KiB).
// NPASS = 3 (fused LN1 + qkv + attention): the workgroup runs the k loop three times -- q, k, v slices of its 136
// channels, accumulators of finished passes parked in registers -- and then finishes Attention.forward :55-64
// exactly like ln_gemm_ng_kernel<ATT>.
// The k order of every output element is fixed (k-tiles ascending, six products in the order above), so results do
// not depend on the batch size or launch geometry.
#include <stdlib.h>

#include "gemm_common.hpp"

namespace mpl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_W = 27 * 1024;             // W bytes per k-tile of a 136-column group
constexpr int X3_GB = SUB_A + X3_W;         // gamma/beta piece
constexpr int X3_STAGE = X3_GB + 1024;      // 36864
constexpr int X3_T0 = 5;                    // column tiles of waves 0..3; waves 4..7 take the other NT - 5

// 8 fp32 -> hi / mid / lo bf16x8 (round to nearest even at every step; the residuals are exact in fp32)
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r = x[i] - (float)h;
        const __bf16 m = (__bf16)r;
        hi[i] = h;
        mid[i] = m;
        lo[i] = (__bf16)(r - (float)m);
    }
}

size_t x3_operand_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || N % BN || K % BK) return 0;
    return (size_t)(N / BN) * (K / BK) * X3_W;
}

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ W, int N, int K,
                                                            bf16x8* __restrict__ dst, size_t total) {
    const int KT = K / BK;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const int tile = (int)((idx >> 6) % NT);
        const int kt = (int)((idx / (64 * NT)) % KT);
        const int g = (int)(idx / ((size_t)64 * NT * KT));
        const int li = lane & 15, kq = lane >> 4;
        const int c = tile * 16 + li;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
        if (c < BN) {
            const float* src = W + (size_t)(g * BN + c) * K + kt * BK + 8 * kq;
            const float4 p = ld4(src), q = ld4(src + 4);
            x[0] = p.x; x[1] = p.y; x[2] = p.z; x[3] = p.w; x[4] = q.x; x[5] = q.y; x[6] = q.z; x[7] = q.w;
        }
        bf16x8 hi, mid, lo;
        split3(x, hi, mid, lo);
        bf16x8* o = dst + ((size_t)(g * KT + kt) * 27 + tile * 3) * 64 + lane;
        o[0] = hi;
        o[64] = mid;
        o[128] = lo;
    }
}

int launch_split_bf16x3(const float* W, int N, int K, unsigned short* dst, hipStream_t s) {
    if (!W || !dst || x3_operand_bytes(N, K) == 0) return MPL_E_INVALID;
    const size_t total = (size_t)(N / BN) * (K / BK) * NT * 64;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(grid), dim3(256), 0, s, W, N, K, reinterpret_cast<bf16x8*>(dst), total);
    return hip_check_launch();
}

struct X3Args {
    const float* A;
    int lda;
    const float* stats;
    const float* ln_w;
    const float* ln_b;
    const char* W3;
    const float* bias;
    const float* R;
    int ldr;
    float* C;
    int ldc;
    int M, N, K;
    int grid_m, grid_n;
    float eps;
    float* stats_out;
    int att_ntok, att_hd;
    float* att_out;
};

// Everything a compute wave does, for its NTW column tiles starting at tile `tile0`.
template <int EPI, bool LN, int NPASS, int NST, int NTW>
__device__ __forceinline__ void x3_body(const X3Args& a, char* smem, int tid, int wave, int tile0) {
    const int lane = tid & 63;
    const int rg = wave & 3;
    const int li = lane & 15, kq = lane >> 4;
    int tm, tn;
    {
        const int b = blockIdx.x;
        if ((a.grid_m & 7) == 0) {  // XCD-aware: blocks b, b+8, .. share an XCD/L2 -> give each XCD a band of m tiles
            const int per = a.grid_m >> 3;
            const int xcd = b & 7, i = b >> 3;
            tm = xcd * per + (i % per);
            tn = i / per;
        } else {
            tm = b % a.grid_m;
            tn = b / a.grid_m;
        }
    }
    const int M = a.M, N = a.N, K = a.K;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Dq = N / 3;                        // NPASS == 3: width of each of q, k, v
    const int KT = K / BK;
    const int T = NPASS * KT;                    // stages
    auto colbase = [&](int pass) -> int { return NPASS == 3 ? pass * Dq + n0 : n0; };

    float mu = 0.f, rs = 1.f;
    if (LN) {
        int m = m0 + rg * 16 + li;
        m = m < M ? m : M - 1;
        const int sl = (K % BN == 0) ? BN : K, ns = K / sl;
        ln_combine(a.stats + (size_t)m * ns * 2, ns, sl, K, a.eps, mu, rs);
        asm volatile("" : "+v"(mu), "+v"(rs));   // consume the loads before the k loop (see ln_gemm.hip)
    }

    // ---- DMA slots of this wave: A piece `wave` (8 rows), W pieces wave, wave + 8, wave + 16 (+ 24 + wave for
    // waves 0..2), gamma/beta from wave 7
    unsigned voA;
    {
        const int r = wave * 8 + (lane >> 3);
        int m = m0 + r;
        m = m < M ? m : M - 1;
        voA = (unsigned)(((size_t)m * a.lda + 4 * ((lane & 7) ^ ((r >> 1) & 5))) * sizeof(float));
        asm volatile("" : "+v"(voA));
    }
    unsigned voW = (unsigned)(lane * 16);
    asm volatile("" : "+v"(voW));
    const float* gb_src = ((lane & 8) ? a.ln_b : a.ln_w) + 4 * (lane & 7);
    const bool w_extra = wave < 3;
    const bool gb_on = LN && wave == 7;
    const int per_st = 4 + (w_extra ? 1 : 0) + (gb_on ? 1 : 0);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue_stage = [&](int t) {
        const int pass = NPASS == 1 ? 0 : t / KT;
        const int kt = t - pass * KT;
        const int grp = colbase(pass) / BN;
        const char* wsrc = a.W3 + ((size_t)grp * KT + kt) * X3_W;
        const unsigned st = lds0 + (unsigned)((t % NST) * X3_STAGE);
        const unsigned keep = dma_m0_save();
        dma16_fast(voA, a.A + kt * BK, st + (unsigned)(wave * 1024));
#pragma unroll
        for (int j = 0; j < 3; ++j)
            dma16_fast(voW, reinterpret_cast<const float*>(wsrc + (wave + 8 * j) * 1024),
                       st + (unsigned)(SUB_A + (wave + 8 * j) * 1024));
        if (w_extra) dma16_fast(voW, reinterpret_cast<const float*>(wsrc + (24 + wave) * 1024), st + (unsigned)(SUB_A + (24 + wave) * 1024));
        if (gb_on) dma16(gb_src + kt * BK, st + (unsigned)X3_GB);
        dma_m0_restore(keep);
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < T) issue_stage(t);

    f32x4 acc[NPASS][NTW];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rv[NTW][4];
    const int row0 = m0 + rg * 16 + 4 * kq;
    const int t_res = T - 2;
    const int key = (li >> 1) & 5;
    const int RES = NTW * 4;

#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        for (int kt = 0; kt < KT; ++kt) {
            const int t = pass * KT + kt;
            int ahead = T - 1 - t;
            ahead = ahead < NST - 2 ? ahead : NST - 2;
            int allow = ahead * per_st;
            if (EPI == MPL_EPI_BIAS_RESIDUAL && t > t_res) allow += RES;   // younger than every DMA piece
            wait_vm(allow);
            __builtin_amdgcn_s_barrier();       // everyone's pieces of stage t landed; everyone is done with t-1
            asm volatile("" ::: "memory");
            if (t + NST - 1 < T) issue_stage(t + NST - 1);
            if (EPI == MPL_EPI_BIAS_RESIDUAL && t == t_res)
                load_residual_w<NTW>(rv, a.R, a.ldr, M, N, row0, n0 + tile0 * 16, li);

            const char* st = smem + (t % NST) * X3_STAGE;
            // A fragment of 16x16x32: lane (i, kq) holds A[i][8 kq .. 8 kq + 7] = logical 16-B columns 2kq, 2kq+1
            const float* as = reinterpret_cast<const float*>(st) + (rg * 16 + li) * BK;
            const float4 a0 = ld4(as + (((2 * kq) ^ key) << 2)), a1 = ld4(as + (((2 * kq + 1) ^ key) << 2));
            float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            if (LN) {
                const float* gb = reinterpret_cast<const float*>(st + X3_GB);
                const float4 g0 = ld4(gb + 8 * kq), g1 = ld4(gb + 8 * kq + 4), e0 = ld4(gb + 32 + 8 * kq), e1 = ld4(gb + 36 + 8 * kq);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = (x[i] - mu) * rs * gg[i] + ee[i];
            }
            bf16x8 ah, am, al;
            split3(x, ah, am, al);
            const bf16x8* bs = reinterpret_cast<const bf16x8*>(st + SUB_A) + tile0 * 3 * 64 + lane;
            // B fragments in two batches (3 tiles, then the rest): 36 instead of 60 live registers, so that the
            // residual variants also fit the 128-register budget of two workgroups per CU
#define MPL_X3(AP, BP, N0, N1)                       \
    _Pragma("unroll") for (int n = N0; n < N1; ++n) \
        acc[pass][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AP, BP[n - N0], acc[pass][n], 0, 0, 0);
#define MPL_X3_BATCH(N0, N1)                                  \
    {                                                         \
        bf16x8 bh[3], bm[3], bl[3];                           \
        _Pragma("unroll") for (int n = N0; n < N1; ++n) {     \
            bh[n - N0] = bs[(n * 3 + 0) * 64];                \
            bm[n - N0] = bs[(n * 3 + 1) * 64];                \
            bl[n - N0] = bs[(n * 3 + 2) * 64];                \
        }                                                     \
        MPL_X3(al, bh, N0, N1)                                \
        MPL_X3(ah, bl, N0, N1)                                \
        MPL_X3(am, bm, N0, N1)                                \
        MPL_X3(am, bh, N0, N1)                                \
        MPL_X3(ah, bm, N0, N1)                                \
        MPL_X3(ah, bh, N0, N1)                                \
    }
            MPL_X3_BATCH(0, 3)
            MPL_X3_BATCH(3, NTW)
#undef MPL_X3_BATCH
#undef MPL_X3
        }
    }

    if (NPASS == 3) {
        // ---- fused attention epilogue.  T[64][412]: q | k | v (+bias) of this workgroup's 136 channels.
        float* Tt = reinterpret_cast<float*>(smem);
        float* SC = Tt + BM * ATT_TS;
        __syncthreads();                        // every wave is done reading the last stage
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int cb = colbase(p);
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int c = (tile0 + n) * 16 + li;
                if (c < BN) {
                    const float bv = a.bias[cb + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tt[(rg * 16 + 4 * kq + r) * ATT_TS + p * BN + c] = acc[p][n][r] + bv;
                }
            }
        }
        attention_on_tile(Tt, SC, tid, 512, a.att_ntok, a.att_hd, a.att_out, m0, n0, M, Dq);
        return;
    }

    float v[NTW][4];
    tile_values_store<EPI, NTW>(acc[0], a.bias, rv, a.C, a.ldc, M, N, row0, n0, n0 + tile0 * 16, li, v);
    if (EPI == MPL_EPI_BIAS_RESIDUAL && a.stats_out) {
        // LayerNorm partials of the 136-column slice: the second half hands its final values to the first through
        // the stage slot nobody reads any more (stage T-2: every wave passed barrier T-1), and the first half
        // reduces all 9 tiles in the order of the fp32 kernels' epilogue
        constexpr int NT1 = NT - X3_T0;
        float4* xfer = reinterpret_cast<float4*>(smem + ((T - 2) % NST) * X3_STAGE);
        if (tile0) {
#pragma unroll
            for (int n = 0; n < NTW; ++n) xfer[(rg * NT1 + n) * 64 + lane] = float4{v[n][0], v[n][1], v[n][2], v[n][3]};
        }
        __syncthreads();
        if (!tile0) {
            float vv[NT][4];
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[n][r] = v[n][r];
#pragma unroll
            for (int n = 0; n < NT1; ++n) {
                const float4 q = xfer[(rg * NT1 + n) * 64 + lane];
                vv[X3_T0 + n][0] = q.x; vv[X3_T0 + n][1] = q.y; vv[X3_T0 + n][2] = q.z; vv[X3_T0 + n][3] = q.w;
            }
            slice_stats_store(vv, a.stats_out, N / BN, M, row0, n0, li);
        }
    }
}

template <int EPI, bool LN, int NPASS, int NST>
__global__ __launch_bounds__(512, (NPASS == 1 && EPI != MPL_EPI_BIAS_RESIDUAL) ? 4 : 2) void x3_gemm_kernel(const X3Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < 4) x3_body<EPI, LN, NPASS, NST, X3_T0>(a, smem, tid, wave, 0);
    else x3_body<EPI, LN, NPASS, NST, NT - X3_T0>(a, smem, tid, wave, X3_T0);
}

template <int EPI, bool LN, int NPASS, int NST>
static int launch_x3(const X3Args& a, hipStream_t s) {
    constexpr int LDS = NST * X3_STAGE;
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    static_assert(NPASS == 1 || (BM * ATT_TS + ATT_SCORE_FLOATS) * 4 <= LDS, "attention epilogue does not fit in the ring");
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)x3_gemm_kernel<EPI, LN, NPASS, NST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev] = true;
    }
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((x3_gemm_kernel<EPI, LN, NPASS, NST>), dim3(a.grid_m * a.grid_n), dim3(512), LDS, s, a);
    return hip_check_launch();
}

template <int EPI, bool LN>
static int launch_x3_auto(const X3Args& a, hipStream_t s) {
    // two workgroups per CU (2-stage rings) once there are more workgroups than CUs, else one with a deeper ring
    static const int force = getenv("MPL_X3_NST") ? atoi(getenv("MPL_X3_NST")) : 0;   // bench-only
    const int wgs = a.grid_m * a.grid_n;
    int nst = wgs > 256 ? 2 : 3;
    if (force) nst = force;
    switch (nst) {
        case 2: return launch_x3<EPI, LN, 1, 2>(a, s);
        case 4: return launch_x3<EPI, LN, 1, 4>(a, s);
        default: return launch_x3<EPI, LN, 1, 3>(a, s);
    }
}

int launch_x3_gemm(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b, float eps,
                   const unsigned short* W3, const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N,
                   int K, int epi, float* stats_out, hipStream_t s) {
    if (M <= 0 || !A || !W3 || !bias || !C || (lda & 3) || x3_operand_bytes(N, K) == 0 || K < 2 * BK) return MPL_E_INVALID;
    const bool ln = ln_w != nullptr;
    if (ln && (!stats || !ln_b)) return MPL_E_INVALID;
    if (epi == MPL_EPI_BIAS_RESIDUAL && !R) return MPL_E_INVALID;
    if (stats_out && epi != MPL_EPI_BIAS_RESIDUAL) return MPL_E_INVALID;
    X3Args a{A, lda, stats, ln_w, ln_b, reinterpret_cast<const char*>(W3), bias, R, ldr, C, ldc, M, N, K,
             (M + BM - 1) / BM, N / BN, eps, stats_out, 0, 0, nullptr};
#define MPL_X3_CASE(E) \
    case E:            \
        return ln ? launch_x3_auto<E, true>(a, s) : launch_x3_auto<E, false>(a, s);
    switch (epi) {
        MPL_X3_CASE(MPL_EPI_BIAS)
        MPL_X3_CASE(MPL_EPI_BIAS_GELU)
        MPL_X3_CASE(MPL_EPI_BIAS_RESIDUAL)
        default:
            return MPL_E_INVALID;
    }
#undef MPL_X3_CASE
}

// LN1 + qkv projection + softmax attention in one launch (see launch_ln_qkv_attention): att[M, D] from x[M, D]
int launch_x3_qkv_attention(const float* x, int M, int D, const float* stats, const float* ln_w, const float* ln_b,
                            float eps, const unsigned short* W3, const float* bias, int n_tok, int heads, float* att,
                            hipStream_t s) {
    if (!qkv_attention_fusable(n_tok, D, heads) || !stats || !ln_w || !ln_b || !W3 || !bias || !att || M <= 0 ||
        x3_operand_bytes(3 * D, D) == 0)
        return MPL_E_INVALID;
    X3Args a{x, D, stats, ln_w, ln_b, reinterpret_cast<const char*>(W3), bias, nullptr, 0, nullptr, 0, M, 3 * D, D,
             (M + BM - 1) / BM, D / BN, eps, nullptr, n_tok, D / heads, att};
    return launch_x3<MPL_EPI_BIAS, true, 3, 4>(a, s);
}

}  // namespace mpl
