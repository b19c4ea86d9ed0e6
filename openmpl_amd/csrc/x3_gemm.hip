// fp32 GEMM on the bf16 matrix cores by exact operand splitting ("fp32x3"), engine v2:
//     C = epi( LN(A) . W^T + bias )      with BOTH operands pre-split and in MFMA fragment order.
//
// Reference ops (MPL/lib/models/multiview_mpl.py): Block.norm1 + Attention.qkv :55 + Attention.forward :55-64,
// Attention.proj :65 + residual :90, Block.norm2 + Mlp.fc1 + GELU :32-33, Mlp.fc2 :35 + residual :91.
//
// Arithmetic (unchanged from v1).  On gfx950 the bf16 matrix pipe is 16x faster than the fp32 one, so each fp32
// operand is the exact sum of three bf16 numbers
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)     (3 x 8 = 24 mantissa bits)
// and a product is accumulated in fp32 (v_mfma_f32_16x16x32_bf16) from the six partial products of weight >= 2^-16,
// in the fixed order  lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi  (A part . W part).  The dropped three are
// <= 2^-24 relative -- below the rounding of an fp32 multiply.
//
// What is new.  v1 kept A in fp32 and every wave normalised (LayerNorm) and split its A fragment inside the k loop:
// ~50 VALU instructions per 54 MFMAs, gamma/beta traffic, and the same rows split again by every column-group
// workgroup.  v2 removes all vector work from the k loop:
//   * activations travel between the GEMMs of a block ALREADY SPLIT, in fragment order ("A3"): the epilogue that
//     produces a row block writes its hi/mid/lo parts once (the MFMA is issued with swapped operands, so a lane owns 4
//     consecutive columns of ONE row and two adjacent column tiles are exactly one consumer fragment: one 16-byte
//     store per part, no cross-lane traffic);
//   * LayerNorm is folded into the GEMM that follows it:
//         LN(x) . W^T + b  =  rstd * ( x . (gamma o W)^T  -  mean * s )  +  c,
//         s_n = sum_k gamma_k W_nk,   c_n = sum_k beta_k W_nk + b_n,
//     gamma is multiplied into W when the binding splits the weights, mean / rstd come from the per-slice statistics
//     the residual epilogues emit (as in v1) and are applied to the accumulator tile.  |mean| / sigma of a residual
//     stream is O(0.1), so the cancellation in (acc - mean * s) costs nothing (measured: 5.0e-7 folded vs 5.5e-7
//     unfolded against fp64 at the deepest LayerNorm of the bench model); it would cost a factor sqrt(1 + (mean /
//     sigma)^2) for a badly centred stream.
// The k loop is then: LDS-DMA of both operands (no VGPRs), ds_read_b128 fragments prefetched one stage ahead, MFMA.
//
// Layouts (K = 136 G columns, G a multiple of 4: 544, 1088, 2176; KT = K / 32 k-tiles):
//   k-tile t < 4G ("full"):  group g = t / 4, quarter p = t % 4; lane (i, kq) element j  <->  column
//                            136 g + 32 p + 16 (j / 4) + 4 kq + (j % 4)
//   k-tile t = 4G + u ("tail", the 8 last columns of groups 4u .. 4u+3):  lane (i, kq) element j  <->  column
//                            136 (4u + kq) + 128 + j
//   A3[row tile][4 row groups][KT][3 parts][64 lanes][8 bf16]   lane = 16 kq + i, i = row in the 16-row group
//       row tile T holds rows T*rpt .. T*rpt + rpt-1 (rpt <= 64 rows per tile; rows >= rpt of a tile are padding)
//   W3[N/136][KT][9 slots][3 parts][64 lanes][8 bf16]           lane = 16 kq + i, i = column in the 16-column tile;
//       slot s holds column tile {0,1,2,3,8,4,5,6,7}[s]: waves 0..3 own slots 0..4 (two tile pairs + the half tile),
//       waves 4..7 slots 5..8 (two pairs); followed by the fold vectors c[N], s[N] (fp32)
// Workgroup = one row tile x 136 columns (x NPASS column groups), 8 waves: wave w owns row group w & 3 and slots
// 0..4 (w < 4) or 5..8: every SIMD sees 9 tiles = 54 MFMAs per k-tile.  Stage = A3 12 KiB + W3 27 KiB, ring of 4.
// The k order of every output element is fixed, so results do not depend on batch size or launch geometry.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "gemm_common.hpp"

namespace mpl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int X3_RG = 3 * 1024;              // A3 bytes per (row group, k-tile)
constexpr int X3_A = 4 * X3_RG;              // A3 bytes per stage (64 rows)
constexpr int X3_W = 27 * 1024;              // W3 bytes per k-tile of a 136-column group
constexpr int X3_STAGE = X3_A + X3_W;        // 39936
constexpr int X3_NST = 4;                    // ring depth (159744 B of LDS, one workgroup per CU)
constexpr int X3_MAX_WGS = 1024;             // workgroups of x3_stack_kernel (one per CU: 256 on MI355X)
constexpr int X3_T0 = 5;
#ifndef X3_DBG
#define X3_DBG 0   // 1: per-wave s_memtime stamps for mpl_x3_debug_buffer (tools/chain_phase.py, tools/x3_phase.py build with
                   // MPL_HIPCC_FLAGS=-DX3_DBG=1); 0: the stamps, their branches and ~14 SGPRs are compiled out of the k loops
#endif
#ifndef X3_ABL
#define X3_ABL 0   // bench-only ablations (results are garbage): 1 no B fragment reads, 2 no DMA refill, 4 no A fragment reads, 8 no MFMA
#endif
#ifndef X3_WT_AUX
#define X3_WT_AUX 17   // cache policy of the hand-off stores: 17 = sc0 sc1 (system scope write-through), 16 = sc1 (agent scope)
#endif
#ifndef X3_STAGGER
#define X3_STAGGER 1   // 1: the waves 4..7 request their DMA pieces three product rows later than the waves 0..3
#endif

__host__ __device__ constexpr int x3_slot_tile(int s) { return s < 4 ? s : (s == 4 ? 8 : s - 1); }

// column of element j of lane quarter kq in k-tile t (G = K / 136)
__host__ __device__ inline int x3_col(int t, int kq, int j, int G) {
    if (t < 4 * G) return 136 * (t >> 2) + 32 * (t & 3) + 16 * (j >> 2) + 4 * kq + (j & 3);
    return 136 * (4 * (t - 4 * G) + kq) + 128 + j;
}

bool x3_shape_ok(int N, int K) { return N > 0 && K > 0 && N % BN == 0 && K % (4 * BN) == 0; }

// NP = parts per operand element: 3 = fp32 as the exact sum of three bf16 (the fp32 engine), 1 = one bf16 (the bf16
// engine, BASELINE.json configs[2]).  A STAGE of the k loop is 3 KiB per 16-row group of A and 27 KiB of W either way:
// one k-tile of 32 in three parts, or three consecutive k-tiles in one part (K padded with zero k-tiles to a multiple of
// 96) -- so both engines share the ring, the DMA schedule and the fragment reads, and differ in which fragments an MFMA
// pairs (six part products of one k-tile / three k-tiles) and in how an epilogue packs its output.
__host__ __device__ constexpr int x3_stages(int K, int NP) { return NP == 3 ? K / BK : (K / BK + 2) / 3; }
// byte offset of (k-tile t, part p) inside the strip of a 16-row group of an A operand
__host__ __device__ constexpr unsigned x3_frag_off(int t, int p, int NP) { return NP == 3 ? (unsigned)(t * X3_RG + p * 1024) : (unsigned)(t * 1024); }

size_t x3_operand_bytes(int N, int K, int NP) {
    if (!x3_shape_ok(N, K)) return 0;
    return (size_t)(N / BN) * x3_stages(K, NP) * X3_W + (size_t)2 * N * sizeof(float);
}

size_t x3_act_bytes(int M, int K, int rpt, int NP) {
    if (M <= 0 || K <= 0 || K % (4 * BN) || rpt <= 0 || rpt > BM) return 0;
    const size_t tiles = ((size_t)M + rpt - 1) / rpt;
    return tiles * 4 * x3_stages(K, NP) * X3_RG;
}

__device__ __forceinline__ bf16x8 to_bf16x8(const float (&x)[8]) {      // round to nearest even
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)x[i];
    return v;
}

// 8 fp32 -> hi / mid / lo packed bf16 (RNE at every step; the residuals are exact in fp32)
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    float r[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = (__bf16)x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = x[i] - (float)hi[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) mid[i] = (__bf16)r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r2[i] = r[i] - (float)mid[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = (__bf16)r2[i];
}

// ---------------------------------------------------------------------------------------------- weight operand
template <int NP>
__global__ __launch_bounds__(256) void split_w3_kernel(const float* __restrict__ W, const float* __restrict__ gamma, int N,
                                                        int K, bf16x8* __restrict__ dst, size_t total) {
    const int G = K / BN, KS = x3_stages(K, NP), KTA = NP == 3 ? KS : 3 * KS;   // k-tiles incl. the zero padding of NP = 1
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const int slot = (int)((idx >> 6) % NT);
        const int kt = (int)((idx / (64 * NT)) % KTA);
        const int g = (int)(idx / ((size_t)64 * NT * KTA));
        const int li = lane & 15, kq = lane >> 4;
        const int c = x3_slot_tile(slot) * 16 + li;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = 0.f;
            if (c < BN && kt < K / BK) {
                const int k = x3_col(kt, kq, j, G);
                const float w = W[(size_t)(g * BN + c) * K + k];
                x[j] = gamma ? w * gamma[k] : w;           // LayerNorm gain folded into the weight (one fp32 rounding)
            }
        }
        if (NP == 3) {
            bf16x8 hi, mid, lo;
            split3(x, hi, mid, lo);
            bf16x8* o = dst + ((size_t)(g * KS + kt) * 27 + slot * 3) * 64 + lane;
            o[0] = hi;
            o[64] = mid;
            o[128] = lo;
        } else {
            dst[((size_t)(g * KS + kt / 3) * 27 + slot * 3 + kt % 3) * 64 + lane] = to_bf16x8(x);
        }
    }
}

// fold vectors: c_n = bias_n + sum_k beta_k W_nk,  s_n = sum_k of the (gamma_k W_nk) the operand holds -- fl32 for the
// split operand, bf16 for the bf16 operand (fp64 sums, one wave per n)
template <int NP>
__global__ __launch_bounds__(256) void fold_vectors_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ bias,
                                                            int N, int K, float* __restrict__ cvec, float* __restrict__ svec) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    double s = 0.0, c = 0.0;
    if (gamma) {
        for (int k = lane; k < K; k += 64) {
            const float w = W[(size_t)n * K + k];
            const float wg = w * gamma[k];
            s += NP == 3 ? (double)wg : (double)(float)(__bf16)wg;
            c += (double)w * (double)beta[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            c += __shfl_xor(c, o, 64);
        }
    }
    if (lane == 0) {
        cvec[n] = (float)(c + (double)bias[n]);
        svec[n] = (float)s;
    }
}

template <int NP>
static int launch_pack(const float* W, int N, int K, const float* ln_w, const float* ln_b, const float* bias, unsigned short* dst,
                       hipStream_t s) {
    const int KS = x3_stages(K, NP), KTA = NP == 3 ? KS : 3 * KS;
    const size_t total = (size_t)(N / BN) * KTA * NT * 64;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split_w3_kernel<NP>, dim3(grid), dim3(256), 0, s, W, ln_w, N, K, reinterpret_cast<bf16x8*>(dst), total);
    float* vec = reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + (size_t)(N / BN) * KS * X3_W);
    hipLaunchKernelGGL(fold_vectors_kernel<NP>, dim3((N + 3) / 4), dim3(256), 0, s, W, ln_w, ln_b, bias, N, K, vec, vec + N);
    return hip_check_launch();
}

int launch_split_bf16x3(const float* W, int N, int K, const float* ln_w, const float* ln_b, const float* bias,
                        unsigned short* dst, int np, hipStream_t s) {
    if (!W || !dst || !bias || !x3_shape_ok(N, K) || ((ln_w != nullptr) != (ln_b != nullptr)) || (np != 1 && np != 3))
        return MPL_E_INVALID;
    ProfScope prof(MPL_K_PACK, s);
    return np == 3 ? launch_pack<3>(W, N, K, ln_w, ln_b, bias, dst, s) : launch_pack<1>(W, N, K, ln_w, ln_b, bias, dst, s);
}

// ---------------------------------------------------------------------------------------------- activation operand
// fp32 rows -> A3 (used for the rows that enter a block stack from outside: the SPT output, the unit-test entry)
// fp32 rows -> packed operand (the rows that enter a block stack from outside: the SPT output, the unit-test entry).
// One launch also does the two other things the entry of a stack needs: blocks >= nb_split compute the LayerNorm slice
// partials {mean, M2} of the rows (one wave per row, two-pass: stats != NULL), and block 0 zeroes the arrival counters of
// the persistent stack kernel (counters != NULL) -- three launches and a memset node in round 2a, one launch now.
template <int NP>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, int M, int K, int ldx, int rpt,
                                                          char* __restrict__ dst, size_t total, int nb_split,
                                                          float* __restrict__ stats, unsigned* __restrict__ counters,
                                                          int n_counters) {
    if ((int)blockIdx.x >= nb_split) {
        const int lane = threadIdx.x & 63;
        const int row = ((int)blockIdx.x - nb_split) * 4 + (threadIdx.x >> 6);
        if (row >= M) return;
        const int ns = K / BN;
        for (int sidx = 0; sidx < ns; ++sidx) {
            const float* xr = X + (size_t)row * ldx + sidx * BN;
            const bool on = lane < BN / 4;                       // 34 float4 per 136-column slice
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (on) v = ld4(xr + 4 * lane);
            const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) / (float)BN;
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            const float ss = wave_sum(on ? (a * a + b * b) + (c * c + d * d) : 0.f);
            if (lane == 0) {
                stats[((size_t)row * ns + sidx) * 2] = mean;
                stats[((size_t)row * ns + sidx) * 2 + 1] = ss;
            }
        }
        return;
    }
    if (blockIdx.x == 0 && counters)
        for (int i = threadIdx.x; i < n_counters; i += 256) counters[i] = 0u;
    const int G = K / BN, KS = x3_stages(K, NP), KTA = NP == 3 ? KS : 3 * KS;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)nb_split * 256) {
        const int lane = (int)(idx & 63);
        const int kt = (int)((idx >> 6) % KTA);
        const size_t rgi = idx / ((size_t)64 * KTA);          // tile * 4 + row group
        const int li = lane & 15, kq = lane >> 4;
        const int rl = (int)(rgi & 3) * 16 + li;
        const size_t row = (rgi >> 2) * rpt + rl;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
        if (rl < rpt && row < (size_t)M && kt < K / BK) {
            const float* src = X + row * ldx;
            const int c0 = x3_col(kt, kq, 0, G), c4 = x3_col(kt, kq, 4, G);
            const float4 p = ld4(src + c0), q = ld4(src + c4);
            x[0] = p.x; x[1] = p.y; x[2] = p.z; x[3] = p.w; x[4] = q.x; x[5] = q.y; x[6] = q.z; x[7] = q.w;
        }
        char* o = dst + rgi * KS * X3_RG + lane * 16;
        if (NP == 3) {
            bf16x8 hi, mid, lo;
            split3(x, hi, mid, lo);
            *reinterpret_cast<bf16x8*>(o + x3_frag_off(kt, 0, 3)) = hi;
            *reinterpret_cast<bf16x8*>(o + x3_frag_off(kt, 1, 3)) = mid;
            *reinterpret_cast<bf16x8*>(o + x3_frag_off(kt, 2, 3)) = lo;
        } else {
            *reinterpret_cast<bf16x8*>(o + x3_frag_off(kt, 0, 1)) = to_bf16x8(x);
        }
    }
}

// stats / counters optional (see the kernel)
int launch_split_rows(const float* X, int M, int K, int ldx, int rpt, unsigned short* dst, int np, float* stats,
                      unsigned* counters, int n_counters, hipStream_t s) {
    if (!X || !dst || x3_act_bytes(M, K, rpt, np) == 0 || (ldx & 3) || (np != 1 && np != 3)) return MPL_E_INVALID;
    const size_t tiles = ((size_t)M + rpt - 1) / rpt;
    const int KS = x3_stages(K, np), KTA = np == 3 ? KS : 3 * KS;
    const size_t total = tiles * 4 * KTA * 64;
    const int nb_split = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const int grid = nb_split + (stats ? (M + 3) / 4 : 0);
    ProfScope prof(MPL_K_ROW_STATS, s);
    if (np == 3)
        hipLaunchKernelGGL(split_rows_kernel<3>, dim3(grid), dim3(256), 0, s, X, M, K, ldx, rpt, reinterpret_cast<char*>(dst), total,
                           nb_split, stats, counters, n_counters);
    else
        hipLaunchKernelGGL(split_rows_kernel<1>, dim3(grid), dim3(256), 0, s, X, M, K, ldx, rpt, reinterpret_cast<char*>(dst), total,
                           nb_split, stats, counters, n_counters);
    return hip_check_launch();
}

// ---------------------------------------------------------------------------------------------- GEMM
struct X3Args {
    const char* A3;          // split activations, K columns
    const char* W3;          // split weights (gamma folded for LNF)
    const float* cvec;       // bias (or folded c) per output column
    const float* svec;       // LNF: s per output column
    const float* stats;      // LNF: per-row slice partials of the K-wide input rows
    const float* R;          // residual (fp32), EPI_BIAS_RESIDUAL
    int ldr;
    float* C;                // fp32 output (optional)
    int ldc;
    char* C3;                // A3 output (optional): the next GEMM's operand
    float* stats_out;        // residual epilogue: slice partials of the rows produced
    int M, N, K, rpt;
    int grid_m, grid_n;
    float eps;
    int att_ntok, att_hd;
    unsigned long long* dbg;   // bench-only (mpl_x3_debug_buffer): per-wave s_memtime stamps, else NULL
    // chain mode only: where a workgroup reports that a partner of its team never arrived (x3_stack_kernel)
    unsigned* err_ws;          // word in the call's workspace (read by fuse_head_kernel: the output is poisoned with NaN)
    unsigned* err_host;        // sticky word in pinned host memory (read by the next API call on this device: it fails)
    int spin_log2;             // polls of the arrival counter before a wait counts as lost
};

static std::atomic<unsigned long long*> g_x3_dbg{nullptr};
void x3_set_debug_buffer(unsigned long long* p) { g_x3_dbg.store(p); }
// polls (s_sleep 2 + one L2 round trip each, ~1 us) before a wait of x3_stack_kernel counts as lost: 2^23 ~ 10 s
static std::atomic<int> g_x3_spin_log2{23};
void x3_set_spin_log2(int v) { g_x3_spin_log2.store(v & 0xff); }

enum { X3_EPI_BIAS = 0, X3_EPI_GELU = 1, X3_EPI_RES = 2, X3_EPI_ATT = 3 };

constexpr int X3_VEC = X3_NST * X3_STAGE;   // the 4 KiB of LDS above the ring: the epilogue vectors [pass][c | s][136] of a phase
constexpr int X3_LDS_BYTES = X3_VEC + 4096;  // = 160 KiB
constexpr int X3_FAIL = X3_VEC + 4092;       // last word of the LDS (behind the <= 3264 B of epilogue vectors): "a wait of this workgroup was lost"
constexpr int X3_ATT_TS = 3 * BN + 4;        // row stride (floats) of the q | k | v tile of the attention epilogue

// ---- stores / loads of data that crosses workgroups INSIDE a launch (chain mode, see x3_stack_kernel): write-through
// (sc0 sc1) stores and L1-bypassing (sc1) loads on both sides -- a valid hand-off for ANY placement of the workgroups
// (MI355X_MICROARCH.md, "Valid forms").  Kernel boundaries make plain accesses sufficient in the one-GEMM launches.
// `base` wave-uniform, `off` < 2 GiB.  wt: sc0 | sc1 = write-through to memory (the line leaves the writer's L2).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void st16(bool wt, void* base, unsigned off, const bf16x8& v) {
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    if (wt) __builtin_amdgcn_raw_buffer_store_b128(w, x3_rsrc(base), off, 0, X3_WT_AUX);
    else __builtin_amdgcn_raw_buffer_store_b128(w, x3_rsrc(base), off, 0, 0);
}
__device__ __forceinline__ void st8(bool wt, void* base, unsigned off, const bf16x8& v) {      // the 4 low bf16 of a fragment
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    const u32x2 h = {w[0], w[1]};
    if (wt) __builtin_amdgcn_raw_buffer_store_b64(h, x3_rsrc(base), off, 0, X3_WT_AUX);
    else __builtin_amdgcn_raw_buffer_store_b64(h, x3_rsrc(base), off, 0, 0);
}
__device__ __forceinline__ void st_f2(bool wt, float* base, unsigned off, float x, float y) {
    const u32x2 h = {__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y)};
    if (wt) __builtin_amdgcn_raw_buffer_store_b64(h, x3_rsrc(base), off, 0, X3_WT_AUX);
    else __builtin_amdgcn_raw_buffer_store_b64(h, x3_rsrc(base), off, 0, 0);
}
// L1-bypassing (sc1) loads the compiler tracks (buffer form; `base` wave-uniform, offsets < 2 GiB)
__device__ __forceinline__ u32x4 ld16_l2(const void* base, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(x3_rsrc(base), off, 0, 16);
}
// Attention.forward :55-64 on the q | k | v tile T[64][X3_ATT_TS] (+bias, LayerNorm applied) of this workgroup's 136
// channels: S whole sequences of nt tokens (rows S*nt.. are padding); output written as A3 of width Dq for proj.
// pack 8 fp32 values of one fragment lane and store them as k-tile t of the strip at `base` (+ lane offset `lo`)
template <int NP>
__device__ __forceinline__ void emit_frag(bool wt, char* base, int t, unsigned lo, const float (&x)[8]) {
    if (NP == 3) {
        bf16x8 hi, mid, l3;
        split3(x, hi, mid, l3);
        st16(wt, base, x3_frag_off(t, 0, 3) + lo, hi);
        st16(wt, base, x3_frag_off(t, 1, 3) + lo, mid);
        st16(wt, base, x3_frag_off(t, 2, 3) + lo, l3);
    } else {
        st16(wt, base, x3_frag_off(t, 0, 1) + lo, to_bf16x8(x));
    }
}
// the same for the 4 values (8 bytes) a lane contributes to the shared tail k-tile
template <int NP>
__device__ __forceinline__ void emit_tail(bool wt, char* base, int t, unsigned lo, const float (&x)[8]) {
    if (NP == 3) {
        bf16x8 hi, mid, l3;
        split3(x, hi, mid, l3);
        st8(wt, base, x3_frag_off(t, 0, 3) + lo, hi);
        st8(wt, base, x3_frag_off(t, 1, 3) + lo, mid);
        st8(wt, base, x3_frag_off(t, 2, 3) + lo, l3);
    } else {
        st8(wt, base, x3_frag_off(t, 0, 1) + lo, to_bf16x8(x));
    }
}
// NP = 1: the k-tiles that pad the strip of row group `rg_strip` to whole stages are zero (read by the next GEMM)
template <int NP>
__device__ __forceinline__ void zero_pad_tiles(bool wt, char* strip, int Kout, int lane) {
    if (NP == 1) {
        const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t = Kout / BK; t < 3 * x3_stages(Kout, 1); ++t) st16(wt, strip, x3_frag_off(t, 0, 1) + lane * 16, to_bf16x8(z));
    }
}

// Generic form (any n_tok <= 32, any head width that divides 136): run-time sizes.  The headline shape (4 tokens, 68-wide
// heads) never comes here: x3_phase finishes its attention in registers.
template <int NP>
__device__ __forceinline__ void x3_attention(bool WT, float* T, float* SC, int tid, int nt, int hd, int S, char* C3, int tile_m,
                                             int g_out, int Dq) {
    const int hd4 = hd >> 2;
    const int HP = BN / hd, nn = nt * nt;
    const float scale = 1.0f / sqrtf((float)hd);
    for (int t = tid; t < S * HP * nn; t += 512) {
        const int j = t % nt, i = (t / nt) % nt, hh = (t / nn) % HP, sq = t / (nn * HP);
        const float* q = T + (sq * nt + i) * X3_ATT_TS + hh * hd;
        const float* k = T + (sq * nt + j) * X3_ATT_TS + BN + hh * hd;
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
    for (int t = tid; t < S * HP * nt; t += 512) {
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
    // P.V and the A3 fragments of the output rows: task = (row, quarter p, lane quarter kq) -> 8 values = the two
    // 4-column chunks 32p + 4kq and 32p + 16 + 4kq; tail tasks (row, kq < 2) -> 4 values at 128 + 4kq
    const int Go = Dq / BN;
    const int strip = x3_stages(Dq, NP) * X3_RG;                 // bytes of one row group of the output operand
    char* cbase = C3 + (size_t)tile_m * 4 * strip;               // this row tile of the output operand (wave-uniform)
    auto pv4 = [&](int row, int c) -> float4 {
        float4 o = {0.f, 0.f, 0.f, 0.f};
        if (row < S * nt) {
            const int sq = row / nt, i = row - sq * nt;
            const int hh = c / hd;
            const float* pr = SC + ((sq * HP + hh) * nt + i) * nt;
            const float* v = T + (sq * nt) * X3_ATT_TS + 2 * BN + c;
            for (int j = 0; j < nt; ++j) {
                const float4 vv = ld4(v + j * X3_ATT_TS);
                const float pj = pr[j];
                o.x = fmaf(pj, vv.x, o.x);
                o.y = fmaf(pj, vv.y, o.y);
                o.z = fmaf(pj, vv.z, o.z);
                o.w = fmaf(pj, vv.w, o.w);
            }
        }
        return o;
    };
    for (int t = tid; t < BM * 16; t += 512) {
        const int li = t & 15, kq = (t >> 4) & 3, rg = (t >> 6) & 3, p = t >> 8;
        const int row = rg * 16 + li;
        const float4 a = pv4(row, 32 * p + 4 * kq), b = pv4(row, 32 * p + 16 + 4 * kq);
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        emit_frag<NP>(WT, cbase, 4 * g_out + p, (unsigned)(rg * strip + (kq * 16 + li) * 16), x);
    }
    for (int t = tid; t < BM * 2; t += 512) {
        const int li = t & 15, kq = (t >> 4) & 1, rg = t >> 5;
        const int row = rg * 16 + li;
        const float4 a = pv4(row, 128 + 4 * kq);
        const float x[8] = {a.x, a.y, a.z, a.w, 0.f, 0.f, 0.f, 0.f};
        emit_tail<NP>(WT, cbase, 4 * Go + (g_out >> 2), (unsigned)(rg * strip + ((g_out & 3) * 16 + li) * 16 + kq * 8), x);
    }
    if (NP == 1 && g_out == 0 && tid < 256) zero_pad_tiles<NP>(WT, cbase + (tid >> 6) * strip, Dq, tid & 63);
}

// One GEMM of one workgroup tile (tm, tn): everything a wave does for its NTW slots starting at slot `slot0`.
// CHAIN = false: the GEMM is a launch of its own.  CHAIN = true: it is one phase of x3_stack_kernel -- the workgroups
// (tm, 0 .. G-1) of a row tile advance together through the GEMMs of the whole block stack; `chain` counts their
// arrivals, the A operand of this phase may be read once it reaches `chain_need`, and this workgroup arrives when its
// outputs are written.  The W operand does not depend on the other workgroups: its first stages are requested BEFORE
// the wait.
// Returns false (chain mode only, every wave of the workgroup alike) when the wait for the team timed out: nothing was
// computed, the error words are set and the caller must leave the kernel -- a workgroup NEVER continues past a failed wait.
template <int NP, int EPI, bool LNF, int NPASS, int NTW, bool CHAIN>
__device__ __forceinline__ bool x3_phase(const X3Args& a, char* smem, int tid, int wave, int slot0, int tm, int tn,
                                         unsigned* chain, unsigned chain_need) {
    constexpr int NST = X3_NST;
    constexpr bool HAS_A = NTW == X3_T0;         // waves 0..3 (slots 0..4) bring the A pieces
    constexpr bool WT = CHAIN;       // chain mode hand-offs: write-through stores (+ L1-bypassing loads): valid for any placement
    const int lane = tid & 63;
    const int rg = wave & 3;
    const int li = lane & 15, kq = lane >> 4;
    const int M = a.M, N = a.N, K = a.K;
    // NPASS = 2: two ADJACENT 136-column groups (tn counts pairs); NPASS = 3: the q, k, v slices of one group
    const int m0 = tm * a.rpt, n0 = NPASS == 2 ? tn * (2 * BN) : tn * BN;
    const int Dq = N / 3;
    const int KT = x3_stages(K, NP);             // stages per pass (k-tiles of 32 in three parts / triples of k-tiles)
    const int T = NPASS * KT;
    auto colbase = [&](int pass) -> int { return NPASS == 3 ? pass * Dq + n0 : n0 + pass * BN; };
    const unsigned long long t_entry = (X3_DBG && a.dbg) ? __builtin_amdgcn_s_memtime() : 0;

    // ---- DMA slots of this wave.  A3: waves 0..3 bring the 3 KiB of row group `wave`; W3: waves 4..7 pieces 4(w-4)..+3,
    // waves 0..2 pieces 16+3w..+2, wave 3 pieces 25, 26.  Source and destination of a run are contiguous: one M0 write.
    const int w_first = HAS_A ? 16 + 3 * wave : 4 * (wave - 4);
    const int w_cnt = HAS_A ? (wave == 3 ? 2 : 3) : 4;
    unsigned voA = (unsigned)(lane * 16), voW = (unsigned)(lane * 16 + w_first * 1024);
    asm volatile("" : "+v"(voA), "+v"(voW));
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // running state of the next W / A stage to request (stage u carries W of pass u % NPASS, and A when that pass is 0)
    int iw_t = 0, iw_g = 0, ia_t = 0;
    unsigned iw_slot = 0, ia_slot = 0;
    const char* is_w[NPASS];
#pragma unroll
    for (int g = 0; g < NPASS; ++g) is_w[g] = a.W3 + (size_t)(colbase(g) / BN) * KT * X3_W;
    const char* is_a = a.A3 + ((size_t)tm * 4 + (wave & 3)) * KT * X3_RG;
    auto issue_w = [&]() {
        const unsigned st = lds0 + iw_slot;
        const unsigned keep = dma_m0_save();
#pragma unroll
        for (int g = 0; g < NPASS; ++g)
            if (g == iw_g) {
                asm volatile(
                    "s_mov_b32 m0, %2\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %0, %1\n\t"
                    "global_load_lds_dwordx4 %0, %1 offset:1024"
                    :
                    : "v"(voW), "s"(is_w[g]), "s"(st + (unsigned)(X3_A + w_first * 1024))
                    : "memory");
                if (w_cnt > 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(voW), "s"(is_w[g]) : "memory");
                if (w_cnt > 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" : : "v"(voW), "s"(is_w[g]) : "memory");
                is_w[g] += X3_W;
            }
        if (++iw_g == NPASS) iw_g = 0;
        dma_m0_restore(keep);
        ++iw_t;
        iw_slot += X3_STAGE;
        if (iw_slot == NST * X3_STAGE) iw_slot = 0;
    };
    // A of stage ia_t (a no-op for the stages of passes 1, 2); in chain mode the operand was written by other
    // workgroups of this launch: L1-bypassing loads
    auto issue_a = [&]() {
        if ((ia_t % NPASS) == 0) {
            if (HAS_A) {
                const unsigned keep = dma_m0_save();
                if (CHAIN)
                    asm volatile(
                        "s_mov_b32 m0, %2\n\t"
                        "s_nop 0\n\t"
                        "global_load_lds_dwordx4 %0, %1 sc1\n\t"
                        "global_load_lds_dwordx4 %0, %1 offset:1024 sc1\n\t"
                        "global_load_lds_dwordx4 %0, %1 offset:2048 sc1"
                        :
                        : "v"(voA), "s"(is_a), "s"(lds0 + ia_slot + (unsigned)(wave * X3_RG))
                        : "memory");
                else
                    asm volatile(
                        "s_mov_b32 m0, %2\n\t"
                        "s_nop 0\n\t"
                        "global_load_lds_dwordx4 %0, %1\n\t"
                        "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                        "global_load_lds_dwordx4 %0, %1 offset:2048"
                        :
                        : "v"(voA), "s"(is_a), "s"(lds0 + ia_slot + (unsigned)(wave * X3_RG))
                        : "memory");
                dma_m0_restore(keep);
            }
            is_a += X3_RG;
        }
        ++ia_t;
        ia_slot += X3_STAGE;
        if (ia_slot == NST * X3_STAGE) ia_slot = 0;
    };
    // The same requests for the steady state of the k loop (see `stage`): the pass of the W stage and whether the stage
    // carries A are compile-time constants of the call site, the destination is the slot the current stage just released
    // (stage t+NST lives where stage t lived): no dispatch, no ring arithmetic, no bookkeeping.
    auto refill_fast = [&](auto wp_c, auto ai_c, unsigned slot) {
        constexpr int g = decltype(wp_c)::value;
        constexpr bool with_a = decltype(ai_c)::value;
        const unsigned keep = dma_m0_save();
        asm volatile(
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:1024"
            :
            : "v"(voW), "s"(is_w[g]), "s"(lds0 + slot + (unsigned)(X3_A + w_first * 1024))
            : "memory");
        if (w_cnt > 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(voW), "s"(is_w[g]) : "memory");
        if (w_cnt > 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" : : "v"(voW), "s"(is_w[g]) : "memory");
        is_w[g] += X3_W;
        if (with_a && HAS_A) {
            if (CHAIN)
                asm volatile(
                    "s_mov_b32 m0, %2\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %0, %1 sc1\n\t"
                    "global_load_lds_dwordx4 %0, %1 offset:1024 sc1\n\t"
                    "global_load_lds_dwordx4 %0, %1 offset:2048 sc1"
                    :
                    : "v"(voA), "s"(is_a), "s"(lds0 + slot + (unsigned)(wave * X3_RG))
                    : "memory");
            else
                asm volatile(
                    "s_mov_b32 m0, %2\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %0, %1\n\t"
                    "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                    "global_load_lds_dwordx4 %0, %1 offset:2048"
                    :
                    : "v"(voA), "s"(is_a), "s"(lds0 + slot + (unsigned)(wave * X3_RG))
                    : "memory");
        }
        if (with_a) is_a += X3_RG;
        dma_m0_restore(keep);
    };
    // generic bookkeeping <- the state after fast stages, before generic stage t_next
    auto resync = [&](int t_next, unsigned slot) {
        iw_t = ia_t = t_next + NST;
        iw_g = iw_t % NPASS;
        iw_slot = ia_slot = slot;
    };
    // ---- the epilogue vectors of this workgroup's columns (c, and s of a folded LayerNorm) into the spare 4 KiB of LDS,
    // one LDS-DMA instruction per wave 0..3 at the FRONT of the queue (older than every counted piece): the epilogue
    // then reads them at LDS latency instead of one global round trip per column tile
    if (HAS_A) {
        constexpr int NV = NPASS * 2 * (BN / 4);                   // float4s: [pass][c | s][34]
        int idx = wave * 64 + lane;
        const bool on = idx < NV && (LNF || !((idx / (BN / 4)) & 1));
        idx = on ? idx : 0;
        const int vp = idx / (2 * (BN / 4)), which = (idx / (BN / 4)) & 1, c4 = idx % (BN / 4);
        const float* src = (which ? a.svec : a.cvec) + colbase(vp) + 4 * c4;
        if (on) dma16(src, lds0 + (unsigned)(X3_VEC + wave * 1024));   // masked lanes write nothing: X3_FAIL stays intact
    }
    // ---- prologue: the first NST stages in stage order, W before A inside a stage.  Chain mode: W(0) does not depend on
    // the other workgroups and is requested BEFORE the wait for them; everything else after it, A(0) first -- the LDS-DMA
    // path of a CU moves ~1 KiB per 24 cycles, and with the whole W prologue in front of it (round 2a) A(0) landed
    // ~2.5 k cycles later than it does now.  In-order queue of this wave:  W(0) A(0) | W(1) [A(1)] | W(2) [A(2)] | W(3) [A(3)].
    // The poll is the job of wave 7 (lane 0): its FIRST look at the counter goes out before any DMA piece of the wave --
    // loads return in order, and behind W(0) the answer would take a DMA round trip (~2 k cycles) even for the workgroup
    // that arrived last and has nothing to wait for (the one that sets the pace of its team).
    bool arrived = !CHAIN;
    if (CHAIN && !HAS_A && wave == 7)
        arrived = __hip_atomic_load(chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= chain_need;
    issue_w();
    if (CHAIN) {
        if (!HAS_A && wave == 7 && !arrived) {
            // relaxed, L2-bypassing; bounded so that a lost partner cannot hang the GPU.  A wait that runs out is an
            // ERROR, never a licence to go on: the workgroup raises the sticky error words and leaves the kernel (its
            // team mates follow when their own waits run out), fuse_head_kernel poisons the poses of this call with NaN
            // and the next API call on the device fails (MPL_E_DEVICE).
            const unsigned lim = 1u << a.spin_log2;
            unsigned spin = 0;
            for (; spin < lim; ++spin) {
                if (__hip_atomic_load(chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= chain_need) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (spin == lim && lane == 0) {
                *reinterpret_cast<volatile unsigned*>(smem + X3_FAIL) = 1u;
                if (a.err_ws) __hip_atomic_store(a.err_ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.err_host) __hip_atomic_store(a.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();      // control dependency only: the A requests below are issued after the poll succeeded
        asm volatile("" ::: "memory");
        if (*reinterpret_cast<volatile unsigned*>(smem + X3_FAIL)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the W(0) pieces in flight land before the LDS is given up
            return false;
        }
    }
    issue_a();
#pragma unroll
    for (int t = 1; t < NST; ++t)
        if (t < T) {
            issue_w();
            issue_a();
        }

    const int row_l = rg * 16 + li;
    const bool row_ok = row_l < a.rpt && m0 + row_l < M;
    const int row = row_ok ? m0 + row_l : (M - 1);
    const int ns = K / BN;                       // LNF: slices of the K-wide row (4 or 8)
    float4 st_raw[4];
    float4 rv[NTW];
    // epilogue operands that do not depend on the accumulators (LayerNorm partials of this lane's row, residual):
    // requested a few stages before the end of the k loop, consumed after it
    auto epilogue_operands = [&]() {
        if (LNF) {
            const float* sp = a.stats + (size_t)row * ns * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (2 * i < ns) {
                    if (CHAIN) {   // written by the other workgroups of the team in this launch: L1-bypassing loads
                        st_raw[i].x = __hip_atomic_load(sp + 4 * i + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        st_raw[i].y = __hip_atomic_load(sp + 4 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        st_raw[i].z = __hip_atomic_load(sp + 4 * i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        st_raw[i].w = __hip_atomic_load(sp + 4 * i + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        st_raw[i] = ld4(sp + 4 * i);
                    }
                } else {
                    st_raw[i] = float4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (EPI == X3_EPI_RES) {
            // fp32 rows this workgroup wrote itself (earlier GEMM of the stack) or a previous launch wrote; in chain mode
            // read past the L1, which own stores do not refresh
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                int c = 16 * x3_slot_tile(slot0 + n) + 4 * kq;
                c = c + 3 < BN ? c : 0;
                const float* rp = a.R + (size_t)row * a.ldr + n0 + c;
                if (CHAIN) rv[n] = __builtin_bit_cast(float4, ld16_l2(a.R + (size_t)m0 * a.ldr, (unsigned)((size_t)(rp - (a.R + (size_t)m0 * a.ldr)) * 4)));
                else rv[n] = ld4(rp);
            }
        }
    };
    const int t_ops = T - 4;                    // always a generic stage (the fast ones end at T - 5; T >= 6)

    f32x4 acc[NPASS][NTW];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment registers, ping-pong: the reads of stage t+1 are issued among the MFMAs of stage t
    bf16x8 A0[3], A1[3];
    bf16x8 B0[NTW][3], B1[NTW][3];
    auto read_a = [&](unsigned slot, bf16x8 (&f)[3]) {
        const bf16x8* as = reinterpret_cast<const bf16x8*>(smem + slot + rg * X3_RG) + lane;
        f[0] = as[0];
        f[1] = as[64];
        f[2] = as[128];
    };
    auto read_b = [&](unsigned slot, bf16x8 (&f)[NTW][3]) {
        const bf16x8* bs = reinterpret_cast<const bf16x8*>(smem + slot + X3_A) + slot0 * 3 * 64 + lane;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            f[n][0] = bs[(n * 3 + 0) * 64];
            f[n][1] = bs[(n * 3 + 1) * 64];
            f[n][2] = bs[(n * 3 + 2) * 64];
        }
    };
    auto slot_after = [](unsigned sl) -> unsigned { return sl + X3_STAGE == NST * X3_STAGE ? 0u : sl + X3_STAGE; };
    // Six products, fixed order (A part . W part): lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi, each over the wave's
    // NTW tiles (an accumulator is touched every NTW MFMAs).  The W fragment is the FIRST MFMA operand: lane (i, kq) then
    // holds C[row i][4 consecutive columns 16 tile + 4 kq ..] (transposed tile).
    // The fragment reads of stage t+1 are spread between the product rows of stage t (3 ds_read_b128 per row): issued
    // as one burst right after the barrier, the 18 reads of a wave overflow the 15-deep LDS queue and the wave sits in
    // the burst -- with its SIMD partner, which is in the same phase -- while the matrix pipe idles.
    auto mfma_row = [&](f32x4 (&accp)[NTW], const bf16x8& af, const bf16x8 (&bf)[NTW][3], int bp) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
            if (!(X3_ABL & 8)) accp[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[n][bp], af, accp[n], 0, 0, 0);
    };
    {   // stage 0 landed for everyone: at most the pieces of stages 1..3 of this wave are outstanding (counted with the
        // fewest pieces a wave of the role has: W pieces 2 (wave 3) / 4, A pieces 3 in the stages of pass 0)
        constexpr int LATER = HAS_A ? (NPASS == 1 ? 15 : 9) : 12;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LATER) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_a(0, A0);
        read_b(0, B0);
    }
    const unsigned long long t_loop = (X3_DBG && a.dbg) ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long t_vm = 0, t_bar = 0;      // bench-only: cycles at the counted DMA wait / at lgkmcnt + barrier
    unsigned slot_c = 0;
    // one stage: publish stage t+1 (which frees the slot of stage t for the DMA of stage t+NST), then the MFMAs of stage
    // t with the fragment reads of stage t+1 in between
    // FAST (tag fast_c): a stage of the steady state -- t + NST < T, so there is a next stage, a refill, and the standard
    // counted wait: none of the run-time case distinctions of the generic form (which the first stages of a one-pass GEMM
    // and the last NST stages use) and of its ~50 scalar instructions and ~15 branches per stage; the pass WP of the
    // refilled stage and whether it carries A (AI) come from the call site.
    auto stage = [&](auto fast_c, auto wp_c, auto ai_c, int t, f32x4 (&accp)[NTW], const bf16x8 (&a_cur)[3], bf16x8 (&a_nxt)[3],
                     const bf16x8 (&b_cur)[NTW][3], bf16x8 (&b_nxt)[NTW][3], bool next_has_a) {
        constexpr bool FAST = decltype(fast_c)::value;
        const unsigned slot_n = slot_after(slot_c);
        const bool more = FAST || t + 1 < T;
        unsigned long long w0 = 0, w1 = 0;
        if (X3_DBG && a.dbg) w0 = __builtin_amdgcn_s_memtime();
        if (more) {
            // own pieces of stage t+1 landed (stages t+2, t+3 may stay in flight); every fragment read of stage t has
            // returned: the barrier must not be passed before, the slot of stage t is refilled right after it.
            // The wait is an immediate, counted with the SMALLEST number of pieces any wave of the role has per stage
            // (conservative: a wave with more pieces then also waits for part of stage t+2, requested two stage times
            // ago).  The queue is in stage order from the prologue on (every GEMM has >= 6 stages: K % 544 == 0).
            constexpr int MINP = NPASS == 1 ? (HAS_A ? 5 : 4) : (HAS_A ? 2 : 4);
            static_assert(NST == 4, "the counted waits assume three stages in flight");
            if (FAST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MINP) : "memory");
            else if (t + 3 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MINP) : "memory");
            else if (t + 2 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MINP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (X3_DBG && a.dbg) { w1 = __builtin_amdgcn_s_memtime(); t_vm += w1 - w0; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (X3_DBG && a.dbg) t_bar += __builtin_amdgcn_s_memtime() - w1;
        }
        const bf16x8* bs = reinterpret_cast<const bf16x8*>(smem + slot_n + X3_A) + slot0 * 3 * 64 + lane;
        auto rd_b = [&](int n) {
            if (more && n < NTW && !(X3_ABL & 1)) {
                b_nxt[n][0] = bs[(n * 3 + 0) * 64];
                b_nxt[n][1] = bs[(n * 3 + 1) * 64];
                b_nxt[n][2] = bs[(n * 3 + 2) * 64];
            }
        };
        auto refill = [&]() {
            if constexpr (FAST) {
                if (!(X3_ABL & 2)) refill_fast(wp_c, ai_c, slot_c);
            } else {
                if (more && iw_t < T && !((X3_ABL & 2) && t > 0)) {   // stage t+NST into the slot of stage t
                    issue_w();
                    issue_a();
                }
                if (t == t_ops) epilogue_operands();
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NP == 3) {
            mfma_row(accp, a_cur[2], b_cur, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(X3_STAGGER && !HAS_A)) refill();
            if (more && next_has_a && !(X3_ABL & 4)) read_a(slot_n, a_nxt);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(accp, a_cur[0], b_cur, 2);
            __builtin_amdgcn_sched_barrier(0);
            rd_b(0);
            if (!next_has_a) rd_b(4);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(accp, a_cur[1], b_cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            rd_b(1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(accp, a_cur[1], b_cur, 0);
            __builtin_amdgcn_sched_barrier(0);
            rd_b(2);
            if (X3_STAGGER && !HAS_A) refill();
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(accp, a_cur[0], b_cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            rd_b(3);
            if (next_has_a) rd_b(4);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(accp, a_cur[0], b_cur, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // bf16 engine: the three fragments of a stage are three consecutive k-tiles, one product each
            mfma_row(accp, a_cur[0], b_cur, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(X3_STAGGER && !HAS_A)) refill();
            if (more && next_has_a && !(X3_ABL & 4)) read_a(slot_n, a_nxt);
            rd_b(0);
            if (!next_has_a) rd_b(4);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(accp, a_cur[1], b_cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            rd_b(1);
            rd_b(2);
            if (X3_STAGGER && !HAS_A) refill();
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(accp, a_cur[2], b_cur, 2);
            __builtin_amdgcn_sched_barrier(0);
            rd_b(3);
            if (next_has_a) rd_b(4);
            __builtin_amdgcn_sched_barrier(0);
        }
        slot_c = slot_n;
    };
    using GEN = std::integral_constant<bool, false>;
    using FST = std::integral_constant<bool, true>;
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;
    using AY = std::integral_constant<bool, true>;
    using AN = std::integral_constant<bool, false>;
    // head (generic) -> steady state (fast) -> tail (generic): three loops one after the other, so that the fragment
    // ping-pong registers never meet at a join of a fast and a generic path
    if constexpr (NPASS == 1) {
        int kt = 0;
        for (; kt + 5 < T; kt += 2) {            // both stages refill: t + NST < T
            stage(FST{}, W0{}, AY{}, kt, acc[0], A0, A1, B0, B1, true);
            stage(FST{}, W0{}, AY{}, kt + 1, acc[0], A1, A0, B1, B0, true);
        }
        resync(kt, slot_c);
        for (; kt + 1 < KT; kt += 2) {
            stage(GEN{}, W0{}, AY{}, kt, acc[0], A0, A1, B0, B1, true);
            stage(GEN{}, W0{}, AY{}, kt + 1, acc[0], A1, A0, B1, B0, true);
        }
        if (kt < KT) stage(GEN{}, W0{}, AY{}, kt, acc[0], A0, A1, B0, B1, true);
    } else if constexpr (NPASS == 2) {
        // two column groups share the A fragment of the k-tile; the second stage prefetches the next k-tile's.  The A
        // ping-pong alternates per k-tile; two stages per k-tile keep the B parity the same in every k-tile.
        // Stage t refills stage t + 4: pass t % 2, with A when that is 0.
        int kt = 0;
        for (; 2 * kt + 7 < T; kt += 2) {
            stage(FST{}, W0{}, AY{}, 2 * kt, acc[0], A0, A0, B0, B1, false);
            stage(FST{}, W1{}, AN{}, 2 * kt + 1, acc[1], A0, A1, B1, B0, true);
            stage(FST{}, W0{}, AY{}, 2 * kt + 2, acc[0], A1, A1, B0, B1, false);
            stage(FST{}, W1{}, AN{}, 2 * kt + 3, acc[1], A1, A0, B1, B0, true);
        }
        resync(2 * kt, slot_c);
        for (; kt + 1 < KT; kt += 2) {
            stage(GEN{}, W0{}, AY{}, 2 * kt, acc[0], A0, A0, B0, B1, false);
            stage(GEN{}, W0{}, AY{}, 2 * kt + 1, acc[1], A0, A1, B1, B0, true);
            stage(GEN{}, W0{}, AY{}, 2 * kt + 2, acc[0], A1, A1, B0, B1, false);
            stage(GEN{}, W0{}, AY{}, 2 * kt + 3, acc[1], A1, A0, B1, B0, true);
        }
        if (kt < KT) {
            stage(GEN{}, W0{}, AY{}, 2 * kt, acc[0], A0, A0, B0, B1, false);
            stage(GEN{}, W0{}, AY{}, 2 * kt + 1, acc[1], A0, A1, B1, B0, true);
        }
    } else {
        // k-tile by k-tile: q, k, v stages share the A fragment; three stages per k-tile flip the B parity every k-tile.
        // Stage t = 3 kt + j refills stage t + 4: pass (j + 1) % 3, with A when that is 0.
        int kt = 0;
        for (; 3 * kt + 9 < T; kt += 2) {
            stage(FST{}, W1{}, AN{}, 3 * kt, acc[0], A0, A0, B0, B1, false);
            stage(FST{}, W2{}, AN{}, 3 * kt + 1, acc[1], A0, A0, B1, B0, false);
            stage(FST{}, W0{}, AY{}, 3 * kt + 2, acc[2], A0, A1, B0, B1, true);
            stage(FST{}, W1{}, AN{}, 3 * kt + 3, acc[0], A1, A1, B1, B0, false);
            stage(FST{}, W2{}, AN{}, 3 * kt + 4, acc[1], A1, A1, B0, B1, false);
            stage(FST{}, W0{}, AY{}, 3 * kt + 5, acc[2], A1, A0, B1, B0, true);
        }
        resync(3 * kt, slot_c);
        for (; kt + 1 < KT; kt += 2) {
            stage(GEN{}, W0{}, AY{}, 3 * kt, acc[0], A0, A0, B0, B1, false);
            stage(GEN{}, W0{}, AY{}, 3 * kt + 1, acc[1], A0, A0, B1, B0, false);
            stage(GEN{}, W0{}, AY{}, 3 * kt + 2, acc[2], A0, A1, B0, B1, true);
            stage(GEN{}, W0{}, AY{}, 3 * kt + 3, acc[0], A1, A1, B1, B0, false);
            stage(GEN{}, W0{}, AY{}, 3 * kt + 4, acc[1], A1, A1, B0, B1, false);
            stage(GEN{}, W0{}, AY{}, 3 * kt + 5, acc[2], A1, A0, B1, B0, true);
        }
        if (kt < KT) {
            stage(GEN{}, W0{}, AY{}, 3 * kt, acc[0], A0, A0, B0, B1, false);
            stage(GEN{}, W0{}, AY{}, 3 * kt + 1, acc[1], A0, A0, B1, B0, false);
            stage(GEN{}, W0{}, AY{}, 3 * kt + 2, acc[2], A0, A1, B0, B1, true);
        }
    }

    // ------------------------------------------------------------------------------------------ epilogue
    const unsigned long long t_epi = (X3_DBG && a.dbg) ? __builtin_amdgcn_s_memtime() : 0;
    float mu = 0.f, rs = 1.f;
    if (LNF) {
        float st[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st[4 * i] = st_raw[i].x; st[4 * i + 1] = st_raw[i].y; st[4 * i + 2] = st_raw[i].z; st[4 * i + 3] = st_raw[i].w;
        }
        // Chan's combination of the per-slice {mean, M2} partials (gemm_common.hpp ln_combine, from registers)
        float msum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) msum += (i < ns) ? st[2 * i] : 0.f;
        const float mean = msum / (float)ns;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = st[2 * i] - mean;
            m2 += (i < ns) ? fmaf((float)BN * d, d, st[2 * i + 1]) : 0.f;
        }
        mu = mean;
        rs = 1.0f / sqrtf(fmaf(m2, 1.0f / (float)K, a.eps));
    }
    // acc[p][n][r] = C[row_l][colbase(p) + 16 tile(n) + 4 kq + r]
    auto tile_of = [&](int n) -> int { return x3_slot_tile(slot0 + n); };
    auto value4 = [&](int p, int n, float (&v)[4]) {
        const int cl = 16 * tile_of(n) + 4 * kq;
        const bool ok = cl + 3 < BN;
        const float* vecs = reinterpret_cast<const float*>(smem + X3_VEC) + p * 2 * BN + (ok ? cl : 0);
        const float4 cv = ld4(vecs);
        const float c4[4] = {cv.x, cv.y, cv.z, cv.w};
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        if (LNF) {
            const float4 sv = ld4(vecs + BN);
            s4[0] = sv.x; s4[1] = sv.y; s4[2] = sv.z; s4[3] = sv.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // explicit fused operations: the same roundings in every instantiation of this function (chain phases and
            // one-GEMM launches must agree bitwise, whatever the optimiser would contract)
            float t = acc[p][n][r];
            if (LNF) t = fmaf(rs, fmaf(-mu, s4[r], t), c4[r]);
            else t += c4[r];
            if (EPI == X3_EPI_GELU) t = gelu_as(t);
            v[r] = ok ? t : 0.f;
        }
    };
    unsigned long long t_st = 0;

    if constexpr (EPI == X3_EPI_ATT) {
      if (a.att_ntok == 4 && (a.att_hd == 68 || a.att_hd == BN) && a.rpt == BM) {
        // ---- Attention.forward :55-64 for 4 tokens per sequence and 68- or 136-wide heads, in REGISTERS.  A sequence is 4
        // consecutive rows = the 4 lanes of a quad (lane (li, kq), li = 4 s + i), and a lane holds 4 consecutive channels of
        // its row per column tile: k_j / v_j of the sequence come from the quad by DPP (quad_perm broadcast), the q.k sum
        // of a head (17 channel quads: tiles 0..3 + the first quad of tile 4 | the rest) is reduced over the tiles of the
        // wave, the 4 kq lanes and the two waves of the row group (one 2-KiB exchange through LDS).  No q|k|v tile in LDS
        // (round 2a wrote 104 KB per workgroup and read it back: 5 k + 7 k cycles of the qkv phase), and the output is
        // emitted from registers exactly like the other epilogues.
        float qv[NTW][4], kv[NTW][4], vv[NTW][4];
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            value4(0, n, qv[n]);
            value4(1, n, kv[n]);
            value4(2, n, vv[n]);
        }
        auto quad = [](float x, int j) -> float {      // value of lane (quad base + j)
            const int xi = __builtin_bit_cast(int, x);
            int r;
            switch (j) {
                case 0: r = __builtin_amdgcn_update_dpp(xi, xi, 0x00, 0xf, 0xf, false); break;
                case 1: r = __builtin_amdgcn_update_dpp(xi, xi, 0x55, 0xf, 0xf, false); break;
                case 2: r = __builtin_amdgcn_update_dpp(xi, xi, 0xaa, 0xf, 0xf, false); break;
                default: r = __builtin_amdgcn_update_dpp(xi, xi, 0xff, 0xf, 0xf, false); break;
            }
            return __builtin_bit_cast(float, r);
        };
        const bool two_heads = a.att_hd == 68;          // 136 channels = two 68-wide heads, or one 136-wide head
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};       // partial q_i . k_j of head 0 / head 1
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const bool h1 = two_heads && 4 * tile_of(n) + kq >= 17;       // channel quad 4 tile + kq: 0..16 head 0, 17..33 head 1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float d = qv[n][0] * quad(kv[n][0], j);
                d = fmaf(qv[n][1], quad(kv[n][1], j), d);
                d = fmaf(qv[n][2], quad(kv[n][2], j), d);
                d = fmaf(qv[n][3], quad(kv[n][3], j), d);
                s0[j] += h1 ? 0.f : d;
                s1[j] += h1 ? d : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = xor16_add(s0[j]); s0[j] = xor32_add(s0[j]);
            s1[j] = xor16_add(s1[j]); s1[j] = xor32_add(s1[j]);
        }
        float* xs = reinterpret_cast<float*>(smem);        // [2 halves][64 rows][8]
        const int half = slot0 ? 1 : 0;
        __syncthreads();                                    // every wave is done reading the last stage
        if (kq == 0) {
            st4(xs + (half * BM + row_l) * 8, float4{s0[0], s0[1], s0[2], s0[3]});
            st4(xs + (half * BM + row_l) * 8 + 4, float4{s1[0], s1[1], s1[2], s1[3]});
        }
        __syncthreads();
        float p0[4], p1[4];
        {
            const float4 a0 = ld4(xs + row_l * 8), a1 = ld4(xs + row_l * 8 + 4);
            const float4 b0 = ld4(xs + (BM + row_l) * 8), b1 = ld4(xs + (BM + row_l) * 8 + 4);
            const float scale = 1.0f / sqrtf((float)a.att_hd);
            const float t0[4] = {(a0.x + b0.x) * scale, (a0.y + b0.y) * scale, (a0.z + b0.z) * scale, (a0.w + b0.w) * scale};
            const float t1[4] = {(a1.x + b1.x) * scale, (a1.y + b1.y) * scale, (a1.z + b1.z) * scale, (a1.w + b1.w) * scale};
            auto softmax4 = [](const float (&t)[4], float (&pr)[4]) {
                const float mx = fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3]));
                const float e0 = __expf(t[0] - mx), e1 = __expf(t[1] - mx), e2 = __expf(t[2] - mx), e3 = __expf(t[3] - mx);
                const float inv = 1.0f / ((e0 + e1) + (e2 + e3));
                pr[0] = e0 * inv; pr[1] = e1 * inv; pr[2] = e2 * inv; pr[3] = e3 * inv;
            };
            softmax4(t0, p0);
            softmax4(t1, p1);
        }
        float ov[NTW][4];
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const bool h1 = two_heads && 4 * tile_of(n) + kq >= 17;
            const float pj[4] = {h1 ? p1[0] : p0[0], h1 ? p1[1] : p0[1], h1 ? p1[2] : p0[2], h1 ? p1[3] : p0[3]};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float o = pj[0] * quad(vv[n][c], 0);
                o = fmaf(pj[1], quad(vv[n][c], 1), o);
                o = fmaf(pj[2], quad(vv[n][c], 2), o);
                o = fmaf(pj[3], quad(vv[n][c], 3), o);
                ov[n][c] = o;
            }
        }
        // the attention output as A3 of width Dq for proj: two adjacent tiles = one fragment of the consumer
        const int Go = Dq / BN, g_out = n0 / BN;
        char* cbase = a.C3 + ((size_t)tm * 4 + rg) * x3_stages(Dq, NP) * X3_RG;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float x[8] = {ov[2 * q][0], ov[2 * q][1], ov[2 * q][2], ov[2 * q][3],
                                ov[2 * q + 1][0], ov[2 * q + 1][1], ov[2 * q + 1][2], ov[2 * q + 1][3]};
            emit_frag<NP>(WT, cbase, 4 * g_out + (slot0 ? 2 : 0) + q, (unsigned)(lane * 16), x);
        }
        if (NTW == X3_T0 && kq < 2) {     // the half tile: 4 values per lane into the shared tail k-tile
            const float x[8] = {ov[NTW - 1][0], ov[NTW - 1][1], ov[NTW - 1][2], ov[NTW - 1][3], 0.f, 0.f, 0.f, 0.f};
            emit_tail<NP>(WT, cbase, 4 * Go + (g_out >> 2), (unsigned)(((g_out & 3) * 16 + li) * 16 + kq * 8), x);
        }
        if (NP == 1 && g_out == 0 && HAS_A) zero_pad_tiles<NP>(WT, cbase, Dq, lane);
      } else {
        float* Tt = reinterpret_cast<float*>(smem);
        float* SC = Tt + BM * X3_ATT_TS;
        __syncthreads();                        // every wave is done reading the last stage
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int cl = 16 * tile_of(n) + 4 * kq;
                if (cl + 3 < BN) {
                    float v[4];
                    value4(p, n, v);
                    st4(Tt + row_l * X3_ATT_TS + p * BN + cl, float4{v[0], v[1], v[2], v[3]});
                }
            }
        __syncthreads();
        x3_attention<NP>(WT, Tt, SC, tid, a.att_ntok, a.att_hd, a.rpt / a.att_ntok, a.C3, tm, n0 / BN, Dq);
      }
    } else {
        const int Go = N / BN;
        char* cbase = a.C3 + ((size_t)tm * 4 + rg) * x3_stages(N, NP) * X3_RG;   // this wave's row group of the output operand
        float vals[NTW][4];
#pragma unroll
        for (int p = 0; p < (NPASS == 2 ? 2 : 1); ++p) {
            const int g_out = colbase(p) / BN;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                value4(p, n, vals[n]);
                const int cl = 16 * tile_of(n) + 4 * kq;
                const bool ok = cl + 3 < BN;
                if (EPI == X3_EPI_RES) {
                    vals[n][0] += rv[n].x; vals[n][1] += rv[n].y; vals[n][2] += rv[n].z; vals[n][3] += rv[n].w;
                    if (!ok) vals[n][0] = vals[n][1] = vals[n][2] = vals[n][3] = 0.f;
                }
                if (a.C && ok && row_ok)
                    st4(a.C + (size_t)row * a.ldc + colbase(p) + cl, float4{vals[n][0], vals[n][1], vals[n][2], vals[n][3]});
            }
            if (a.C3) {
                // two adjacent column tiles = one fragment of the consumer: slots (0,1) (2,3) of each half
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float x[8] = {vals[2 * q][0], vals[2 * q][1], vals[2 * q][2], vals[2 * q][3],
                                        vals[2 * q + 1][0], vals[2 * q + 1][1], vals[2 * q + 1][2], vals[2 * q + 1][3]};
                    emit_frag<NP>(WT, cbase, 4 * g_out + (slot0 ? 2 : 0) + q, (unsigned)(lane * 16), x);
                }
                if (NTW == X3_T0 && kq < 2) {     // the half tile: 4 values per lane into the shared tail k-tile
                    const float x[8] = {vals[NTW - 1][0], vals[NTW - 1][1], vals[NTW - 1][2], vals[NTW - 1][3], 0.f, 0.f, 0.f, 0.f};
                    emit_tail<NP>(WT, cbase, 4 * Go + (g_out >> 2), (unsigned)(((g_out & 3) * 16 + li) * 16 + kq * 8), x);
                }
                if (NP == 1 && g_out == 0 && HAS_A) zero_pad_tiles<NP>(WT, cbase, N, lane);
            }
        }
        if (X3_DBG && a.dbg) t_st = __builtin_amdgcn_s_memtime();
        if constexpr (EPI == X3_EPI_RES) {
            if (a.stats_out) {
                // LayerNorm partials {mean, M2} of the 136-column slice of each row: a row's values sit in the 4 kq lanes
                // of BOTH waves of a SIMD pair (w, w+4): two exchanges through LDS (sum, then centred squares), each
                // combined in the fixed order (half 0) + (half 1)
                float* xch = reinterpret_cast<float*>(smem);       // [2 phases][2 halves][64 rows]
                const int half = slot0 ? 1 : 0;
                float sum = 0.f;
#pragma unroll
                for (int n = 0; n < NTW; ++n) sum += (vals[n][0] + vals[n][1]) + (vals[n][2] + vals[n][3]);
                sum = xor16_add(sum);
                sum = xor32_add(sum);
                __syncthreads();                    // every wave is done with the ring
                if (kq == 0) xch[half * 64 + row_l] = sum;
                __syncthreads();
                const float mean = (xch[row_l] + xch[64 + row_l]) * (1.0f / (float)BN);
                float q = 0.f;
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const bool ok = 16 * tile_of(n) + 4 * kq + 3 < BN;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d = vals[n][r] - mean;
                        q = ok ? fmaf(d, d, q) : q;
                    }
                }
                q = xor16_add(q);
                q = xor32_add(q);
                if (kq == 0) xch[128 + half * 64 + row_l] = q;
                __syncthreads();
                if (half == 0 && kq == 0 && row_ok)
                    st_f2(WT, a.stats_out + (size_t)m0 * Go * 2, (unsigned)(((row - m0) * Go + n0 / BN) * 8), mean, xch[128 + row_l] + xch[192 + row_l]);
            }
        }
    }
    if (CHAIN) {
        // arrive: every store of this workgroup has been acknowledged (write-through), nobody touches the ring any more
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(chain, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (X3_DBG && a.dbg) {   // bench-only: entry, loop start, loop end, stores issued, stores drained (shader clock), wait sums
        if (!t_st) t_st = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* o = a.dbg + (size_t)(blockIdx.x * 8 + wave) * 8;
            o[0] = t_entry; o[1] = t_loop; o[2] = t_epi; o[3] = t_st; o[4] = t_end; o[5] = t_vm; o[6] = t_bar;
        }
    }
    return true;
}

template <int NP, int EPI, bool LNF, int NPASS>
__global__ __launch_bounds__(512, 2) void x3_gemm_kernel(const X3Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    if (wave < 4) x3_phase<NP, EPI, LNF, NPASS, X3_T0, false>(a, smem, tid, wave, 0, tm, tn, nullptr, 0u);
    else x3_phase<NP, EPI, LNF, NPASS, NT - X3_T0, false>(a, smem, tid, wave, X3_T0, tm, tn, nullptr, 0u);
}

// ---------------------------------------------------------------------------------------------- whole block stack
// ONE launch for all Block applications of a stack.  The GEMM chain of a block only couples the workgroups of one ROW
// TILE: proj of row tile tm needs the attention output of the G = D/136 workgroups (tm, 0..G-1) and nothing else, and so
// on down the stack.  So G workgroups form a team that walks one row tile through every GEMM of every block
// application, synchronising only among themselves (a monotonic arrival counter per row tile); teams never wait for
// each other and drift apart, which spreads the operand-fetch and store bursts that a grid of lock-stepped one-GEMM
// launches concentrates at its start and end -- and the weight stages of the next GEMM are requested before the wait.
// All teams are resident at once (one workgroup per CU, grid <= CU count): nobody waits for a workgroup that has not
// started.  With more row tiles than teams a team takes several tiles one after the other.
struct X3StackArgs {
    char *x3, *att3, *hid3;
    float *x, *stats;
    unsigned* counters;          // one per row tile, zeroed before the launch
    int M, D, n_tok, heads, rpt, n_tiles, n_teams, G, n_apps, n_phases;
    float eps;
    unsigned long long* dbg;
    unsigned *err_ws, *err_host;      // see X3Args
    int spin_log2;
    int inject;                       // test hook (mpl_x3_spin_limit): > 0 = workgroup (row tile 0, column group 0) leaves before phase `inject`
    const char* w[MPL_MAX_APPS][4];   // per application: qkv (norm1 folded), proj, fc1 (norm2 folded), fc2 operands
};

template <int NP>
__global__ __launch_bounds__(512, 2) void x3_stack_kernel(const X3StackArgs s) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = s.G, D = s.D;
    int team, tn;
    {
        const int b = blockIdx.x;
        // blocks b, b+8, .. share an XCD (observed, never relied on): a team takes blocks of one residue class.  Fewer than
        // 8 teams: the launch still has 8 G blocks and the classes without a team leave at once.
        team = (b & 7) + 8 * ((b >> 3) / G);
        tn = (b >> 3) % G;
        if (team >= s.n_teams) return;
    }
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(smem + X3_FAIL) = 0u;
    __syncthreads();
    auto vecs = [&](const char* w3, int N, int K) -> const float* {
        return reinterpret_cast<const float*>(w3 + (size_t)(N / BN) * x3_stages(K, NP) * X3_W);
    };
    // One loop over the 4 * n_apps GEMM phases of a row tile.  Every phase starts from OPAQUE copies of the thread id and
    // the tile index: nothing derived from them is loop invariant, so the compiler does not keep the address arithmetic
    // of all eight phase bodies alive across the loop (which spilled ~200 VGPRs).
    for (int tile0 = team; tile0 < s.n_tiles; tile0 += s.n_teams) {
        unsigned need = 0;           // arrivals that complete the phase whose output the next phase reads
        for (int ph = 0; ph < s.n_phases; ++ph, need += G) {
            int tidp = tid, tile = tile0, tnp = tn;
            asm volatile("" : "+v"(tidp));
            asm volatile("" : "+s"(tile), "+s"(tnp));
            const int wv = __builtin_amdgcn_readfirstlane(tidp >> 6);
            unsigned* ctr = s.counters + tile;
            const char* const* w = s.w[ph >> 2];
            bool ok = true;
            // fault injection for the failure-path test: one workgroup deserts its team, which must then REPORT the lost
            // hand-off (error words, NaN poses, MPL_E_DEVICE) instead of computing on stale operands
            if (s.inject > 0 && ph == s.inject && tile == 0 && tnp == 0) return;
            switch (ph & 3) {
                case 0: {   // x = x + proj(attn(qkv(norm1(x))))   (Block.forward :84-90)
                    const float* v = vecs(w[0], 3 * D, D);
                    const X3Args a{s.x3, w[0], v, v + 3 * D, s.stats, nullptr, 0, nullptr, 0, s.att3, nullptr, s.M, 3 * D, D, s.rpt,
                                   s.n_tiles, G, s.eps, s.n_tok, D / s.heads, s.dbg, s.err_ws, s.err_host, s.spin_log2};
                    if (wv < 4) ok = x3_phase<NP, X3_EPI_ATT, true, 3, X3_T0, true>(a, smem, tidp, wv, 0, tile, tnp, ctr, need);
                    else ok = x3_phase<NP, X3_EPI_ATT, true, 3, NT - X3_T0, true>(a, smem, tidp, wv, X3_T0, tile, tnp, ctr, need);
                    break;
                }
                case 2: {   // x = x + fc2(gelu(fc1(norm2(x))))    (Block.forward :91, Mlp.forward :31-37)
                    const float* v = vecs(w[2], 2 * D, D);
                    const X3Args a{s.x3, w[2], v, v + 2 * D, s.stats, nullptr, 0, nullptr, 0, s.hid3, nullptr, s.M, 2 * D, D, s.rpt,
                                   s.n_tiles, G, s.eps, 0, 0, s.dbg, s.err_ws, s.err_host, s.spin_log2};
                    if (wv < 4) ok = x3_phase<NP, X3_EPI_GELU, true, 2, X3_T0, true>(a, smem, tidp, wv, 0, tile, tnp, ctr, need);
                    else ok = x3_phase<NP, X3_EPI_GELU, true, 2, NT - X3_T0, true>(a, smem, tidp, wv, X3_T0, tile, tnp, ctr, need);
                    break;
                }
                default: {  // proj (ph & 3 == 1, A = attention output, K = D) and fc2 (A = hidden, K = 2D): one body for both
                    const bool fc2 = (ph & 3) == 3;
                    const int K = fc2 ? 2 * D : D;
                    const char* w3 = fc2 ? w[3] : w[1];
                    const float* v = vecs(w3, D, K);
                    const X3Args a{fc2 ? s.hid3 : s.att3, w3, v, v + D, nullptr, s.x, D, s.x, D, s.x3, s.stats, s.M, D, K, s.rpt, s.n_tiles,
                                   G, s.eps, 0, 0, s.dbg, s.err_ws, s.err_host, s.spin_log2};
                    if (wv < 4) ok = x3_phase<NP, X3_EPI_RES, false, 1, X3_T0, true>(a, smem, tidp, wv, 0, tile, tnp, ctr, need);
                    else ok = x3_phase<NP, X3_EPI_RES, false, 1, NT - X3_T0, true>(a, smem, tidp, wv, X3_T0, tile, tnp, ctr, need);
                    break;
                }
            }
            if (!ok) return;          // a partner of this team was lost (error words are set): never go on with stale operands
        }
    }
}

template <int NP, int EPI, bool LNF, int NPASS>
static int launch_x3(const X3Args& a, hipStream_t s) {
    constexpr int LDS = X3_LDS_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    static std::atomic<bool> attr_set[64];   // set-once flags: a racing second hipFuncSetAttribute is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)x3_gemm_kernel<NP, EPI, LNF, NPASS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
            hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((x3_gemm_kernel<NP, EPI, LNF, NPASS>), dim3(a.grid_m * a.grid_n), dim3(512), LDS, s, a);
    return hip_check_launch();
}

// C (fp32, optional) and / or C3 (operand of width N for the next GEMM, optional) = epi( LN?(A) . W^T + bias ) from
// packed operands (np = 3: split fp32, np = 1: bf16).  ln: the weight operand was built with LayerNorm folded
// (launch_split_bf16x3 with ln_w) and `stats` holds the slice partials of the K-wide rows behind A3.  rpt = rows per
// row tile of A3 (and of C3).
template <int NP>
static int launch_x3_gemm_np(const X3Args& a0, bool ln, int epi, hipStream_t s) {
    X3Args a = a0;
    // two adjacent column groups per workgroup whenever the groups pair up (one A stage feeds two W stages): chosen by
    // the SHAPE only, never by the row count, so that a row's arithmetic does not depend on the batch size
    const bool pair = epi != MPL_EPI_BIAS_RESIDUAL && (a.grid_n & 1) == 0;
    if (pair) a.grid_n /= 2;
    switch (epi) {
        case MPL_EPI_BIAS:
            if (pair) return ln ? launch_x3<NP, X3_EPI_BIAS, true, 2>(a, s) : launch_x3<NP, X3_EPI_BIAS, false, 2>(a, s);
            return ln ? launch_x3<NP, X3_EPI_BIAS, true, 1>(a, s) : launch_x3<NP, X3_EPI_BIAS, false, 1>(a, s);
        case MPL_EPI_BIAS_GELU:
            if (pair) return ln ? launch_x3<NP, X3_EPI_GELU, true, 2>(a, s) : launch_x3<NP, X3_EPI_GELU, false, 2>(a, s);
            return ln ? launch_x3<NP, X3_EPI_GELU, true, 1>(a, s) : launch_x3<NP, X3_EPI_GELU, false, 1>(a, s);
        case MPL_EPI_BIAS_RESIDUAL:
            return ln ? launch_x3<NP, X3_EPI_RES, true, 1>(a, s) : launch_x3<NP, X3_EPI_RES, false, 1>(a, s);
        default:
            return MPL_E_INVALID;
    }
}

int launch_x3_gemm(const unsigned short* A3, const unsigned short* W3, bool ln, const float* stats, float eps, const float* R,
                   int ldr, float* C, int ldc, unsigned short* C3, float* stats_out, int M, int N, int K, int rpt, int epi, int np,
                   hipStream_t s) {
    if (M <= 0 || !A3 || !W3 || (!C && !C3) || !x3_shape_ok(N, K) || rpt <= 0 || rpt > BM || (np != 1 && np != 3)) return MPL_E_INVALID;
    if (ln && !stats) return MPL_E_INVALID;
    if (epi == MPL_EPI_BIAS_RESIDUAL && !R) return MPL_E_INVALID;
    if (stats_out && epi != MPL_EPI_BIAS_RESIDUAL) return MPL_E_INVALID;
    if (C3 && N % (4 * BN)) return MPL_E_INVALID;           // an operand output must itself be a valid operand width
    const char* w3 = reinterpret_cast<const char*>(W3);
    const float* vec = reinterpret_cast<const float*>(w3 + (size_t)(N / BN) * x3_stages(K, np) * X3_W);
    const X3Args a{reinterpret_cast<const char*>(A3), w3, vec, vec + N, stats, R, ldr, C, ldc, reinterpret_cast<char*>(C3), stats_out,
                   M, N, K, rpt, (M + rpt - 1) / rpt, N / BN, eps, 0, 0, g_x3_dbg.load(), nullptr, nullptr, 0};
    return np == 3 ? launch_x3_gemm_np<3>(a, ln, epi, s) : launch_x3_gemm_np<1>(a, ln, epi, s);
}

// The whole block stack in one launch (see x3_stack_kernel).  `ops` = n_apps x {qkv, proj, fc1, fc2} packed operands.
template <int NP>
static int launch_stack_np(const X3StackArgs& a, int dev, hipStream_t s) {
    constexpr int LDS = X3_LDS_BYTES;
    static std::atomic<bool> attr_set[64];
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)x3_stack_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    // Every workgroup of this kernel must be resident while it runs (teams spin on their partners), and a workgroup
    // takes a whole CU (156 KiB of LDS): two of these launches on different streams of one device would each get part
    // of the CUs and starve each other.  In-stream order does not cover other streams, so the library serialises ITS OWN
    // stack launches per device: each one waits for the event recorded behind the previous one, whatever stream that was
    // on (a no-op on the same stream).  Other processes on the same GPU are not covered: single tenant;
    // a wait that runs out anyway is reported, never ignored (X3Args::err_*).
    if (int rcc = refuse_stream_capture(s)) return rcc;     // the cross-stream event chain cannot be captured into a graph
    hipEvent_t ev = stack_chain_event(dev);
    if (!ev) return MPL_E_LAUNCH;
    std::lock_guard<std::mutex> g(stack_chain_mutex(dev));
    if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) return MPL_E_LAUNCH;
    int rc;
    {
        ProfScope prof(MPL_K_GEMM, s);
        hipLaunchKernelGGL(x3_stack_kernel<NP>, dim3(((a.n_teams + 7) / 8) * 8 * a.G), dim3(512), LDS, s, a);
        rc = hip_check_launch();
    }
    if (hipEventRecord(ev, s) != hipSuccess) return MPL_E_LAUNCH;
    return rc;
}

int launch_x3_stack(float* x, int M, int D, int n_tok, int heads, const unsigned short* const* ops, int n_apps,
                    unsigned short* x3, unsigned short* att3, unsigned short* hid3, float* stats, unsigned* counters, float eps,
                    int stop_after, int np, bool counters_zeroed, hipStream_t s) {
    if (!x || !ops || !x3 || !att3 || !hid3 || !stats || !counters || M <= 0 || n_apps <= 0 || n_apps > MPL_MAX_APPS ||
        !x3_attention_fusable(n_tok, D, heads) || !x3_shape_ok(D, 2 * D) || M % n_tok || (np != 1 && np != 3))
        return MPL_E_INVALID;
    static std::atomic<int> resident[64];   // workgroups of this kernel the device holds at once (0 = not asked yet)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!resident[dev].load(std::memory_order_acquire)) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return MPL_E_LAUNCH;
        // the 160 KiB of LDS admit exactly one workgroup per CU -- asked, not assumed: a device (or a runtime limit) that
        // cannot hold even one makes the persistent form impossible
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)x3_stack_kernel<3>, 512, X3_LDS_BYTES) != hipSuccess || per_cu < 1)
            return MPL_E_UNSUPPORTED;
        resident[dev].store(cus, std::memory_order_release);
    }
    X3StackArgs a;
    a.x3 = reinterpret_cast<char*>(x3);
    a.att3 = reinterpret_cast<char*>(att3);
    a.hid3 = reinterpret_cast<char*>(hid3);
    a.x = x;
    a.stats = stats;
    a.counters = counters;
    a.M = M; a.D = D; a.n_tok = n_tok; a.heads = heads;
    a.rpt = x3_rows_per_tile(n_tok);
    a.n_tiles = (M + a.rpt - 1) / a.rpt;
    a.G = D / BN;
    const int cap = resident[dev].load() / a.G;
    if (cap < 1) return MPL_E_UNSUPPORTED;
    a.n_teams = a.n_tiles < cap ? a.n_tiles : cap;
    if (a.n_teams * a.G > X3_MAX_WGS) a.n_teams = X3_MAX_WGS / a.G;
    a.n_apps = n_apps;
    a.n_phases = (stop_after > 0 && stop_after < 4 * n_apps) ? stop_after : 4 * n_apps;
    a.eps = eps;
    a.dbg = g_x3_dbg.load();
    a.err_ws = counters + a.n_tiles;
    a.err_host = device_error_word(dev);
    a.spin_log2 = g_x3_spin_log2.load();
    a.inject = take_fault_injection();       // one-shot test hook (mpl_x3_spin_limit)
    for (int i = 0; i < n_apps; ++i)
        for (int j = 0; j < 4; ++j) {
            if (!ops[4 * i + j]) return MPL_E_INVALID;
            a.w[i][j] = reinterpret_cast<const char*>(ops[4 * i + j]);
        }
    // per-call state: the arrival counter of every row tile (the entry kernel of the stack has already zeroed them when
    // it ran in front of this launch)
    if (!counters_zeroed && hipMemsetAsync(counters, 0, (size_t)(a.n_tiles + 1) * sizeof(unsigned), s) != hipSuccess) return MPL_E_LAUNCH;
    return np == 3 ? launch_stack_np<3>(a, dev, s) : launch_stack_np<1>(a, dev, s);
}

// rows per tile for the fused attention: whole sequences of n_tok tokens in at most 64 rows
int x3_rows_per_tile(int n_tok) { return (n_tok >= 1 && n_tok <= BM) ? (BM / n_tok) * n_tok : 0; }

bool x3_attention_fusable(int n_tok, int dim, int heads) {
    if (n_tok < 1 || n_tok > 32 || heads <= 0 || dim % heads || !x3_shape_ok(3 * dim, dim)) return false;
    const int hd = dim / heads;
    if (BN % hd || (hd & 3)) return false;
    const int S = BM / n_tok, HP = BN / hd;
    return (size_t)(BM * X3_ATT_TS + S * HP * n_tok * n_tok) * sizeof(float) <= (size_t)X3_NST * X3_STAGE;
}

// LN1 + qkv projection + softmax attention in one launch: att3 (A3 of width D) from x3 (A3 of width D)
int launch_x3_qkv_attention(const unsigned short* A3, const unsigned short* W3, const float* stats, float eps, int M, int D,
                            int n_tok, int heads, unsigned short* att3, int np, hipStream_t s) {
    if (!x3_attention_fusable(n_tok, D, heads) || !A3 || !W3 || !stats || !att3 || M <= 0 || M % n_tok || (np != 1 && np != 3))
        return MPL_E_INVALID;
    const int N = 3 * D, rpt = x3_rows_per_tile(n_tok);
    const char* w3 = reinterpret_cast<const char*>(W3);
    const float* vec = reinterpret_cast<const float*>(w3 + (size_t)(N / BN) * x3_stages(D, np) * X3_W);
    X3Args a{reinterpret_cast<const char*>(A3), w3, vec, vec + N, stats, nullptr, 0, nullptr, 0, reinterpret_cast<char*>(att3),
             nullptr, M, N, D, rpt, (M + rpt - 1) / rpt, D / BN, eps, n_tok, D / heads, g_x3_dbg.load(), nullptr, nullptr, 0};
    return np == 3 ? launch_x3<3, X3_EPI_ATT, true, 3>(a, s) : launch_x3<1, X3_EPI_ATT, true, 3>(a, s);
}

}  // namespace mpl
