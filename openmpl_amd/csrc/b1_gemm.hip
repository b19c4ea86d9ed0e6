// bf16 GEMMs of the FPT block stack ("bf16" matmul precision, BASELINE.json configs[2]: CMU Panoptic, V = 8, bf16):
//     C = epi( LN(A) . W^T + bias ),   operands rounded to bf16, products exact, fp32 accumulation
//
// Reference ops (MPL/lib/models/multiview_mpl.py): Block.norm1 + Attention.qkv :55 + Attention.forward :55-64,
// Attention.proj :65 + residual :90, Block.norm2 + Mlp.fc1 + GELU :32-33, Mlp.fc2 :35 + residual :91.
//
// The engine is the stage of h2_gemm.hip (h2_phase.hpp) with ONE part per operand (NP = 1).  The second KiB of every 2-KiB
// fragment slot, which holds the lo part of an fp16x2 operand, holds the NEXT k-tile here: a stage is a PAIR of k-tiles
// (K = 64), moves exactly the pieces of an h2 stage through the same ring and issues one v_mfma_f32_16x16x32_bf16 per k-tile
// and column tile.  Round 2's bf16 engine (removed in round 5; one part: three k-tiles per 39-KiB stage, ring of 4, one row tile
// per team, 1.44 KiB of operand per MFMA) was bound by the L2 -> LDS path; this one walks PAIRS of row tiles in EVERY phase
// whenever a team owns two (V = 8, B = 1024: 128 row tiles = exactly two per team) -- every W k-tile pair is fetched and read
// from LDS once for both tiles, 34 KiB per 36 MFMAs per SIMD = 0.94 KiB per MFMA -- and, with one part per operand, the
// fragments of a two-tile stage fit the register file beside the THREE accumulator sets of the qkv + attention phase and the
// two of fc1, which the fp16x2 engine has to run tile by tile.
//
// Rounding points (oracle/mpl_oracle.py block_bf16, unchanged from round 2):
//   * weights: bf16(gamma_k W_nk) for the LayerNorm GEMMs (one fp32 product, then the rounding), bf16(W_nk) for proj / fc2;
//   * activations: bf16 when an epilogue hands them to the next GEMM -- the attention output, the GELU output, and x itself:
//     the residual epilogues write x twice, fp32 (the residual stream, LayerNorm statistics) and as the packed bf16 operand
//     x16 of the next LayerNorm GEMM (2 B per element on the wire instead of the 4 B of the raw rows the h2 engine reads);
//   * the LayerNorm is FOLDED: LN(x) . W^T + b = rstd (x16 . (gamma o W)^T - mean s_n) + c_n, s_n = sum_k bf16(gamma_k W_nk),
//     c_n = b_n + sum_k beta_k W_nk; mean / rstd from the fp32 slice partials the residual epilogues emit;
//   * statistics, softmax, GELU, residual stream, output: fp32.
//
// Layouts (K = 136 G columns; KT = K / 32 k-tiles in the k permutation of h2_gemm.hip; KS = ceil(KT / 2) stages, an odd KT is
// padded with one zero k-tile):
//   A1[row tile][4 row groups][KS][2 k-tiles][64 lanes][8 bf16]      (x16, att1, hid1)
//   W1[N/136][KS][9 slots][2 k-tiles][64 lanes][8 bf16]  followed by fp32 vectors c[N], s[N] (and unused space up to the
//       trailer size of an h2 operand, so that both engines address the trailer alike)
#include <stdlib.h>

#include <mutex>

#include "h2_phase.hpp"

namespace mpl {

// ---------------------------------------------------------------------------------------------- weight operand
// one wave per output column n: c_n = b_n + sum_k beta_k W_nk and s_n = sum_k of the ROUNDED gamma_k W_nk (fp64 sums)
__global__ __launch_bounds__(256) void b1_fold_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ bias, int N, int K,
                                                       float* __restrict__ tr) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    double sv = 0.0, c = 0.0;
    if (gamma) {
        for (int k = lane; k < K; k += 64) {
            const float w = W[(size_t)n * K + k];
            sv += (double)(float)(__bf16)(w * gamma[k]);
            c += (double)w * (double)beta[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sv += __shfl_xor(sv, o, 64);
            c += __shfl_xor(c, o, 64);
        }
    }
    if (lane == 0) {
        tr[n] = (float)(c + (double)bias[n]);
        tr[N + n] = (float)sv;
    }
}
__global__ __launch_bounds__(256) void b1_pack_w_kernel(const float* __restrict__ W, const float* __restrict__ gamma, int N, int K,
                                                         bf16x8* __restrict__ dst, size_t total) {
    const int G = K / BN, KT = K / BK, KS = h2_ksteps(K, 1);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const int par = (int)((idx >> 6) & 1);
        const int slot = (int)((idx >> 7) % NT);
        const int ks = (int)((idx / (128 * NT)) % KS);
        const int g = (int)(idx / ((size_t)128 * NT * KS));
        const int li = lane & 15, kq = lane >> 4;
        const int c = h2_slot_tile(slot) * 16 + li;
        const int kt = 2 * ks + par;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = 0.f;
            if (c < BN && kt < KT) {
                const int k = h2_col(kt, kq, j, G);
                const float w = W[(size_t)(g * BN + c) * K + k];
                x[j] = gamma ? w * gamma[k] : w;           // LayerNorm gain folded into the weight (one fp32 rounding)
            }
        }
        dst[idx] = to_bf16x8(x);        // idx = ((g KS + ks) 18 + slot 2 + par) 64 + lane
    }
}
int launch_pack_b1(const float* W, int N, int K, const float* ln_w, const float* ln_b, const float* bias, unsigned short* dst,
                   hipStream_t s) {
    if (!W || !dst || !bias || !h2_shape_ok(N, K) || ((ln_w != nullptr) != (ln_b != nullptr))) return MPL_E_INVALID;
    if (ln_w && K > 1088) return MPL_E_UNSUPPORTED;      // the kernel combines at most 8 slice partials per row (K = 136 x 8)
    const int KS = h2_ksteps(K, 1);
    float* tr = reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + (size_t)(N / BN) * KS * H2_W);
    ProfScope prof(MPL_K_PACK, s);
    hipLaunchKernelGGL(b1_fold_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, ln_w, ln_b, bias, N, K, tr);
    const size_t total = (size_t)(N / BN) * KS * NT * 128;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(b1_pack_w_kernel, dim3(grid), dim3(256), 0, s, W, ln_w, N, K, reinterpret_cast<bf16x8*>(dst), total);
    return hip_check_launch();
}

// ---------------------------------------------------------------------------------------------- entry of a stack
// One launch in front of the persistent kernel: blocks < nb_pack write the rows as the packed operand x16 (every strip of
// every allocated row tile -- an even number, see h2_act_bytes -- incl. the rows and the k-tile that pad it: zeros); the
// blocks behind them the LayerNorm slice partials {mean, M2} of the rows (one wave per row, two-pass per 136-column slice);
// block 0 also zeroes the arrival counters + the error word of the call.
__global__ __launch_bounds__(256) void b1_entry_kernel(const float* __restrict__ X, int M, int K, int ldx, int rpt, char* __restrict__ dst,
                                                        size_t total, int nb_pack, float* __restrict__ stats,
                                                        unsigned* __restrict__ counters, int n_counters) {
    if ((int)blockIdx.x >= nb_pack) {
        const int lane = threadIdx.x & 63;
        const int row = ((int)blockIdx.x - nb_pack) * 4 + (threadIdx.x >> 6);
        if (row >= M) return;
        const int ns = K / BN;
        const bool on = lane < BN / 4;
        for (int s0 = 0; s0 < ns; s0 += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = float4{0.f, 0.f, 0.f, 0.f};
                if (on && s0 + u < ns) v[u] = ld4(X + (size_t)row * ldx + (s0 + u) * BN + 4 * lane);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (s0 + u >= ns) break;
                const float mean = wave_sum((v[u].x + v[u].y) + (v[u].z + v[u].w)) / (float)BN;
                const float a = v[u].x - mean, b = v[u].y - mean, c = v[u].z - mean, d = v[u].w - mean;
                const float ss = wave_sum(on ? (a * a + b * b) + (c * c + d * d) : 0.f);
                if (lane == 0) {
                    stats[((size_t)row * ns + s0 + u) * 2] = mean;
                    stats[((size_t)row * ns + s0 + u) * 2 + 1] = ss;
                }
            }
        }
        return;
    }
    if (blockIdx.x == 0 && counters)
        for (int i = threadIdx.x; i < n_counters; i += 256) counters[i] = 0u;
    const int G = K / BN, KT = K / BK, KS = h2_ksteps(K, 1);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)nb_pack * 256) {
        const int lane = (int)(idx & 63);
        const int kt = (int)((idx >> 6) % (2 * KS));
        const size_t rgi = idx / ((size_t)128 * KS);          // tile * 4 + row group
        const int li = lane & 15, kq = lane >> 4;
        const int rl = (int)(rgi & 3) * 16 + li;
        const size_t row = (rgi >> 2) * rpt + rl;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
        if (rl < rpt && row < (size_t)M && kt < KT) {
            const float* src = X + row * ldx;
            const float4 p = ld4(src + h2_col(kt, kq, 0, G)), q = ld4(src + h2_col(kt, kq, 4, G));
            x[0] = p.x; x[1] = p.y; x[2] = p.z; x[3] = p.w; x[4] = q.x; x[5] = q.y; x[6] = q.z; x[7] = q.w;
        }
        *reinterpret_cast<bf16x8*>(dst + idx * 16) = to_bf16x8(x);       // byte (rgi KS + kt / 2) 2048 + (kt & 1) 1024 + lane 16
    }
}
int launch_b1_entry(const float* X, int M, int K, int ldx, int rpt, unsigned short* x16, float* stats, unsigned* counters,
                    int n_counters, hipStream_t s) {
    const size_t bytes = h2_act_bytes(M, K, rpt, 1);
    if (!X || !x16 || bytes == 0 || (ldx & 3) || (counters && !stats)) return MPL_E_INVALID;
    const size_t total = bytes / 16;
    const int nb_pack = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const int grid = nb_pack + (stats ? (M + 3) / 4 : 0);
    ProfScope prof(MPL_K_ROW_STATS, s);
    hipLaunchKernelGGL(b1_entry_kernel, dim3(grid), dim3(256), 0, s, X, M, K, ldx, rpt, reinterpret_cast<char*>(x16), total, nb_pack,
                       stats, counters, n_counters);
    return hip_check_launch();
}

// ---------------------------------------------------------------------------------------------- launches
template <int EPI, bool LNF, int NPASS>
static int launch_b1(const H2Args& a, hipStream_t s) {
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)h2_gemm_kernel<EPI, LNF, NPASS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((h2_gemm_kernel<EPI, LNF, NPASS, 1>), dim3(a.grid_m * a.grid_n), dim3(512), H2_LDS_BYTES, s, a);
    return hip_check_launch();
}

// One GEMM of a block application as a launch of its own (mpl_x3_stack_mode(1): the A/B form of the stack): fc1 (ln, GELU, two
// column groups per workgroup) or proj / fc2 (residual; C = the fp32 rows, C1 = their packed copy, stats_out = their partials)
int launch_b1_gemm(const unsigned short* A1, const unsigned short* W1, bool ln, const float* stats, float eps, const float* R, int ldr,
                   float* C, int ldc, unsigned short* C1, float* stats_out, int M, int N, int K, int rpt, int epi, hipStream_t s) {
    if (M <= 0 || !A1 || !W1 || (!C && !C1) || !h2_shape_ok(N, K) || rpt <= 0 || rpt > BM) return MPL_E_INVALID;
    if (ln && (!stats || K > 1088)) return MPL_E_INVALID;
    const char* w1 = reinterpret_cast<const char*>(W1);
    const float* vec = reinterpret_cast<const float*>(w1 + (size_t)(N / BN) * h2_ksteps(K, 1) * H2_W);
    H2Args a{reinterpret_cast<const char*>(A1), nullptr, 0, w1, vec, vec + N, stats, nullptr, nullptr, R, ldr, C, ldc, reinterpret_cast<char*>(C1),
             stats_out, M, N, K, rpt, (M + rpt - 1) / rpt, N / BN, eps, 0, 0, h2_debug_buffer(), nullptr, nullptr, 0};
    if (epi == MPL_EPI_BIAS_GELU && ln && (a.grid_n & 1) == 0 && C1 && !C) {
        a.grid_n /= 2;
        return launch_b1<H2_EPI_GELU, true, 2>(a, s);
    }
    if (epi == MPL_EPI_BIAS_RESIDUAL && !ln && R && C) return launch_b1<H2_EPI_RES, false, 1>(a, s);
    return MPL_E_UNSUPPORTED;
}

// LN1 + qkv projection + softmax attention in one launch: att1 (packed bf16, width D) from the packed rows x16
int launch_b1_qkv_attention(const unsigned short* x16, const unsigned short* W1, const float* stats, float eps, int M, int D, int n_tok,
                            int heads, unsigned short* att1, hipStream_t s) {
    if (!h2_attention_fusable(n_tok, D, heads) || !x16 || !W1 || !stats || !att1 || M <= 0 || M % n_tok) return MPL_E_INVALID;
    const int N = 3 * D, rpt = h2_rows_per_tile(n_tok);
    const char* w1 = reinterpret_cast<const char*>(W1);
    const float* vec = reinterpret_cast<const float*>(w1 + (size_t)(N / BN) * h2_ksteps(D, 1) * H2_W);
    H2Args a{reinterpret_cast<const char*>(x16), nullptr, 0, w1, vec, vec + N, stats, nullptr, nullptr, nullptr, 0, nullptr, 0,
             reinterpret_cast<char*>(att1), nullptr, M, N, D, rpt, (M + rpt - 1) / rpt, D / BN, eps, n_tok, D / heads, h2_debug_buffer(),
             nullptr, nullptr, 0};
    return launch_b1<H2_EPI_ATT, true, 3>(a, s);
}

int launch_b1_stack(float* x, unsigned short* x16, int M, int D, int n_tok, int heads, const unsigned short* const* ops, int n_apps,
                    unsigned short* att1, unsigned short* hid1, float* stats, unsigned* counters, float eps, int stop_after,
                    hipStream_t s) {
    return h2_launch_stack<1>(x, x16, M, D, n_tok, heads, ops, n_apps, att1, hid1, stats, counters, eps, stop_after, s);
}

}  // namespace mpl
