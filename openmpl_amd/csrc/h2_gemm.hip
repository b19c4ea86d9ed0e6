// fp32 GEMM on the fp16 matrix cores by two-way operand splitting ("h2"), the default engine of the FPT block stack:
//     C = epi( LN(A) . W^T + bias )
//
// Reference ops (MPL/lib/models/multiview_mpl.py): Block.norm1 + Attention.qkv :55 + Attention.forward :55-64,
// Attention.proj :65 + residual :90, Block.norm2 + Mlp.fc1 + GELU :32-33, Mlp.fc2 :35 + residual :91.
//
// Arithmetic.  The packed 16-bit matrix pipe of gfx950 is 16x faster than the fp32 one, so an fp32 product is formed from
// 16-bit pieces.  Round 2 (removed in round 5) used three bf16 pieces per operand and six partial
// products.  This engine uses TWO fp16 pieces and THREE partial products -- the "3xTF32" scheme of the CUDA world on the
// fp16 pipe (fp16 has the 11-bit significand of TF32):
//     x = hi + lo (+ eps),  hi = fp16(x),  lo = fp16(x - hi)      |eps| <= 2^-22 |x|   (x - hi is exact in fp32)
//     x . w  ~=  lo.hi + hi.lo + hi.hi        (A part . W part, fp32 accumulation in v_mfma_f32_16x16x32_f16)
// the dropped lo.lo term is <= 2^-22 relative.  Measured against fp64 on the bench model the result is as far from the
// truth as an fp32 GEMM is (the fp32 ACCUMULATION error dominates both): single GEMM 5.5e-7 (h2) / 5.2e-7 (fp32) /
// 2.8e-7 (x3) max-scaled; whole forward 7.9e-7 / 7.9e-7 / 7.9e-7 (tools/h2_emulate.py, tests/test_h2_gpu.py).
// Half the matrix instructions and two thirds of the operand bytes of x3 -- and the operand bytes through the LDS-DMA path
// are what bounds the k loop (tools/h2_probe.hip: 19-42 cycles per KiB and CU, by source).
//
// fp16 has a 5-bit exponent, so every operand is brought into its window by an EXACT power-of-two scale that the epilogue
// takes out again (no rounding anywhere):
//   * weights: one scale per output column n, max_k |W_nk| sw_n in [2^13, 2^14)   (computed when the binding packs them);
//   * the input of a LayerNorm GEMM (qkv, fc1) is normalised BEFORE it is split: z = (x - mean) rstd 2^10, |z| <= sqrt(K)
//     2^10 < 65504 for every K <= 2048 -- the gain gamma is folded into W, beta and the bias into c:
//         LN(x) . W^T + b  =  2^-10 sw_n^-1 ( z . (sw_n gamma o W_n) ) + c_n,     c_n = b_n + sum_k beta_k W_nk ;
//   * the inputs of proj and fc2 (attention output, GELU output) use one STATIC scale PER COLUMN (= per k of the consumer)
//     from a bound that needs no data: |LN(x) . W_n + b_n| <= bound_n = sqrt(K) |gamma o W_n|_2 + |c_n| (Cauchy-Schwarz,
//     |LN(x)|_2 <= sqrt(K)); the attention output is a convex combination of v rows (column by column) and |gelu(t)| <= |t|,
//     so the bound of the producing column holds for both.  so_n = the largest power of two with so_n bound_n <= 2^15 is
//     stored behind the producer's fragments; the producing epilogue multiplies column n by so_n, and the CONSUMER's weights
//     are packed as W_mk / so_k (exact), their column scale sw_m taken afterwards -- the operands are column-equilibrated, so
//     one outlier channel (a huge v bias, one fc1 row 1e4 x the others) costs no other channel anything (round 3 used one
//     scale per layer: the outlier set the window of every column; tests/test_robust_gpu.py).  Typical values sit sqrt(K)
//     below their bound, i.e. at >= 2^10 of a 2^15 window; fp16 subnormals are honoured by the MFMA and by v_cvt
//     (tools/h2_probe.hip), so the absolute resolution is 2^-24 2^-11 of the window.
// Nothing overflows for ANY input; the unit-test entry (mpl_ln_linear_h2) scales a plain A operand by its measured amax.
//
// Data flow of a block application (x = fp32 residual stream, the ONLY activation kept in fp32):
//   qkv : A = x (fp32 rows, LDS-DMA'd raw into the stage; each of the two waves that multiply a row group reads its lane's
//         32 B of fp32 and normalises + splits them in registers), epilogue = attention in registers -> att2 (packed)
//   proj: A = att2 (packed, LDS-DMA), epilogue: x += ..., LayerNorm slice partials
//   fc1 : A = x as above, epilogue GELU -> hid2 (packed);   fc2: A = hid2, epilogue: x += ..., partials
// x3 handed x from GEMM to GEMM as a split copy; here the residual epilogues write only fp32 x (write-through) and the
// consumers split it themselves: 4 B per element on the wire instead of 4 + 6, and no two-step scale exchange.
//
// Layouts (K = 136 G columns, G a multiple of 4; KT = K / 32 k-tiles; the k permutation is x3's):
//   k-tile t < 4G:  lane (i, kq) element j <-> column 136 (t/4) + 32 (t%4) + 16 (j/4) + 4 kq + (j%4);  t = 4G + u: 136 (4u + kq) + 128 + j
//   A2[row tile][4 row groups][KT][2 parts][64 lanes][8 fp16]     lane = 16 kq + i, i = row in the 16-row group
//   W2[N/136][KT][9 slots][2 parts][64 lanes][8 fp16]             slot s = column tile {0,1,2,3,8,4,5,6,7}[s]
//       followed by fp32 vectors c[N], sc[N] (epilogue multiplier), sw[N], bound[N], so[N] (static scale of output column n
//       when it travels on as a packed operand) and meta[8] = {.., fingerprints of so / of the input scales, see h2_meta_kernel}
// Workgroup = one row tile x 136 columns (x NPASS column groups), 8 waves: wave w owns row group w & 3 and slots 0..4
// (w < 4) or 5..8.  Stage = A 8 KiB + W 18 KiB = 26 KiB, ring of 6 (156 KiB, one workgroup per CU).
// The k order of every output element is fixed: results do not depend on batch size or launch geometry.
#include <stdlib.h>

#include <mutex>

#include "h2_phase.hpp"

namespace mpl {

// A folded LayerNorm needs K <= 1088 (the kernel combines at most 8 slice partials per row; the static 2^10 scale of a normalised row, |z| <= sqrt(K) 2^10, would allow K <= 2048);
// launch_pack_h2 checks that; the packed layout itself exists for every multiple of 544
bool h2_shape_ok(int N, int K) { return N > 0 && K > 0 && N % BN == 0 && K % (4 * BN) == 0 && K <= 8704; }

size_t h2_operand_bytes(int N, int K, int np) {
    if (!h2_shape_ok(N, K) || (np != 1 && np != 2)) return 0;
    return (size_t)(N / BN) * h2_ksteps(K, np) * H2_W + ((size_t)H2_TRV * N + 8) * sizeof(float);
}
size_t h2_act_bytes(int M, int K, int rpt, int np) {
    if (M <= 0 || K <= 0 || K % (4 * BN) || rpt <= 0 || rpt > BM || (np != 1 && np != 2)) return 0;
    // an even number of row tiles: the two-tile stage (h2_stack2_kernel) requests the strips of a pair, also of an absent partner
    const size_t tiles = ((((size_t)M + rpt - 1) / rpt) + 1) / 2 * 2;
    return tiles * 4 * h2_ksteps(K, np) * H2_RG;
}
int h2_rows_per_tile(int n_tok) { return (n_tok >= 1 && n_tok <= BM) ? (BM / n_tok) * n_tok : 0; }
bool h2_attention_fusable(int n_tok, int dim, int heads) {
    if (n_tok < 1 || n_tok > 32 || heads <= 0 || dim % heads || !h2_shape_ok(3 * dim, dim)) return false;
    const int hd = dim / heads;
    if (BN % hd || (hd & 3)) return false;
    const int S = BM / n_tok, HP = BN / hd;
    return (size_t)(BM * H2_ATT_TS + S * HP * n_tok * n_tok) * sizeof(float) <= (size_t)H2_NST * H2_STAGE;
}

// ---------------------------------------------------------------------------------------------- weight operand
// trailer floats behind the fragments: c[N] | sc[N] | sw[N] | bound[N] | so[N] | meta[8]
// One wave per output column n: the column scale sw_n (max |f o W_n| sw_n in [2^13, 2^14); f = the LayerNorm gain, or the
// reciprocal of the static scales the A operand arrives with, or 1), c_n = b_n + sum_k beta_k W_nk (fp64 sum), the epilogue
// multiplier sc_n = 1 / (sa sw_n) (sa = the static scale of a normalised LayerNorm input, or 1), the data-free bound of
// |out_n| and the static scale so_n it implies for column n as an operand (LayerNorm GEMMs only; see the head of the file).
__device__ __forceinline__ float h2_in_factor(const float* gamma, const float* in_scale, int k) {
    return gamma ? gamma[k] : (in_scale ? 1.0f / in_scale[k] : 1.0f);      // in_scale: powers of two, the reciprocal is exact
}
__global__ __launch_bounds__(256) void h2_fold_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ bias,
                                                       const float* __restrict__ in_scale, int N, int K, float* __restrict__ tr) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    double ss = 0.0, c = 0.0;
    float amax = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)n * K + k];
        const float wg = (gamma || in_scale) ? w * h2_in_factor(gamma, in_scale, k) : w;
        amax = fmaxf(amax, fabsf(wg));
        ss += (double)wg * (double)wg;
        if (gamma) c += (double)w * (double)beta[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o, 64);
        c += __shfl_xor(c, o, 64);
        amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    }
    if (lane == 0) {
        float sw = 1.0f;
        if (amax > 0.f && amax < 3.0e38f) {
            int e;
            (void)frexpf(amax, &e);            // amax = m 2^e, m in [0.5, 1): amax 2^(14 - e) in [2^13, 2^14)
            e = 14 - e;
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            sw = ldexpf(1.0f, e);
        }
        const float cn = (float)(c + (double)bias[n]);
        const float sa = gamma ? H2_SA : 1.0f;
        tr[n] = cn;
        tr[N + n] = 1.0f / (sa * sw);
        tr[2 * N + n] = sw;
        const float bound = gamma ? (float)(sqrt((double)K) * sqrt(ss)) + fabsf(cn) : 0.f;
        tr[3 * N + n] = bound;
        tr[4 * N + n] = gamma ? h2_window_scale(bound) : 1.0f;
    }
}
// meta[0..5]: the largest bound over all columns / over the last third of the columns (v of qkv), the window scales they
// imply and their reciprocals (informational since the scales went per column).  meta[6], meta[7]: FINGERPRINTS that let the
// stack check that a consumer operand was packed against the scales its producer applies -- 0.5 + the sum of the binary
// exponents (exact in fp32) of so over all columns [6] and over the last third [7] for a LayerNorm operand; for a plain
// operand [6] = the same sum over the in_scale vector it was packed with (0 without one) and [7] = 0.
__global__ __launch_bounds__(256) void h2_meta_kernel(int N, int K, int has_ln, const float* __restrict__ in_scale, float* __restrict__ tr) {
    __shared__ float red[4][256];
    float ball = 0.f, bv = 0.f, fall = 0.f, fv = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float b = tr[3 * N + n];
        const float e = (float)ilogbf(tr[4 * N + n]);
        ball = fmaxf(ball, b);
        fall += e;
        if (3 * n >= 2 * N) {
            bv = fmaxf(bv, b);
            fv += e;
        }
    }
    if (!has_ln) {
        fall = fv = 0.f;
        if (in_scale)
            for (int k = threadIdx.x; k < K; k += 256) fall += (float)ilogbf(in_scale[k]);
    }
    red[0][threadIdx.x] = ball;
    red[1][threadIdx.x] = bv;
    red[2][threadIdx.x] = fall;
    red[3][threadIdx.x] = fv;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = fmaxf(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + s]);
            red[2][threadIdx.x] += red[2][threadIdx.x + s];          // integers far below 2^24: exact in any order
            red[3][threadIdx.x] += red[3][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* m = tr + H2_TRV * N;
        const float sa = h2_window_scale(red[0][0]), sv = h2_window_scale(red[1][0]);
        m[0] = sa; m[1] = sv; m[2] = 1.0f / sa; m[3] = 1.0f / sv;
        m[4] = red[0][0]; m[5] = red[1][0];
        m[6] = (has_ln || in_scale) ? red[2][0] + 0.5f : 0.f;
        m[7] = has_ln ? red[3][0] + 0.5f : 0.f;
    }
}
__global__ __launch_bounds__(256) void h2_pack_w_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                         const float* __restrict__ in_scale, int N, int K,
                                                         const float* __restrict__ tr, f16x8* __restrict__ dst, size_t total) {
    const int G = K / BN, KT = K / BK;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const int slot = (int)((idx >> 6) % NT);
        const int kt = (int)((idx / (64 * NT)) % KT);
        const int g = (int)(idx / ((size_t)64 * NT * KT));
        const int li = lane & 15, kq = lane >> 4;
        const int c = h2_slot_tile(slot) * 16 + li;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = 0.f;
            if (c < BN) {
                const int k = h2_col(kt, kq, j, G);
                const float w = W[(size_t)(g * BN + c) * K + k];
                // gain folded (one fp32 rounding; the reciprocal input scale is exact), then the exact column scale
                x[j] = ((gamma || in_scale) ? w * h2_in_factor(gamma, in_scale, k) : w) * tr[2 * N + g * BN + c];
            }
        }
        f16x8 hi, lo;
        split2(x, hi, lo);
        f16x8* o = dst + ((size_t)(g * KT + kt) * 18 + slot * 2) * 64 + lane;
        o[0] = hi;
        o[64] = lo;
    }
}

// in_scale (optional, plain operands only): device vector of K powers of two, the static scales the A operand's columns
// arrive with (the so vector of the producing layer, h2_out_scale): the weights are packed as W_nk / in_scale_k
const float* h2_out_scale(const unsigned short* op, int N, int K) {
    if (!op || !h2_shape_ok(N, K)) return nullptr;
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(op) + (size_t)(N / BN) * (K / BK) * H2_W) + 4 * (size_t)N;
}
int launch_pack_h2(const float* W, int N, int K, const float* ln_w, const float* ln_b, const float* bias, const float* in_scale,
                   unsigned short* dst, hipStream_t s) {
    if (!W || !dst || !bias || !h2_shape_ok(N, K) || ((ln_w != nullptr) != (ln_b != nullptr))) return MPL_E_INVALID;
    if (ln_w && in_scale) return MPL_E_INVALID;          // a LayerNorm input is normalised in the k loop: it has no static scale
    if (ln_w && K > 1088) return MPL_E_UNSUPPORTED;      // the kernel combines at most 8 slice partials per row (K = 136 x 8)
    float* tr = reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + (size_t)(N / BN) * (K / BK) * H2_W);
    ProfScope prof(MPL_K_PACK, s);
    hipLaunchKernelGGL(h2_fold_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, ln_w, ln_b, bias, in_scale, N, K, tr);
    hipLaunchKernelGGL(h2_meta_kernel, dim3(1), dim3(256), 0, s, N, K, ln_w ? 1 : 0, in_scale, tr);
    const size_t total = (size_t)(N / BN) * (K / BK) * NT * 64;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(h2_pack_w_kernel, dim3(grid), dim3(256), 0, s, W, ln_w, in_scale, N, K, tr, reinterpret_cast<f16x8*>(dst), total);
    return hip_check_launch();
}

// ---------------------------------------------------------------------------------------------- entry of a stack
// One launch in front of the persistent kernel: the LayerNorm slice partials {mean, M2} of the incoming rows (one wave per
// row, two-pass per 136-column slice) and -- block 0 -- the zeroed arrival counters + error word of the call.
// With `ops` (the stack's operands) block 0 also checks that every proj / fc2 operand was packed against the static scales its
// producer applies (fingerprints in meta[6], meta[7], h2_meta_kernel): a mismatch sets the error words of the call -- the
// poses come out NaN and the device reports MPL_E_DEVICE -- instead of multiplying under the wrong scales.
__global__ __launch_bounds__(256) void h2_entry_kernel(const float* __restrict__ X, int M, int K, int ldx, float* __restrict__ stats,
                                                        unsigned* __restrict__ counters, int n_counters, const H2Ops ops) {
    if (blockIdx.x == 0 && counters) {
        for (int i = threadIdx.x; i < n_counters; i += 256) counters[i] = 0u;
        __syncthreads();
        const int D = ops.D;
        for (int i = threadIdx.x; i < ops.n_apps; i += 256) {
            auto meta = [&](const char* w2, int N, int Kw) -> const float* {
                return reinterpret_cast<const float*>(w2 + (size_t)(N / BN) * (Kw / BK) * H2_W) + H2_TRV * N;
            };
            const bool ok = meta(ops.w[i][1], D, D)[6] == meta(ops.w[i][0], 3 * D, D)[7] &&
                            meta(ops.w[i][3], D, 2 * D)[6] == meta(ops.w[i][2], 2 * D, D)[6];
            if (!ok) {
                __hip_atomic_store(counters + n_counters - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ops.err_host) __hip_atomic_store(ops.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const int row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int ns = K / BN;
    const bool on = lane < BN / 4;
    // the slices of the row are requested together (one memory round trip; one per slice made this kernel 8 us long)
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
}
// ops (optional): n_apps x {qkv, proj, fc1, fc2} operands of the stack that follows (width K), checked as described above;
// the error word of the call is counters[n_counters - 1]
int launch_h2_entry(const float* X, int M, int K, int ldx, float* stats, unsigned* counters, int n_counters,
                    const unsigned short* const* ops, int n_apps, hipStream_t s) {
    if (!X || !stats || M <= 0 || K % BN || (ldx & 3) || n_apps < 0 || n_apps > MPL_MAX_APPS || (ops && !counters)) return MPL_E_INVALID;
    H2Ops o;
    o.n_apps = ops ? n_apps : 0;
    o.D = K;
    o.err_host = nullptr;
    int dev = 0;
    if (ops && hipGetDevice(&dev) == hipSuccess) o.err_host = device_error_word(dev);
    for (int i = 0; i < o.n_apps; ++i)
        for (int j = 0; j < 4; ++j) {
            if (!ops[4 * i + j]) return MPL_E_INVALID;
            o.w[i][j] = reinterpret_cast<const char*>(ops[4 * i + j]);
        }
    ProfScope prof(MPL_K_ROW_STATS, s);
    hipLaunchKernelGGL(h2_entry_kernel, dim3((M + 3) / 4), dim3(256), 0, s, X, M, K, ldx, stats, counters, n_counters, o);
    return hip_check_launch();
}

// plain (not normalised) fp32 rows -> packed A2 with ONE measured scale (unit-test entry of a GEMM without LayerNorm):
// sc[0] <- amax bits (atomicMax over |x|), then sc[1] = scale, sc[2] = 1 / scale, operand = split2(x scale)
__global__ __launch_bounds__(256) void h2_amax_kernel(const float* __restrict__ X, size_t n, unsigned* __restrict__ sc) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(X[i]));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(sc, __builtin_bit_cast(unsigned, m));      // non-negative floats order like their bits
}
__global__ __launch_bounds__(256) void h2_pack_rows_kernel(const float* __restrict__ X, int M, int K, int ldx, int rpt,
                                                            char* __restrict__ dst, size_t total, float* __restrict__ sc) {
    const float amax = __builtin_bit_cast(float, reinterpret_cast<const unsigned*>(sc)[0]);
    const float scale = h2_window_scale(amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[1] = scale;
        sc[2] = 1.0f / scale;
    }
    const int G = K / BN, KT = K / BK;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const int kt = (int)((idx >> 6) % KT);
        const size_t rgi = idx / ((size_t)64 * KT);          // tile * 4 + row group
        const int li = lane & 15, kq = lane >> 4;
        const int rl = (int)(rgi & 3) * 16 + li;
        const size_t row = (rgi >> 2) * rpt + rl;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
        if (rl < rpt && row < (size_t)M) {
            const float* src = X + row * ldx;
            const float4 p = ld4(src + h2_col(kt, kq, 0, G)), q = ld4(src + h2_col(kt, kq, 4, G));
            x[0] = p.x * scale; x[1] = p.y * scale; x[2] = p.z * scale; x[3] = p.w * scale;
            x[4] = q.x * scale; x[5] = q.y * scale; x[6] = q.z * scale; x[7] = q.w * scale;
        }
        f16x8 hi, lo;
        split2(x, hi, lo);
        char* o = dst + (rgi * KT + kt) * H2_RG + lane * 16;
        *reinterpret_cast<f16x8*>(o) = hi;
        *reinterpret_cast<f16x8*>(o + 1024) = lo;
    }
}
// sc: 4 floats of device scratch ({amax bits, scale, 1 / scale, -}); the GEMM reads sc + 2 as its a_inv
int launch_h2_pack_rows(const float* X, int M, int K, int ldx, int rpt, unsigned short* dst, float* sc, hipStream_t s) {
    if (!X || !dst || !sc || h2_act_bytes(M, K, rpt) == 0 || (ldx & 3) || ldx != K) return MPL_E_INVALID;
    if (hipMemsetAsync(sc, 0, 16, s) != hipSuccess) return MPL_E_LAUNCH;
    const size_t n = (size_t)M * K;
    hipLaunchKernelGGL(h2_amax_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, X, n,
                       reinterpret_cast<unsigned*>(sc));
    const size_t tiles = ((size_t)M + rpt - 1) / rpt;
    const size_t total = tiles * 4 * (K / BK) * 64;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(h2_pack_rows_kernel, dim3(grid), dim3(256), 0, s, X, M, K, ldx, rpt, reinterpret_cast<char*>(dst), total, sc);
    return hip_check_launch();
}

static std::atomic<unsigned long long*> g_h2_dbg{nullptr};
void h2_set_debug_buffer(unsigned long long* p) { g_h2_dbg.store(p); }
static std::atomic<int> g_h2_spin_log2{23};
static std::atomic<int> g_h2_rt{0};          // 0: by shape; 1 / 2: force the one- / two-tile stage (A/B switch)
void h2_set_row_tiles(int rt) { g_h2_rt.store(rt); }
void h2_set_spin_log2(int v) { g_h2_spin_log2.store(v & 0xff); }
unsigned long long* h2_debug_buffer() { return g_h2_dbg.load(); }
int h2_spin_log2() { return g_h2_spin_log2.load(); }
int h2_row_tiles() { return g_h2_rt.load(); }
static std::atomic<int> g_h2_narrow{0};
void h2_set_narrow(int mode) { g_h2_narrow.store(mode & 3); }
int h2_narrow_mode() { return g_h2_narrow.load(); }
static std::atomic<int> g_h2_wt_always{getenv("MPL_WRITE_THROUGH") != nullptr ? 1 : 0};
void h2_set_write_through(int always) { g_h2_wt_always.store(always & 1); }
int h2_write_through_always() { return g_h2_wt_always.load(); }
static std::atomic<int> g_h2_direct_w{1};
void h2_set_direct_w(int on) { g_h2_direct_w.store(on & 1); }
int h2_direct_w() { return g_h2_direct_w.load(); }


template <int EPI, bool LNF, int NPASS>
static int launch_h2(const H2Args& a, hipStream_t s) {
    constexpr int LDS = H2_LDS_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)h2_gemm_kernel<EPI, LNF, NPASS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((h2_gemm_kernel<EPI, LNF, NPASS>), dim3(a.grid_m * a.grid_n), dim3(512), LDS, s, a);
    return hip_check_launch();
}

// One GEMM as a launch of its own.  ln: A = fp32 rows X (ld = K) normalised with `stats`; else A = packed A2 whose scale
// reciprocal sits at a_inv (device; NULL = the operand carries the per-column static scales W2 was packed against).
// Outputs: C fp32 (optional) and / or C2 packed under the static scales of W2's own columns (its so vector).
int launch_h2_gemm(const float* X, const unsigned short* A2, const float* a_inv, const unsigned short* W2, bool ln, const float* stats,
                   float eps, const float* R, int ldr, float* C, int ldc, unsigned short* C2, float* stats_out,
                   int M, int N, int K, int rpt, int epi, hipStream_t s) {
    if (M <= 0 || !W2 || (!C && !C2) || !h2_shape_ok(N, K) || rpt <= 0 || rpt > BM) return MPL_E_INVALID;
    if (ln ? (!X || !stats || K > 1088) : !A2) return MPL_E_INVALID;
    if (epi == MPL_EPI_BIAS_RESIDUAL && !R) return MPL_E_INVALID;
    if (stats_out && epi != MPL_EPI_BIAS_RESIDUAL) return MPL_E_INVALID;
    if (C2 && (N % (4 * BN) || !ln || epi == MPL_EPI_BIAS_RESIDUAL)) return MPL_E_INVALID;     // static scales exist for LayerNorm GEMMs only
    const char* w2 = reinterpret_cast<const char*>(W2);
    const float* vec = reinterpret_cast<const float*>(w2 + (size_t)(N / BN) * (K / BK) * H2_W);
    H2Args a{reinterpret_cast<const char*>(A2), X, K, w2, vec, vec + N, stats, a_inv, C2 ? vec + 4 * N : nullptr, R, ldr, C, ldc, reinterpret_cast<char*>(C2),
             stats_out, M, N, K, rpt, (M + rpt - 1) / rpt, N / BN, eps, 0, 0, g_h2_dbg.load(), nullptr, nullptr, 0};
    const bool pair = epi != MPL_EPI_BIAS_RESIDUAL && (a.grid_n & 1) == 0;     // by the SHAPE only (batch invariance)
    if (pair) a.grid_n /= 2;
    switch (epi) {
        case MPL_EPI_BIAS:
            if (pair) return ln ? launch_h2<H2_EPI_BIAS, true, 2>(a, s) : launch_h2<H2_EPI_BIAS, false, 2>(a, s);
            return ln ? launch_h2<H2_EPI_BIAS, true, 1>(a, s) : launch_h2<H2_EPI_BIAS, false, 1>(a, s);
        case MPL_EPI_BIAS_GELU:
            if (pair) return ln ? launch_h2<H2_EPI_GELU, true, 2>(a, s) : launch_h2<H2_EPI_GELU, false, 2>(a, s);
            return ln ? launch_h2<H2_EPI_GELU, true, 1>(a, s) : launch_h2<H2_EPI_GELU, false, 1>(a, s);
        case MPL_EPI_BIAS_RESIDUAL:
            return ln ? launch_h2<H2_EPI_RES, true, 1>(a, s) : launch_h2<H2_EPI_RES, false, 1>(a, s);
        default:
            return MPL_E_INVALID;
    }
}

// LN1 + qkv projection + softmax attention in one launch: att2 (packed, width D, scaled by meta[1] of W2) from fp32 rows
int launch_h2_qkv_attention(const float* X, const unsigned short* W2, const float* stats, float eps, int M, int D, int n_tok,
                            int heads, unsigned short* att2, hipStream_t s) {
    if (!h2_attention_fusable(n_tok, D, heads) || !X || !W2 || !stats || !att2 || M <= 0 || M % n_tok) return MPL_E_INVALID;
    const int N = 3 * D, rpt = h2_rows_per_tile(n_tok);
    const char* w2 = reinterpret_cast<const char*>(W2);
    const float* vec = reinterpret_cast<const float*>(w2 + (size_t)(N / BN) * (D / BK) * H2_W);
    H2Args a{nullptr, X, D, w2, vec, vec + N, stats, nullptr, vec + 4 * N, nullptr, 0, nullptr, 0, reinterpret_cast<char*>(att2),
             nullptr, M, N, D, rpt, (M + rpt - 1) / rpt, D / BN, eps, n_tok, D / heads, g_h2_dbg.load(), nullptr, nullptr, 0};
    return launch_h2<H2_EPI_ATT, true, 3>(a, s);
}

// The whole block stack in one launch.  `ops` = n_apps x {qkv, proj, fc1, fc2} packed operands; counters: n_tiles arrival
// counters + 1 error word, zeroed by the caller (launch_h2_entry).
// the form a stack launch of this shape takes (MPL_FORM_* of mpl_hip.h), np = operand parts (2 = fp16x2, 1 = bf16)
int h2_stack_form_code(int M, int D, int n_tok, int np, int cus) {
    const H2Form f = np == 2 ? h2_stack_form<2>(M, D, n_tok, cus) : h2_stack_form<1>(M, D, n_tok, cus);
    if (f.pairs) return MPL_FORM_PAIRS;
    if (f.rgs == 2) return MPL_FORM_ROWS32;
    if (f.rgs == 1) return f.direct ? MPL_FORM_ROWS16_DIRECT : MPL_FORM_ROWS16;
    return MPL_FORM_TEAMS;
}

int launch_h2_stack(float* x, int M, int D, int n_tok, int heads, const unsigned short* const* ops, int n_apps,
                    unsigned short* att2, unsigned short* hid2, float* stats, unsigned* counters, float eps, int stop_after,
                    hipStream_t s) {
    return h2_launch_stack<2>(x, nullptr, M, D, n_tok, heads, ops, n_apps, att2, hid2, stats, counters, eps, stop_after, s);
}

}  // namespace mpl
