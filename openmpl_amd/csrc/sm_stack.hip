// Block stack for SMALL batches (round 4): at most 32 token rows (B V <= 32: a single frame, a few frames / persons).
//
// Reference ops (MPL/lib/models/multiview_mpl.py): the `for blk in self.blocks` loop :420-423 -- Block.forward :84-92
// (x += proj(attn(qkv(norm1(x)))); x += fc2(gelu(fc1(norm2(x))))), Attention.forward :55-64, Mlp.forward :31-37.
//
// The persistent team kernel of h2_gemm.hip walks a 64-row tile through the 52 GEMMs of a stack with ONE team of D / 136
// workgroups: 4 (8) CUs stream all 120 MB of packed weights through their LDS-DMA path, 0.8 ms however few rows there are -- a
// single frame (V = 2, B = 1) costs 1.56 ms.  With so few rows the GEMMs are weight-streaming problems, and the MI355X shape of
// that is the WHOLE chip on every GEMM: a GEMM of N output columns is N / 16 independent column tiles (102 for qkv at D = 544),
// one workgroup each, separated by grid barriers (all workgroups resident: grid <= CU count, one arrival counter behind
// write-through stores, L1-bypassing loads on the consuming side, spins bounded and a lost barrier REPORTED like a lost
// hand-off of the team kernel).  Inside a workgroup:
//   * the 16 weight rows of its tile (the nn.Linear tensor in place: no packed copy) come by LDS-DMA straight into FRAGMENT
//     order -- one 1-KiB piece per 16-deep k step, lane (j, kq) fetching W[n0 + j][16 u + 4 kq ..] -- and, because weights do
//     not depend on anybody, the tile of the NEXT GEMM is requested BEFORE the grid barrier: its latency (HBM / Infinity Cache:
//     all 114 MB are touched once per forward) hides behind the barrier;
//   * the four waves split K (wave w takes the k steps u = w mod 4, exactly the pieces it requested itself: its own counted
//     wait, no workgroup barrier for the weights), the A fragments (one or two 16-row tiles of x / att / hid against every weight
//     fragment) are requested
//     together up front, LayerNorm statistics are reduced across the waves through LDS (two-pass, the row values stay in
//     registers), the four partial accumulators are added in a fixed order, wave 0 applies the epilogue.
// Arithmetic: exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32 = an fmaf chain per output), exact-erf GELU, fp32 softmax:
// the accuracy of the "fp32_mfma" engine.  The result of a pose is therefore NOT bitwise the one the fp16x2 engine gives the
// same pose in a large batch (both are within 1e-6 of the fp64 oracle); inside this engine results are bitwise independent of
// the batch.  Per block application: [LN1 + qkv] | [attention: one (sequence, head) per workgroup] | [proj + residual] |
// [LN2 + fc1 + GELU] | [fc2 + residual], five grid barriers; activations in global memory (L2): x in place, qkv, att, hid.
// (Measured and not kept: the attention of one or two frames computed redundantly by every proj workgroup into LDS, which removes
// one step and one barrier per application -- 0.401 ms per stack against 0.400 at V = 2, B = 1, slower from B = 4 on: the
// dependent trips to L2 for q | k and then v cost what the barrier step costs.)
#include <stdlib.h>

#include <mutex>

#include "gemm_common.hpp"

namespace mpl {

constexpr int SM_MAX_BLOCKS = 24;
constexpr int SM_MAX_MT = 2;        // 16-row MFMA tiles of token rows per launch.  (Four were built and measured: every one of the N / 16 workgroups
                                    // of a GEMM reads ALL of A past its L1 -- 102 x 139 KB per qkv at 64 rows -- and a further row tile costs +2.5 us
                                    // per step: 0.59-0.62 ms per stack at 32 rows, 0.79 at 48, 0.90-0.93 at 64 against 0.80-0.83 for the team kernels.)
constexpr int SM_MAX_ROWS = 16 * SM_MAX_MT;
constexpr int SM_MAX_TOK = 16;
constexpr int SM_UMAX = 17;      // k steps of 16 a wave holds as A fragments at a time: K <= 1088 in one go (LayerNorm GEMMs: K = D)
constexpr int SM_LDS_W = 136 * 1024;            // weight tiles (two buffers when they fit); 16 x K x 4 B each
constexpr int SM_LDS_X = SM_LDS_W;              // exchange area: accumulators [row tile][4 waves][256] | LayerNorm partials [row tile][4][16]
constexpr int SM_LDS_RED = SM_LDS_X + SM_MAX_MT * 4 * 1024;
constexpr int SM_LDS_FAIL = SM_LDS_RED + SM_MAX_MT * 256 + 256;
constexpr int SM_LDS_BYTES = SM_LDS_FAIL + 256;

struct SmBlock {
    const float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};
struct SmArgs {
    float *x, *qkv, *att, *hid;
    unsigned *bar, *err_ws, *err_host;
    int M, D, n_tok, H, n_apps, n_wg, spin_log2;
    float eps;
    unsigned char sched[MPL_MAX_APPS];
    SmBlock blk[SM_MAX_BLOCKS];
};

bool sm_stack_ok(int M, int D, int n_tok, int H, int n_apps, int n_blocks) {
    return D <= 16 * 4 * SM_UMAX && M >= 1 && M <= SM_MAX_ROWS && n_tok >= 1 && n_tok <= SM_MAX_TOK && M % n_tok == 0 && D % 16 == 0 && H > 0 && D % H == 0 &&
           ((D / H) & 3) == 0 && 16 * 2 * D * 4 <= SM_LDS_W && n_apps >= 1 && n_apps <= MPL_MAX_APPS && n_blocks >= 1 &&
           n_blocks <= SM_MAX_BLOCKS;
}
constexpr int SM_BAR_WORDS = 32 * 17 + 32;    // 8 group counters, 8 generation words, the top counter (128 bytes apart), the error word
int sm_stack_max_rows() { return SM_MAX_ROWS; }
size_t sm_stack_ws_bytes(int M, int D) { return ((size_t)M * 6 * D + SM_BAR_WORDS) * sizeof(float) + 256; }

enum { SM_EPI_STORE = 0, SM_EPI_GELU = 1, SM_EPI_RES = 2 };

// Activations cross workgroups between the steps.  The guide's fence-free form (MI355X_MICROARCH.md, "inter-workgroup
// visibility": write-through `sc0 sc1` stores AND L1-bypassing `sc1` loads on both sides, the arrival behind a drained store
// queue) instead of an agent release + acquire per barrier (buffer_wbl2 + buffer_inv: ~3.4 us of the ~9.7 us a step took with
// them): a workgroup writes 1 KiB per step, so the per-store price of write-through is nothing here.
// (Measured and not kept: ONE agent-scope invalidate (`buffer_inv sc1`) behind every grid barrier + plain cacheable loads, so that the
// workgroups of an XCD would share one fetch of A through their L2: +2.3 us per step at 16 rows, nothing gained at 32-64.)
typedef unsigned sm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sm_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 sm_ld4(const float* base, unsigned float_off) {      // L1-bypassing 16-byte load
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sm_rsrc(base), float_off * 4u, 0, 16));
}
__device__ __forceinline__ float sm_ld1(const float* base, unsigned float_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sm_rsrc(base), float_off * 4u, 0, 16));
}
__device__ __forceinline__ void sm_st1(float* base, unsigned float_off, float v) {     // write-through store
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), sm_rsrc(base), float_off * 4u, 0, 17);
}
__device__ __forceinline__ void sm_st4(float* base, unsigned float_off, const float4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sm_u32x4, v), sm_rsrc(base), float_off * 4u, 0, 17);
}

// A weight tile to request: W[n0 .. n0 + 15][0 .. K) into the LDS buffer at byte offset `buf`, in fragment order: the piece of k
// step u holds, at lane (j, kq), W[n0 + j][16 u + 4 kq .. + 3].  Wave w requests (and later multiplies) the k steps u = w mod 4.
struct SmTile {
    const float* W;
    int K, n0;
    unsigned buf;
    bool on;
};
__device__ __forceinline__ void sm_request_w(char* smem, const SmTile& t, int wave, int li, int kq) {
    if (!t.on) return;
    const float* src = t.W + (size_t)(t.n0 + li) * t.K + 4 * kq;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + t.buf;
    for (int u = wave; 16 * u < t.K; u += 4) dma16(src + 16 * u, lds0 + (unsigned)(u * 1024));
}

// Grid barrier: every workgroup arrives on one monotonic counter (zeroed by the launcher); lane 0 of wave 0 arrives and polls.
// Weights depend on nobody, so the tile of the NEXT GEMM travels THROUGH the barrier: the waves 1..3 have requested their pieces
// already (a raw s_barrier does not drain the VM queue), wave 0 -- whose queue must be empty for the release fence -- requests
// its pieces between its arrival and its first look at the counter.  all_store: every wave has global stores to retire (the
// attention step); else only wave 0 has (it retired them before it came here).
__device__ __forceinline__ bool sm_grid_sync(const SmArgs& a, char* smem, unsigned& target, int tid, volatile unsigned* s_fail,
                                             const SmTile& next, bool all_store) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (all_store || wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave == 0) {
        // every store of this workgroup was write-through and has been waited for by its wave (in front of the barrier above).
        // Two levels, as the guide's "barrier-xcd" row: the workgroups b mod 8 = x (one XCD, as dispatched today: for speed only, any
        // partition is correct) arrive on counter x; the last of them arrives on the top counter for its group, waits for the
        // other groups there and releases its own group through generation word x.  `target` counts barriers passed.
        const int grp = (int)(blockIdx.x & 7u);
        const unsigned n_grp = ((unsigned)a.n_wg - (unsigned)grp + 7u) >> 3, n_top = a.n_wg < 8 ? (unsigned)a.n_wg : 8u;
        unsigned* cnt = a.bar + 32 * grp;           // 128 bytes apart: no two hot words on one line
        unsigned* gen = a.bar + 32 * (8 + grp);
        unsigned* top = a.bar + 32 * 16;
        bool leader = false;
        if (lane == 0) leader = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == n_grp * (target + 1u);
        if (lane == 0 && leader) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm_request_w(smem, next, 0, lane & 15, lane >> 4);
        if (lane == 0) {
            const unsigned lim = 1u << a.spin_log2;
            unsigned spin = 0;
            if (leader) {
                const unsigned want = n_top * (target + 1u);
                for (; spin < lim; ++spin) {
                    if ((int)(__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (spin < lim) __hip_atomic_store(gen, target + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (; spin < lim; ++spin) {
                    if ((int)(__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (target + 1u)) >= 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (spin == lim) {  // a workgroup that never arrived is an ERROR (the GPU was shared for longer than the bound): report, leave
                *s_fail = 1u;
                if (a.err_ws) __hip_atomic_store(a.err_ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.err_host) __hip_atomic_store(a.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    target += 1u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    return *s_fail == 0u;
}

// (Measured and not kept: NO barrier -- every workgroup that worked in a step publishes "step k done" in a word of its own (one
// write-through store behind its drained data stores) and a workgroup that will work in step k + 1 polls the words of the workers of
// step k with one or two vector loads; workgroups without a tile in a step neither publish nor wait; write-after-read holds
// transitively.  0.424 ms per stack against 0.397 with the two-level barrier at V = 2, B = 1 (0.442 with the polling wave's weight
// requests moved behind the wait): a hundred workgroups polling the same four lines cost more than the barrier's three dependent trips.)
// One 16-column tile of  C = epi( LN?(A) . W^T + bias ), M <= 16 MT rows (MT row tiles of 16 against every weight fragment): the
// four waves split K, wave 0 finishes.  The weight tile `cur` was requested earlier (by this same wave for its own k steps: its
// own wait covers them); `next` is requested by the waves 1..3 as soon as the workgroup is done with the multiply-adds (wave 0's
// share: sm_grid_sync).  The arithmetic of a row does not depend on MT (same k split, same order of additions).
template <int EPI, bool LN, int MT>
__device__ __forceinline__ void sm_tile(char* smem, const SmTile& cur, const SmTile& next, const float* __restrict__ A, int lda,
                                        const float* __restrict__ g, const float* __restrict__ be, float eps, const float* __restrict__ bias,
                                        float* __restrict__ C, int ldc, int M, int N, int wave, int lane) {
    const int li = lane & 15, kq = lane >> 4;
    const int K = cur.K;
    const bool active = cur.on;
    float* xch = reinterpret_cast<float*>(smem + SM_LDS_X);          // [MT][4 waves][256]
    float* red = reinterpret_cast<float*>(smem + SM_LDS_RED);        // [MT][4 waves][16] LayerNorm partials
    unsigned ao[MT];                                                 // A was written by other workgroups of this launch: sm_ld4
#pragma unroll
    for (int rt = 0; rt < MT; ++rt) {
        const int row = 16 * rt + li < M ? 16 * rt + li : M - 1;
        ao[rt] = (unsigned)(row * lda + 4 * kq);
    }
    const int nu_all = (K / 16 - wave + 3) / 4;                      // k steps of this wave
    const int nu = nu_all < SM_UMAX ? nu_all : SM_UMAX;              // ... of the first go (all of them unless K > 1088: fc2 at D = 1088)
    float4 a4[MT][SM_UMAX];
#pragma unroll
    for (int rt = 0; rt < MT; ++rt)
#pragma unroll
        for (int i = 0; i < SM_UMAX; ++i)
            a4[rt][i] = (i < nu && active) ? sm_ld4(A, ao[rt] + 16 * (wave + 4 * i)) : float4{0.f, 0.f, 0.f, 0.f};
    // everything else the step will need from memory is requested NOW, beside the A fragments (one memory round trip for the whole
    // step instead of three dependent ones): the LayerNorm gain / offset of this lane's columns, wave 0's bias and residual values
    float4 g4[LN ? SM_UMAX : 1], b4[LN ? SM_UMAX : 1];
    if (LN) {
#pragma unroll
        for (int i = 0; i < SM_UMAX; ++i) {
            const int k = 16 * (wave + 4 * i) + 4 * kq;
            g4[i] = (i < nu && active) ? ld4(g + k) : float4{0.f, 0.f, 0.f, 0.f};
            b4[i] = (i < nu && active) ? ld4(be + k) : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float bn = 0.f, rsd[MT][4];
#pragma unroll
    for (int rt = 0; rt < MT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) rsd[rt][r] = 0.f;
    if (wave == 0 && active) {
        bn = bias[cur.n0 + li];
        if (EPI == SM_EPI_RES) {
#pragma unroll
            for (int rt = 0; rt < MT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // this workgroup's own columns of x (or the SPT kernel's rows)
                    const int m = 16 * rt + 4 * kq + r;
                    rsd[rt][r] = m < M ? sm_ld1(C, (unsigned)(m * ldc + cur.n0 + li)) : 0.f;
                }
        }
    }
    if (LN) {
        // two-pass statistics of the row over ALL K columns: this lane holds 1/16 of the row (its kq quarter of its wave's k steps)
#pragma unroll
        for (int rt = 0; rt < MT; ++rt) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < SM_UMAX; ++i) s += (a4[rt][i].x + a4[rt][i].y) + (a4[rt][i].z + a4[rt][i].w);
            s = xor32_add(xor16_add(s));
            if (kq == 0) red[(rt * 4 + wave) * 16 + li] = s;
        }
        __syncthreads();
        float mean[MT];
#pragma unroll
        for (int rt = 0; rt < MT; ++rt) {
            const float* rr = red + rt * 64;
            mean[rt] = ((rr[li] + rr[16 + li]) + (rr[32 + li] + rr[48 + li])) / (float)K;
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < MT; ++rt) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < SM_UMAX; ++i)
                if (i < nu) {
                    const float d0 = a4[rt][i].x - mean[rt], d1 = a4[rt][i].y - mean[rt], d2 = a4[rt][i].z - mean[rt], d3 = a4[rt][i].w - mean[rt];
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            q = xor32_add(xor16_add(q));
            if (kq == 0) red[(rt * 4 + wave) * 16 + li] = q;
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < MT; ++rt) {
            const float* rr = red + rt * 64;
            const float rstd = 1.0f / sqrtf(((rr[li] + rr[16 + li]) + (rr[32 + li] + rr[48 + li])) / (float)K + eps);
            const float sh = -mean[rt] * rstd;
#pragma unroll
            for (int i = 0; i < SM_UMAX; ++i)
                if (i < nu) {
                    a4[rt][i].x = fmaf(fmaf(a4[rt][i].x, rstd, sh), g4[i].x, b4[i].x);
                    a4[rt][i].y = fmaf(fmaf(a4[rt][i].y, rstd, sh), g4[i].y, b4[i].y);
                    a4[rt][i].z = fmaf(fmaf(a4[rt][i].z, rstd, sh), g4[i].z, b4[i].z);
                    a4[rt][i].w = fmaf(fmaf(a4[rt][i].w, rstd, sh), g4[i].w, b4[i].w);
                }
        }
    }
    f32x4 acc[MT];
#pragma unroll
    for (int rt = 0; rt < MT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's weight pieces (its own requests) have landed
        const float4* wf = reinterpret_cast<const float4*>(smem + cur.buf) + lane;
#pragma unroll
        for (int i = 0; i < SM_UMAX; ++i)
            if (i < nu) {
                const float4 w = wf[(wave + 4 * i) * 64];
#pragma unroll
                for (int rt = 0; rt < MT; ++rt) acc[rt] = mfma16_k16(a4[rt][i], w, acc[rt]);
            }
        for (int i0 = SM_UMAX; i0 < nu_all; i0 += SM_UMAX) {        // K > 1088 (never a LayerNorm GEMM): the rest in further goes
#pragma unroll
            for (int rt = 0; rt < MT; ++rt)
#pragma unroll
                for (int i = 0; i < SM_UMAX; ++i)
                    a4[rt][i] = (i0 + i < nu_all) ? sm_ld4(A, ao[rt] + 16 * (wave + 4 * (i0 + i))) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < SM_UMAX; ++i)
                if (i0 + i < nu_all) {
                    const float4 w = wf[(wave + 4 * (i0 + i)) * 64];
#pragma unroll
                    for (int rt = 0; rt < MT; ++rt) acc[rt] = mfma16_k16(a4[rt][i], w, acc[rt]);
                }
        }
    }
    // the four K quarters, added in a fixed order by wave 0
#pragma unroll
    for (int rt = 0; rt < MT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[(rt * 4 + wave) * 256 + r * 64 + lane] = acc[rt][r];
    __syncthreads();                                                 // every wave is done with the tile in LDS, too
    if (wave != 0) sm_request_w(smem, next, wave, li, kq);
    if (wave == 0 && active) {
        const int n = cur.n0 + li;
#pragma unroll
        for (int rt = 0; rt < MT; ++rt) {
            const float* xr = xch + rt * 1024;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * rt + 4 * kq + r;                  // D[row = 4 kq + r][col = li] of row tile rt
                float v = ((xr[r * 64 + lane] + xr[256 + r * 64 + lane]) + (xr[512 + r * 64 + lane] + xr[768 + r * 64 + lane])) + bn;
                if (EPI == SM_EPI_GELU) v = gelu_erf(v);
                if (m < M) {
                    const unsigned co = (unsigned)(m * ldc + n);
                    if (EPI == SM_EPI_RES) v += rsd[rt][r];
                    sm_st1(C, co, v);
                }
            }
        }
    }
}

template <int MT>
__global__ __launch_bounds__(256) void sm_stack_kernel(const SmArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float (*S)[SM_MAX_TOK + 1] = reinterpret_cast<float (*)[SM_MAX_TOK + 1]>(smem + SM_LDS_X);      // attention scores (between GEMMs)
    volatile unsigned* s_fail = reinterpret_cast<volatile unsigned*>(smem + SM_LDS_FAIL);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int M = a.M, D = a.D, G = a.n_wg;
    if (tid == 0) *s_fail = 0u;
    __syncthreads();
    unsigned target = 0;
    const int hd = D / a.H;
    const float scale = 1.0f / sqrtf((float)hd);
    const int t = blockIdx.x;                             // this workgroup's column tile in every GEMM that has that many
    const bool has_qkv = t < 3 * D / 16, has_d = t < D / 16, has_fc1 = t < 2 * D / 16;
    // Weight-tile buffers (a tile = 16 K 4 bytes: 64 D for K = D, 128 D for fc2): qkv and fc1 at 0, proj at 64 D, fc2 at 64 D when
    // fc1's tile fits below it (192 D bytes in all: D = 544), else at 0 -- then fc2's tile cannot travel through the barrier in front
    // of it and the next qkv tile waits until fc2's multiply-adds are done (D = 1088).
    const bool roomy = 192 * D <= SM_LDS_W;
    const unsigned b_lo = 0, b_hi = 64 * D, b_fc2 = roomy ? 64 * D : 0;
    const SmTile none{nullptr, 0, 0, 0, false};
    {
        const SmTile first{a.blk[a.sched[0]].qkv_w, D, 16 * t, b_lo, has_qkv};
        sm_request_w(smem, first, wave, li, kq);
    }
    for (int app = 0; app < a.n_apps; ++app) {
        const SmBlock& b = a.blk[a.sched[app]];
        const bool more = app + 1 < a.n_apps;
        const SmTile t_qkv{b.qkv_w, D, 16 * t, b_lo, has_qkv}, t_proj{b.proj_w, D, 16 * t, b_hi, has_d}, t_fc1{b.fc1_w, D, 16 * t, b_lo, has_fc1},
            t_fc2{b.fc2_w, 2 * D, 16 * t, b_fc2, has_d};
        const SmTile t_nq{more ? a.blk[a.sched[app + 1]].qkv_w : nullptr, D, 16 * t, b_lo, more && has_qkv};
        // ---- qkv = norm1(x) . Wqkv^T + b                                            (Attention.forward :55)
        sm_tile<SM_EPI_STORE, true, MT>(smem, t_qkv, t_proj, a.x, D, b.ln1_w, b.ln1_b, a.eps, b.qkv_b, a.qkv, 3 * D, M, 3 * D, wave, lane);
        if (!sm_grid_sync(a, smem, target, tid, s_fail, t_proj, false)) return;
        // ---- attention, one (sequence, head) at a time                              (:56-64)
        {
            const int nt = a.n_tok, n_pairs = (M / nt) * a.H;
            for (int p = blockIdx.x; p < n_pairs; p += G) {
                const int sq = p / a.H, h = p % a.H;
                const unsigned base = (unsigned)(sq * nt * 3 * D + h * hd);
                if (tid < nt * nt) {
                    const int i = tid / nt, j = tid % nt;
                    const unsigned q = base + (unsigned)(i * 3 * D), k = base + (unsigned)(j * 3 * D + D);
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    for (int e = 0; e < hd; e += 4) {
                        const float4 qa = sm_ld4(a.qkv, q + e), kb = sm_ld4(a.qkv, k + e);
                        s0 = fmaf(qa.x, kb.x, s0);
                        s1 = fmaf(qa.y, kb.y, s1);
                        s2 = fmaf(qa.z, kb.z, s2);
                        s3 = fmaf(qa.w, kb.w, s3);
                    }
                    S[i][j] = ((s0 + s1) + (s2 + s3)) * scale;      // the scale AFTER the product (:58)
                }
                __syncthreads();
                if (tid < nt) {
                    float mx = S[tid][0];
                    for (int j = 1; j < nt; ++j) mx = fmaxf(mx, S[tid][j]);
                    float l = 0.f;
                    for (int j = 0; j < nt; ++j) {
                        const float e = __expf(S[tid][j] - mx);
                        S[tid][j] = e;
                        l += e;
                    }
                    const float inv = 1.0f / l;
                    for (int j = 0; j < nt; ++j) S[tid][j] *= inv;
                }
                __syncthreads();
                for (int o = tid; o < nt * (hd / 4); o += 256) {
                    const int i = o / (hd / 4), e = 4 * (o % (hd / 4));
                    float4 acc = {0.f, 0.f, 0.f, 0.f};
                    for (int j = 0; j < nt; ++j) {
                        const float4 v = sm_ld4(a.qkv, base + (unsigned)(j * 3 * D + 2 * D + e));
                        const float pj = S[i][j];
                        acc.x = fmaf(pj, v.x, acc.x);
                        acc.y = fmaf(pj, v.y, acc.y);
                        acc.z = fmaf(pj, v.z, acc.z);
                        acc.w = fmaf(pj, v.w, acc.w);
                    }
                    sm_st4(a.att, (unsigned)((sq * nt + i) * D + h * hd + e), acc);      // out channel = h hd + e (:64)
                }
                __syncthreads();
            }
        }
        if (!sm_grid_sync(a, smem, target, tid, s_fail, none, true)) return;
        // ---- x += att . Wproj^T + b                                                 (:65, Block.forward :90)
        sm_tile<SM_EPI_RES, false, MT>(smem, t_proj, t_fc1, a.att, D, nullptr, nullptr, 0.f, b.proj_b, a.x, D, M, D, wave, lane);
        if (!sm_grid_sync(a, smem, target, tid, s_fail, t_fc1, false)) return;
        // ---- hid = gelu(norm2(x) . W1^T + b)                                        (Mlp.forward :32-33)
        sm_tile<SM_EPI_GELU, true, MT>(smem, t_fc1, roomy ? t_fc2 : none, a.x, D, b.ln2_w, b.ln2_b, a.eps, b.fc1_b, a.hid, 2 * D, M, 2 * D, wave, lane);
        if (!sm_grid_sync(a, smem, target, tid, s_fail, roomy ? t_fc2 : none, false)) return;
        // ---- x += hid . W2^T + b                                                    (:35, Block.forward :91)
        if (!roomy) sm_request_w(smem, t_fc2, wave, li, kq);
        sm_tile<SM_EPI_RES, false, MT>(smem, t_fc2, t_nq, a.hid, 2 * D, nullptr, nullptr, 0.f, b.fc2_b, a.x, D, M, D, wave, lane);
        if (!sm_grid_sync(a, smem, target, tid, s_fail, t_nq, false)) return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static std::atomic<int> g_sm_off{getenv("MPL_NO_SMALL_STACK") != nullptr ? 1 : 0};
void sm_stack_disable(int off) { g_sm_off.store(off); }
bool sm_stack_enabled() { return g_sm_off.load() == 0; }

// blocks: HOST array; schedule[a] indexes it.  ws: sm_stack_ws_bytes(M, D) bytes; *err_ws receives the error word of this call.
int launch_sm_stack(float* x, int n_seq, int n_tok, int D, int H, const mpl_block_weights* blocks, const uint8_t* schedule, int n_apps,
                    void* ws, size_t ws_bytes, const unsigned** err_ws, int spin_log2, hipStream_t s) {
    const int M = n_seq * n_tok;
    int n_blocks = 0;
    for (int i = 0; i < n_apps; ++i) n_blocks = schedule[i] + 1 > n_blocks ? schedule[i] + 1 : n_blocks;
    if (!x || !blocks || !schedule || !sm_stack_ok(M, D, n_tok, H, n_apps, n_blocks)) return MPL_E_INVALID;
    if (!ws || ws_bytes < sm_stack_ws_bytes(M, D)) return MPL_E_WORKSPACE;
    (void)take_fault_injection();       // the one-shot test hook deserts a workgroup of a TEAM launch: an armed one must not outlive
                                        // this (unrelated) launch and hit the next team launch of the process
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return MPL_E_LAUNCH;
    SmArgs a;
    a.x = x;
    a.qkv = reinterpret_cast<float*>(ws);
    a.att = a.qkv + (size_t)M * 3 * D;
    a.hid = a.att + (size_t)M * D;
    a.bar = reinterpret_cast<unsigned*>(a.hid + (size_t)M * 2 * D);
    a.err_ws = a.bar + 32 * 17;
    a.err_host = device_error_word(dev);
    a.M = M; a.D = D; a.n_tok = n_tok; a.H = H; a.n_apps = n_apps;
    // every workgroup must be resident (grid barrier, ~150 KiB of LDS each: one per CU) and every column tile of the widest GEMM
    // needs a workgroup of its own: a device with fewer CUs than that runs the team kernels instead
    if (3 * D / 16 > cus) return MPL_E_UNSUPPORTED;
    a.n_wg = 3 * D / 16;
    a.spin_log2 = spin_log2;
    a.eps = 1e-6f;      // norm_layer = partial(nn.LayerNorm, eps=1e-6), multiview_mpl.py:139
    for (int i = 0; i < MPL_MAX_APPS; ++i) a.sched[i] = i < n_apps ? schedule[i] : 0;
    for (int i = 0; i < n_blocks; ++i) {
        const mpl_block_weights& b = blocks[i];
        if (!b.ln1_w || !b.ln1_b || !b.qkv_w || !b.qkv_b || !b.proj_w || !b.proj_b || !b.ln2_w || !b.ln2_b || !b.fc1_w || !b.fc1_b ||
            !b.fc2_w || !b.fc2_b)
            return MPL_E_INVALID;
        a.blk[i] = SmBlock{b.ln1_w, b.ln1_b, b.qkv_w, b.qkv_b, b.proj_w, b.proj_b, b.ln2_w, b.ln2_b, b.fc1_w, b.fc1_b, b.fc2_w, b.fc2_b};
    }
    if (err_ws) *err_ws = a.err_ws;
    if (int rc = refuse_stream_capture(s)) return rc;
    // row tiles of 16 per weight fragment: 1 or 2 (two instantiations; the result of a row does not depend on the choice)
    const int mi = M <= 16 ? 0 : 1;
    void (*kernel)(const SmArgs) = mi == 0 ? sm_stack_kernel<1> : sm_stack_kernel<2>;
    static std::atomic<bool> attr_set[64][2];
    if (!attr_set[dev][mi].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS_BYTES) != hipSuccess)
            return MPL_E_LAUNCH;
        // the grid barrier needs grid <= resident workgroups.  The LDS footprint (~150 KiB of 160) allows ONE workgroup per CU
        // whatever the occupancy API says, so the API's known over-count of one block per CU at 81 .. 112 SGPRs (256-thread
        // blocks, MI355X_MICROARCH.md "Correctness boundaries"; these kernels spill ~340 SGPRs and sit in that bucket) cannot
        // strand a workgroup here: per_cu is clamped to 1 and the grid (3 D / 16 <= cus, checked above) to per_cu x CUs
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, 256, SM_LDS_BYTES) != hipSuccess || per_cu < 1)
            return MPL_E_UNSUPPORTED;
        attr_set[dev][mi].store(true, std::memory_order_release);
    }
    if (a.n_wg > cus) return MPL_E_UNSUPPORTED;            // grid <= min(per_cu, 1) x CUs
    if (hipMemsetAsync(a.bar, 0, SM_BAR_WORDS * sizeof(unsigned), s) != hipSuccess) return MPL_E_LAUNCH;
    // a grid barrier needs the chip like the team kernels do: serialised with them per device (api.hip)
    hipEvent_t ev = stack_chain_event(dev);
    if (!ev) return MPL_E_LAUNCH;
    std::lock_guard<std::mutex> g(stack_chain_mutex(dev));
    if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) return MPL_E_LAUNCH;
    int rc;
    {
        ProfScope prof(MPL_K_GEMM, s);
        hipLaunchKernelGGL(kernel, dim3(a.n_wg), dim3(256), SM_LDS_BYTES, s, a);
        rc = hip_check_launch();
    }
    if (hipEventRecord(ev, s) != hipSuccess) return MPL_E_LAUNCH;
    return rc;
}

}  // namespace mpl
