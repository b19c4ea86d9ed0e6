// Block stack for SMALL batches: up to 80 token rows (a single frame, a few frames / persons), in groups of sequences of at most
// 16 rows that run side by side.
//
// Reference ops (MPL/lib/models/multiview_mpl.py): the `for blk in self.blocks` loop :420-423 -- Block.forward :84-92
// (x += proj(attn(qkv(norm1(x)))); x += fc2(gelu(fc1(norm2(x))))), Attention.forward :55-64, Mlp.forward :31-37.
//
// The persistent team kernel of h2_gemm.hip walks a 64-row tile through the 52 GEMMs of a stack with ONE team of D / 136
// workgroups: 4 (8) CUs stream all 120 MB of packed weights through their LDS-DMA path, 0.6-0.8 ms however few rows there are.
// With so few rows the GEMMs are weight-streaming problems, and the MI355X shape of that is the WHOLE chip on every GEMM: a GEMM
// of N output columns is N / 16 independent column tiles (102 for qkv at D = 544), one 512-thread workgroup each (all resident:
// grid <= CU count).  Rows of different sequences never meet (a GEMM treats rows independently, the attention stays inside a
// sequence): with two sequences or more the launch is SEVERAL such problems side by side, each group of sequences on its own
// workgroups (sm_stack_groups: 2 x 102 workgroups up to 32 rows at D = 544; 3-5 x 51 with two column tiles per workgroup up to 80
// rows) -- a workgroup then polls and multiplies a fraction of the rows.  Inside a workgroup:
//   * the 16 weight rows of its tile (the nn.Linear tensor in place: no packed copy) come by LDS-DMA straight into FRAGMENT
//     order -- one 1-KiB piece per 16-deep k step, lane (j, kq) fetching W[n0 + j][16 u + 4 kq ..] -- and, because weights do
//     not depend on anybody, the tile of the NEXT GEMM is requested as soon as the multiply-adds of this one are done: its
//     latency (HBM / Infinity Cache: all 114 MB are touched once per forward) hides behind the hand-off of the activations;
//   * the eight waves split K (wave w takes the k steps u = w mod 8, exactly the pieces it requested itself: its own counted
//     wait, no workgroup barrier for the weights), LayerNorm statistics are reduced across the waves through LDS (two-pass, the
//     row values stay in registers), the eight partial accumulators are added in a fixed order, the waves 0..3 apply the epilogue
//     (one result register = four token rows each).
// Arithmetic: exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32 = an fmaf chain per output), exact-erf GELU, fp32 softmax:
// the accuracy of the "fp32_mfma" engine.  The result of a pose is therefore NOT bitwise the one the fp16x2 engine gives the
// same pose in a large batch (both are within 1e-6 of the fp64 oracle); inside this engine results are bitwise independent of
// the batch.  Per block application: [LN1 + qkv] | [attention: one (sequence, head) per workgroup] | [proj + residual] |
// [LN2 + fc1 + GELU] | [fc2 + residual].
//
// HAND-OFF BETWEEN THE STEPS (round 6): no grid barrier.  Every activation element travels as an 8-byte pair {value, tag} in one
// write-through store (the guide's granule: observed untorn), tag = the number of the step that produced it; a consumer loads the
// pairs it needs past its L1 and repeats the load until every tag is the one of the producing step.  The data IS the flag: one
// store and one load trip per step instead of the store drain + two counter trips + generation word + poll + operand load of the
// two-level barrier of rounds 4-5.  Reuse of a buffer is safe without a barrier because every step consumes ALL outputs of the
// step in front of it: a workgroup that writes qkv / att / hid / x of step s has seen every output of step s - 1, whose producers
// had each finished reading what step s overwrites (transitively: the buffer's previous readers are those producers or earlier).
// Tags are unique per launch (the workspace is zeroed by the launcher: tag 0 = stale).  Polls are bounded; a hand-off that never
// arrives is REPORTED (error word, like a lost hand-off of the team kernels).
//
// What the round measured (V = 2, B = 1, us per stack of 65 steps; tools/sm_steps.py stamps every phase of a step, tools/sm_time.py,
// tools/micro/mfma4_chain.hip): rounds 4-5 (four waves, 17 predicated k steps in four specialised copies of the tile code, grid
// barriers) 395.  With one wave per SIMD every instruction is on the critical path and the step was ~2000 of them: LayerNorm,
// multiply-adds and epilogue took 1-2 us EACH (36 dependent MFMAs = 1250 cycles alone, 3500-4900 in the step).  Pairs instead of
// barriers on that code: 427-469 (slower: the polls add instructions).  + one run-time copy of the tile code in a loop over the
// steps (67 -> 29 KB): 484 (it was not the instruction cache).  + eight waves and straight-line code over 5 / 9 k steps per wave
// (steps beyond K: zeroed operands instead of branches): 301.  + no probe poll in front of the fragment loads: 289.  Measured and
// not kept: the epilogue by wave 0 alone (the same); the attention of <= 4 rows inside the proj workgroups (a hand-off less, but
// every proj workgroup polls all of q | k | v and walks all heads: 301); two row tiles PER WORKGROUP (32 rows: 139 KB of pairs per
// workgroup and step, 0.96-1.07 ms against 0.55-0.63 with barriers and 0.62-0.63 for the team kernels).  + two groups of sequences
// on twice the workgroups: 17-32 rows 354-413 us per stack (V = 2 B = 12: 354, V = 4 B = 8: 388), 16 rows 375 -> 332, 8 rows 327 -> 313;
// + three to five groups of 51 workgroups with two column tiles each: 33-80 rows 386-447 us (the team kernels: 616-640).
// A step is now ~4.4 us: ~1.5 hand-off (write-through store -> fabric -> L1-bypassing load: the guide's all-to-all edge), ~1.0
// LayerNorm (three workgroup barriers behind the slowest wave's arrival), ~0.6 multiply-adds (34 fp32 MFMAs per SIMD), ~0.6
// epilogue + weight requests.
#include <stdlib.h>

#include <mutex>

#include "gemm_common.hpp"

#ifndef MPL_LAB
#ifdef SM_DBG
#error "SM_DBG is a laboratory switch: build with -DMPL_LAB (tools/build_variants.sh)"
#endif
#endif

namespace mpl {

#ifdef SM_DBG
// laboratory: shader-clock stamps of the steps of workgroups 0 and 50, waves 0 and 1 (tools/sm_steps.py)
__device__ unsigned long long sm_dbg_buf[2][2][400][8];
#define SM_STAMP(k)                                                                                      \
    do {                                                                                                 \
        if ((blockIdx.x == 0 || blockIdx.x == 50) && wave < 2 && (threadIdx.x & 63) == 0 && dbg_step < 397) \
            sm_dbg_buf[blockIdx.x == 0 ? 0 : 1][wave][dbg_step][k] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define SM_STAMP(k)
#endif

constexpr int SM_MAX_BLOCKS = 24;
constexpr int SM_TILE_ROWS = 16; // ONE 16-row MFMA tile of token rows per workgroup.  More were built and measured in every round: every one of the N / 16
                                 // workgroups of a GEMM reads ALL of A past its L1, and as pairs that is 139 KB per workgroup and step at 32 rows -- two
                                 // tiles per workgroup: 0.96-1.07 ms per stack in this form, 0.55-0.63 with the grid barriers of rounds 4-5, 0.62-0.63
                                 // for the team kernels; four tiles 0.90-0.93 (round 4).
constexpr int SM_MAX_ROWS = 6 * SM_TILE_ROWS;      // more rows: GROUPS of sequences of at most 16 rows, side by side (rows of different sequences never meet: a
                                 // GEMM treats rows independently, the attention stays inside a sequence), each group on its own workgroups --
                                 // the launch of a 16-row problem several times on disjoint compute units (sm_stack_groups)
constexpr int SM_MAX_TOK = 16;
constexpr int SM_NW = 8;         // waves per workgroup: they split K (two per SIMD: each hides the other's LDS / memory latencies)
constexpr int SM_NT = 64 * SM_NW;
constexpr int SM_NU_MAX = 9;     // k steps of 16 a wave holds as A fragments at a time: K <= 16 x 8 x 9 = 1152 in one go (LayerNorm GEMMs: K = D)
constexpr int SM_LDS_W = 136 * 1024;            // weight tiles (two buffers when they fit); 16 x K x 4 B each
constexpr int SM_LDS_X = SM_LDS_W;              // exchange area: accumulators [waves][256] | LayerNorm partials [waves][16]
constexpr int SM_LDS_RED = SM_LDS_X + SM_NW * 1024;
constexpr int SM_LDS_FAIL = SM_LDS_RED + 2 * SM_NW * 64 + 256;
constexpr int SM_LDS_BYTES = SM_LDS_FAIL + 256;

struct SmBlock {
    const float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};
struct SmArgs {
    float *x;                           // [M][D] plain fp32: input of the stack, and its output (written by the last fc2 step)
    float *xp, *qkvp, *attp, *hidp;     // {value, tag} pairs: [M][D], [M][3 D], [M][D], [M][2 D]
    unsigned *err_ws, *err_host;
    int M, M0, D, n_tok, H, n_apps, n_wg, spin_log2;      // M0: rows of a (full) group of sequences; n_wg: workgroups per group
    float eps;
    unsigned char sched[MPL_MAX_APPS];
    SmBlock blk[SM_MAX_BLOCKS];
};

// Layout of a launch: groups of sequences (each of at most 16 rows, on *wpg workgroups of its own) -- 0 = not a launch for this
// engine.  Up to two 16-row groups: one workgroup per column tile of the widest GEMM and group (3 D / 16 = 102 at D = 544), and TWO
// groups whenever there are two sequences or more and the device has the compute units for them -- also below 17 rows: a
// workgroup then polls and multiplies half the rows (V = 2: 16 rows 375 -> 332 us per stack, 8 rows 327 -> 313, same box).  More
// rows (or a device too small for that): the two-tile layout, half as many workgroups per group, each owning the column tiles t and
// t + wpg -- as many groups as there are sequences and compute units (D = 544: 51 workgroups per group, five groups = 80 rows on
// 256 CUs); it needs the LDS for two qkv / fc1 weight tiles beside fc2's (256 D bytes: D <= 544).
int sm_stack_groups(int M, int D, int n_tok, int cus, int* wpg_out, int* rows_out) {
    if (n_tok <= 0 || n_tok > SM_TILE_ROWS || M <= 0 || M % n_tok || D % 16) return 0;
    const int n_seq = M / n_tok, tiles = 3 * D / 16, fit = SM_TILE_ROWS / n_tok;      // fit: sequences in a 16-row group
    const int g_min = (n_seq + fit - 1) / fit;
    int g = 0, wpg = 0;
    if (g_min <= 2 && (n_seq >= 2 ? 2 : 1) * tiles <= cus) {      // (the two-tile layout below 33 rows, measured: 16 rows 343 -> 389 us, 32 rows 393 -> 384)
        g = n_seq >= 2 ? 2 : 1;
        wpg = tiles;
    } else if (g_min == 1 && tiles <= cus) {
        g = 1;
        wpg = tiles;
    } else {
        const int half = (tiles + 1) / 2;
        if (256 * D > SM_LDS_W || D / 16 > half || g_min * half > cus) return 0;
        g = cus / half < n_seq ? cus / half : n_seq;
        wpg = half;
    }
    const int spg = (n_seq + g - 1) / g;                  // sequences per group; the last group may hold fewer
    if (wpg_out) *wpg_out = wpg;
    if (rows_out) *rows_out = spg * n_tok;
    return (n_seq + spg - 1) / spg;
}
bool sm_stack_ok(int M, int D, int n_tok, int H, int n_apps, int n_blocks, int cus) {
    return D <= 16 * SM_NW * SM_NU_MAX && D >= 16 * SM_NW && M >= 1 && sm_stack_groups(M, D, n_tok, cus, nullptr, nullptr) > 0 && n_tok >= 1 && n_tok <= SM_MAX_TOK && M % n_tok == 0 && D % 16 == 0 && H > 0 && D % H == 0 &&
           ((D / H) & 3) == 0 && 16 * 2 * D * 4 <= SM_LDS_W && n_apps >= 1 && n_apps <= MPL_MAX_APPS && n_blocks >= 1 &&
           n_blocks <= SM_MAX_BLOCKS && n_tok * 3 * (D / H) * 4 <= 64 * D /* the q | k | v slice of a (sequence, head) in the first tile buffer */;
}
constexpr int SM_TAIL_WORDS = 64;             // behind the pairs: the error word of the launch
int sm_stack_max_rows() { return SM_MAX_ROWS; }
size_t sm_stack_ws_bytes(int M, int D) { return ((size_t)M * 14 * D + SM_TAIL_WORDS) * sizeof(float) + 256; }

enum { SM_EPI_STORE = 0, SM_EPI_GELU = 1, SM_EPI_RES = 2 };

// Activations cross workgroups (and XCDs) between the steps: write-through `sc0 sc1` stores, L1 / L2-bypassing `sc1` loads (the
// guide's fence-free form, MI355X_MICROARCH.md "inter-workgroup visibility").  A pair is ONE 8-byte store: value and tag become
// visible together, so no drain, no fence and no counter orders anything here.
typedef unsigned sm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sm_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sm_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
// (`fresh`: a zero the optimiser cannot see through, made anew in every round of a poll loop -- sm_fresh() -- so that the loads of
// a round stay inside the loop: to the optimiser the builtin is a pure read of memory nobody in this kernel writes.)
__device__ __forceinline__ unsigned sm_fresh() {
    unsigned z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}
__device__ __forceinline__ sm_u32x4 sm_ld2p(const float* base, unsigned pair_off, unsigned fresh) {     // two pairs {v0, t0, v1, t1}, past L1
    return __builtin_amdgcn_raw_buffer_load_b128(sm_rsrc(base), pair_off * 8u, fresh, 16);
}
__device__ __forceinline__ void sm_stp(float* base, unsigned pair_off, float v, unsigned tag) {     // one pair, write-through
    __builtin_amdgcn_raw_buffer_store_b64(sm_u32x2{__builtin_bit_cast(unsigned, v), tag}, sm_rsrc(base), pair_off * 8u, 0, 17);
}
__device__ __forceinline__ void sm_st2p(float* base, unsigned pair_off, float v0, float v1, unsigned tag) {
    __builtin_amdgcn_raw_buffer_store_b128(sm_u32x4{__builtin_bit_cast(unsigned, v0), tag, __builtin_bit_cast(unsigned, v1), tag}, sm_rsrc(base),
                                           pair_off * 8u, 0, 17);
}
__device__ __forceinline__ float sm_f(unsigned u) { return __builtin_bit_cast(float, u); }

// A weight tile to request: W[n0 .. n0 + 15][0 .. K) into the LDS buffer at byte offset `buf`, in fragment order: the piece of k
// step u holds, at lane (j, kq), W[n0 + j][16 u + 4 kq .. + 3].  Wave w requests (and later multiplies) the k steps u = w mod SM_NW.
// (LDS-typed: with a generic pointer to this word inside the poll loops the gfx950 backend of ROCm 7.2 stops with "Illegal instruction
// detected: V_CMP_NE_U32_e32 0, $src_shared_base")
typedef volatile __attribute__((address_space(3))) unsigned* sm_fail_t;
// A workgroup owns `cnt` column tiles of a GEMM (0: none in this step; 2 in the two-tile layout of sm_stack_groups): columns
// n0 + i dn .., weight buffers buf + i x 64 K bytes.
struct SmTile {
    const float* W;
    int K, n0;
    unsigned buf;
    int cnt, dn;
};
// (lds_base: the LDS address of the dynamic shared array, taken once from the symbol itself)
__device__ __forceinline__ void sm_request_w(unsigned lds_base, const SmTile& t, int wave, int li, int kq) {
    for (int i = 0; i < t.cnt; ++i) {
        const float* src = t.W + (size_t)(t.n0 + i * t.dn + li) * t.K + 4 * kq;
        const unsigned lds0 = lds_base + t.buf + (unsigned)(i * 64 * t.K);
        for (int u = wave; 16 * u < t.K; u += SM_NW) dma16(src + 16 * u, lds0 + (unsigned)(u * 1024));
    }
}

__device__ __forceinline__ void sm_report_lost(const SmArgs& a, sm_fail_t s_fail) {
    *s_fail = 1u;
    if (a.err_ws) __hip_atomic_store(a.err_ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.err_host) __hip_atomic_store(a.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// This wave's A fragments of a go: a4[i] = A[li][16 u_i + 4 kq .. + 3], u_i = wave + SM_NW (i0 + i), i < NU, from the pairs the
// producing step (tag) wrote; a step beyond K reads the wave's first step again (the caller zeroes it).  The fragments are loaded
// and re-loaded until every tag matches (measured and not kept: ONE pair polled first and the fragments fetched when it shows the
// tag -- a round trip more per step: 305 against 289 us per stack at V = 2, B = 1, 393 against 374 at 16 rows).  Bounded: a
// producer that never comes is an ERROR (the GPU was shared for longer than the bound): reported, the launch leaves.
// Straight-line per round: no per-step branches.
template <int NU>
__device__ __forceinline__ void sm_fetch_a(const SmArgs& a, const float* __restrict__ Ap, int lda, int K, unsigned ao, int wave, int i0, unsigned tag,
                                           float4 (&a4)[NU], sm_fail_t s_fail) {
    const unsigned lim = 1u << a.spin_log2;
    unsigned spin = 0;
    unsigned uo[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = wave + SM_NW * (i0 + i);
        uo[i] = ao + 16u * (unsigned)(16 * u < K ? u : wave);
    }
    for (;;) {
        unsigned bad = 0;
        const unsigned z = sm_fresh();
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const sm_u32x4 p0 = sm_ld2p(Ap, uo[i], z), p1 = sm_ld2p(Ap, uo[i] + 2u, z);
            a4[i] = float4{sm_f(p0.x), sm_f(p0.z), sm_f(p1.x), sm_f(p1.z)};
            bad |= (p0.y ^ tag) | (p0.w ^ tag) | (p1.y ^ tag) | (p1.w ^ tag);
        }
        if (!__any(bad != 0u)) break;
        if (++spin >= lim) {
            if ((threadIdx.x & 63) == 0) sm_report_lost(a, s_fail);
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// One 16-column tile of  C = epi( LN?(A) . W^T + bias ), M <= 16 rows: the SM_NW waves split K (wave w: the k steps u = w mod
// SM_NW, NU of them per go -- K <= 16 SM_NW NU in one go, which a LayerNorm GEMM needs), the waves 0..3 finish (wave r: result
// register r of the accumulators = the rows 4 kq + r).  The weight tile `cur` was requested earlier (by this same wave for its own
// k steps: its own wait covers them); `next` is requested when the multiply-adds are done (the finishing waves: behind their
// epilogue).  A: pairs tagged `tin`; C: pairs tagged `tout`; the residual (EPI_RES: this workgroup's own columns of x) lives in
// the registers of the finishing waves across the steps; Cplain: x as plain fp32 for the kernel behind the stack (the last step
// only).  Returns false when a hand-off was lost (workgroup-uniform).
// EPI and LN are run-time values and the loops are straight-line code over NU (steps beyond K: operands zeroed, no branches): with
// one or two waves per SIMD every instruction of a step is on its critical path (rounds 4-5: four waves with 17 predicated steps
// each in four specialised copies, ~2000 instructions per step; tools/sm_steps.py).
template <int NU>
__device__ __forceinline__ bool sm_tile(int EPI, bool LN, const SmArgs& a, char* smem, unsigned lds_base, const SmTile& cur, const SmTile& next,
                                        const float* __restrict__ Ap, int lda, unsigned tin, const float* __restrict__ g, const float* __restrict__ be, float eps,
                                        const float* __restrict__ bias, float* __restrict__ Cp, int ldc, unsigned tout, float& rsd, float* __restrict__ Cplain, int M,
                                        int wave, int lane, sm_fail_t s_fail, int dbg_step = 0) {
    const int li = lane & 15, kq = lane >> 4;
    const int K = cur.K;
    const bool active = cur.cnt > 0;
    SM_STAMP(0);
    float* xch = reinterpret_cast<float*>(smem + SM_LDS_X);          // [SM_NW waves][256]
    float* red = reinterpret_cast<float*>(smem + SM_LDS_RED);        // [sums | squares][SM_NW waves][16] LayerNorm partials
    const unsigned ao = (unsigned)((li < M ? li : M - 1) * lda + 4 * kq);
    const int n_go = (K / 16 + SM_NW * NU - 1) / (SM_NW * NU);      // 1 for every LayerNorm GEMM (sm_stack_ok)
    float bn[2] = {0.f, 0.f};
    if (wave < 4 && active) {
        bn[0] = bias[cur.n0 + li];
        if (cur.cnt > 1) bn[1] = bias[cur.n0 + cur.dn + li];
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int go = 0; go < n_go; ++go) {
        const int i0 = go * NU;
        bool val[NU];                                                // step i of this go exists (16 u < K)
#pragma unroll
        for (int i = 0; i < NU; ++i) val[i] = 16 * (wave + SM_NW * (i0 + i)) < K;
        // what the go needs from memory beside A is requested first (LayerNorm vectors: zero for a step that does not exist)
        float4 g4[NU], b4[NU];
        if (LN) {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int k = 16 * (val[i] ? wave + SM_NW * i : wave) + 4 * kq;
                const float4 gv = ld4(g + k), bv = ld4(be + k);
                g4[i] = val[i] ? gv : float4{0.f, 0.f, 0.f, 0.f};
                b4[i] = val[i] ? bv : float4{0.f, 0.f, 0.f, 0.f};
            }
        }
        float4 a4[NU];
        if (active) sm_fetch_a<NU>(a, Ap, lda, K, ao, wave, i0, tin, a4, s_fail);
#pragma unroll
        for (int i = 0; i < NU; ++i)
            if (!(val[i] && active)) a4[i] = float4{0.f, 0.f, 0.f, 0.f};
        SM_STAMP(1);
        if (LN) {
            // two-pass statistics of the row over ALL K columns: this lane holds 1/(4 SM_NW) of the row (its kq quarter of its wave's k steps)
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NU; ++i) sum += (a4[i].x + a4[i].y) + (a4[i].z + a4[i].w);
            sum = xor32_add(xor16_add(sum));
            if (kq == 0) red[wave * 16 + li] = sum;
            __syncthreads();
            const float* rr = red + li;
            const float mean = (((rr[0] + rr[16]) + (rr[32] + rr[48])) + ((rr[64] + rr[80]) + (rr[96] + rr[112]))) / (float)K;
            float q = 0.f;      // (the squares go to a second array: no barrier between reading the sums and writing them)
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const float d0 = a4[i].x - mean, d1 = a4[i].y - mean, d2 = a4[i].z - mean, d3 = a4[i].w - mean;
                q = fmaf(val[i] ? 1.0f : 0.0f, (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3), q);      // (1 x t + q = t + q exactly)
            }
            q = xor32_add(xor16_add(q));
            if (kq == 0) red[SM_NW * 16 + wave * 16 + li] = q;
            __syncthreads();
            rr += SM_NW * 16;
            const float rstd = 1.0f / sqrtf((((rr[0] + rr[16]) + (rr[32] + rr[48])) + ((rr[64] + rr[80]) + (rr[96] + rr[112]))) / (float)K + eps);
            const float sh = -mean * rstd;
#pragma unroll
            for (int i = 0; i < NU; ++i) {      // (a step that does not exist: g = b = 0 -> 0)
                a4[i].x = fmaf(fmaf(a4[i].x, rstd, sh), g4[i].x, b4[i].x);
                a4[i].y = fmaf(fmaf(a4[i].y, rstd, sh), g4[i].y, b4[i].y);
                a4[i].z = fmaf(fmaf(a4[i].z, rstd, sh), g4[i].z, b4[i].z);
                a4[i].w = fmaf(fmaf(a4[i].w, rstd, sh), g4[i].w, b4[i].w);
            }
        }
        SM_STAMP(2);
        if (active) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's weight pieces (its own requests) have landed
            SM_STAMP(3);
            const float4* wf = reinterpret_cast<const float4*>(smem + cur.buf) + lane;
            float4 wv[NU];                                               // all fragments first: one LDS latency, not one per k step
#pragma unroll
            for (int i = 0; i < NU; ++i) wv[i] = wf[(val[i] ? wave + SM_NW * (i0 + i) : wave) * 64];      // (a step that does not exist: A = 0 against a finite W)
#pragma unroll
            for (int i = 0; i < NU; ++i) acc[0] = mfma16_k16(a4[i], wv[i], acc[0]);
            if (cur.cnt > 1) {                                           // the second column tile of this workgroup: the same A fragments
#pragma unroll
                for (int i = 0; i < NU; ++i) wv[i] = wf[4 * K + (val[i] ? wave + SM_NW * (i0 + i) : wave) * 64];      // + 64 K bytes
#pragma unroll
                for (int i = 0; i < NU; ++i) acc[1] = mfma16_k16(a4[i], wv[i], acc[1]);
            }
        }
    }
    // the K parts, added in a fixed order by the waves 0..3 -- which may still be reading the exchange area of the step (or tile) in
    // front of this one (no grid barrier separates the steps any more)
    SM_STAMP(4);
    bool ok = true;
    for (int ti = 0; ti < (cur.cnt > 1 ? 2 : 1); ++ti) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[wave * 256 + r * 64 + lane] = ti ? acc[1][r] : acc[0][r];
        __syncthreads();                                             // every wave is done with the tile in LDS, too
        if (ti == 0) {
            SM_STAMP(5);
            ok = *s_fail == 0u;
            // the next weight tile: a wave stands in the issue of its pieces (the CU accepts a fragment-order piece per ~60 cycles): the
            // waves 4.. now, the finishing waves behind their epilogue -- the output pairs are what the other workgroups wait for
            if (wave >= 4) sm_request_w(lds_base, next, wave, li, kq);
        }
        if (wave < 4 && active) {
            const int m = 4 * kq + wave;                             // D[row = 4 kq + r][col = li], r = this wave
            const float* x0 = xch + wave * 64 + lane;
            float v = (((x0[0] + x0[256]) + (x0[512] + x0[768])) + ((x0[1024] + x0[1280]) + (x0[1536] + x0[1792]))) + (ti ? bn[1] : bn[0]);
            if (EPI == SM_EPI_GELU) v = gelu_erf(v);
            if (EPI == SM_EPI_RES) {                                 // (never two tiles: D / 16 column tiles <= workgroups per group)
                v += rsd;
                rsd = v;
            }
            if (m < M) {
                const unsigned co = (unsigned)(m * ldc + cur.n0 + ti * cur.dn + li);
                sm_stp(Cp, co, v, tout);
                if (EPI == SM_EPI_RES && Cplain) Cplain[co] = v;
            }
        }
    }
    if (wave < 4) sm_request_w(lds_base, next, wave, li, kq);
    SM_STAMP(6);
    return ok;
}

__global__ __launch_bounds__(SM_NT) void sm_stack_kernel(const SmArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float (*S)[SM_MAX_TOK + 1] = reinterpret_cast<float (*)[SM_MAX_TOK + 1]>(smem + SM_LDS_X);      // attention scores (between GEMMs)
    sm_fail_t s_fail = (sm_fail_t)((__attribute__((address_space(3))) char*)smem + SM_LDS_FAIL);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int D = a.D, G = a.n_wg;
    // the group of sequences this workgroup belongs to: its rows of x and of every pair buffer (the groups share nothing else)
    const int grp = (int)blockIdx.x / G, r0 = grp * a.M0;
    const int M = a.M - r0 < a.M0 ? a.M - r0 : a.M0;
    float* const xg = a.x + (size_t)r0 * D;
    float* const xp = a.xp + (size_t)r0 * 2 * D;
    float* const qkvp = a.qkvp + (size_t)r0 * 6 * D;
    float* const attp = a.attp + (size_t)r0 * 2 * D;
    float* const hidp = a.hidp + (size_t)r0 * 4 * D;
    if (tid == 0) *s_fail = 0u;
    __syncthreads();
#ifdef SM_DBG
    if ((blockIdx.x == 0 || blockIdx.x == 50) && lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        sm_dbg_buf[blockIdx.x == 0 ? 0 : 1][0][397][wave & 7] = hw;
    }
    if (blockIdx.x == 0 && tid == 0) {
        sm_dbg_buf[0][0][399][0] = __builtin_amdgcn_s_memtime();
        sm_dbg_buf[0][0][399][1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const int hd = D / a.H;
    const float scale = 1.0f / sqrtf((float)hd);
    const int t = (int)blockIdx.x - grp * G;              // this workgroup's column tiles in a GEMM of n tiles: t and, below 2 G workgroups
                                                           // per widest GEMM (the two-tile layout of sm_stack_groups), t + G
    auto owned = [&](int n_tiles) -> int { return t < n_tiles ? (t + G < n_tiles ? 2 : 1) : 0; };
    const int c_qkv = owned(3 * D / 16), c_fc1 = owned(2 * D / 16), c_d = owned(D / 16);      // c_d <= 1: D / 16 <= G
    const bool has_d = c_d > 0;
    // Weight-tile buffers (a tile = 16 K 4 bytes: 64 D for K = D, 128 D for fc2): qkv and fc1 at 0 (one tile, or two in the two-tile
    // layout), proj behind them, fc2 at proj's place when it fits (192 D bytes in all with one tile per workgroup, 256 D with two:
    // D = 544), else at 0 -- then fc2's tile is requested when fc1's multiply-adds are done and the next qkv tile when fc2's are
    // (D = 1088).
    const unsigned b_lo = 0, b_hi = (3 * D / 16 > G ? 128 : 64) * D;
    const bool roomy = b_hi + 128 * D <= (unsigned)SM_LDS_W;
    const unsigned b_fc2 = roomy ? b_hi : 0;
    const SmTile none{nullptr, 0, 0, 0, 0, 0};
    {
        const SmTile first{a.blk[a.sched[0]].qkv_w, D, 16 * t, b_lo, c_qkv, 16 * G};
        sm_request_w(lds_base, first, wave, li, kq);
    }
    // step 0: x (plain fp32, written by the kernel in front of this launch) becomes pairs; the workgroup that owns 16 columns of x
    // in proj / fc2 keeps them in the registers of its waves 0..3 from here on (the residual of Block.forward :90-91)
    float rsd = 0.f;
    if (wave < 4 && has_d) {
        const int m = 4 * kq + wave;
        if (m < M) {
            const unsigned co = (unsigned)(m * D + 16 * t + li);
            rsd = xg[co];
            sm_stp(xp, co, rsd, 1u);
        }
    }
    for (int app = 0; app < a.n_apps; ++app) {
        const SmBlock& b = a.blk[a.sched[app]];
        const bool more = app + 1 < a.n_apps;
#ifdef SM_DBG
        if (blockIdx.x == 101 && wave == 7 && (app == 0 || app == 6)) {     // calibration: 64 dependent MFMAs in shader-clock ticks, on a workgroup off the critical path
            f32x4 cacc = {0.f, 0.f, 0.f, 0.f};
            const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int i = 0; i < 64; ++i) cacc = mfma16((float)lane, 1.0f, cacc);
            if (cacc[0] == 12345.f) a.x[0] = cacc[1];
            const unsigned long long c1 = __builtin_amdgcn_s_memtime();
            if (lane == 0) sm_dbg_buf[0][0][398][app == 0 ? 0 : 1] = c1 - c0;
        }
#endif
        const unsigned tg = 2u + 5u * (unsigned)app;      // tags of this application's five steps: tg .. tg + 4
        const SmTile t_qkv{b.qkv_w, D, 16 * t, b_lo, c_qkv, 16 * G}, t_proj{b.proj_w, D, 16 * t, b_hi, c_d, 0}, t_fc1{b.fc1_w, D, 16 * t, b_lo, c_fc1, 16 * G},
            t_fc2{b.fc2_w, 2 * D, 16 * t, b_fc2, c_d, 0};
        const SmTile t_nq{more ? a.blk[a.sched[app + 1]].qkv_w : nullptr, D, 16 * t, b_lo, more ? c_qkv : 0, 16 * G};
        // the five steps of Block.forward :84-92 as a LOOP with one call site of the tile code (see sm_tile: instruction cache)
#pragma nounroll
        for (int ph = 0; ph < 5; ++ph) {
        if (ph != 1) {
            // ph 0: qkv = norm1(x) . Wqkv^T + b (Attention.forward :55) | 2: x += att . Wproj^T + b (:65, Block.forward :90)
            //    3: hid = gelu(norm2(x) . W1^T + b) (Mlp.forward :32-33) | 4: x += hid . W2^T + b (:35, Block.forward :91)
            const SmTile cur = ph == 0 ? t_qkv : ph == 2 ? t_proj : ph == 3 ? t_fc1 : t_fc2;
            const SmTile nxt = ph == 0 ? t_proj : ph == 2 ? t_fc1 : ph == 3 ? (roomy ? t_fc2 : none) : t_nq;
            const float* Ap = ph == 0 || ph == 3 ? xp : ph == 2 ? attp : hidp;
            float* Cp = ph == 0 ? qkvp : ph == 3 ? hidp : xp;
            const int lda = ph == 4 ? 2 * D : D, ldc = ph == 0 ? 3 * D : ph == 3 ? 2 * D : D;
            const float* g = ph == 0 ? b.ln1_w : b.ln2_w;
            const float* be = ph == 0 ? b.ln1_b : b.ln2_b;
            const float* bias = ph == 0 ? b.qkv_b : ph == 2 ? b.proj_b : ph == 3 ? b.fc1_b : b.fc2_b;
            const int epi = ph == 0 ? SM_EPI_STORE : ph == 3 ? SM_EPI_GELU : SM_EPI_RES;
            if (ph == 4 && !roomy) sm_request_w(lds_base, t_fc2, wave, li, kq);
            // (two instantiations by the k steps per wave: K <= 640 | longer -- K = D and K = 2 D at D = 544)
            const bool okk = cur.K <= 16 * SM_NW * 5
                                 ? sm_tile<5>(epi, ph == 0 || ph == 3, a, smem, lds_base, cur, nxt, Ap, lda, tg + (unsigned)ph - 1u, g, be, a.eps, bias, Cp, ldc,
                                                  tg + (unsigned)ph, rsd, ph == 4 && !more ? xg : nullptr, M, wave, lane, s_fail, 5 * app + ph)
                                 : sm_tile<SM_NU_MAX>(epi, ph == 0 || ph == 3, a, smem, lds_base, cur, nxt, Ap, lda, tg + (unsigned)ph - 1u, g, be, a.eps, bias, Cp,
                                                          ldc, tg + (unsigned)ph, rsd, ph == 4 && !more ? xg : nullptr, M, wave, lane, s_fail, 5 * app + ph);
            if (!okk) return;
        } else
        // ---- attention, one (sequence, head) at a time                              (:56-64)
        {
#ifdef SM_DBG
            const int dbg_step = 5 * app + 1;
#endif
            SM_STAMP(0);
            const int nt = a.n_tok, n_pairs = (M / nt) * a.H;
            float* T = reinterpret_cast<float*>(smem);      // [q | k | v][token][hd]: the first tile buffer (qkv's tile is spent, fc1's comes later)
            const int hh = hd / 2, per_row = 3 * hh, items = nt * per_row;
            for (int p = t; p < n_pairs; p += G) {
                const int sq = p / a.H, h = p % a.H;
                const unsigned base = (unsigned)(sq * nt * 3 * D + h * hd);
                const unsigned lim = 1u << a.spin_log2;
                for (unsigned spin = 0;; ++spin) {
                    unsigned bad = 0;
                    const unsigned z = sm_fresh();
                    for (int o = tid; o < items; o += SM_NT) {
                        const int i = o / per_row, rem = o - i * per_row, seg = rem / hh, e = 2 * (rem - seg * hh);
                        const sm_u32x4 pr = sm_ld2p(qkvp, base + (unsigned)(i * 3 * D + seg * D + e), z);
                        *reinterpret_cast<float2*>(T + (seg * nt + i) * hd + e) = float2{sm_f(pr.x), sm_f(pr.z)};
                        bad |= (pr.y ^ (tg)) | (pr.w ^ (tg));
                    }
                    if (!__syncthreads_or(bad != 0u)) break;
                    if (spin + 1u >= lim) {
                        if (tid == 0) sm_report_lost(a, s_fail);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                SM_STAMP(2);
                if (tid < nt * nt) {
                    const int i = tid / nt, j = tid % nt;
                    const float *q = T + i * hd, *k = T + (nt + j) * hd;
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    for (int e0 = 0; e0 < hd; e0 += 32) {      // eight fragments of q and k in flight (the additions in the order e = 0, 4, 8, ...)
                        float4 qa[8], kb[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + 4 * u < hd ? e0 + 4 * u : 0;
                            qa[u] = ld4(q + e);
                            kb[u] = ld4(k + e);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (e0 + 4 * u < hd) {
                                s0 = fmaf(qa[u].x, kb[u].x, s0);
                                s1 = fmaf(qa[u].y, kb[u].y, s1);
                                s2 = fmaf(qa[u].z, kb[u].z, s2);
                                s3 = fmaf(qa[u].w, kb[u].w, s3);
                            }
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
                for (int o = tid; o < nt * (hd / 4); o += SM_NT) {
                    const int i = o / (hd / 4), e = 4 * (o % (hd / 4));
                    float4 acc = {0.f, 0.f, 0.f, 0.f};
                    for (int j = 0; j < nt; ++j) {
                        const float4 v = ld4(T + (2 * nt + j) * hd + e);
                        const float pj = S[i][j];
                        acc.x = fmaf(pj, v.x, acc.x);
                        acc.y = fmaf(pj, v.y, acc.y);
                        acc.z = fmaf(pj, v.z, acc.z);
                        acc.w = fmaf(pj, v.w, acc.w);
                    }
                    const unsigned co = (unsigned)((sq * nt + i) * D + h * hd + e);      // out channel = h hd + e (:64)
                    sm_st2p(attp, co, acc.x, acc.y, tg + 1u);
                    sm_st2p(attp, co + 2u, acc.z, acc.w, tg + 1u);
                }
                __syncthreads();
            }
            SM_STAMP(6);
            if (*s_fail != 0u) return;
        }
        }      // steps
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SM_DBG
    if (blockIdx.x == 0 && tid == 0) {
        sm_dbg_buf[0][0][399][2] = __builtin_amdgcn_s_memtime();
        sm_dbg_buf[0][0][399][3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef SM_DBG
extern "C" int mpl_sm_dbg(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sm_dbg_buf), sizeof(sm_dbg_buf)); }
#endif
static std::atomic<int> g_sm_off{getenv("MPL_NO_SMALL_STACK") != nullptr ? 1 : 0};
void sm_stack_disable(int off) { g_sm_off.store(off); }
bool sm_stack_enabled() { return g_sm_off.load() == 0; }

// blocks: HOST array; schedule[a] indexes it.  ws: sm_stack_ws_bytes(M, D) bytes; *err_ws receives the error word of this call.
int launch_sm_stack(float* x, int n_seq, int n_tok, int D, int H, const mpl_block_weights* blocks, const uint8_t* schedule, int n_apps,
                    void* ws, size_t ws_bytes, const unsigned** err_ws, int spin_log2, hipStream_t s) {
    const int M = n_seq * n_tok;
    int n_blocks = 0;
    for (int i = 0; i < n_apps; ++i) n_blocks = schedule[i] + 1 > n_blocks ? schedule[i] + 1 : n_blocks;
    if (!x || !blocks || !schedule) return MPL_E_INVALID;
    if (!ws || ws_bytes < sm_stack_ws_bytes(M, D)) return MPL_E_WORKSPACE;
    (void)take_fault_injection();       // the one-shot test hook deserts a workgroup of a TEAM launch: an armed one must not outlive
                                        // this (unrelated) launch and hit the next team launch of the process
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return MPL_E_LAUNCH;
    if (!sm_stack_ok(M, D, n_tok, H, n_apps, n_blocks, cus)) return MPL_E_INVALID;
    int wpg = 0, rows = 0;
    const int groups = sm_stack_groups(M, D, n_tok, cus, &wpg, &rows);
    SmArgs a;
    a.x = x;
    a.xp = reinterpret_cast<float*>(ws);
    a.qkvp = a.xp + (size_t)M * 2 * D;
    a.attp = a.qkvp + (size_t)M * 6 * D;
    a.hidp = a.attp + (size_t)M * 2 * D;
    a.err_ws = reinterpret_cast<unsigned*>(a.hidp + (size_t)M * 4 * D);
    a.err_host = device_error_word(dev);
    a.M = M; a.M0 = rows; a.D = D; a.n_tok = n_tok; a.H = H; a.n_apps = n_apps;
    // every workgroup must be resident (they poll each other's output; ~150 KiB of LDS each: one per CU) and every column tile of
    // the widest GEMM of every group of sequences needs a workgroup of its own: a device with fewer CUs than that runs the team
    // kernels instead
    a.n_wg = wpg;
    const int grid = groups * wpg;                         // <= cus (sm_stack_groups)
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
    void (*kernel)(const SmArgs) = sm_stack_kernel;
    static std::atomic<bool> attr_set[64];
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS_BYTES) != hipSuccess)
            return MPL_E_LAUNCH;
        // workgroups that poll each other's output need grid <= resident workgroups.  The LDS footprint (~150 KiB of 160) allows ONE workgroup per CU
        // whatever the occupancy API says, so the API's known over-count of one block per CU at 81 .. 112 SGPRs (256-thread
        // blocks, MI355X_MICROARCH.md "Correctness boundaries"; these kernels spill ~340 SGPRs and sit in that bucket) cannot
        // strand a workgroup here: per_cu is clamped to 1 and the grid (<= cus, checked above) to per_cu x CUs
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, SM_NT, SM_LDS_BYTES) != hipSuccess || per_cu < 1)
            return MPL_E_UNSUPPORTED;
        attr_set[dev].store(true, std::memory_order_release);
    }
    // tag 0 = "not of this launch": the pairs of an earlier launch on this workspace carry the same step numbers
    if (hipMemsetAsync(ws, 0, ((size_t)M * 14 * D + SM_TAIL_WORDS) * sizeof(float), s) != hipSuccess) return MPL_E_LAUNCH;
    // workgroups that wait for each other need the chip like the team kernels do: serialised with them per device (api.hip)
    hipEvent_t ev = stack_chain_event(dev);
    if (!ev) return MPL_E_LAUNCH;
    std::lock_guard<std::mutex> g(stack_chain_mutex(dev));
    if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) return MPL_E_LAUNCH;
    int rc;
    {
        ProfScope prof(MPL_K_GEMM, s);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(SM_NT), SM_LDS_BYTES, s, a);
        rc = hip_check_launch();
    }
    if (hipEventRecord(ev, s) != hipSuccess) return MPL_E_LAUNCH;
    return rc;
}

}  // namespace mpl
