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
// The k loop is software pipelined around a barrier in the MIDDLE of the iteration (see x3_body): MFMA batch 0 of
// stage t | barrier | LayerNorm + split of the A fragment of stage t+1 interleaved with MFMA batch 1 of stage t.
// NPASS = 3 (fused LN1 + qkv + attention): three accumulator sets for the q, k, v slices of the workgroup's 136
// channels; the stages run k-tile by k-tile (q, k, v of k-tile 0, q, k, v of k-tile 1, ...) so that ONE A fragment
// (LayerNorm + split) serves three stages, and the A tile is staged only with the q stage.  Afterwards the workgroup
// finishes Attention.forward :55-64 exactly like ln_gemm_ng_kernel<ATT> (attention_on_tile).
// Special values: an operand is reproduced exactly unless |x| > 3.389e38 (bf16(x) overflows) or its low parts fall
// below the bf16 normal range (|x| < ~1e-33 loses trailing bits) -- outside anything a LayerNorm-ed activation or a
// trained weight takes; inf / nan propagate as nan, as inf - inf does in the residual.
// The k order of every output element is fixed (k-tiles ascending, six products in the order above), so results do
// not depend on the batch size or launch geometry.
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.hpp"

namespace mpl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_W = 27 * 1024;             // W bytes per k-tile of a 136-column group
constexpr int X3_GB = SUB_A + X3_W;         // gamma/beta piece
constexpr int X3_STAGE = X3_GB + 1024;      // 36864
constexpr int X3_T0 = 5;                    // column tiles of waves 0..3; waves 4..7 take the other NT - 5
// Run the second wave of every SIMD half a stage out of phase (see `stage` in x3_body).  Measured on MI355X at
// M = 4096, D = 544: 444-446 k poses/s with, 450-455 k without -- the phases do not overlap better, the lagging wave
// just holds more registers across the barrier.  Kept for experiments, off in the product build.
constexpr bool X3_DEPHASE = false;
#ifndef X3_LEAD
#define X3_LEAD 2
#endif
#ifndef X3_B1_EARLY
#define X3_B1_EARLY 0   // 1: fetch the second B batch of stage t+1 at the end of stage t -- measured -10 % (the reads hit the WAR
                        // hazard on registers the batch-1 MFMAs still read and cannot slip under the next batch-0 MFMAs)
#endif

// 8 fp32 -> hi / mid / lo bf16x8 (round to nearest even at every step; the residuals are exact in fp32).  Written
// stage by stage over the 8 elements so that the four packed chains interleave instead of stalling on each other.
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
    int abl;                                     // bench-only ablation mask (MPL_X3_ABL): 1 no ring refill
};

// Everything a compute wave does, for its NTW column tiles starting at tile `tile0`.
template <int EPI, bool LN, int NPASS, int NST, int NTW, bool LAG, bool DBG = false>
__device__ __forceinline__ void x3_body(const X3Args& a, char* smem, int tid, int wave, int tile0) {
    const unsigned long long t_entry = DBG ? __builtin_amdgcn_s_memtime() : 0;
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
    // NPASS = 2: the workgroup owns two ADJACENT 136-column groups (tn counts pairs); NPASS = 3: the q, k, v slices
    const int m0 = tm * BM, n0 = NPASS == 2 ? tn * (2 * BN) : tn * BN;
    const int Dq = N / 3;                        // NPASS == 3: width of each of q, k, v
    const int KT = K / BK;
    const int T = NPASS * KT;                    // stages
    auto colbase = [&](int pass) -> int { return NPASS == 3 ? pass * Dq + n0 : n0 + pass * BN; };

    // ---- DMA slots of this wave: A piece `wave` (8 rows); a run of ADJACENT W pieces (waves 0..2: four starting at
    // 4 w, waves 3..7: three starting at 3 w + 3) -- source and destination are both contiguous, so one M0 write
    // serves the run and the pieces differ only by the instruction's immediate offset; gamma/beta from wave 7.
    unsigned voA;
    {
        const int r = wave * 8 + (lane >> 3);
        int m = m0 + r;
        m = m < M ? m : M - 1;
        voA = (unsigned)(((size_t)(m - m0) * a.lda + 4 * ((lane & 7) ^ ((r >> 1) & 5))) * sizeof(float));
        asm volatile("" : "+v"(voA));
    }
    const bool w_four = wave < 3;
    const int w_first = w_four ? 4 * wave : 3 * wave + 3;
    unsigned voW = (unsigned)(lane * 16 + w_first * 1024);
    asm volatile("" : "+v"(voW));
    const float* gb_src = ((lane & 8) ? a.ln_b : a.ln_w) + 4 * (lane & 7);
    const bool gb_on = LN && wave == 7;
    // every wave issues at least this many pieces per stage (A + 3 W; the q|k|v instance stages A only with q)
    constexpr int MIN_PIECES = NPASS == 1 ? 4 : 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // Running state of the NEXT stage to issue (stages are issued strictly in order).  NPASS = 3 interleaves the
    // three column groups k-tile by k-tile: stage t = 3 kt + g carries W of group g (q, k, v) for k-tile kt, and the
    // A tile + gamma/beta of k-tile kt ride along with g = 0 only.
    int is_t = 0, is_kt = 0, is_g = 0;
    unsigned is_slot = 0;                        // byte offset of its ring slot
    const char* is_w[NPASS];
#pragma unroll
    for (int g = 0; g < NPASS; ++g) is_w[g] = a.W3 + (size_t)(colbase(g) / BN) * KT * X3_W;
    const float* is_a = a.A + (size_t)m0 * a.lda;   // tile row base in the 64-bit DMA base: per-lane offsets are tile relative
    auto issue_w = [&](const char* wsrc, unsigned st) {
        asm volatile(
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:2048"
            :
            : "v"(voW), "s"(wsrc), "s"(st + (unsigned)(SUB_A + w_first * 1024))
            : "memory");
        if (w_four) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" : : "v"(voW), "s"(wsrc) : "memory");
    };
    auto issue_next = [&]() {
        const unsigned st = lds0 + is_slot;
        const unsigned keep = dma_m0_save();
        if (NPASS == 1) {
            dma16_fast(voA, is_a, st + (unsigned)(wave * 1024));
            issue_w(is_w[0], st);
            if (gb_on) dma16(gb_src + is_kt * BK, st + (unsigned)X3_GB);
            is_w[0] += X3_W;
            is_a += BK;
            ++is_kt;
        } else {
            if (is_g == 0) {
                dma16_fast(voA, is_a, st + (unsigned)(wave * 1024));
                if (gb_on) dma16(gb_src + is_kt * BK, st + (unsigned)X3_GB);
                is_a += BK;
            }
#pragma unroll
            for (int g = 0; g < NPASS; ++g)
                if (g == is_g) { issue_w(is_w[g], st); is_w[g] += X3_W; }
            if (++is_g == NPASS) { is_g = 0; ++is_kt; }
        }
        dma_m0_restore(keep);
        ++is_t;
        is_slot += X3_STAGE;
        if (is_slot == NST * X3_STAGE) is_slot = 0;
    };
#pragma unroll
    for (int t = 0; t < NST; ++t)
        if (t < T) issue_next();

    // LayerNorm statistics of this lane's row: loaded AFTER the first stages are in flight, so that the two cold
    // latencies (statistics from the previous kernel's epilogue, first k-tiles) overlap instead of adding up
    float mu = 0.f, rs = 1.f;
    if (LN) {
        int m = m0 + rg * 16 + li;
        m = m < M ? m : M - 1;
        const int sl = (K % BN == 0) ? BN : K, ns = K / sl;
        ln_combine(a.stats + (size_t)m * ns * 2, ns, sl, K, a.eps, mu, rs);
        asm volatile("" : "+v"(mu), "+v"(rs));   // consume the loads before the k loop (see ln_gemm.hip)
    }


    f32x4 acc[NPASS][NTW];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rv[NTW][4];
    const int row0 = m0 + rg * 16 + 4 * kq;
    const int t_res = T >= 4 ? T - 4 : 0;        // residual prefetch: ~4 stages (~3 us) ahead of the epilogue
    const int key = (li >> 1) & 5;
    constexpr int NB0 = (NTW + 1) / 2, NB1 = NTW - NB0;   // column tiles of the two B batches (3+2 or 2+2)

    // Software pipeline.  The barrier sits in the MIDDLE of an iteration: by then a wave has read everything of
    // stage t into registers, so the barrier both frees slot t % NST for the DMA of stage t + NST and publishes
    // stage t + 1, whose A fragment (LayerNorm + 3-way split: ~50 VALU ops) is prepared while the matrix pipe works
    // on the second B batch of stage t (sched_group_barrier: one MFMA, then four VALU ops, repeated).  The last
    // iteration prepares a stage that does not exist (stale LDS, results unused): no special cases in the body.
    bf16x8 A0[3], A1[3];                         // split A fragment (hi, mid, lo): current / next, ping-pong
    bf16x8 b0h[NB0], b0m[NB0], b0l[NB0];         // B batch 0 of the current stage
    bf16x8 b1h[NB1], b1m[NB1], b1l[NB1];         // B batch 1 (LAG waves keep it across the barrier)
    auto read_a = [&](unsigned slot, float (&x)[8], float (&gg)[8], float (&ee)[8]) {
        const char* st = smem + slot;
        // A fragment of 16x16x32: lane (i, kq) holds A[i][8 kq .. 8 kq + 7] = logical 16-B columns 2kq, 2kq+1
        const float* as = reinterpret_cast<const float*>(st) + (rg * 16 + li) * BK;
        const float4 a0 = ld4(as + (((2 * kq) ^ key) << 2)), a1 = ld4(as + (((2 * kq + 1) ^ key) << 2));
        x[0] = a0.x; x[1] = a0.y; x[2] = a0.z; x[3] = a0.w; x[4] = a1.x; x[5] = a1.y; x[6] = a1.z; x[7] = a1.w;
        if (LN) {
            const float* gb = reinterpret_cast<const float*>(st + X3_GB);
            const float4 g0 = ld4(gb + 8 * kq), g1 = ld4(gb + 8 * kq + 4), e0 = ld4(gb + 32 + 8 * kq), e1 = ld4(gb + 36 + 8 * kq);
            gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
            ee[0] = e0.x; ee[1] = e0.y; ee[2] = e0.z; ee[3] = e0.w; ee[4] = e1.x; ee[5] = e1.y; ee[6] = e1.z; ee[7] = e1.w;
        }
    };
    auto norm_split = [&](float (&x)[8], const float (&gg)[8], const float (&ee)[8], bf16x8 (&o)[3]) {
        if (LN) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = (x[i] - mu) * rs * gg[i] + ee[i];
        }
        split3(x, o[0], o[1], o[2]);
    };
    auto b_base = [&](unsigned slot) -> const bf16x8* {
        return reinterpret_cast<const bf16x8*>(smem + slot + SUB_A) + tile0 * 3 * 64 + lane;
    };
    auto read_b0 = [&](unsigned slot) {
        const bf16x8* bs = b_base(slot);
#pragma unroll
        for (int n = 0; n < NB0; ++n) { b0h[n] = bs[(n * 3 + 0) * 64]; b0m[n] = bs[(n * 3 + 1) * 64]; b0l[n] = bs[(n * 3 + 2) * 64]; }
    };
    auto read_b1 = [&](unsigned slot) {
        const bf16x8* bs = b_base(slot) + NB0 * 3 * 64;
#pragma unroll
        for (int n = 0; n < NB1; ++n) { b1h[n] = bs[(n * 3 + 0) * 64]; b1m[n] = bs[(n * 3 + 1) * 64]; b1l[n] = bs[(n * 3 + 2) * 64]; }
    };
    {   // prologue: stage 0 landed for everyone
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float x[8], gg[8], ee[8];
        read_a(0, x, gg, ee);
        read_b0(0);
        if (LAG || X3_B1_EARLY) read_b1(0);
        norm_split(x, gg, ee, A0);
    }
    unsigned slot_c = 0;                         // ring slot (byte offset) of the current stage
    auto slot_after = [](unsigned sl) -> unsigned { return sl + X3_STAGE == NST * X3_STAGE ? 0u : sl + X3_STAGE; };
#define MPL_X3(AP, BP, NN, OFF)                   \
    _Pragma("unroll") for (int n = 0; n < NN; ++n) \
        accp[OFF + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AP, BP[n], accp[OFF + n], 0, 0, 0);
    // one stage: MFMAs of stage t from `cur`, split fragment of stage t + 1 into `nxt`
    unsigned long long dbg[5] = {0, 0, 0, 0, 0};
    const unsigned long long t_loop = DBG ? __builtin_amdgcn_s_memtime() : 0;
    auto now = [] {
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return v;
    };
    // With X3_DEPHASE the two waves of a SIMD run the stage in different orders so that one is in its MFMA-only
    // batch while the other interleaves MFMAs with the VALU-heavy split:
    //   waves 0..3 (LAG = false):  batch 0 of stage t | barrier t | split(t+1) x batch 1 of stage t
    //   waves 4..7 (LAG = true):   barrier t | batch 0 of stage t | split(t+1) x batch 1 of stage t | B reads of t+1
    // (a LAG wave holds both B batches of a stage in registers across the barrier that frees the stage's slot).
    // SPLIT (std::true_type / false_type): whether this stage prepares a new A fragment for the next one (always for
    // NPASS = 1; only the last of the three q|k|v stages of a k-tile for NPASS = 3, which share one fragment)
    auto stage = [&](auto split_c, int t, f32x4 (&accp)[NTW], const bf16x8 (&cur)[3], bf16x8 (&nxt)[3]) {
        constexpr bool do_split = decltype(split_c)::value;
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        if (DBG) s0 = now();
        const unsigned slot_n = slot_after(slot_c);
        auto batch0 = [&]() {
            MPL_X3(cur[2], b0h, NB0, 0)
            __builtin_amdgcn_sched_barrier(0);
            // refill the slot the last barrier freed under the matrix pipe's shadow: the batch-0 MFMAs have no
            // VALU work to pair with, the DMA issue is scalar + 4..6 VMEM instructions
            if ((LAG || t >= 1) && is_t < T && !(a.abl & 1)) issue_next();
            __builtin_amdgcn_sched_barrier(0);
            MPL_X3(cur[0], b0l, NB0, 0)
            MPL_X3(cur[1], b0m, NB0, 0)
            MPL_X3(cur[1], b0h, NB0, 0)
            MPL_X3(cur[0], b0m, NB0, 0)
            MPL_X3(cur[0], b0h, NB0, 0)
        };
        auto sync = [&]() {
            // own pieces of stage t+1 landed: with a full ring, NST-2 later stages (>= MIN_PIECES pieces each) may
            // stay in flight; near the end (and for NST = 2) simply drain.  Every read of stage t has returned.
            if (NST > 2 && t + NST - 1 < T) wait_vm((NST - 2) * MIN_PIECES);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (DBG) s2 = now();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (DBG) s3 = now();
        };
        if (!LAG) {
            if (!X3_B1_EARLY) read_b1(slot_c);
            batch0();
            if (DBG) s1 = now();
            sync();
        } else {
            sync();
            batch0();
            if (DBG) s1 = now();
        }
        if (EPI == MPL_EPI_BIAS_RESIDUAL && t == t_res)
            load_residual_w<NTW>(rv, a.R, a.ldr, M, N, row0, n0 + tile0 * 16, li);
        float x[8], gg[8], ee[8];
        read_a(slot_n, x, gg, ee);
        // LayerNorm variants fetch their next B batch only after the split (measured: fetching it up front costs 5 % on the
        // fused attention instance even where registers are plentiful -- the A fragment then queues behind 9 more reads)
        constexpr bool B0_EARLY = !LAG && !LN;
        if (B0_EARLY) read_b0(slot_n);
        // Hand-interleaved: the 6 * NB1 MFMAs of batch 1 alternate with the split of the next A fragment, one MFMA
        // (16 cycles of matrix pipe) per ~4 VALU ops; sched_barrier(0) pins the order hipcc would otherwise undo
        // (it groups all VALU first and lets the wave sit on the ds_read latency with the matrix pipe idle).
        unsigned wh[4], wm[4], wl[4];            // packed bf16 pairs of the next fragment
        float r0[4], r1[4];
        constexpr int NM = 6 * NB1;              // MFMAs to place
        int mi = 0;
        auto mfma_next = [&]() {                 // the mi-th MFMA of batch 1 in the canonical product order
            if (mi < NM) {
                const int prod = mi / NB1, n = mi % NB1;
                const bf16x8& ap = cur[prod == 0 ? 2 : (prod == 2 || prod == 3) ? 1 : 0];
                const bf16x8& bp = (prod == 0 || prod == 3 || prod == 5) ? b1h[n] : (prod == 1) ? b1l[n] : b1m[n];
                accp[NB0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap, bp, accp[NB0 + n], 0, 0, 0);
                ++mi;
            }
        };
        auto pack = [](float lo_, float hi_) -> unsigned {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const bf16x2 v = {(__bf16)lo_, (__bf16)hi_};
            return __builtin_bit_cast(unsigned, v);
        };
        auto lo_f = [](unsigned w) -> float { return __builtin_bit_cast(float, w << 16); };
        auto hi_f = [](unsigned w) -> float { return __builtin_bit_cast(float, w & 0xffff0000u); };
        if constexpr (do_split) {
            // X3_LEAD MFMAs ahead of the first VALU group cover the latency of the A fragment reads just issued
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < X3_LEAD; ++k) mfma_next();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {            // hi part + first residual of pair j
                float x0 = x[2 * j], x1 = x[2 * j + 1];
                if (LN) {
                    x0 = (x0 - mu) * rs * gg[2 * j] + ee[2 * j];
                    x1 = (x1 - mu) * rs * gg[2 * j + 1] + ee[2 * j + 1];
                }
                wh[j] = pack(x0, x1);
                r0[j] = x0 - lo_f(wh[j]);
                r1[j] = x1 - hi_f(wh[j]);
                mfma_next();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {            // mid part + second residual
                wm[j] = pack(r0[j], r1[j]);
                r0[j] -= lo_f(wm[j]);
                r1[j] -= hi_f(wm[j]);
                mfma_next();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 4; j += 2) {         // lo part
                wl[j] = pack(r0[j], r1[j]);
                wl[j + 1] = pack(r0[j + 1], r1[j + 1]);
                mfma_next();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < NM; ++k) mfma_next();   // whatever is left (mi is a compile-time value here)
            __builtin_amdgcn_sched_barrier(0);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            nxt[0] = __builtin_bit_cast(bf16x8, (u32x4){wh[0], wh[1], wh[2], wh[3]});
            nxt[1] = __builtin_bit_cast(bf16x8, (u32x4){wm[0], wm[1], wm[2], wm[3]});
            nxt[2] = __builtin_bit_cast(bf16x8, (u32x4){wl[0], wl[1], wl[2], wl[3]});
        } else {
#pragma unroll
            for (int k = 0; k < NM; ++k) mfma_next();
        }
        // LayerNorm variants keep gamma/beta live above: their next B batch is fetched only now (128-register budget)
        if (!B0_EARLY) read_b0(slot_n);
        if (LAG || X3_B1_EARLY) read_b1(slot_n);
        slot_c = slot_n;
        if (DBG) {
            const unsigned long long s4 = now();
            if (!LAG) { dbg[0] += s1 - s0; dbg[1] += s2 - s1; dbg[2] += s3 - s2; dbg[3] += s4 - s3; }
            else { dbg[0] += s1 - s3; dbg[1] += s2 - s0; dbg[2] += s3 - s2; dbg[3] += s4 - s1; }
            dbg[4] += 1;
        }
    };
    constexpr std::true_type SPLIT{};
    constexpr std::false_type KEEP{};
    if constexpr (NPASS == 1) {
        int kt = 0;
        for (; kt + 1 < KT; kt += 2) {
            stage(SPLIT, kt, acc[0], A0, A1);
            stage(SPLIT, kt + 1, acc[0], A1, A0);
        }
        if (kt < KT) stage(SPLIT, kt, acc[0], A0, A1);
    } else if constexpr (NPASS == 2) {
        // two column groups per k-tile share one fragment; the second stage prepares the next one
        int kt = 0;
        for (; kt + 1 < KT; kt += 2) {
            stage(KEEP, 2 * kt, acc[0], A0, A0);
            stage(SPLIT, 2 * kt + 1, acc[1], A0, A1);
            stage(KEEP, 2 * kt + 2, acc[0], A1, A1);
            stage(SPLIT, 2 * kt + 3, acc[1], A1, A0);
        }
        if (kt < KT) {
            stage(KEEP, 2 * kt, acc[0], A0, A0);
            stage(SPLIT, 2 * kt + 1, acc[1], A0, A1);
        }
    } else {
        // k-tile by k-tile: q, k, v stages share the fragment of the k-tile; the v stage prepares the next one
        int kt = 0;
        for (; kt + 1 < KT; kt += 2) {
            stage(KEEP, 3 * kt, acc[0], A0, A0);
            stage(KEEP, 3 * kt + 1, acc[1], A0, A0);
            stage(SPLIT, 3 * kt + 2, acc[2], A0, A1);
            stage(KEEP, 3 * kt + 3, acc[0], A1, A1);
            stage(KEEP, 3 * kt + 4, acc[1], A1, A1);
            stage(SPLIT, 3 * kt + 5, acc[2], A1, A0);
        }
        if (kt < KT) {
            stage(KEEP, 3 * kt, acc[0], A0, A0);
            stage(KEEP, 3 * kt + 1, acc[1], A0, A0);
            stage(SPLIT, 3 * kt + 2, acc[2], A0, A1);
        }
    }
#undef MPL_X3

    const unsigned long long t_epi = DBG ? now() : 0;
    if constexpr (NPASS == 3) {
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
    if constexpr (NPASS == 2) {   // two plain (bias / GELU) tiles side by side; no residual, no statistics
        tile_values_store<EPI, NTW>(acc[0], a.bias, rv, a.C, a.ldc, M, N, row0, n0, n0 + tile0 * 16, li, v);
        tile_values_store<EPI, NTW>(acc[1], a.bias, rv, a.C, a.ldc, M, N, row0, n0 + BN, n0 + BN + tile0 * 16, li, v);
        return;
    }
    tile_values_store<EPI, NTW>(acc[0], a.bias, rv, a.C, a.ldc, M, N, row0, n0, n0 + tile0 * 16, li, v);
    const unsigned long long t_st = DBG ? now() : 0;
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
    if (DBG) {   // bench-only: per-wave phase cycles into the (otherwise unused) att_out pointer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = now();
        if (lane == 0) {
            float* o = a.att_out + (size_t)(blockIdx.x * 8 + wave) * 16;
            for (int i = 0; i < 5; ++i) o[i] = (float)dbg[i];
            o[5] = (float)(t_loop - t_entry);
            o[6] = (float)(t_epi - t_loop);
            o[7] = (float)(t_st - t_epi);
            o[8] = (float)(t_end - t_st);
        }
    }
}

template <int EPI, bool LN, int NPASS, int NST, bool DBG = false>
__global__ __launch_bounds__(512, (NPASS == 1 && EPI != MPL_EPI_BIAS_RESIDUAL) ? 4 : 2) void x3_gemm_kernel(const X3Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < 4) x3_body<EPI, LN, NPASS, NST, X3_T0, false, DBG>(a, smem, tid, wave, 0);
    else x3_body<EPI, LN, NPASS, NST, NT - X3_T0, X3_DEPHASE, DBG>(a, smem, tid, wave, X3_T0);
}

static int x3_abl() {
    static const int v = getenv("MPL_X3_ABL") ? atoi(getenv("MPL_X3_ABL")) : 0;
    return v;
}

template <int EPI, bool LN, int NPASS, int NST>
static int launch_x3(const X3Args& a, hipStream_t s) {
    constexpr int LDS = NST * X3_STAGE;
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    static_assert(NPASS != 3 || (BM * ATT_TS + ATT_SCORE_FLOATS) * 4 <= LDS, "attention epilogue does not fit in the ring");
    static_assert(NPASS != 2 || EPI != MPL_EPI_BIAS_RESIDUAL, "the paired instance has no residual epilogue");
    static std::atomic<bool> attr_set[64];   // set-once flags: a racing second hipFuncSetAttribute is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)x3_gemm_kernel<EPI, LN, NPASS, NST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL((x3_gemm_kernel<EPI, LN, NPASS, NST>), dim3(a.grid_m * a.grid_n), dim3(512), LDS, s, a);
    return hip_check_launch();
}

template <int EPI, bool LN>
static int launch_x3_auto(const X3Args& a, hipStream_t s) {
    // two workgroups per CU (2-stage rings) once there are more workgroups than CUs, else one with a deeper ring.  The
    // residual instances hold their prefetched residual + LayerNorm hand-over in ~190 registers, so only one of them
    // fits a CU whatever the ring: always the deeper one.
    static const int force = getenv("MPL_X3_NST") ? atoi(getenv("MPL_X3_NST")) : 0;   // bench-only
    const int wgs = a.grid_m * a.grid_n;
    int nst = (wgs > 256 && EPI != MPL_EPI_BIAS_RESIDUAL) ? 2 : 3;
    if (force) nst = force;
    static const bool dbg = getenv("MPL_X3_DBG") != nullptr;   // bench-only phase timing into stats_out
    if (dbg && EPI == MPL_EPI_BIAS_RESIDUAL && !LN && a.stats_out) {
        X3Args b = a;
        b.att_out = a.stats_out;      // timing dump
        b.stats_out = nullptr;
        ProfScope prof(MPL_K_GEMM, s);
        hipFuncSetAttribute((const void*)x3_gemm_kernel<MPL_EPI_BIAS_RESIDUAL, false, 1, 3, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 3 * X3_STAGE);
        hipLaunchKernelGGL((x3_gemm_kernel<MPL_EPI_BIAS_RESIDUAL, false, 1, 3, true>), dim3(a.grid_m * a.grid_n), dim3(512),
                           3 * X3_STAGE, s, b);
        return hip_check_launch();
    }
    // More 136-column tiles than CUs (LN2 + fc1 + GELU, the unfused qkv): give each workgroup two adjacent column
    // groups instead -- one LayerNorm + split of the A fragment then serves two stages, at one workgroup per CU with a
    // 4-stage ring.  Measured on MI355X: fc1 at M = 4096, D = 544 39.4 -> 37.4 us, FULL flag set (D = 1088) +2.4 %.
    // Same k order per output element: bitwise identical results (tools/x3_pair_check.py).
    static const bool nopair = getenv("MPL_X3_NOPAIR") != nullptr;   // bench-only A/B switch
    if constexpr (EPI != MPL_EPI_BIAS_RESIDUAL) {
        if (!nopair && !force && (a.grid_n & 1) == 0 && wgs > 256) {
            X3Args b = a;
            b.grid_n = a.grid_n / 2;
            return launch_x3<EPI, LN, 2, 4>(b, s);
        }
    }
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
             (M + BM - 1) / BM, N / BN, eps, stats_out, 0, 0, nullptr, x3_abl()};
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
             (M + BM - 1) / BM, D / BN, eps, nullptr, n_tok, D / heads, att, x3_abl()};
    return launch_x3<MPL_EPI_BIAS, true, 3, 4>(a, s);
}

}  // namespace mpl
