// Device side of the packed-operand GEMM engines of the FPT block stack (h2_gemm.hip: fp32 arithmetic from two fp16 parts,
// NP = 2; b1_gemm.hip: bf16 operands, NP = 1): tile constants, the GEMM phase (h2_phase), the one-GEMM kernel and the two
// persistent stack kernels, as templates that each engine's translation unit instantiates for its own NP.
#pragma once
#include <mutex>
#include <type_traits>

#include "gemm_common.hpp"

namespace mpl {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int H2_RG = 2 * 1024;              // A bytes per (row group, k-tile): hi | lo fragment
constexpr int H2_A = 4 * H2_RG;              // A bytes per stage (64 rows)
constexpr int H2_W = 18 * 1024;              // W bytes per k-tile of a 136-column group: 9 slots x {hi, lo}
constexpr int H2_STAGE = H2_A + H2_W;        // 26624
constexpr int H2_NST = 6;                    // ring depth (159744 B of LDS, one workgroup per CU)
constexpr int H2_VEC = H2_NST * H2_STAGE;    // the 4 KiB above the ring: epilogue vectors [pass][c | sc][136] of a phase
constexpr int H2_LDS_BYTES = H2_VEC + 4096;  // = 160 KiB
constexpr int H2_DW_RES = 80 * 1024;       // direct-W form: the residual rows of the two multiplying waves (9 KiB)
constexpr int H2_FAIL = H2_VEC + 4092;       // last word of the LDS: "a wait of this workgroup was lost"
constexpr int H2_T0 = 5;
constexpr int H2_MAX_WGS = 1024;
constexpr int H2_ATT_TS = 3 * BN + 4;        // row stride (floats) of the q | k | v tile of the generic attention epilogue
constexpr float H2_SA = 1024.0f;             // scale of a normalised LayerNorm input (|z| <= sqrt(K): fine up to K = 2048)
constexpr int H2_TRV = 5;                    // fp32 vectors of length N behind the fragments: c | sc | sw | bound | so
// ---- laboratory switches.  The product library is built WITHOUT -DMPL_LAB: every switch below then has the one value the
// shipped kernels were measured and tested with, and a stray -DH2_* on the command line is a compile error instead of a
// silently different (or, for H2_ABL, deliberately WRONG) library.  tools/build_variants.sh passes -DMPL_LAB together with the
// switch it varies; the values and what they do are unchanged from rounds 3-5.
#ifndef MPL_LAB
#if defined(H2_DBG) || defined(H2_ABL) || defined(H2_DW_PIN) || defined(H2_WT_AUX) || defined(H2_WSPLIT) || defined(H2_R2_AB) || \
    defined(H2_R2_AB2) || defined(H2_TAIL_LOOP) || defined(H2_KPS2)
#error "H2_* experiment switches are laboratory-only: build with -DMPL_LAB (tools/build_variants.sh does)"
#endif
#endif
#ifndef H2_DBG
#define H2_DBG 0        // 1 / 2: per-wave shader-clock stamps of every phase (tools/chain_phase.py)
#endif
#ifndef H2_ABL
#define H2_ABL 0        // bench-only ablations (results are GARBAGE): 1 no B fragment reads, 2 no DMA refill, 4 no A fragment reads, 8 no MFMA, 16 no LayerNorm conversion, 32 no barrier, 64 W out of a hot L2
#endif
#ifndef H2_DW_PIN
#define H2_DW_PIN 1     // direct-W form: pin the MFMA / load interleave of a stage (sched_group_barrier)
#endif
#ifndef H2_WT_AUX
#define H2_WT_AUX 17    // cache policy of the hand-off stores: 17 = sc0 sc1 (write-through), 16 = sc1
#endif
#ifndef H2_WSPLIT
#define H2_WSPLIT 1     // W pieces per stage and wave: 0 = 1 (waves 0..3) / 4 (waves 4, 5) / 3 (waves 6, 7); 1 = 2 / 3 / 2
#endif
#ifndef H2_R2_AB
#define H2_R2_AB 1      // bf16 pair form: A pieces by the waves 4..7 (see h2_phase)
#endif
#define H2_RP_WC0 (H2_R2_AB ? 4 : 1)      // W pieces per wave role of the bf16 pair form
#define H2_RP_WC1 (H2_R2_AB ? 1 : 4)
#define H2_RP_WC2 (H2_R2_AB ? 0 : 3)
#ifndef H2_R2_AB2
#define H2_R2_AB2 0     // the same duty for the two-tile stage of the fp16x2 engine (h2_stack2_kernel)
#endif
#define H2_R2_WC0 (H2_R2_AB2 ? 4 : 1)      // W pieces per wave role of the two-tile stage (the waves 0..3 carry four A pieces)
#define H2_R2_WC1 (H2_R2_AB2 ? 1 : 4)
#define H2_R2_WC2 (H2_R2_AB2 ? 0 : 3)
#define H2_WC0 (H2_WSPLIT == 1 ? 2 : 1)
#define H2_WC1 (H2_WSPLIT == 1 ? 3 : 4)
#define H2_WC2 (H2_WSPLIT == 1 ? 2 : 3)
#ifndef H2_TAIL_LOOP
#define H2_TAIL_LOOP 1
#endif
#ifndef H2_KPS2
#define H2_KPS2 1      // 1: one barrier per TWO stages: it publishes two stages at once, the refill then targets 5 stages ahead (one
                       // ring slot of slack); 0: a barrier in front of every stage, refill 6 stages ahead
#endif

// stages (k steps) of a K-wide operand: NP = 2 (fp16 hi | lo) one 32-deep k-tile per stage; NP = 1 (bf16) a PAIR of k-tiles per
// stage -- the second KiB of every 2-KiB fragment slot holds the odd k-tile instead of the lo part, so both engines move the
// same pieces through the same ring; an odd k-tile count is padded with one zero k-tile
__host__ __device__ constexpr int h2_ksteps(int K, int NP) { return NP == 2 ? K / BK : (K / BK + 1) / 2; }
__host__ __device__ constexpr int h2_slot_tile(int s) { return s < 4 ? s : (s == 4 ? 8 : s - 1); }
__host__ __device__ inline int h2_col(int t, int kq, int j, int G) {
    if (t < 4 * G) return 136 * (t >> 2) + 32 * (t & 3) + 16 * (j >> 2) + 4 * kq + (j & 3);
    return 136 * (4 * (t - 4 * G) + kq) + 128 + j;
}
// 8 fp32 -> hi / lo packed fp16 (RNE; the residual is exact in fp32; subnormal results are kept)
__device__ __forceinline__ void split2(const float (&x)[8], f16x8& hi, f16x8& lo) { split2_f16(x, hi, lo); }      // common.hpp
// largest power of two p with p * v <= 2^15 (v > 0, finite); 1 for v == 0
__host__ __device__ inline float h2_window_scale(float v) {
    if (!(v > 0.f)) return 1.0f;
    int e;
    (void)frexpf(32768.0f / v, &e);            // 32768 / v = m 2^e, m in [0.5, 1)  ->  2^(e-1) <= 32768 / v
    e = e - 1 < -120 ? -120 : (e - 1 > 120 ? 120 : e - 1);   // both the scale and its reciprocal stay normal fp32 numbers
    return ldexpf(1.0f, e);
}

struct H2Ops {
    int n_apps, D;
    unsigned* err_host;
    const char* w[MPL_MAX_APPS][4];
};

// ---------------------------------------------------------------------------------------------- GEMM
struct H2Args {
    const char* A2;          // packed activations (GEMMs without LayerNorm); unused when LNF
    const float* X;          // LNF: the fp32 rows, normalised and split in the k loop
    int ldx;
    const char* W2;          // packed weights (gamma folded for LNF)
    const float* cvec;       // c per output column
    const float* svec;       // sc per output column: 1 / (sa sw)
    const float* stats;      // LNF: per-row slice partials of the K-wide input rows
    const float* a_inv;      // !LNF: device scalar, reciprocal of the scale of A2 (NULL = 1)
    const float* ovec;       // C2 != NULL: so per output column (absolute column index), the static scales of the packed output
    const float* R;          // residual (fp32), EPI_RES
    int ldr;
    float* C;                // fp32 output (optional)
    int ldc;
    char* C2;                // packed output (optional): the next GEMM's operand
    float* stats_out;        // residual epilogue: slice partials of the rows produced
    int M, N, K, rpt;
    int grid_m, grid_n;
    float eps;
    int att_ntok, att_hd;
    unsigned long long* dbg;
    unsigned* err_ws;        // chain mode: the error word of the call (workspace), set when a hand-off wait is lost
    unsigned* err_host;
    int spin_log2;
    int plain;               // chain mode: the team sits on ONE XCD (h2_team_placement): hand-off stores stay in its L2 (plain
                             // stores) instead of writing through; consumers read past their L1 either way
};

enum { H2_EPI_BIAS = 0, H2_EPI_GELU = 1, H2_EPI_RES = 2, H2_EPI_ATT = 3 };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t h2_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void h2_st16(bool wt, void* base, unsigned off, const u32x4& w) {
    if (wt) __builtin_amdgcn_raw_buffer_store_b128(w, h2_rsrc(base), off, 0, H2_WT_AUX);
    else __builtin_amdgcn_raw_buffer_store_b128(w, h2_rsrc(base), off, 0, 0);
}
__device__ __forceinline__ void h2_st8(bool wt, void* base, unsigned off, const u32x2& h) {
    if (wt) __builtin_amdgcn_raw_buffer_store_b64(h, h2_rsrc(base), off, 0, H2_WT_AUX);
    else __builtin_amdgcn_raw_buffer_store_b64(h, h2_rsrc(base), off, 0, 0);
}
__device__ __forceinline__ u32x4 h2_ld16_l2(const void* base, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(h2_rsrc(base), off, 0, 16);
}
// 8 fp32 -> packed bf16 (round to nearest even)
__device__ __forceinline__ bf16x8 to_bf16x8(const float (&x)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)x[i];
    return v;
}
// pack 8 fp32 values (already multiplied by the output scale) of one fragment lane: k-tile t of the strip at `base`.
// NP = 2: hi | lo fp16 in the two KiB of k-tile t; NP = 1: bf16 in KiB t (k-tile t is half t & 1 of stage t >> 1)
template <int NP>
__device__ __forceinline__ void h2_emit_frag(bool wt, char* base, int t, unsigned lo_off, const float (&x)[8]) {
    if constexpr (NP == 2) {
        f16x8 hi, lo;
        split2(x, hi, lo);
        h2_st16(wt, base, (unsigned)(t * H2_RG) + lo_off, __builtin_bit_cast(u32x4, hi));
        h2_st16(wt, base, (unsigned)(t * H2_RG + 1024) + lo_off, __builtin_bit_cast(u32x4, lo));
    } else {
        h2_st16(wt, base, (unsigned)(t * 1024) + lo_off, __builtin_bit_cast(u32x4, to_bf16x8(x)));
    }
}
// the 4 values (8 bytes per part) a lane contributes to the shared tail k-tile.  NP = 1 with `pad`: the zero k-tile behind it
// (an odd k-tile count is padded to whole stages) gets this lane's 8 bytes of zeros
template <int NP>
__device__ __forceinline__ void h2_emit_tail(bool wt, char* base, int t, unsigned lo_off, const float (&x)[8], bool pad = false) {
    if constexpr (NP == 2) {
        f16x8 hi, lo;
        split2(x, hi, lo);
        const u32x4 h = __builtin_bit_cast(u32x4, hi), l = __builtin_bit_cast(u32x4, lo);
        h2_st8(wt, base, (unsigned)(t * H2_RG) + lo_off, u32x2{h[0], h[1]});
        h2_st8(wt, base, (unsigned)(t * H2_RG + 1024) + lo_off, u32x2{l[0], l[1]});
    } else {
        const u32x4 h = __builtin_bit_cast(u32x4, to_bf16x8(x));
        h2_st8(wt, base, (unsigned)(t * 1024) + lo_off, u32x2{h[0], h[1]});
        if (pad) h2_st8(wt, base, (unsigned)((t + 1) * 1024) + lo_off, u32x2{0u, 0u});
    }
}

// Attention.forward :55-64 on the q | k | v tile T[64][H2_ATT_TS] (+bias, LayerNorm applied) of this workgroup's 136
// channels, generic form (any n_tok <= 32, any head width that divides 136): S whole sequences of nt tokens; the output
// is written as packed A2 of width Dq (column c scaled by so[c], the static scales of this workgroup's 136 v columns) for proj.  The 4-token shapes never come here (registers).
template <int NP>
// rg_lo / rgs: the row groups (16 rows each) of the tile this workgroup owns (row-narrow teams; 0 / 4 = the whole tile): only their
// sequences are scored and written -- 16 must then be a multiple of nt, the launcher sees to it
__device__ __forceinline__ void h2_attention(bool WT, float* T, float* SC, int tid, int nt, int hd, int S, char* C2, int tile_m,
                                             int g_out, int Dq, const float* so, int rg_lo = 0, int rgs = 4) {
    const int hd4 = hd >> 2;
    const int HP = BN / hd, nn = nt * nt;
    const float scale = 1.0f / sqrtf((float)hd);
    const int sq_lo = rgs == 4 ? 0 : (16 * rg_lo) / nt, sq_hi = rgs == 4 ? S : (16 * (rg_lo + rgs)) / nt;
    for (int t = sq_lo * HP * nn + tid; t < sq_hi * HP * nn; t += 512) {
        const int j = t % nt, i = (t / nt) % nt, hh = (t / nn) % HP, sq = t / (nn * HP);
        const float* q = T + (sq * nt + i) * H2_ATT_TS + hh * hd;
        const float* k = T + (sq * nt + j) * H2_ATT_TS + BN + hh * hd;
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
    for (int t = sq_lo * HP * nt + tid; t < sq_hi * HP * nt; t += 512) {
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
    const int Go = Dq / BN;
    const int strip = h2_ksteps(Dq, NP) * H2_RG;                 // bytes of one row group of the output operand
    char* cbase = C2 + (size_t)tile_m * 4 * strip;
    auto pv4 = [&](int row, int c) -> float4 {
        float4 o = {0.f, 0.f, 0.f, 0.f};
        if (row < S * nt) {
            const int sq = row / nt, i = row - sq * nt;
            const int hh = c / hd;
            const float* pr = SC + ((sq * HP + hh) * nt + i) * nt;
            const float* v = T + (sq * nt) * H2_ATT_TS + 2 * BN + c;
            for (int j = 0; j < nt; ++j) {
                const float4 vv = ld4(v + j * H2_ATT_TS);
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
        if (rg < rg_lo || rg >= rg_lo + rgs) continue;
        const int row = rg * 16 + li;
        const float4 a = pv4(row, 32 * p + 4 * kq), b = pv4(row, 32 * p + 16 + 4 * kq);
        float4 sa = {1.f, 1.f, 1.f, 1.f}, sb = {1.f, 1.f, 1.f, 1.f};
        if (NP == 2) { sa = ld4(so + 32 * p + 4 * kq); sb = ld4(so + 32 * p + 16 + 4 * kq); }      // static scale per column
        const float x[8] = {a.x * sa.x, a.y * sa.y, a.z * sa.z, a.w * sa.w, b.x * sb.x, b.y * sb.y, b.z * sb.z, b.w * sb.w};
        h2_emit_frag<NP>(WT, cbase, 4 * g_out + p, (unsigned)(rg * strip + (kq * 16 + li) * 16), x);
    }
    for (int t = tid; t < BM * 2; t += 512) {
        const int li = t & 15, kq = (t >> 4) & 1, rg = t >> 5;
        if (rg < rg_lo || rg >= rg_lo + rgs) continue;
        const int row = rg * 16 + li;
        const float4 a = pv4(row, 128 + 4 * kq);
        float4 sa = {1.f, 1.f, 1.f, 1.f};
        if (NP == 2) sa = ld4(so + 128 + 4 * kq);
        const float x[8] = {a.x * sa.x, a.y * sa.y, a.z * sa.z, a.w * sa.w, 0.f, 0.f, 0.f, 0.f};
        h2_emit_tail<NP>(WT, cbase, 4 * Go + (g_out >> 2), (unsigned)(rg * strip + ((g_out & 3) * 16 + li) * 16 + kq * 8), x,
                         ((Dq / BK) & 1) && (g_out >> 2) == (Go >> 2) - 1);
    }
}

// One GEMM of one workgroup tile (tm, tn): everything a wave does for its NTW slots starting at slot `slot0`.
// CHAIN = false: the GEMM is a launch of its own.  CHAIN = true: one phase of h2_stack_kernel (see there): `chain` counts
// the arrivals of the team, the A operand (and the LayerNorm partials) may be read once it reaches `chain_need`, and this
// workgroup arrives when its outputs are written.  Returns false when the wait timed out (error words set, nothing computed).
// WC = W pieces this wave requests per stage (2 for the waves 0..3, 3 for the waves 4, 5, 2 for the waves 6, 7): a template
// parameter, so that neither the requests nor the counted waits need a branch in the k loop.
// RT = row tiles per workgroup.  RT = 2 (h2_stack2_kernel: teams that own two or more row tiles): the stage carries the A
// pieces of TWO row tiles (8 row groups, 16 KiB) against the same 18 KiB of W -- 34 KiB for 54 MFMAs per SIMD instead of 26 KiB
// for 27, the W fragments are read from LDS once for both tiles -- in a ring of 4 with one barrier per stage (a stage is as long
// as two of the RT = 1 stages, so that IS the two-stage barrier period).  `tm` then counts pairs of row tiles.  The arithmetic of
// every output element is the same in both forms (same k order, same product order per accumulator, same epilogue).
// NP = operand parts: 2 = fp16 hi | lo, three products per k-tile (fp32 arithmetic, h2_gemm.hip); 1 = bf16, a stage carries a
// PAIR of k-tiles (even | odd in the two KiB of every fragment slot) and issues one product per k-tile (b1_gemm.hip).  NP = 1
// takes every A operand packed (also the LayerNorm GEMMs: the residual epilogues emit x as a packed bf16 copy beside the
// fp32 rows) and applies a folded LayerNorm in the epilogue: rstd (acc - mean s_n) + c_n with s_n = sum_k of the packed
// gamma o W_n.
// ACT / rg_lo / rgs -- row-narrow teams (h2_stackn_kernel: launches that leave most of the chip idle, e.g. the reference's shipped
// call shape of 256 frames x 2 views = 8 row tiles): a workgroup owns only the row groups rg_lo .. rg_lo + rgs - 1 (16 or 32
// rows) of its 64-row tile, so 4 / rgs times as many workgroups share the launch.  The waves of those row groups run the phase
// exactly as in the full-tile form (ACT = true: same fragments, same product order, same epilogue -- the poses are bitwise the
// same); the waves of the other row groups (ACT = false) only keep their share of the W pieces moving and meet the barriers.
// the attention of the qkv phase runs in registers (a sequence = 2, 4 or 8 lanes of a 16-row group) for these shapes; else in LDS
__device__ __forceinline__ bool h2_att_in_registers(int ntok, int hd, int rpt) {
    return (ntok == 2 || ntok == 4 || ntok == 8) && (hd == 68 || hd == BN) && rpt == BM;
}

template <int EPI, bool LNF, int NPASS, int NTW, bool CHAIN, int WC, int RT = 1, int NP = 2, bool ACT = true, bool DW = false>
__device__ __forceinline__ bool h2_phase(const H2Args& a, char* smem, int tid, int wave, int slot0, int tm, int tn,
                                         unsigned* chain, unsigned chain_need, bool arrive = true, int rg_lo = 0, int rgs = 4) {
    static_assert(ACT || (RT == 1 && CHAIN), "loader-only waves exist in the row-narrow stack only");
    static_assert(!DW || (RT == 1 && CHAIN && NP == 2), "direct-W form: 16-row teams of the fp16x2 stack");
    constexpr int ABYTES = RT * 4 * H2_RG;       // A bytes per stage
    constexpr int STAGE = ABYTES + H2_W;
    constexpr int NST = RT == 1 ? H2_NST : 4;
    constexpr int APW = 2 * RT;                  // A pieces a wave 0..3 requests per A stage
    static_assert(NST * STAGE <= H2_VEC, "ring too large");
    static_assert(RT == 1 || (RT == 2 && CHAIN && (NP == 1 || (EPI != H2_EPI_ATT && NPASS <= 2))), "two row tiles per stage: proj / fc1 / fc2 of a stack (NP = 1: every phase)");
    constexpr bool RAWX = LNF && NP == 2;        // the A operand is the raw fp32 rows, normalised + split in the k loop
    // P2: one barrier per TWO stages.  At the barrier in front of an even stage e every wave has its pieces of the stages
    // <= e + 2 landed and has finished reading the fragments of the stages <= e, so stage e may refill the slot of stage e - 1
    // and stage e + 1 the slot of stage e: a refill goes DIST = NST - 1 stages ahead.  Nothing but the DMA landing has to be
    // published: the LayerNorm operand stays RAW in LDS and every wave that multiplies it normalises + splits its lane's eight
    // values in registers (both waves of a row group do the same arithmetic; the in-place conversion by the requesting wave
    // that this replaced needed every barrier and made the waves 0..3 the slow half of the stage).
    constexpr bool P2 = H2_KPS2 != 0 && RT == 1;
    constexpr int DIST = P2 ? NST - 1 : NST;
    constexpr bool LEAD = NTW == H2_T0;          // waves 0..3 (slots 0..4): multiply first, load afterwards; bring the epilogue vectors
    // A pieces: the waves 0..3 (the second row tile's pieces moved to the waves 4..7, with the 2 / 3 / 2 W split: measured 0 at
    // M = 8192 on the fp16x2 engine, tools/ab_rt2.sh, and removed again).  AB (bf16 pair form): the waves 4..7, which request at
    // the HEAD of a stage, bring ALL A pieces (four each) and the waves 0..3, which request behind their product rows, W only:
    // the A strips come from beyond L2 (their producers store write-through) and need the longer flight
    constexpr bool AB = RT == 2 && (NP == 1 ? H2_R2_AB != 0 : H2_R2_AB2 != 0);
    constexpr bool HAS_A = (AB ? !LEAD : LEAD) && ACT;
    const bool WT = CHAIN && !a.plain;
    const int lane = tid & 63;
    // DW: every wave works for row group rg_lo (the two multiplying waves sit on different SIMDs)
    const int rg = DW ? rg_lo : (wave & 3);
    const int li = lane & 15, kq = lane >> 4;
    const int M = a.M, N = a.N, K = a.K;
    const int m0 = tm * RT * a.rpt, n0 = NPASS == 2 ? tn * (2 * BN) : tn * BN;
    const int Dq = N / 3;
    const int KT = h2_ksteps(K, NP), G = K / BN;   // stages per pass
    const int T = NPASS * KT;                    // stages: stage u carries W of pass u % NPASS, and A when that pass is 0
    auto colbase = [&](int pass) -> int { return NPASS == 3 ? pass * Dq + n0 : n0 + pass * BN; };
    const unsigned long long t_entry = (H2_DBG && a.dbg) ? __builtin_amdgcn_s_memtime() : 0;

    const int row_l = rg * 16 + li;
    bool row_ok[RT];
    int row[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        row_ok[rt] = row_l < a.rpt && m0 + rt * a.rpt + row_l < M;
        row[rt] = row_ok[rt] ? m0 + rt * a.rpt + row_l : (M - 1);
    }

    // ---- DMA pieces of this wave.  W (18 per stage): waves 4, 5 pieces 0..2 / 3..5, waves 6, 7 pieces 6, 7 / 8, 9, wave
    // w < 4 pieces 10 + 2 w, 11 + 2 w (H2_WSPLIT 0: 4 / 4 / 3 / 3 and one each).  A (8 per A stage): waves 0..3 the two pieces
    // of row group `wave`.  (Moving ALL W pieces to the waves 4..7 paid while the waves 0..3 also converted the LayerNorm
    // operand in place, -3 %; it costs 1-5 % now.)
    // RT = 2: the waves 0..3 carry four A pieces, so they take one W piece each and the waves 4..7 four / three (the 1 / 4 / 3 split)
    constexpr bool WS1 = H2_WSPLIT == 1 && RT == 1;
    const int w_first = AB ? (LEAD ? 4 * wave : 12 + wave)
                      : WS1 ? (LEAD ? 10 + 2 * wave : (wave < 6 ? 3 * (wave - 4) : 6 + 2 * (wave - 6)))
                            : (LEAD ? 14 + wave : (wave < 6 ? 4 * (wave - 4) : 8 + 3 * (wave - 6)));
    static_assert(AB ? (LEAD ? WC == 4 : (WC == 1 || WC == 0))
                     : WS1 ? (LEAD ? WC == 2 : (WC == 3 || WC == 2)) : (LEAD ? WC == 1 : (WC == 3 || WC == 4)), "W pieces per wave");
    constexpr int w_cnt = WC;
    unsigned voW = (unsigned)(lane * 16 + w_first * 1024);
    // A source offsets of this lane.  Packed operand: 16 B per lane and part.  fp32 rows (LNF): the lane's 4 + 4 columns of
    // its row; full k-tiles start at column 4 kq (second piece +16 columns), tail k-tiles at 136 kq + 128 (second piece +4)
    unsigned voA[RT], voT[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        voA[rt] = (unsigned)(lane * 16);
        voT[rt] = 0;
        if (RAWX) {
            voA[rt] = (unsigned)(((size_t)(row[rt] - m0) * a.ldx + 4 * kq) * 4);
            voT[rt] = (unsigned)(((size_t)(row[rt] - m0) * a.ldx + 136 * kq + 128) * 4);
        }
        asm volatile("" : "+v"(voA[rt]), "+v"(voT[rt]));
    }
    asm volatile("" : "+v"(voW));
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    int iw_t = 0, iw_g = 0, ia_t = 0, ia_kt = 0;
    unsigned iw_slot = 0, ia_slot = 0;
    const char* is_w[NPASS];
#pragma unroll
    for (int g = 0; g < NPASS; ++g) is_w[g] = a.W2 + (size_t)(colbase(g) / BN) * KT * H2_W;
    // packed operand: the strip of row group (wave & 3) of row tile tm RT + rt; fp32 rows: the first row of the (pair of) tile(s)
    const char* is_a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
        is_a[rt] = RAWX ? reinterpret_cast<const char*>(a.X + (size_t)m0 * a.ldx)
                       : a.A2 + (((size_t)tm * RT + rt) * 4 + (wave & 3)) * KT * H2_RG;
    // bench-only (H2_ABL & 64): every phase streams the same 16 k-tiles of one column group of the first operand of the launch,
    // i.e. W out of a hot L2 (results are garbage): what the stream from beyond L2 costs
    int hotc[NPASS];
#pragma unroll
    for (int g = 0; g < NPASS; ++g) {
        hotc[g] = 0;
        if (H2_ABL & 64) is_w[g] = a.W2 + (size_t)tn * 17 * H2_W;
    }
    auto adv_w = [&](int g) {
        is_w[g] += H2_W;
        if (H2_ABL & 64) {
            if (++hotc[g] == (NP == 2 ? 16 : 8)) {
                hotc[g] = 0;
                is_w[g] -= (NP == 2 ? 16 : 8) * H2_W;
            }
        }
    };
    auto w_pieces = [&](const char* src, unsigned dst) {
        if constexpr (w_cnt == 0) return;
        asm volatile(
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1"
            :
            : "v"(voW), "s"(src), "s"(dst)
            : "memory");
        if (w_cnt > 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(voW), "s"(src) : "memory");
        if (w_cnt > 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(voW), "s"(src) : "memory");
        if (w_cnt > 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" : : "v"(voW), "s"(src) : "memory");
    };
    // the two A pieces of this wave's row group (of each of the RT row tiles) for A k-tile ia_kt into stage slot `slot` (waves
    // 0..3; M0 is the caller's); row tile rt lives in the row groups 4 rt .. 4 rt + 3 of the stage
    auto a_pieces = [&](unsigned slot) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned dst = lds0 + slot + (unsigned)(((wave & 3) + 4 * rt) * H2_RG);
            if (!RAWX) {
                if (CHAIN)
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 sc1"
                                 : : "v"(voA[rt]), "s"(is_a[rt]), "s"(dst) : "memory");
                else
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                                 : : "v"(voA[rt]), "s"(is_a[rt]), "s"(dst) : "memory");
                is_a[rt] += H2_RG;
            } else {
                // raw fp32: the lane's columns j = 0..3 land in the first KiB (where the hi fragment will be), j = 4..7 in the second;
                // the instruction offset would move BOTH addresses, so the second piece gets its own source base and M0
                const bool full = ia_kt < 4 * G;
                const char* src = full ? is_a[rt] + (size_t)(136 * (ia_kt >> 2) + 32 * (ia_kt & 3)) * 4 : is_a[rt] + (size_t)(136 * 4 * (ia_kt - 4 * G)) * 4;
                const char* src2 = src + (full ? 64 : 16);
                const unsigned vo = full ? voA[rt] : voT[rt];
                if (CHAIN)
                    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 sc1"
                                 : : "v"(vo), "s"(src), "s"(src2), "s"(dst), "s"(dst + 1024u) : "memory");
                else
                    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                                 : : "v"(vo), "s"(src), "s"(src2), "s"(dst), "s"(dst + 1024u) : "memory");
            }
        }
        ++ia_kt;
    };
    auto issue_w = [&]() {
        const unsigned keep = dma_m0_save();
#pragma unroll
        for (int g = 0; g < NPASS; ++g)
            if (g == iw_g) {
                w_pieces(is_w[g], lds0 + iw_slot + (unsigned)(ABYTES + w_first * 1024));
                adv_w(g);
            }
        if (++iw_g == NPASS) iw_g = 0;
        dma_m0_restore(keep);
        ++iw_t;
        iw_slot += STAGE;
        if (iw_slot == NST * STAGE) iw_slot = 0;
    };
    auto issue_a = [&]() {
        if ((ia_t % NPASS) == 0 && HAS_A) {
            const unsigned keep = dma_m0_save();
            a_pieces(ia_slot);
            dma_m0_restore(keep);
        }
        ++ia_t;
        ia_slot += STAGE;
        if (ia_slot == NST * STAGE) ia_slot = 0;
    };
    // steady state: stage t + NST lives where stage t lived; its pass (g + DIST) mod NPASS is a compile-time number at the call
    // site, and it carries A exactly when that pass is 0
    auto refill_fast = [&](auto wp_c, auto ai_c, unsigned slot) {
        constexpr int g = decltype(wp_c)::value;
        constexpr bool with_a = decltype(ai_c)::value;
        const unsigned keep = dma_m0_save();
        w_pieces(is_w[g], lds0 + slot + (unsigned)(ABYTES + w_first * 1024));
        adv_w(g);
        if (with_a && HAS_A) a_pieces(slot);
        dma_m0_restore(keep);
    };
    static_assert(RT == 2 || NST % NPASS == 0, "RT = 1 parks the row statistics in the A region of slot 1, which must only ever hold pass-1 stages");

    // ---- epilogue vectors c, sc of this workgroup's columns into the spare 4 KiB of LDS (front of the DMA queue), and behind
    // them the static scales so of the columns that leave as a packed operand (attention: the v pass; else every pass)
    constexpr int NSO = NP == 1 ? 0 : (EPI == H2_EPI_ATT ? 1 : (EPI == H2_EPI_RES ? 0 : NPASS));      // bf16 operands carry no scales
    constexpr int VSO = NPASS * 2 * BN;                            // float offset of the so block inside the vector region
    static_assert((NPASS * 2 + NSO) * BN * 4 <= 4092, "epilogue vectors overflow the spare LDS");
    if (LEAD) {
        constexpr int NV = NPASS * 2 * (BN / 4);                   // float4s: [pass][c | sc][34]
        int idx = wave * 64 + lane;
        const bool vec = idx < NV;
        const bool on = vec || (idx < NV + NSO * (BN / 4) && a.ovec != nullptr);
        const int q = idx - NV;                                    // so block: [pass (attention: the v pass only)][34]
        idx = vec ? idx : 0;
        const int vp = idx / (2 * (BN / 4)), which = (idx / (BN / 4)) & 1, c4 = idx % (BN / 4);
        const float* src = (which ? a.svec : a.cvec) + colbase(vp) + 4 * c4;
        if (!vec && on) src = a.ovec + colbase(EPI == H2_EPI_ATT ? 2 : q / (BN / 4)) + 4 * (q % (BN / 4));
        if (on) dma16(src, lds0 + (unsigned)(H2_VEC + wave * 1024));
    }
    // ---- prologue.  Chain mode: W(0) does not depend on the other workgroups and is requested BEFORE the wait for them;
    // the poll is the job of wave 7 (lane 0), its first look goes out before any DMA piece of the wave.
    bool arrived = !CHAIN;
    if (CHAIN && !LEAD && wave == 7)
        arrived = __hip_atomic_load(chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= chain_need;
    // ---- DW (direct-W form, 16-row teams): no ring.  The two multiplying waves take their W fragments straight from global
    // memory (L2) into registers, PD stages ahead; the whole A operand of the row group (2 KT KiB) is brought into LDS once,
    // by all eight waves, behind the hand-off.  Same fragments, same product order, same epilogue as the ring form.
    // Measured around it (round 5, V = 2 B = 256, stack 0.77 ms in the ring form -> 0.63 ms here; tools/ab.sh):
    //  * the k loop runs at 36 B/clk of W per CU -- the rate of TWO waves' vector loads whatever their depth (PD 2 = PD 4;
    //    tools/micro/l2_stream.hip: 2 waves 36, 4 or 8 waves 52-64 B/clk out of L2);
    //  * W through an LDS ring kept filled by the six idle waves, one barrier per stage (no vector loads in the multiplying
    //    waves): 0.62-0.64 ms; two slots per wave direct and the rest through that ring: 0.66 ms -- the stage of every form is
    //    430-490 cycles against 240 of its 15 MFMAs (tools/chain_phase.py): barrier round trip, fragment reads and more issue
    //    slots than the MFMA gaps hide;
    //  * more multiplying waves (one per pass and column half in the qkv / fc1 phases, accumulators handed to the q wave through
    //    LDS; a helper wave per column half in the one-pass phases): bitwise, 0.62 ms.  The W stream of the workgroup then comes in
    //    at 48-63 B/clk, i.e. at the 64 B/clk of a compute unit's vector L1: the k loops of a block application need 38 k cycles
    //    at that rate and take 46 k (two multiplying waves: 52 k); the other half of its ~100 k cycles are the phase entries
    //    (hand-off, A operand, LayerNorm conversion: 27 k) and the epilogues (25 k).  Every 16-row workgroup streams ALL of W for
    //    its 136 columns: only narrower column groups (more compute units per row tile) would cut that, and they would change
    //    the slice statistics and the attention sums, i.e. the bits;
    //  * nt loads 0.88 ms (every team then fetches W from beyond L2); L2 warming by the idle waves (one dword per line, by
    //    vector load or LDS-DMA) 0.86 ms: a sparse request costs the L1 as much as a full line; W served out of a hot L2 (every
    //    phase re-reading one 1.2 MB window, H2_ABL 64): -2 %, 0 for the whole-tile kernels, -7 % for bf16 at depth 12 -- the stream from beyond L2 is not the bound.
    constexpr int PD = NPASS == 3 ? 3 : 4;
    f16x8 Bb[DW && ACT ? PD : 1][NTW][2];
    auto dw_ld = [](const char* p) -> f16x8 { return *reinterpret_cast<const f16x8*>(p); };
    auto dw_fetch = [&](auto j_c, int kt) {
        constexpr int j = decltype(j_c)::value;
        if constexpr (DW && ACT) {
            const char* src = is_w[j % NPASS] + (size_t)kt * H2_W + (size_t)(slot0 * 2) * 1024 + lane * 16;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                Bb[j][n][0] = dw_ld(src + (n * 2 + 0) * 1024);
                Bb[j][n][1] = dw_ld(src + (n * 2 + 1) * 1024);
            }
        }
    };
    if constexpr (DW && EPI == H2_EPI_RES && ACT) {
        // the residual rows of this wave's outputs (fp32 rows this workgroup wrote two phases ago: beyond L2 by now, ~3.5 k
        // cycles away) by LDS-DMA, one KiB per column tile, requested FIRST: they do not depend on the partners, so they fly
        // during the hand-off; no register is held for them across the k loop (as ordinary loads near its end they cost its
        // first tail stage 3.5 k cycles), and the wait in front of the loop covers them
        const unsigned keep_r = dma_m0_save();
        const float* rb = a.R + (size_t)m0 * a.ldr;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            int c = 16 * h2_slot_tile(slot0 + n) + 4 * kq;
            c = c + 3 < BN ? c : 0;
            const unsigned off = (unsigned)(((size_t)(row[0] - m0) * a.ldr + n0 + c) * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1"
                         : : "v"(off), "s"(rb), "s"(lds0 + (unsigned)(H2_DW_RES + ((slot0 ? H2_T0 : 0) + n) * 1024)) : "memory");
        }
        dma_m0_restore(keep_r);
    }
    if constexpr (DW) {
        dw_fetch(std::integral_constant<int, 0>{}, 0);
        dw_fetch(std::integral_constant<int, 1>{}, 1 / NPASS);
        dw_fetch(std::integral_constant<int, 2>{}, 2 / NPASS);
        if constexpr (PD == 4) dw_fetch(std::integral_constant<int, 3>{}, 3 / NPASS);
    } else {
        issue_w();
    }
    if (CHAIN) {
        if (!LEAD && wave == 7 && !arrived) {
            const unsigned lim = 1u << a.spin_log2;
            unsigned spin = 0;
            for (; spin < lim; ++spin) {
                if (__hip_atomic_load(chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= chain_need) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (spin == lim && lane == 0) {     // a lost partner is an ERROR, never a licence to go on
                *reinterpret_cast<volatile unsigned*>(smem + H2_FAIL) = 1u;
                if (a.err_ws) __hip_atomic_store(a.err_ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.err_host) __hip_atomic_store(a.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (*reinterpret_cast<volatile unsigned*>(smem + H2_FAIL)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return false;
        }
    }
    const unsigned long long t_chain = (H2_DBG == 2 && a.dbg) ? __builtin_amdgcn_s_memtime() : 0;
    if constexpr (!DW) {
        issue_a();
    } else {
        // the A operand of row group rg_lo, k-tile kt at LDS offset 2048 kt: piece i = 2 kt + half, spread over the eight waves
        const int rows_lo = m0 + rg_lo * 16 + li;
        const int rowa = (rg_lo * 16 + li < a.rpt && rows_lo < M) ? rows_lo : (M - 1);
        const unsigned vA = RAWX ? (unsigned)(((size_t)(rowa - m0) * a.ldx + 4 * kq) * 4) : (unsigned)(lane * 16);
        const unsigned vT = RAWX ? (unsigned)(((size_t)(rowa - m0) * a.ldx + 136 * kq + 128) * 4) : 0u;
        const char* abase = RAWX ? reinterpret_cast<const char*>(a.X + (size_t)m0 * a.ldx) : a.A2 + ((size_t)tm * 4 + rg_lo) * KT * H2_RG;
        const unsigned keep = dma_m0_save();
        for (int i = wave; i < 2 * KT; i += 8) {
            const int kt = i >> 1, half = i & 1;
            const char* src;
            unsigned vo;
            if (RAWX) {
                const bool full = kt < 4 * G;
                src = full ? abase + (size_t)(136 * (kt >> 2) + 32 * (kt & 3)) * 4 + half * 64 : abase + (size_t)(136 * 4 * (kt - 4 * G)) * 4 + half * 16;
                vo = full ? vA : vT;
            } else {
                src = abase + (size_t)i * 1024;
                vo = vA;
            }
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" : : "v"(vo), "s"(src), "s"(lds0 + (unsigned)i * 1024u) : "memory");
        }
        dma_m0_restore(keep);
    }
    // LNF: the row statistics (slice partials {mean, M2} written by the producers of x) of this wave's 16 rows, requested
    // right behind A(0).  NPASS >= 2: by LDS-DMA (L1-bypassing in chain mode) into the A region of stage slot 1, which a
    // pass-1 stage never uses -- an ordinary load here would make the compiler drain the WHOLE queue (it cannot see the
    // LDS-DMA requests in it) in front of the first conversion.  NPASS == 1 (one-GEMM launches only): ordinary loads.
    // RT = 2: into the 20 KiB its ring of 4 x 34 KiB leaves free below the vector region (every stage of a one-pass GEMM carries A)
    constexpr unsigned ST_LDS = DW ? 72u * 1024u /* above the A operand: the launcher keeps 2 KiB x k-tiles below it */ : (RT == 2 ? NST * STAGE : STAGE);           // + 1 KiB per wave, + 8 KiB per row tile
    constexpr bool ST_DMA = NPASS >= 2 || RT == 2;
    static_assert(RT == 1 || ST_LDS + 16384 <= H2_VEC, "statistics rows overlap the epilogue vectors");
    float4 st_raw[4];
    if (LNF) {
        const int ns = K / BN;
        if constexpr (ST_DMA) {
            // lane l brings the partials of slices 2q, 2q+1 (16 B) of row 16 wave + l / (ns / 2), q = l % (ns / 2)
            const int hpr = ns >> 1;
            if (lane < 16 * hpr) {
                // lane / hpr by a multiplication (lane < 64, hpr <= 8: exact): a division by a run-time number costs a
                // reciprocal sequence that the compiler hoists out of the phase loop and keeps alive (it spilled)
                const unsigned inv = 65536u / (unsigned)hpr + 1u;
                const int lq = (int)(((unsigned)lane * inv) >> 16);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    int r = m0 + rt * a.rpt + 16 * rg + lq;
                    r = r < M ? r : M - 1;
                    const float* g = a.stats + ((size_t)r * ns + 2 * (lane - lq * hpr)) * 2;
                    unsigned keep;
                    if (CHAIN)
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(g), "s"(lds0 + ST_LDS + (unsigned)(wave * 1024 + rt * 8192)) : "memory");
                    else
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(g), "s"(lds0 + ST_LDS + (unsigned)(wave * 1024 + rt * 8192)) : "memory");
                }
            }
        } else {

            const float* sp = a.stats + (size_t)row[0] * ns * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) st_raw[i] = (2 * i < ns) ? ld4(sp + 4 * i) : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if constexpr (!DW) {
#pragma unroll
        for (int t = 1; t < DIST; ++t) {
            issue_w();
            issue_a();
        }
    }
    const float ainv = (!LNF && a.a_inv) ? a.a_inv[0] : 1.0f;

    // residual of this lane's outputs (fp32 rows this workgroup wrote itself two phases ago, or a previous launch wrote)
    float4 rv[RT][NTW];
    auto epilogue_operands = [&]() {
        if constexpr (EPI == H2_EPI_RES && DW) {
            // direct-W form: the residual rows were brought into LDS behind the hand-off (see there), a lane's 16 bytes per tile
#pragma unroll
            for (int n = 0; n < NTW; ++n)
                rv[0][n] = *reinterpret_cast<const float4*>(smem + H2_DW_RES + ((slot0 ? H2_T0 : 0) + n) * 1024 + lane * 16);
        } else if (EPI == H2_EPI_RES) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    int c = 16 * h2_slot_tile(slot0 + n) + 4 * kq;
                    c = c + 3 < BN ? c : 0;
                    const float* rp = a.R + (size_t)row[rt] * a.ldr + n0 + c;
                    if (CHAIN) rv[rt][n] = __builtin_bit_cast(float4, h2_ld16_l2(a.R + (size_t)m0 * a.ldr, (unsigned)((size_t)(rp - (a.R + (size_t)m0 * a.ldr)) * 4)));
                    else rv[rt][n] = ld4(rp);
                }
        }
    };

    f32x4 acc[NPASS][RT][NTW];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[p][rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- LNF: z = x cv_a + cv_b = (x - mean) rstd 2^10 of this lane's row, applied when the raw values are taken out of LDS
    // (NP = 1: cv_a = rstd, cv_b = mean of the row, applied by the epilogue)
    float cv_a[RT], cv_b[RT];
    float4 cvr0[RT], cvr1[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        cv_a[rt] = cv_b[rt] = 0.f;
        cvr0[rt] = cvr1[rt] = float4{0.f, 0.f, 0.f, 0.f};
    }

    f16x8 A0[RT][2], A1[RT][2];
    f16x8 B0[NTW][2], B1[NTW][2];
    // LNF: the lane's eight raw values of its row wait in cvr0 / cvr1 until
    // finish_a() normalises and splits them at the end of the stage
    auto read_a = [&](unsigned slot, f16x8 (&f)[RT][2]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if constexpr (RAWX) {
                const char* p = smem + slot + (rg + 4 * rt) * H2_RG + lane * 16;
                cvr0[rt] = *reinterpret_cast<const float4*>(p);
                cvr1[rt] = *reinterpret_cast<const float4*>(p + 1024);
            } else {
                const f16x8* as = reinterpret_cast<const f16x8*>(smem + slot + (rg + 4 * rt) * H2_RG) + lane;
                f[rt][0] = as[0];
                f[rt][1] = as[64];
            }
        }
    };
    auto finish_a = [&](f16x8 (&f)[RT][2]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float z[8] = {cvr0[rt].x, cvr0[rt].y, cvr0[rt].z, cvr0[rt].w, cvr1[rt].x, cvr1[rt].y, cvr1[rt].z, cvr1[rt].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                z[j] = fmaf(z[j], cv_a[rt], cv_b[rt]);        // rows beyond the tile: cv_a = cv_b = 0 (set with the statistics): z = 0
                z[j] = __builtin_amdgcn_fmed3f(z[j], -65000.0f, 65000.0f);     // never an inf in an operand, whatever the statistics
            }
            split2(z, f[rt][0], f[rt][1]);
        }
    };
    auto read_b = [&](unsigned slot, f16x8 (&f)[NTW][2]) {
        const f16x8* bs = reinterpret_cast<const f16x8*>(smem + slot + ABYTES) + slot0 * 2 * 64 + lane;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            f[n][0] = bs[(n * 2 + 0) * 64];
            f[n][1] = bs[(n * 2 + 1) * 64];
        }
    };
    auto slot_after = [&](unsigned sl) -> unsigned { return sl + STAGE == NST * STAGE ? 0u : sl + STAGE; };
    // Three products, fixed order (A part . W part): lo.hi, hi.lo, hi.hi, each over the wave's NTW tiles.  The W fragment is
    // the FIRST MFMA operand: lane (i, kq) then holds C[row i][4 consecutive columns 16 tile + 4 kq ..].
    // (per accumulator; with RT = 2 the same W fragment serves both row tiles)
    auto mfma_row = [&](f32x4 (&accp)[RT][NTW], const f16x8 (&af)[RT][2], int ap, const f16x8 (&bf)[NTW][2], int bp) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
                if (!(H2_ABL & 8)) {
                    if constexpr (NP == 2)
                        accp[rt][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[n][bp], af[rt][ap], accp[rt][n], 0, 0, 0);
                    else
                        accp[rt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[n][bp]), __builtin_bit_cast(bf16x8, af[rt][ap]),
                                                                              accp[rt][n], 0, 0, 0);
                }
    };
    // counted waits.  Per stage a wave of the waves 4..7 requests WC (4 or 3) pieces, a wave 0..3 one W piece plus two A
    // pieces in the A stages (every NPASS-th).  One barrier per stage: the stages t+1 .. t+5 are in flight when stage t
    // starts and stage t+1 is needed, so four stages may stay in flight -- at least 4 WC + 2 (A stages among four consecutive
    // ones: 4, 2, 1 for NPASS 1, 2, 3) pieces of this wave.  P2: see the stage.
    // In general (no P2): the stages t+2 .. t+NST-1 may stay in flight at the barrier in front of stage t.
    unsigned long long t_land = 0;
    {   // stage 0 (P2: the stages 0, 1, 2) landed; later ones may stay in flight: of the stages 1 .. 5, 5 / 2 / 1 carry A for
        // NPASS 1 / 2 / 3, of the stages 3, 4 (P2) 2 / 1 / 1
        if constexpr (DW) {
            // the A operand and the statistics rows landed (DMA); the W fragments of the first PD stages are ordinary loads the
            // compiler counts itself: the NTW * 2 * PD of them were requested BEFORE the DMA pieces, so they have landed too
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (HAS_A) {
            // A stages among the stages 1 .. DIST - 1: every NPASS-th
            constexpr int LATER = P2 ? 2 * WC + APW * (NPASS == 1 ? 2 : 1) : (DIST - 1) * WC + APW * ((DIST - 1) / NPASS);
            static_assert(LATER < 64, "vmcnt is 6 bits");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LATER) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P2 ? 2 * WC : (DIST - 1) * WC) : "memory");
        }
        if (H2_DBG == 2 && a.dbg) t_land = __builtin_amdgcn_s_memtime();
        {
            if (LNF) {
                // Chan's combination of the per-slice {mean, M2} partials (fixed order)
                const int ns = K / BN;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if constexpr (ST_DMA) {
                        const float* sl = reinterpret_cast<const float*>(smem + ST_LDS + wave * 1024 + rt * 8192) + li * ns * 2;
#pragma unroll
                        for (int i = 0; i < 4; ++i) st_raw[i] = (2 * i < ns) ? ld4(sl + 4 * i) : float4{0.f, 0.f, 0.f, 0.f};
                    }
                    float st[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        st[4 * i] = st_raw[i].x; st[4 * i + 1] = st_raw[i].y; st[4 * i + 2] = st_raw[i].z; st[4 * i + 3] = st_raw[i].w;
                    }
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
                    const float rs = 1.0f / sqrtf(fmaf(m2, 1.0f / (float)K, a.eps));
                    if constexpr (NP == 2) {
                        // a row beyond the tile (its loads are clamped to the last row) multiplies as zeros: 0 * x + 0 in finish_a
                        // instead of a select per value there (8 v_cndmask per conversion and wave, round 6)
                        cv_a[rt] = row_ok[rt] ? rs * H2_SA : 0.f;
                        cv_b[rt] = row_ok[rt] ? -mean * (rs * H2_SA) : 0.f;
                    } else {            // folded LayerNorm, applied by the epilogue: rstd (acc - mean s_n) + c_n
                        cv_a[rt] = rs;
                        cv_b[rt] = mean;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (ACT && !DW) {
            read_a(0, A0);
            if (RAWX) finish_a(A0);
            read_b(0, B0);
        }
        if constexpr (DW && RAWX) {
            // the raw rows become the hi | lo fragments IN PLACE (a lane's eight values are its own 16 + 16 bytes), one k-tile per
            // wave and turn: the conversion leaves the serial chain of the two multiplying waves (it was ~300 cycles per k-tile there)
            for (int kt = wave; kt < KT; kt += 8) {
                char* p = smem + kt * 2048 + lane * 16;
                cvr0[0] = *reinterpret_cast<const float4*>(p);
                cvr1[0] = *reinterpret_cast<const float4*>(p + 1024);
                finish_a(A0);
                *reinterpret_cast<f16x8*>(p) = A0[0][0];
                *reinterpret_cast<f16x8*>(p + 1024) = A0[0][1];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    const unsigned long long t_loop = (H2_DBG && a.dbg) ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long t_vm = 0, t_bar = 0, t_mm = 0;      // bench-only sums: counted DMA wait, lgkm + barrier, the MFMA rows of a stage
    unsigned slot_c = 0;
    // one stage: publish the landed stages (which frees a slot for the DMA of stage t + DIST), then the MFMAs of stage t and the
    // fragment reads of stage t+1.  REM = 0: a stage of the steady state (t + NST < T); REM > 0: a stage of the tail with REM
    // stages left including this one -- a compile-time number, so that the counted waits, the last refills and the end of the
    // fragment reads need neither bookkeeping nor branches (a tail stage with run-time bookkeeping cost ~1500 cycles against
    // ~800 of a steady-state one, and a quarter of all stages are tail stages).
    auto stage = [&](auto rem_c, auto wp_c, f32x4 (&accp)[RT][NTW], const f16x8 (&a_cur)[RT][2], f16x8 (&a_nxt)[RT][2],
                     const f16x8 (&b_cur)[NTW][2], f16x8 (&b_nxt)[NTW][2], auto nha_c, auto sync_c) {
        constexpr int REM = decltype(rem_c)::value;
        constexpr bool FAST = REM == 0;
        constexpr bool SYNC = !(P2 && FAST) || decltype(sync_c)::value;      // P2: the odd fast stages run without a barrier
        constexpr bool next_has_a = decltype(nha_c)::value;
        constexpr bool more = FAST || REM > 1;
        constexpr bool RF = FAST || REM > DIST;                              // this stage requests stage t + DIST
        constexpr int g0 = decltype(wp_c)::value;                            // pass of this stage
        // the ring position is opaque here: in the straight-line tail the compiler otherwise forms the LDS addresses of all the
        // remaining stages up front (+40 VGPRs, spills)
        asm volatile("" : "+s"(slot_c));
        const unsigned slot_n = slot_after(slot_c);
        const unsigned slot_p = slot_c == 0 ? (unsigned)((NST - 1) * STAGE) : slot_c - STAGE;   // ring slot of stage t - 1
        unsigned long long w0 = 0, w1 = 0;
        if (H2_DBG && a.dbg) w0 = __builtin_amdgcn_s_memtime();
        if (more && SYNC) {
            if (FAST) {
                if (P2) {   // stages <= t + 2 landed; t + 3, t + 4 may stay in flight (the refills of this period come later)
                    constexpr int na = ((g0 + 3) % NPASS == 0 ? 1 : 0) + ((g0 + 4) % NPASS == 0 ? 1 : 0);   // A stages among them
                    if (HAS_A) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WC + APW * na) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WC) : "memory");
                } else {    // stages <= t + 1 landed; t + 2 .. t + NST - 1 may stay in flight
                    constexpr int na = ((g0 + 2) % NPASS == 0 ? 1 : 0) + ((g0 + 3) % NPASS == 0 ? 1 : 0) +
                                       (NST > 4 ? ((g0 + 4) % NPASS == 0 ? 1 : 0) + ((g0 + 5) % NPASS == 0 ? 1 : 0) : 0);
                    static_assert(NST == 4 || NST == 6, "in-flight stages of the one-barrier-per-stage form");
                    static_assert((NST - 2) * WC + APW * na < 64, "vmcnt is 6 bits");
                    if (HAS_A) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * WC + APW * na) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * WC) : "memory");
                }
            } else {
                // tail (a barrier in front of every stage).  Requested so far: the stages <= min(T - 1, t + DIST - 1); needed:
                // <= t + 1.  The stages in between may stay in flight, counted with the FEWEST pieces this wave has per stage
                constexpr int ahead = (REM - 1 < DIST - 1 ? REM - 1 : DIST - 1) - 1;
                constexpr int per = HAS_A ? WC + (NPASS == 1 ? APW : 0) : WC;
                constexpr int allow = ahead > 0 ? ahead * per : 0;
                static_assert(allow < 64, "vmcnt is 6 bits");
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow) : "memory");
            }
            if (H2_DBG && a.dbg) { w1 = __builtin_amdgcn_s_memtime(); t_vm += w1 - w0; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(H2_ABL & 32)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (H2_DBG && a.dbg) t_bar += __builtin_amdgcn_s_memtime() - w1;
        }
        const f16x8* bs = reinterpret_cast<const f16x8*>(smem + slot_n + ABYTES) + slot0 * 2 * 64 + lane;
        auto rd_b = [&](int n) {
            if (more && n < NTW && !(H2_ABL & 1)) {
                b_nxt[n][0] = bs[(n * 2 + 0) * 64];
                b_nxt[n][1] = bs[(n * 2 + 1) * 64];
            }
        };
        auto loads = [&]() {
            if constexpr (ACT) {
                if (more && next_has_a && !(H2_ABL & 4)) read_a(slot_n, a_nxt);
                rd_b(0); rd_b(1); rd_b(2); rd_b(3); rd_b(4);
            }
            // the requested stage t + DIST has the pass (g + DIST) mod NPASS (= g when DIST = NST) and carries A when that is 0
            constexpr int gr = (g0 + DIST) % NPASS;
            if (RF && !(H2_ABL & 2))
                refill_fast(std::integral_constant<int, gr>{}, std::integral_constant<bool, gr == 0>{}, P2 ? slot_p : slot_c);
            if (REM == 4 && ACT) epilogue_operands();
        };
        // The two waves of a SIMD run out of phase: the waves 0..3 multiply first and load afterwards, the waves 4..7 the other
        // way round -- while one wave of a SIMD sits in its DMA requests and fragment reads, the other one has the matrix pipe.
        // (Interleaving rows, reads and requests -- down to one load operation behind every MFMA --, or reads first in both
        // halves: +1 .. +11 % time)
        __builtin_amdgcn_sched_barrier(0);
        if (!LEAD) {
            loads();
            __builtin_amdgcn_sched_barrier(0);
        }
        unsigned long long m0 = 0;
        if (H2_DBG && a.dbg) m0 = __builtin_amdgcn_s_memtime();
        if constexpr (!ACT) {
        } else if constexpr (NP == 2) {
            mfma_row(accp, a_cur, 1, b_cur, 0);         // lo . hi
            mfma_row(accp, a_cur, 0, b_cur, 1);         // hi . lo
            mfma_row(accp, a_cur, 0, b_cur, 0);         // hi . hi
        } else {
            mfma_row(accp, a_cur, 0, b_cur, 0);         // even k-tile of the pair
            mfma_row(accp, a_cur, 1, b_cur, 1);         // odd k-tile
        }
        __builtin_amdgcn_sched_barrier(0);
        if (H2_DBG && a.dbg) t_mm += __builtin_amdgcn_s_memtime() - m0;
        if (LEAD) loads();
        if (RAWX && ACT && more && next_has_a && !(H2_ABL & 16)) finish_a(a_nxt);
        __builtin_amdgcn_sched_barrier(0);
        slot_c = slot_n;
    };
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;
    using YES = std::integral_constant<bool, true>;
    using NO = std::integral_constant<bool, false>;
    // Stage t = NPASS kt + j: pass j, carries A when j == 0; the fragment registers alternate per k-tile (A) and per stage (B);
    // the last two arguments: the NEXT stage carries A / (P2) a barrier stands in front of this stage.  POS = t mod 2 NPASS
    // picks the row of that table.
    constexpr int U = 2 * NPASS;
    auto step = [&](auto rem_c, auto pos_c) {
        constexpr int POS = decltype(pos_c)::value;
        if constexpr (NPASS == 1) {
            if constexpr (POS == 0) stage(rem_c, W0{}, acc[0], A0, A1, B0, B1, YES{}, YES{});
            else stage(rem_c, W0{}, acc[0], A1, A0, B1, B0, YES{}, NO{});
        } else if constexpr (NPASS == 2) {
            if constexpr (POS == 0) stage(rem_c, W0{}, acc[0], A0, A0, B0, B1, NO{}, YES{});
            else if constexpr (POS == 1) stage(rem_c, W1{}, acc[1], A0, A1, B1, B0, YES{}, NO{});
            else if constexpr (POS == 2) stage(rem_c, W0{}, acc[0], A1, A1, B0, B1, NO{}, YES{});
            else stage(rem_c, W1{}, acc[1], A1, A0, B1, B0, YES{}, NO{});
        } else {
            // three stages per k-tile flip the B parity every k-tile
            if constexpr (POS == 0) stage(rem_c, W0{}, acc[0], A0, A0, B0, B1, NO{}, YES{});
            else if constexpr (POS == 1) stage(rem_c, W1{}, acc[1], A0, A0, B1, B0, NO{}, NO{});
            else if constexpr (POS == 2) stage(rem_c, W2{}, acc[2], A0, A1, B0, B1, YES{}, YES{});
            else if constexpr (POS == 3) stage(rem_c, W0{}, acc[0], A1, A1, B1, B0, NO{}, NO{});
            else if constexpr (POS == 4) stage(rem_c, W1{}, acc[1], A1, A1, B0, B1, NO{}, YES{});
            else stage(rem_c, W2{}, acc[2], A1, A0, B1, B0, YES{}, NO{});
        }
    };
    auto tail = [&](auto self, auto rem_c, auto pos_c) -> void {
        constexpr int REM = decltype(rem_c)::value, POS = decltype(pos_c)::value;
        step(rem_c, pos_c);
        if constexpr (REM > 1) self(self, std::integral_constant<int, REM - 1>{}, std::integral_constant<int, (POS + 1) % U>{});
    };
    using R0 = std::integral_constant<int, 0>;
    // steady state: whole groups of U stages while the last of them still has a stage to request; what is left is one of two
    // compile-time stage counts (T = NPASS KT is 0 or NPASS modulo U), both starting at POS 0
    if constexpr (DW) {
        if constexpr (ACT) {
            // barrier-free k loop: A fragment of k-tile kt out of LDS, W fragments out of the register ring.  One wave per SIMD
            // multiplies here, so the refill of a fragment register goes out right behind the LAST product that reads it (one
            // load behind every MFMA of the second and third product row): the request overlaps the matrix pipe instead of
            // queueing behind the stage (a block of 8 / 10 loads behind the rows cost 0.18 of 0.71 ms, tools/ab.sh)
            // JJ = position of the stage behind u0 (a multiple of PD): register slot JJ % PD, pass JJ % NPASS; MORE = a stage PD
            // further on exists and is requested.  Everything is a compile-time number -- also in the tail, whose length is one
            // of PD .. 2 PD - 1: behind a run-time branch the compiler waits for ALL loads in flight (the tail stages of the first
            // version took 600-1 300 cycles against 420 of a steady one: ~2.7 k cycles per phase, tools-side stamps, round 5)
            auto dw_stage = [&](auto jj_c, auto more_c, int u0) {
                constexpr int JJ = decltype(jj_c)::value;
                constexpr bool MORE = decltype(more_c)::value;
                constexpr int j = JJ % PD, g = JJ % NPASS;
                const int kt = u0 / NPASS + JJ / NPASS;
                if (g == 0 && !(H2_ABL & 4)) {
                    // (reading the fragment one k-tile ahead: 0 -- and two spilled registers)
                    const char* p = smem + kt * 2048 + lane * 16;
                    A0[0][0] = *reinterpret_cast<const f16x8*>(p);
                    A0[0][1] = *reinterpret_cast<const f16x8*>(p + 1024);
                }
                constexpr bool more = MORE && !(H2_ABL & 2);
                const char* src = is_w[g] + (size_t)(kt + PD / NPASS) * H2_W + (size_t)(slot0 * 2) * 1024 + lane * 16;
                auto mm = [&](int n, int ap, int bp) {
                    if (!(H2_ABL & 8)) acc[g][0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bb[j][n][bp], A0[0][ap], acc[g][0][n], 0, 0, 0);
                };
#pragma unroll
                for (int n = 0; n < NTW; ++n) mm(n, 1, 0);                   // lo . hi
#pragma unroll
                for (int n = 0; n < NTW; ++n) {                              // hi . lo
                    mm(n, 0, 1);
                    if constexpr (more) Bb[j][n][1] = dw_ld(src + (n * 2 + 1) * 1024);
                }
#pragma unroll
                for (int n = 0; n < NTW; ++n) {                              // hi . hi
                    mm(n, 0, 0);
                    if constexpr (more) Bb[j][n][0] = dw_ld(src + (n * 2 + 0) * 1024);
                }
                if constexpr (more && H2_DW_PIN) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);
#pragma unroll
                    for (int i = 0; i < 2 * NTW; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            };
            static_assert(PD % NPASS == 0, "a register slot always holds the same pass");
            using YES = std::integral_constant<bool, true>;
            int u0 = 0;
            for (; u0 + 2 * PD <= T; u0 += PD) {                 // every stage of the group exists and has a successor to request
                dw_stage(std::integral_constant<int, 0>{}, YES{}, u0);
                dw_stage(std::integral_constant<int, 1>{}, YES{}, u0);
                dw_stage(std::integral_constant<int, 2>{}, YES{}, u0);
                if constexpr (PD == 4) dw_stage(std::integral_constant<int, 3>{}, YES{}, u0);
            }
            // the tail: R = T - u0 stages are left, PD <= R < 2 PD, R a multiple of NPASS
            auto tail = [&](auto self, auto r_c, auto jj_c) -> void {
                constexpr int R = decltype(r_c)::value, JJ = decltype(jj_c)::value;
                dw_stage(jj_c, std::integral_constant<bool, (JJ + PD < R)>{}, u0);
                if constexpr (JJ + 1 < R) self(self, r_c, std::integral_constant<int, JJ + 1>{});
            };
            using Z = std::integral_constant<int, 0>;
            const int rest = T - u0;
            // each tail inside a loop that runs exactly once behind an opaque trip count: with the back edge the accumulators are
            // loop-carried values that end where they started, which keeps the allocator from renaming them MFMA by MFMA
            // (out-of-place v_mfma + hundreds of spills; the same device as H2_TAIL_LOOP of the ring form)
            int tail_rep = 1;
            asm volatile("" : "+s"(tail_rep));
            if (rest == PD) {
                do tail(tail, std::integral_constant<int, PD>{}, Z{}); while (--tail_rep);
            }
            if constexpr ((PD + 1) % NPASS == 0) {
                if (rest == PD + 1) {
                    do tail(tail, std::integral_constant<int, PD + 1>{}, Z{}); while (--tail_rep);
                }
            }
            if constexpr ((PD + 2) % NPASS == 0) {
                if (rest == PD + 2) {
                    do tail(tail, std::integral_constant<int, PD + 2>{}, Z{}); while (--tail_rep);
                }
            }
            if constexpr (PD == 4 && (PD + 3) % NPASS == 0) {
                if (rest == PD + 3) {
                    do tail(tail, std::integral_constant<int, PD + 3>{}, Z{}); while (--tail_rep);
                }
            }
            epilogue_operands();
        }
    } else {
    int t = 0;
    for (; t + U - 1 + NST < T; t += U) {
        step(R0{}, std::integral_constant<int, 0>{});
        step(R0{}, std::integral_constant<int, 1 % U>{});
        if constexpr (NPASS >= 2) {
            step(R0{}, std::integral_constant<int, 2 % U>{});
            step(R0{}, std::integral_constant<int, 3 % U>{});
        }
        if constexpr (NPASS == 3) {
            step(R0{}, std::integral_constant<int, 4 % U>{});
            step(R0{}, std::integral_constant<int, 5 % U>{});
        }
    }
    // T - t is in [NST, NST + U - 1] and congruent to T, i.e. to 0 or NPASS, modulo U: two possible tail lengths
    constexpr int TAIL_A = (NST + U - 1) / U * U;
    constexpr int TAIL_B = (NST - NPASS + U - 1) / U * U + NPASS;
    // H2_TAIL_LOOP: the straight-line tail sits inside a loop that runs exactly once behind an opaque trip count.  With the
    // back edge the accumulators are loop-carried values that must end where they started, which keeps the allocator from
    // renaming them MFMA by MFMA (out-of-place v_mfma + spills of the pass that is idle) towards the epilogue's assignments.
    constexpr bool TAIL_LOOP = H2_TAIL_LOOP && RT == 2 && NP == 1;
    int tail_rep = 1;
    if (TAIL_LOOP) asm volatile("" : "+s"(tail_rep));
    if (T - t == TAIL_A) {
        do tail(tail, std::integral_constant<int, TAIL_A>{}, std::integral_constant<int, 0>{});
        while (TAIL_LOOP && --tail_rep);
    } else {
        do tail(tail, std::integral_constant<int, TAIL_B>{}, std::integral_constant<int, 0>{});
        while (TAIL_LOOP && --tail_rep);
    }
    }

    // ------------------------------------------------------------------------------------------ epilogue
    const unsigned long long t_epi = (H2_DBG && a.dbg) ? __builtin_amdgcn_s_memtime() : 0;
    // acc[p][rt][n][r] = scaled C[row_l of row tile rt][colbase(p) + 16 tile(n) + 4 kq + r];   value = acc * sc_n (* 1 / a_scale) + c_n
    auto tile_of = [&](int n) -> int { return h2_slot_tile(slot0 + n); };
    auto value4 = [&](int p, int rt, int n, float (&v)[4]) {
        const int cl = 16 * tile_of(n) + 4 * kq;
        const bool ok = cl + 3 < BN;
        const float* vecs = reinterpret_cast<const float*>(smem + H2_VEC) + p * 2 * BN + (ok ? cl : 0);
        const float4 cv = ld4(vecs), sv = ld4(vecs + BN);
        const float c4[4] = {cv.x, cv.y, cv.z, cv.w};
        const float s4[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // explicit fused operations: the same roundings in every instantiation (chain phases and one-GEMM launches agree
            // bitwise); sc_n, 1 / a_scale are powers of two: their product is exact
            float t;
            if constexpr (NP == 2) t = fmaf(acc[p][rt][n][r], LNF ? s4[r] : s4[r] * ainv, c4[r]);
            else if constexpr (LNF) t = fmaf(cv_a[rt], fmaf(-cv_b[rt], s4[r], acc[p][rt][n][r]), c4[r]);
            else t = acc[p][rt][n][r] + c4[r];
            if (EPI == H2_EPI_GELU) t = gelu_as(t);
            v[r] = ok ? t : 0.f;
        }
    };
    // static scales of the 4 columns of tile n as they leave packed (so block p of the vector region)
    auto oscale4 = [&](int p, int n, float (&o)[4]) {
        const int cl = 16 * tile_of(n) + 4 * kq;
        const float4 sv = ld4(reinterpret_cast<const float*>(smem + H2_VEC) + VSO + p * BN + (cl + 3 < BN ? cl : 0));
        o[0] = sv.x; o[1] = sv.y; o[2] = sv.z; o[3] = sv.w;
    };
    unsigned long long t_st = 0;

    if constexpr (!ACT) {
        // loader-only wave of a row-narrow workgroup: nothing to compute or store; it meets the barriers of the epilogue below
        // (and lends its threads to the LDS form of the attention, which every thread of the workgroup walks)
        if constexpr (EPI == H2_EPI_ATT) {
            if (h2_att_in_registers(a.att_ntok, a.att_hd, a.rpt)) {
                __syncthreads();
                __syncthreads();
            } else {
                float* Tt = reinterpret_cast<float*>(smem);
                __syncthreads();
                __syncthreads();
                h2_attention<NP>(WT, Tt, Tt + BM * H2_ATT_TS, tid, a.att_ntok, a.att_hd, a.rpt / a.att_ntok, a.C2, tm, n0 / BN, Dq,
                                 reinterpret_cast<const float*>(smem + H2_VEC) + VSO, rg_lo, rgs);
            }
        } else if constexpr (EPI == H2_EPI_RES) {
            if (a.stats_out) {
                __syncthreads();
                __syncthreads();
                __syncthreads();
            }
        }
    } else {
    if constexpr (EPI == H2_EPI_ATT) {
      if (h2_att_in_registers(a.att_ntok, a.att_hd, a.rpt)) {
        // ---- Attention.forward :55-64 for 4 tokens per sequence and 68- or 136-wide heads, in REGISTERS: a
        // sequence is the 4 lanes of a quad, k_j / v_j come by DPP quad broadcast, the q.k sums are reduced over the tiles of
        // the wave, the 4 kq lanes and -- through 2 KiB of LDS -- the two waves of the row group.
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
        float qv[NTW][4], kv[NTW][4], vv[NTW][4];
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            value4(0, rt, n, qv[n]);
            value4(1, rt, n, kv[n]);
            value4(2, rt, n, vv[n]);
        }
        float ov[NTW][4];
        // 2 or 8 tokens per sequence (two views: the reference's shipped configuration; eight: BASELINE configs[2]): the same scheme
        // with the partner i ^ r of token i instead of a broadcast -- quad permutes for r < 4, the half-row mirror (i -> 7 - i) in
        // front of them for r >= 4 -- so that a lane meets every key / value of its sequence in 1 + (NTK - 1) DPP moves per
        // value; the probabilities of a row are indexed by r (key token i ^ r), a permutation the softmax does not care about.
        // (Round 4 ran these shapes through the q | k | v tile in LDS: 20.7 k of the 45.6 k cycles of a qkv phase at 8 views.)
        auto att_xor = [&](auto ntk_c) {
            constexpr int NTK = decltype(ntk_c)::value;
            auto mirror = [](float x) -> float {
                const int xi = __builtin_bit_cast(int, x);
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x141, 0xf, 0xf, false));      // row_half_mirror
            };
            auto xpart = [](float x, float xm, int r) -> float {       // the value lane i ^ r of the 8-lane half row holds
                const int xi = __builtin_bit_cast(int, x), mi = __builtin_bit_cast(int, xm);
                int o;
                switch (r) {
                    case 0: o = xi; break;
                    case 1: o = __builtin_amdgcn_update_dpp(xi, xi, 0xb1, 0xf, 0xf, false); break;     // quad_perm [1,0,3,2]
                    case 2: o = __builtin_amdgcn_update_dpp(xi, xi, 0x4e, 0xf, 0xf, false); break;     // [2,3,0,1]
                    case 3: o = __builtin_amdgcn_update_dpp(xi, xi, 0x1b, 0xf, 0xf, false); break;     // [3,2,1,0]
                    case 4: o = __builtin_amdgcn_update_dpp(mi, mi, 0x1b, 0xf, 0xf, false); break;     // (i ^ 7) ^ 3
                    case 5: o = __builtin_amdgcn_update_dpp(mi, mi, 0x4e, 0xf, 0xf, false); break;     // (i ^ 7) ^ 2
                    case 6: o = __builtin_amdgcn_update_dpp(mi, mi, 0xb1, 0xf, 0xf, false); break;     // (i ^ 7) ^ 1
                    default: o = mi; break;                                                            // i ^ 7
                }
                return __builtin_bit_cast(float, o);
            };
            const bool two_heads = a.att_hd == 68;
            float s0[NTK], s1[NTK];
#pragma unroll
            for (int r = 0; r < NTK; ++r) s0[r] = s1[r] = 0.f;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const bool h1 = two_heads && 4 * tile_of(n) + kq >= 17;
                float km[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (NTK > 4) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) km[c] = mirror(kv[n][c]);
                }
#pragma unroll
                for (int r = 0; r < NTK; ++r) {
                    float d = qv[n][0] * xpart(kv[n][0], km[0], r);
                    d = fmaf(qv[n][1], xpart(kv[n][1], km[1], r), d);
                    d = fmaf(qv[n][2], xpart(kv[n][2], km[2], r), d);
                    d = fmaf(qv[n][3], xpart(kv[n][3], km[3], r), d);
                    s0[r] += h1 ? 0.f : d;
                    s1[r] += h1 ? d : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < NTK; ++r) {
                s0[r] = xor16_add(s0[r]); s0[r] = xor32_add(s0[r]);
                s1[r] = xor16_add(s1[r]); s1[r] = xor32_add(s1[r]);
            }
            float* xs = reinterpret_cast<float*>(smem) + rt * (2 * BM * 2 * NTK);     // [row tile][2 halves][64 rows][2 heads][NTK]
            const int half = slot0 ? 1 : 0;
            __syncthreads();                                    // every wave is done reading the last stage
            if (kq == 0) {
#pragma unroll
                for (int r = 0; r < NTK; ++r) {
                    xs[(half * BM + row_l) * 2 * NTK + r] = s0[r];
                    xs[(half * BM + row_l) * 2 * NTK + NTK + r] = s1[r];
                }
            }
            __syncthreads();
            float p0[NTK], p1[NTK];
            {
                const float scale = 1.0f / sqrtf((float)a.att_hd);
                float t0[NTK], t1[NTK];
#pragma unroll
                for (int r = 0; r < NTK; ++r) {
                    t0[r] = (xs[row_l * 2 * NTK + r] + xs[(BM + row_l) * 2 * NTK + r]) * scale;
                    t1[r] = (xs[row_l * 2 * NTK + NTK + r] + xs[(BM + row_l) * 2 * NTK + NTK + r]) * scale;
                }
                auto softmax = [](const float (&t)[NTK], float (&pr)[NTK]) {
                    float mx = t[0];
#pragma unroll
                    for (int r = 1; r < NTK; ++r) mx = fmaxf(mx, t[r]);
                    float l = 0.f;
#pragma unroll
                    for (int r = 0; r < NTK; ++r) {
                        pr[r] = __expf(t[r] - mx);
                        l += pr[r];
                    }
                    const float inv = 1.0f / l;
#pragma unroll
                    for (int r = 0; r < NTK; ++r) pr[r] *= inv;
                };
                softmax(t0, p0);
                softmax(t1, p1);
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const bool h1 = two_heads && 4 * tile_of(n) + kq >= 17;
                float so4[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (NP == 2) oscale4(0, n, so4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float vm = NTK > 4 ? mirror(vv[n][c]) : 0.f;
                    float o = (h1 ? p1[0] : p0[0]) * vv[n][c];
#pragma unroll
                    for (int r = 1; r < NTK; ++r) o = fmaf(h1 ? p1[r] : p0[r], xpart(vv[n][c], vm, r), o);
                    ov[n][c] = o * so4[c];
                }
            }
        };
        if (a.att_ntok == 4) {
            auto quad = [](float x, int j) -> float {
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
            const bool two_heads = a.att_hd == 68;
            float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const bool h1 = two_heads && 4 * tile_of(n) + kq >= 17;
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
            float* xs = reinterpret_cast<float*>(smem) + rt * (2 * BM * 8);     // [row tile][2 halves][64 rows][8]
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
    #pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const bool h1 = two_heads && 4 * tile_of(n) + kq >= 17;
                const float pj[4] = {h1 ? p1[0] : p0[0], h1 ? p1[1] : p0[1], h1 ? p1[2] : p0[2], h1 ? p1[3] : p0[3]};
                float so4[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (NP == 2) oscale4(0, n, so4);
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float o = pj[0] * quad(vv[n][c], 0);
                    o = fmaf(pj[1], quad(vv[n][c], 1), o);
                    o = fmaf(pj[2], quad(vv[n][c], 2), o);
                    o = fmaf(pj[3], quad(vv[n][c], 3), o);
                    ov[n][c] = o * so4[c];
                }
            }
        } else if (a.att_ntok == 8) {
            att_xor(std::integral_constant<int, 8>{});
        } else {
            att_xor(std::integral_constant<int, 2>{});
        }
        const int Go = Dq / BN, g_out = n0 / BN;
        char* cbase = a.C2 + (((size_t)tm * RT + rt) * 4 + rg) * h2_ksteps(Dq, NP) * H2_RG;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float x[8] = {ov[2 * q][0], ov[2 * q][1], ov[2 * q][2], ov[2 * q][3],
                                ov[2 * q + 1][0], ov[2 * q + 1][1], ov[2 * q + 1][2], ov[2 * q + 1][3]};
            h2_emit_frag<NP>(WT, cbase, 4 * g_out + (slot0 ? 2 : 0) + q, (unsigned)(lane * 16), x);
        }
        if (NTW == H2_T0 && kq < 2) {
            const float x[8] = {ov[NTW - 1][0], ov[NTW - 1][1], ov[NTW - 1][2], ov[NTW - 1][3], 0.f, 0.f, 0.f, 0.f};
            h2_emit_tail<NP>(WT, cbase, 4 * Go + (g_out >> 2), (unsigned)(((g_out & 3) * 16 + li) * 16 + kq * 8), x,
                             ((Dq / BK) & 1) && (g_out >> 2) == (Go >> 2) - 1);
        }
        }
      } else {
        float* Tt = reinterpret_cast<float*>(smem);
        float* SC = Tt + BM * H2_ATT_TS;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {       // the row tiles of the stage one after the other through the q | k | v tile in LDS
            __syncthreads();
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const int cl = 16 * tile_of(n) + 4 * kq;
                    if (cl + 3 < BN) {
                        float v[4];
                        value4(p, rt, n, v);
                        st4(Tt + row_l * H2_ATT_TS + p * BN + cl, float4{v[0], v[1], v[2], v[3]});
                    }
                }
            __syncthreads();
            h2_attention<NP>(WT, Tt, SC, tid, a.att_ntok, a.att_hd, a.rpt / a.att_ntok, a.C2, tm * RT + rt, n0 / BN, Dq,
                             reinterpret_cast<const float*>(smem + H2_VEC) + VSO, rg_lo, rgs);
        }
      }
    } else {
        const int Go = N / BN;
        float vals[RT][NTW][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            char* cbase = a.C2 + (((size_t)tm * RT + rt) * 4 + rg) * h2_ksteps(N, NP) * H2_RG;
#pragma unroll
            for (int p = 0; p < (NPASS == 2 ? 2 : 1); ++p) {
                const int g_out = colbase(p) / BN;
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    value4(p, rt, n, vals[rt][n]);
                    const int cl = 16 * tile_of(n) + 4 * kq;
                    const bool ok = cl + 3 < BN;
                    if (EPI == H2_EPI_RES) {
                        vals[rt][n][0] += rv[rt][n].x; vals[rt][n][1] += rv[rt][n].y; vals[rt][n][2] += rv[rt][n].z; vals[rt][n][3] += rv[rt][n].w;
                        if (!ok) vals[rt][n][0] = vals[rt][n][1] = vals[rt][n][2] = vals[rt][n][3] = 0.f;
                    }
                    if (a.C && ok && row_ok[rt]) {
                        // fp32 rows: in chain mode the WHOLE team reads them (LayerNorm GEMM of the next phase): write-through
                        const float4 o4 = {vals[rt][n][0], vals[rt][n][1], vals[rt][n][2], vals[rt][n][3]};
                        h2_st16(WT, a.C + (size_t)m0 * a.ldc, (unsigned)(((size_t)(row[rt] - m0) * a.ldc + colbase(p) + cl) * 4),
                                __builtin_bit_cast(u32x4, o4));
                    }
                }
                if constexpr (NSO > 0 || NP == 1) {     // the packed copy: the next GEMM's operand (NP = 1: also of the fp32 rows x)
                  if (a.C2) {
                    if constexpr (NSO > 0) {
#pragma unroll
                        for (int n = 0; n < NTW; ++n) {
                            float so4[4];
                            oscale4(p, n, so4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) vals[rt][n][r] *= so4[r];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float x[8] = {vals[rt][2 * q][0], vals[rt][2 * q][1], vals[rt][2 * q][2], vals[rt][2 * q][3],
                                            vals[rt][2 * q + 1][0], vals[rt][2 * q + 1][1], vals[rt][2 * q + 1][2], vals[rt][2 * q + 1][3]};
                        h2_emit_frag<NP>(WT, cbase, 4 * g_out + (slot0 ? 2 : 0) + q, (unsigned)(lane * 16), x);
                    }
                    if (NTW == H2_T0 && kq < 2) {
                        const float x[8] = {vals[rt][NTW - 1][0], vals[rt][NTW - 1][1], vals[rt][NTW - 1][2], vals[rt][NTW - 1][3], 0.f, 0.f, 0.f, 0.f};
                        h2_emit_tail<NP>(WT, cbase, 4 * Go + (g_out >> 2), (unsigned)(((g_out & 3) * 16 + li) * 16 + kq * 8), x,
                                         ((N / BK) & 1) && (g_out >> 2) == (Go >> 2) - 1);
                    }
                  }
                }
            }
        }
        if (H2_DBG && a.dbg) t_st = __builtin_amdgcn_s_memtime();
        if constexpr (EPI == H2_EPI_RES) {
            if (a.stats_out) {
                // LayerNorm partials {mean, M2} of the 136-column slice of each row: two exchanges through LDS
                constexpr int XR = RT * 64;
                float* xch = reinterpret_cast<float*>(smem);       // [2 phases][2 halves][RT x 64 rows]
                const int half = slot0 ? 1 : 0;
                float sum[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    sum[rt] = 0.f;
#pragma unroll
                    for (int n = 0; n < NTW; ++n) sum[rt] += (vals[rt][n][0] + vals[rt][n][1]) + (vals[rt][n][2] + vals[rt][n][3]);
                    sum[rt] = xor16_add(sum[rt]);
                    sum[rt] = xor32_add(sum[rt]);
                }
                __syncthreads();
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    if (kq == 0) xch[half * XR + rt * 64 + row_l] = sum[rt];
                __syncthreads();
                float mean[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    mean[rt] = (xch[rt * 64 + row_l] + xch[XR + rt * 64 + row_l]) * (1.0f / (float)BN);
                    float q = 0.f;
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        const bool ok = 16 * tile_of(n) + 4 * kq + 3 < BN;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = vals[rt][n][r] - mean[rt];
                            q = ok ? fmaf(d, d, q) : q;
                        }
                    }
                    q = xor16_add(q);
                    q = xor32_add(q);
                    if (kq == 0) xch[2 * XR + half * XR + rt * 64 + row_l] = q;
                }
                __syncthreads();
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    if (half == 0 && kq == 0 && row_ok[rt]) {
                        const u32x2 h = {__builtin_bit_cast(unsigned, mean[rt]),
                                         __builtin_bit_cast(unsigned, xch[2 * XR + rt * 64 + row_l] + xch[3 * XR + rt * 64 + row_l])};
                        h2_st8(WT, a.stats_out + (size_t)m0 * Go * 2, (unsigned)(((row[rt] - m0) * Go + n0 / BN) * 8), h);
                    }
            }
        }
    }
    }
    if (CHAIN) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && arrive) __hip_atomic_fetch_add(chain, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (H2_DBG && a.dbg) {
        if (!t_st) t_st = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* o = a.dbg + (size_t)(blockIdx.x * 8 + wave) * 8;
            o[0] = t_entry; o[1] = t_loop; o[2] = t_epi; o[3] = t_st; o[4] = t_end; o[5] = t_vm; o[6] = t_bar; o[7] = t_mm;
            if (H2_DBG == 2) { o[5] = t_chain; o[6] = t_land; }      // prologue split: hand-off wait | first operands landed
        }
    }
    return true;
}

template <int EPI, bool LNF, int NPASS, int NP = 2>
__global__ __launch_bounds__(512, 2) void h2_gemm_kernel(const H2Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    {
        const int b = blockIdx.x;
        if ((a.grid_m & 7) == 0) {
            const int per = a.grid_m >> 3;
            const int xcd = b & 7, i = b >> 3;
            tm = xcd * per + (i % per);
            tn = i / per;
        } else {
            tm = b % a.grid_m;
            tn = b / a.grid_m;
        }
    }
    if (wave < 4) h2_phase<EPI, LNF, NPASS, H2_T0, false, H2_WC0, 1, NP>(a, smem, tid, wave, 0, tm, tn, nullptr, 0u);
    else if (wave < 6) h2_phase<EPI, LNF, NPASS, NT - H2_T0, false, H2_WC1, 1, NP>(a, smem, tid, wave, H2_T0, tm, tn, nullptr, 0u);
    else h2_phase<EPI, LNF, NPASS, NT - H2_T0, false, H2_WC2, 1, NP>(a, smem, tid, wave, H2_T0, tm, tn, nullptr, 0u);
}

// ---------------------------------------------------------------------------------------------- whole block stack
// ONE launch for all Block applications of a stack (the team protocol of x3_stack_kernel): G = D / 136 workgroups form a
// team that walks one row tile through every GEMM of every application, synchronising only among themselves through a
// monotonic arrival counter per row tile.
struct H2StackArgs {
    char *att2, *hid2;
    char* x16;                   // NP = 1: the packed bf16 copy of x (the A operand of qkv / fc1), rewritten by every residual epilogue
    float *x, *stats;
    unsigned* counters;          // one per row tile (+ the error word), zeroed before the launch
    int M, D, n_tok, heads, rpt, n_tiles, n_teams, G, n_apps, n_phases;
    unsigned* xcc;               // one word per team: bit x set = a workgroup of the team runs on XCD x (h2_team_placement)
    int plain_ok;                // 0: write-through hand-off stores whatever the placement (A/B switch)
    int rgs;                     // row-narrow stack: row groups (16 rows) per workgroup, 1 or 2 (else 4 = whole tiles)
    float eps;
    unsigned long long* dbg;
    unsigned *err_ws, *err_host;
    int spin_log2;
    int inject;
    const char* w[MPL_MAX_APPS][4];   // per application: qkv (norm1 folded), proj, fc1 (norm2 folded), fc2 operands
};

// the trailer vectors behind the fragments of a packed weight operand (both engines: H2_TRV N + 8 floats)
template <int NP>
__device__ __forceinline__ const float* h2_trailer(const char* w2, int N, int K) {
    return reinterpret_cast<const float*>(w2 + (size_t)(N / BN) * h2_ksteps(K, NP) * H2_W);
}
// H2Args of the four GEMMs of a block application inside a stack.  NP = 2: x (fp32) is the A operand of qkv / fc1, the
// attention / GELU outputs leave under the static scales so.  NP = 1: x16 is the A operand, proj / fc2 rewrite it beside x.
template <int NP>
__device__ __forceinline__ H2Args h2_args_qkv(const H2StackArgs& s, const char* w, int D, int G, int plain) {
    if (H2_ABL & 64) w = s.w[0][0];
    const float* v = h2_trailer<NP>(w, 3 * D, D);
    return H2Args{NP == 1 ? s.x16 : nullptr, NP == 1 ? nullptr : s.x, D, w, v, v + 3 * D, s.stats, nullptr, NP == 1 ? nullptr : v + 12 * D,
                  nullptr, 0, nullptr, 0, s.att2, nullptr, s.M, 3 * D, D, s.rpt, s.n_tiles, G, s.eps, s.n_tok, D / s.heads, s.dbg,
                  s.err_ws, s.err_host, s.spin_log2, plain};
}
template <int NP>
__device__ __forceinline__ H2Args h2_args_fc1(const H2StackArgs& s, const char* w, int D, int G, int plain) {
    if (H2_ABL & 64) w = s.w[0][0];
    const float* v = h2_trailer<NP>(w, 2 * D, D);
    return H2Args{NP == 1 ? s.x16 : nullptr, NP == 1 ? nullptr : s.x, D, w, v, v + 2 * D, s.stats, nullptr, NP == 1 ? nullptr : v + 8 * D,
                  nullptr, 0, nullptr, 0, s.hid2, nullptr, s.M, 2 * D, D, s.rpt, s.n_tiles, G, s.eps, 0, 0, s.dbg, s.err_ws, s.err_host,
                  s.spin_log2, plain};
}
// proj (A = attention output, K = D) and fc2 (A = hidden, K = 2D): one body for both.  NP = 2: the operand arrives under the
// producer's per-column static scales, which these weights were packed against (mpl_pack_h2_scaled; h2_entry_kernel checks
// the fingerprints): nothing to take out here
template <int NP>
__device__ __forceinline__ H2Args h2_args_res(const H2StackArgs& s, const char* w, bool fc2, int D, int G, int plain) {
    if (H2_ABL & 64) w = s.w[0][0];
    const int K = fc2 ? 2 * D : D;
    const float* v = h2_trailer<NP>(w, D, K);
    return H2Args{fc2 ? s.hid2 : s.att2, nullptr, 0, w, v, v + D, nullptr, nullptr, nullptr, s.x, D, s.x, D, NP == 1 ? s.x16 : nullptr,
                  s.stats, s.M, D, K, s.rpt, s.n_tiles, G, s.eps, 0, 0, s.dbg, s.err_ws, s.err_host, s.spin_log2, plain};
}

// Where a team runs.  The hand-off stores of a team are write-through by default (sc0 sc1: placement-independent, but the line
// leaves the L2 and every consumer fetches it across the fabric: tools/h2_probe.hip, 42 against 24 cycles per KiB).  The XCDs'
// L2s are coherent for the compute units of ONE XCD, so a team whose workgroups all sit on one XCD can keep its operands there:
// plain stores, consumers read past their L1 (sc1) as before.  Nothing is assumed about placement: every workgroup publishes the
// XCD it actually runs on (HW_REG_XCC_ID) in a word of its team at the start of the kernel; once a workgroup has seen all its
// partners arrive at the end of the first phase -- each arrival is ordered behind that workgroup's publication -- the word is
// complete, and from the third phase on the team stores plain iff exactly one bit is set.  (The blocks of a team are b, b + 8,
// ...: on this part they land on XCD b mod 8 and the answer is yes; elsewhere the kernel simply keeps writing through.)
__device__ __forceinline__ void h2_publish_xcd(const H2StackArgs& s, int team, int tid) {
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(s.xcc + team, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ int h2_team_on_one_xcd(const H2StackArgs& s, int team) {
    const unsigned m = __hip_atomic_load(s.xcc + team, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (s.plain_ok && m != 0u && (m & (m - 1u)) == 0u) ? 1 : 0;
}

template <int NP>
__global__ __launch_bounds__(512, 2) void h2_stack_kernel(const H2StackArgs s) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = s.G, D = s.D;
    int team, tn;
    {
        const int b = blockIdx.x;
        team = (b & 7) + 8 * ((b >> 3) / G);
        tn = (b >> 3) % G;
        if (team >= s.n_teams) return;
    }
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(smem + H2_FAIL) = 0u;
    h2_publish_xcd(s, team, tid);
    int plain = 0, seen = 0;       // plain hand-off stores once the team is known to sit on one XCD (h2_publish_xcd)
    __syncthreads();
    for (int tile0 = team; tile0 < s.n_tiles; tile0 += s.n_teams) {
        unsigned need = 0;
        for (int ph = 0; ph < s.n_phases; ++ph, need += G) {
            if (!seen && ph >= 2) {       // the proj phase has seen every partner arrive: the team's placement word is complete
                plain = __builtin_amdgcn_readfirstlane(h2_team_on_one_xcd(s, team));
                seen = 1;
            }
            // the thread id is rebuilt from the wave index (a scalar) and the lane number every phase: kept in a register
            // across the phases it was the one value the 256-register budget spilled to scratch
            int wvp = wave_s, tile = tile0, tnp = tn;
            asm volatile("" : "+s"(wvp), "+s"(tile), "+s"(tnp));
            int tidp = wvp * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(tidp));
            const int wv = wvp;
            unsigned* ctr = s.counters + tile;
            const char* const* w = s.w[ph >> 2];
            bool ok = true;
            if (s.inject > 0 && ph == s.inject && tile == 0 && tnp == 0) return;     // fault injection (test hook)
            switch (ph & 3) {
                case 0: {   // x = x + proj(attn(qkv(norm1(x))))   (Block.forward :84-90)
                    H2Args a = h2_args_qkv<NP>(s, w[0], D, G, plain);
                    if (wv < 4) ok = h2_phase<H2_EPI_ATT, true, 3, H2_T0, true, H2_WC0, 1, NP>(a, smem, tidp, wv, 0, tile, tnp, ctr, need);
                    else if (wv < 6) ok = h2_phase<H2_EPI_ATT, true, 3, NT - H2_T0, true, H2_WC1, 1, NP>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need);
                    else ok = h2_phase<H2_EPI_ATT, true, 3, NT - H2_T0, true, H2_WC2, 1, NP>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need);
                    break;
                }
                case 2: {   // x = x + fc2(gelu(fc1(norm2(x))))    (Block.forward :91, Mlp.forward :31-37)
                    H2Args a = h2_args_fc1<NP>(s, w[2], D, G, plain);
                    if (wv < 4) ok = h2_phase<H2_EPI_GELU, true, 2, H2_T0, true, H2_WC0, 1, NP>(a, smem, tidp, wv, 0, tile, tnp, ctr, need);
                    else if (wv < 6) ok = h2_phase<H2_EPI_GELU, true, 2, NT - H2_T0, true, H2_WC1, 1, NP>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need);
                    else ok = h2_phase<H2_EPI_GELU, true, 2, NT - H2_T0, true, H2_WC2, 1, NP>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need);
                    break;
                }
                default: {
                    const bool fc2 = (ph & 3) == 3;
                    H2Args a = h2_args_res<NP>(s, fc2 ? w[3] : w[1], fc2, D, G, plain);
                    if (wv < 4) ok = h2_phase<H2_EPI_RES, false, 1, H2_T0, true, H2_WC0, 1, NP>(a, smem, tidp, wv, 0, tile, tnp, ctr, need);
                    else if (wv < 6) ok = h2_phase<H2_EPI_RES, false, 1, NT - H2_T0, true, H2_WC1, 1, NP>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need);
                    else ok = h2_phase<H2_EPI_RES, false, 1, NT - H2_T0, true, H2_WC2, 1, NP>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need);
                    break;
                }
            }
            if (!ok) return;
        }
    }
}

// The stack for teams that own two or more row tiles, bf16 operands (NP = 1): a team walks PAIRS of row tiles through EVERY
// phase -- with one part per operand the fragments of a two-tile stage and the three accumulator sets of the qkv phase fit the
// register file together, so the qkv + attention phase and the two-pass fc1 run the two-tile stage too (every W k-tile pair is
// fetched and read from LDS once for both tiles).  Same poses as h2_stack_kernel<1> bit for bit.
template <int NP>
__global__ __launch_bounds__(512, 2) void h2_stackp_kernel(const H2StackArgs s) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = s.G, D = s.D;
    int team, tn;
    {
        const int b = blockIdx.x;
        team = (b & 7) + 8 * ((b >> 3) / G);
        tn = (b >> 3) % G;
        if (team >= s.n_teams) return;
    }
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(smem + H2_FAIL) = 0u;
    h2_publish_xcd(s, team, tid);
    int plain = 0, seen = 0;       // plain hand-off stores once the team is known to sit on one XCD (h2_publish_xcd)
    __syncthreads();
    const int n_pairs = (s.n_tiles + 1) >> 1;
    for (int pair0 = team; pair0 < n_pairs; pair0 += s.n_teams) {
        unsigned need = 0;
        for (int ph = 0; ph < s.n_phases; ++ph, need += G) {
            if (!seen && ph >= 2) {       // the proj phase has seen every partner arrive: the team's placement word is complete
                plain = __builtin_amdgcn_readfirstlane(h2_team_on_one_xcd(s, team));
                seen = 1;
            }
            int wvp = wave_s, pair = pair0, tnp = tn;
            asm volatile("" : "+s"(wvp), "+s"(pair), "+s"(tnp));
            int tidp = wvp * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(tidp));
            const int wv = wvp;
            unsigned* ctr = s.counters + 2 * pair;       // the arrival counter of the pair = the one of its first row tile
            const char* const* w = s.w[ph >> 2];
            bool ok = true;
            if (s.inject > 0 && ph == s.inject && pair == 0 && tnp == 0) return;     // fault injection (test hook)
            switch (ph & 3) {
                case 0: {
                    H2Args a = h2_args_qkv<NP>(s, w[0], D, G, plain);
                    if (wv < 4) ok = h2_phase<H2_EPI_ATT, true, 3, H2_T0, true, H2_RP_WC0, 2, NP>(a, smem, tidp, wv, 0, pair, tnp, ctr, need);
                    else if (wv < 6) ok = h2_phase<H2_EPI_ATT, true, 3, NT - H2_T0, true, H2_RP_WC1, 2, NP>(a, smem, tidp, wv, H2_T0, pair, tnp, ctr, need);
                    else ok = h2_phase<H2_EPI_ATT, true, 3, NT - H2_T0, true, H2_RP_WC2, 2, NP>(a, smem, tidp, wv, H2_T0, pair, tnp, ctr, need);
                    break;
                }
                case 2: {
                    H2Args a = h2_args_fc1<NP>(s, w[2], D, G, plain);
                    if (wv < 4) ok = h2_phase<H2_EPI_GELU, true, 2, H2_T0, true, H2_RP_WC0, 2, NP>(a, smem, tidp, wv, 0, pair, tnp, ctr, need);
                    else if (wv < 6) ok = h2_phase<H2_EPI_GELU, true, 2, NT - H2_T0, true, H2_RP_WC1, 2, NP>(a, smem, tidp, wv, H2_T0, pair, tnp, ctr, need);
                    else ok = h2_phase<H2_EPI_GELU, true, 2, NT - H2_T0, true, H2_RP_WC2, 2, NP>(a, smem, tidp, wv, H2_T0, pair, tnp, ctr, need);
                    break;
                }
                default: {
                    const bool fc2 = (ph & 3) == 3;
                    H2Args a = h2_args_res<NP>(s, fc2 ? w[3] : w[1], fc2, D, G, plain);
                    if (wv < 4) ok = h2_phase<H2_EPI_RES, false, 1, H2_T0, true, H2_RP_WC0, 2, NP>(a, smem, tidp, wv, 0, pair, tnp, ctr, need);
                    else if (wv < 6) ok = h2_phase<H2_EPI_RES, false, 1, NT - H2_T0, true, H2_RP_WC1, 2, NP>(a, smem, tidp, wv, H2_T0, pair, tnp, ctr, need);
                    else ok = h2_phase<H2_EPI_RES, false, 1, NT - H2_T0, true, H2_RP_WC2, 2, NP>(a, smem, tidp, wv, H2_T0, pair, tnp, ctr, need);
                    break;
                }
            }
            if (!ok) return;
        }
    }
}

// The same stack for teams that own TWO OR MORE row tiles (M > 64 x the number of teams the chip holds: the FULL flag set at
// B = 1024, eight views, B >= 2048): a team walks PAIRS of row tiles.  proj, fc1 and fc2 run the two-tile stage (h2_phase RT = 2:
// every W k-tile is fetched and read once for both tiles); the qkv + attention phase, whose three accumulator sets leave no
// registers for a second row tile, runs the one-tile form for the two tiles one after the other.  Bitwise the same poses as
// h2_stack_kernel (tested: the batch-split and shard equalities of tests/test_gpu_parity.py cross the switch).
template <int NP>
__global__ __launch_bounds__(512, 2) void h2_stack2_kernel(const H2StackArgs s) {
    static_assert(NP == 2, "the six-step pair form of the fp16x2 engine (NP = 1: h2_stackp_kernel)");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = s.G, D = s.D;
    int team, tn;
    {
        const int b = blockIdx.x;
        team = (b & 7) + 8 * ((b >> 3) / G);
        tn = (b >> 3) % G;
        if (team >= s.n_teams) return;
    }
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(smem + H2_FAIL) = 0u;
    h2_publish_xcd(s, team, tid);
    int plain = 0, seen = 0;       // plain hand-off stores once the team is known to sit on one XCD (h2_publish_xcd)
    __syncthreads();
    auto vecs = [&](const char* w2, int N, int K) -> const float* {
        return reinterpret_cast<const float*>(w2 + (size_t)(N / BN) * (K / BK) * H2_W);
    };
    const int n_pairs = (s.n_tiles + 1) >> 1;
    for (int pair0 = team; pair0 < n_pairs; pair0 += s.n_teams) {
        unsigned need = 0;
        // six steps per block application: qkv of the first / the second row tile of the pair, proj, fc1 of the workgroup's
        // first / second column group, fc2.  A step that has a partner step waits for the team only in the first and arrives
        // only in the second of the two.
        for (int e = 0; e < 6 * s.n_apps; ++e) {
            if (!seen && e >= 3) {        // the proj step has seen every partner arrive: the team's placement word is complete
                plain = __builtin_amdgcn_readfirstlane(h2_team_on_one_xcd(s, team));
                seen = 1;
            }
            int wvp = wave_s, pair = pair0, tnp = tn;
            asm volatile("" : "+s"(wvp), "+s"(pair), "+s"(tnp));
            int tidp = wvp * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(tidp));
            const int wv = wvp;
            const int app = e / 6, k = e - 6 * app;
            const int ph = 4 * app + (k < 2 ? 0 : (k == 2 ? 1 : (k < 5 ? 2 : 3)));      // the GEMM this step belongs to
            if (ph >= s.n_phases) break;
            const bool two = 2 * pair + 1 < s.n_tiles;
            if (k == 1 && !two) continue;
            const bool second = k == 1 || k == 4;
            const bool arr = !((k == 0 && two) || k == 3);
            const unsigned nd = second ? 0u : need;
            unsigned* ctr = s.counters + 2 * pair;       // the arrival counter of the pair = the one of its first row tile
            const char* const* w = s.w[app];
            bool ok = true;
            if (s.inject > 0 && ph == s.inject && !second && pair == 0 && tnp == 0) return;     // fault injection (test hook)
            if (k < 2) {   // x = x + proj(attn(qkv(norm1(x)))): one row tile at a time
                const float* v = vecs(w[0], 3 * D, D);
                const H2Args a{nullptr, s.x, D, w[0], v, v + 3 * D, s.stats, nullptr, v + 12 * D, nullptr, 0, nullptr, 0, s.att2,
                               nullptr, s.M, 3 * D, D, s.rpt, s.n_tiles, G, s.eps, s.n_tok, D / s.heads, s.dbg, s.err_ws, s.err_host,
                               s.spin_log2, plain};
                const int tile = 2 * pair + k;
                if (wv < 4) ok = h2_phase<H2_EPI_ATT, true, 3, H2_T0, true, H2_WC0>(a, smem, tidp, wv, 0, tile, tnp, ctr, nd, arr);
                else if (wv < 6) ok = h2_phase<H2_EPI_ATT, true, 3, NT - H2_T0, true, H2_WC1>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, nd, arr);
                else ok = h2_phase<H2_EPI_ATT, true, 3, NT - H2_T0, true, H2_WC2>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, nd, arr);
            } else if (k == 3 || k == 4) {   // fc1 + GELU: the workgroup's two column groups one after the other (two accumulator sets
                                             // for two row tiles do not fit the register file beside the double-buffered fragments)
                const float* v = vecs(w[2], 2 * D, D);
                const H2Args a{nullptr, s.x, D, w[2], v, v + 2 * D, s.stats, nullptr, v + 8 * D, nullptr, 0, nullptr, 0, s.hid2,
                               nullptr, s.M, 2 * D, D, s.rpt, s.n_tiles, G, s.eps, 0, 0, s.dbg, s.err_ws, s.err_host, s.spin_log2, plain};
                const int cg = 2 * tnp + (k - 3);
                if (wv < 4) ok = h2_phase<H2_EPI_GELU, true, 1, H2_T0, true, H2_R2_WC0, 2>(a, smem, tidp, wv, 0, pair, cg, ctr, nd, arr);
                else if (wv < 6) ok = h2_phase<H2_EPI_GELU, true, 1, NT - H2_T0, true, H2_R2_WC1, 2>(a, smem, tidp, wv, H2_T0, pair, cg, ctr, nd, arr);
                else ok = h2_phase<H2_EPI_GELU, true, 1, NT - H2_T0, true, H2_R2_WC2, 2>(a, smem, tidp, wv, H2_T0, pair, cg, ctr, nd, arr);
            } else {  // proj (A = attention output, K = D) and fc2 (A = hidden, K = 2D)
                const bool fc2 = k == 5;
                const int K = fc2 ? 2 * D : D;
                const char* w2 = fc2 ? w[3] : w[1];
                const float* v = vecs(w2, D, K);
                const H2Args a{fc2 ? s.hid2 : s.att2, nullptr, 0, w2, v, v + D, nullptr, nullptr, nullptr, s.x, D, s.x, D, nullptr, s.stats,
                               s.M, D, K, s.rpt, s.n_tiles, G, s.eps, 0, 0, s.dbg, s.err_ws, s.err_host, s.spin_log2, plain};
                if (wv < 4) ok = h2_phase<H2_EPI_RES, false, 1, H2_T0, true, H2_R2_WC0, 2>(a, smem, tidp, wv, 0, pair, tnp, ctr, nd, arr);
                else if (wv < 6) ok = h2_phase<H2_EPI_RES, false, 1, NT - H2_T0, true, H2_R2_WC1, 2>(a, smem, tidp, wv, H2_T0, pair, tnp, ctr, nd, arr);
                else ok = h2_phase<H2_EPI_RES, false, 1, NT - H2_T0, true, H2_R2_WC2, 2>(a, smem, tidp, wv, H2_T0, pair, tnp, ctr, nd, arr);
            }
            if (!ok) return;
            if (arr) need += G;
        }
    }
}

// The stack for launches that would leave most of the chip idle (fewer 64-row tiles x column groups than compute units: the
// reference's shipped call shape TEST.BATCH_SIZE 256 x 2 views = 8 tiles = 32 workgroups of h2_stack_kernel): ROW-NARROW teams.
// A 64-row tile is split into 4 / rgs sub-tiles of rgs row groups (16 or 32 rows); a team of G workgroups walks ONE sub-tile
// through every phase, so 4 / rgs times as many compute units stream the same weights, each for a quarter / half of the rows.
// Inside a workgroup the waves keep their full-tile roles: wave w works on row group w & 3 if the workgroup owns it (h2_phase
// ACT = true: the very code of h2_stack_kernel, so the poses are bitwise the same), otherwise it only requests its W pieces and
// meets the barriers (ACT = false).  Column groups are untouched: heads stay whole, the in-register attention is unchanged.
// Measured and dropped (round 5, tools/ab.sh): in the 16-row form seven waves wait ~450 cycles per stage at the barrier for the one
// wave pair that multiplies (tools/chain_phase.py), so its W pieces were given to the six loader-only waves (three each, by
// rank): no gain for the stage, and the extra instantiations cost every form of this kernel 8 % (SGPR spills 492 -> 768).
template <int NP>
__global__ __launch_bounds__(512, 2) void h2_stackn_kernel(const H2StackArgs s) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = s.G, D = s.D;
    int team, tn;
    {
        const int b = blockIdx.x;
        team = (b & 7) + 8 * ((b >> 3) / G);
        tn = (b >> 3) % G;
        if (team >= s.n_teams) return;
    }
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(smem + H2_FAIL) = 0u;
    h2_publish_xcd(s, team, tid);
    int plain = 0, seen = 0;       // plain hand-off stores once the team is known to sit on one XCD (h2_publish_xcd)
    __syncthreads();
    const int rs = 4 / s.rgs;                    // sub-tiles per row tile
    const int n_units = s.n_tiles * rs;
    for (int unit0 = team; unit0 < n_units; unit0 += s.n_teams) {
        unsigned need = 0;
        for (int ph = 0; ph < s.n_phases; ++ph, need += G) {
            if (!seen && ph >= 2) {       // the proj phase has seen every partner arrive: the team's placement word is complete
                plain = __builtin_amdgcn_readfirstlane(h2_team_on_one_xcd(s, team));
                seen = 1;
            }
            int wvp = wave_s, unit = unit0, tnp = tn;
            asm volatile("" : "+s"(wvp), "+s"(unit), "+s"(tnp));
            int tidp = wvp * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(tidp));
            const int wv = wvp;
            const int tile = unit / rs, rg_lo = (unit - tile * rs) * s.rgs;
            const bool act = (wv & 3) >= rg_lo && (wv & 3) < rg_lo + s.rgs;
            unsigned* ctr = s.counters + H2_CTR_PER_TILE * tile + rg_lo;      // one arrival counter per sub-tile
            const char* const* w = s.w[ph >> 2];
            bool ok = true;
            if (s.inject > 0 && ph == s.inject && unit == 0 && tnp == 0) return;     // fault injection (test hook)
#define H2N_PHASE(EPI, LNF, NPASS)                                                                                                          \
    do {                                                                                                                                    \
        if (wv < 4) ok = act ? h2_phase<EPI, LNF, NPASS, H2_T0, true, H2_WC0, 1, NP, true>(a, smem, tidp, wv, 0, tile, tnp, ctr, need, true, rg_lo, s.rgs)   \
                             : h2_phase<EPI, LNF, NPASS, H2_T0, true, H2_WC0, 1, NP, false>(a, smem, tidp, wv, 0, tile, tnp, ctr, need, true, rg_lo, s.rgs); \
        else if (wv < 6) ok = act ? h2_phase<EPI, LNF, NPASS, NT - H2_T0, true, H2_WC1, 1, NP, true>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need, true, rg_lo, s.rgs)   \
                                  : h2_phase<EPI, LNF, NPASS, NT - H2_T0, true, H2_WC1, 1, NP, false>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need, true, rg_lo, s.rgs); \
        else ok = act ? h2_phase<EPI, LNF, NPASS, NT - H2_T0, true, H2_WC2, 1, NP, true>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need, true, rg_lo, s.rgs)   \
                      : h2_phase<EPI, LNF, NPASS, NT - H2_T0, true, H2_WC2, 1, NP, false>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need, true, rg_lo, s.rgs); \
    } while (0)
            switch (ph & 3) {
                case 0: {
                    H2Args a = h2_args_qkv<NP>(s, w[0], D, G, plain);
                    H2N_PHASE(H2_EPI_ATT, true, 3);
                    break;
                }
                case 2: {
                    H2Args a = h2_args_fc1<NP>(s, w[2], D, G, plain);
                    H2N_PHASE(H2_EPI_GELU, true, 2);
                    break;
                }
                default: {
                    const bool fc2 = (ph & 3) == 3;
                    H2Args a = h2_args_res<NP>(s, fc2 ? w[3] : w[1], fc2, D, G, plain);
                    H2N_PHASE(H2_EPI_RES, false, 1);
                    break;
                }
            }
#undef H2N_PHASE
            if (!ok) return;
        }
    }
}

// Direct-W form of the 16-row teams (h2d_gemm.hip).  With 16 rows a stage of the ring form is bound by the serial chain of the
// one wave pair that multiplies -- fragment reads, its DMA requests, a barrier per two stages, both waves on ONE SIMD -- while six
// waves only move W through LDS for them.  Here the two multiplying waves sit on different SIMDs (wave rg_lo and wave
// 4 + (rg_lo + 2) % 4), take their W fragments straight from L2 into a register ring and run the k loop without barriers; all
// eight waves bring the A operand of the 16 rows into LDS once per phase.  Fragments, product order and epilogue are those of the
// ring form: bitwise the same poses.
template <int NP>
__global__ __launch_bounds__(512, 2) void h2_stackd_kernel(const H2StackArgs s) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = s.G, D = s.D;
    int team, tn;
    {
        const int b = blockIdx.x;
        team = (b & 7) + 8 * ((b >> 3) / G);
        tn = (b >> 3) % G;
        if (team >= s.n_teams) return;
    }
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(smem + H2_FAIL) = 0u;
    h2_publish_xcd(s, team, tid);
    int plain = 0, seen = 0;
    __syncthreads();
    const int n_units = s.n_tiles * 4;
    for (int unit0 = team; unit0 < n_units; unit0 += s.n_teams) {
        unsigned need = 0;
        for (int ph = 0; ph < s.n_phases; ++ph, need += G) {
            if (!seen && ph >= 2) {
                plain = __builtin_amdgcn_readfirstlane(h2_team_on_one_xcd(s, team));
                seen = 1;
            }
            int wvp = wave_s, unit = unit0, tnp = tn;
            asm volatile("" : "+s"(wvp), "+s"(unit), "+s"(tnp));
            int tidp = wvp * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(tidp));
            const int wv = wvp;
            const int tile = unit >> 2, rg_lo = unit & 3;
            const bool act = wv < 4 ? wv == rg_lo : (wv & 3) == ((rg_lo + 2) & 3);
            unsigned* ctr = s.counters + H2_CTR_PER_TILE * tile + rg_lo;
            const char* const* w = s.w[ph >> 2];
            bool ok = true;
            if (s.inject > 0 && ph == s.inject && unit == 0 && tnp == 0) return;     // fault injection (test hook)
#define H2D_PHASE(EPI, LNF, NPASS)                                                                                                          \
    do {                                                                                                                                    \
        if (wv < 4) ok = act ? h2_phase<EPI, LNF, NPASS, H2_T0, true, H2_WC0, 1, NP, true, true>(a, smem, tidp, wv, 0, tile, tnp, ctr, need, true, rg_lo, 1)   \
                             : h2_phase<EPI, LNF, NPASS, H2_T0, true, H2_WC0, 1, NP, false, true>(a, smem, tidp, wv, 0, tile, tnp, ctr, need, true, rg_lo, 1); \
        else ok = act ? h2_phase<EPI, LNF, NPASS, NT - H2_T0, true, H2_WC1, 1, NP, true, true>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need, true, rg_lo, 1)   \
                      : h2_phase<EPI, LNF, NPASS, NT - H2_T0, true, H2_WC1, 1, NP, false, true>(a, smem, tidp, wv, H2_T0, tile, tnp, ctr, need, true, rg_lo, 1); \
    } while (0)
            switch (ph & 3) {
                case 0: {
                    H2Args a = h2_args_qkv<NP>(s, w[0], D, G, plain);
                    H2D_PHASE(H2_EPI_ATT, true, 3);
                    break;
                }
                case 2: {
                    H2Args a = h2_args_fc1<NP>(s, w[2], D, G, plain);
                    H2D_PHASE(H2_EPI_GELU, true, 2);
                    break;
                }
                default: {
                    const bool fc2 = (ph & 3) == 3;
                    H2Args a = h2_args_res<NP>(s, fc2 ? w[3] : w[1], fc2, D, G, plain);
                    H2D_PHASE(H2_EPI_RES, false, 1);
                    break;
                }
            }
#undef H2D_PHASE
            if (!ok) return;
        }
    }
}

// the kernel of the pair form: six steps per block application for fp16x2 operands, every phase on pairs for bf16 operands
template <int NP>
static auto h2_pair_kernel() -> void (*)(const H2StackArgs) {
    if constexpr (NP == 2) return h2_stack2_kernel<2>;
    else return h2_stackp_kernel<1>;
}

// ---------------------------------------------------------------------------------------------- host side of the stack launch
unsigned long long* h2_debug_buffer();       // h2_gemm.hip (A/B switches and test hooks shared by both engines)
int h2_spin_log2();
int h2_row_tiles();
int h2_narrow_mode();
int h2_write_through_always();
int h2_direct_w();
struct H2StackArgs;
int launch_h2n_stack(const H2StackArgs& a, int grid, hipStream_t s);      // h2n_gemm.hip: the row-narrow stack kernel
int launch_h2d_stack(const H2StackArgs& a, int grid, hipStream_t s);      // h2d_gemm.hip: its direct-W form for 16-row teams

// THE rule by which a stack launch picks its kernel form, in one place: h2_launch_stack follows it and mpl_block_stack_form
// (api.hip) reports it (bench.py names the kernel after it, tests pin it).  By the shape of the launch, the device and the A/B
// switches only; every form yields bitwise the same poses.
struct H2Form {
    int pairs;      // teams walk pairs of row tiles (two-tile stage)
    int rgs;        // row groups per workgroup: 4 = whole tiles, 2 / 1 = 32- / 16-row teams
    int direct;     // the 16-row teams in their direct-W form (h2d_gemm.hip)
};
template <int NP>
static H2Form h2_stack_form(int M, int D, int n_tok, int cus) {
    H2Form f{0, 4, 0};
    const int rpt = h2_rows_per_tile(n_tok), n_tiles = (M + rpt - 1) / rpt, G = D / BN;
    const int cap = cus / G;
    // more row tiles than teams the chip holds: teams walk PAIRS of row tiles with the two-tile stage.  h2_row_tiles(): A/B switch
    // (mpl_x3_stack_mode bits 1, 2).  bf16 operands (NP = 1): one tile at a time by default -- the pair form of every phase
    // (h2_stackp_kernel) was built and measured in round 5 and is NOT faster (V = 8, B = 1024: 332 against 311 us for depth 2):
    // the 34-KiB stage leaves a ring of four, too shallow for the flight time of the A strips, and a wave's product rows and its
    // requests do not overlap
    const int force = h2_row_tiles();
    f.pairs = (force == 2 || (force == 0 && NP == 2 && n_tiles > cap)) ? 1 : 0;
    // row-narrow teams (fp16x2 operands): when whole-tile teams would leave compute units idle, a tile is split into sub-tiles of
    // 16 or 32 rows -- the narrowest form that still gives every workgroup a compute unit of its own.  Sequences must not straddle
    // row groups: 16 a multiple of n_tok
    if (NP == 2 && !f.pairs && rpt == BM && (16 % n_tok) == 0) {
        const int nm = h2_narrow_mode();
        if (nm == 3) f.rgs = 1;
        else if (nm == 2) f.rgs = 2;
        else if (nm == 0 && force == 0) {
            if (n_tiles * 4 * G <= cus) f.rgs = 1;
            else if (n_tiles * 2 * G <= cus) f.rgs = 2;
        }
    }
    // 16-row teams: the direct-W form while the A operand of the widest GEMM (fc2, K = 2 D: 2 KiB per k-tile) fits below the
    // statistics rows in LDS (mpl_x3_stack_mode bit 4: the ring form, for A/B)
    f.direct = (f.rgs == 1 && h2_direct_w() && h2_ksteps(2 * D, 2) * 2048 <= 72 * 1024) ? 1 : 0;
    return f;
}

// The whole block stack in one launch (both engines).  `ops` = n_apps x {qkv, proj, fc1, fc2} packed operands of engine NP;
// counters: n_tiles arrival counters + 1 error word, zeroed by the caller (the entry kernel of the engine); x16: NP = 1 only.
template <int NP>
static int h2_launch_stack(float* x, unsigned short* x16, int M, int D, int n_tok, int heads, const unsigned short* const* ops, int n_apps,
                           unsigned short* att2, unsigned short* hid2, float* stats, unsigned* counters, float eps, int stop_after,
                           hipStream_t s) {
    if ((NP == 1) != (x16 != nullptr)) return MPL_E_INVALID;
    if (!x || !ops || !att2 || !hid2 || !stats || !counters || M <= 0 || n_apps <= 0 || n_apps > MPL_MAX_APPS ||
        !h2_attention_fusable(n_tok, D, heads) || !h2_shape_ok(D, 2 * D) || M % n_tok)
        return MPL_E_INVALID;
    static std::atomic<int> resident[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!resident[dev].load(std::memory_order_acquire)) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return MPL_E_LAUNCH;
        if (hipFuncSetAttribute((const void*)h2_stack_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES) != hipSuccess)
            return MPL_E_LAUNCH;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)h2_stack_kernel<NP>, 512, H2_LDS_BYTES) != hipSuccess || per_cu < 1)
            return MPL_E_UNSUPPORTED;
        const void* pk = (const void*)h2_pair_kernel<NP>();
        if (hipFuncSetAttribute(pk, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES) != hipSuccess) return MPL_E_LAUNCH;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pk, 512, H2_LDS_BYTES) != hipSuccess || per_cu < 1) return MPL_E_UNSUPPORTED;
        resident[dev].store(cus, std::memory_order_release);
    }
    H2StackArgs a;
    a.att2 = reinterpret_cast<char*>(att2);
    a.hid2 = reinterpret_cast<char*>(hid2);
    a.x16 = reinterpret_cast<char*>(x16);
    a.x = x;
    a.stats = stats;
    a.counters = counters;
    a.M = M; a.D = D; a.n_tok = n_tok; a.heads = heads;
    a.rpt = h2_rows_per_tile(n_tok);
    a.n_tiles = (M + a.rpt - 1) / a.rpt;
    a.G = D / BN;
    const int cap = resident[dev].load() / a.G;
    if (cap < 1) return MPL_E_UNSUPPORTED;
    const H2Form form = h2_stack_form<NP>(M, D, n_tok, resident[dev].load());
    const bool pairs = form.pairs != 0;
    const int n_units = pairs ? (a.n_tiles + 1) / 2 : a.n_tiles;
    a.rgs = form.rgs;
    // plain hand-off stores for teams on one XCD: everywhere but the two-tile stage of the fp16x2 engine.  Same-process A/B
    // (tools/wt_ab.py): bf16 engine -2 .. -4 % stack time (its stage is bound by the L2 -> LDS path, half of its bytes are
    // activations), row-narrow teams -2 %, headline fp16x2 stack 0 % in time but 9 % less fabric traffic (2.43 -> 2.22 GB per
    // launch, L2 hit rate 80 -> 89 %: profiles/r05_gemm_traffic.json); FULL on the two-tile stage +1.7 % (slower): write-through there
    a.plain_ok = (!h2_write_through_always() && (NP == 1 || !pairs)) ? 1 : 0;
    const int n_units_n = a.rgs == 4 ? n_units : a.n_tiles * (4 / a.rgs);
    a.n_teams = n_units_n < cap ? n_units_n : cap;
    if (a.n_teams * a.G > H2_MAX_WGS) a.n_teams = H2_MAX_WGS / a.G;
    // residency: every workgroup that takes part (n_teams x G <= cap x G <= CUs) needs a CU of its own -- 160 KiB of LDS make that
    // ONE per CU whatever the occupancy API reports (it is only asked whether the kernel fits at all); the grid is rounded up to
    // whole XCD rounds, the surplus blocks leave at once (team >= n_teams)
    if (a.n_teams * a.G > resident[dev].load()) return MPL_E_UNSUPPORTED;
    a.n_apps = n_apps;
    a.n_phases = (stop_after > 0 && stop_after < 4 * n_apps) ? stop_after : 4 * n_apps;
    a.eps = eps;
    a.dbg = h2_debug_buffer();
    a.err_ws = counters + h2_err_index(a.n_tiles);
    a.xcc = counters + H2_CTR_PER_TILE * a.n_tiles;
    a.err_host = device_error_word(dev);
    a.spin_log2 = h2_spin_log2();
    a.inject = take_fault_injection();
    for (int i = 0; i < n_apps; ++i)
        for (int j = 0; j < 4; ++j) {
            if (!ops[4 * i + j]) return MPL_E_INVALID;
            a.w[i][j] = reinterpret_cast<const char*>(ops[4 * i + j]);
        }
    // the library serialises ITS OWN persistent launches per device (h2_launch_stack below); not under graph capture
    if (int rc = refuse_stream_capture(s)) return rc;
    hipEvent_t ev = stack_chain_event(dev);
    if (!ev) return MPL_E_LAUNCH;
    std::lock_guard<std::mutex> g(stack_chain_mutex(dev));
    if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) return MPL_E_LAUNCH;
    int rc;
    {
        ProfScope prof(MPL_K_GEMM, s);
        rc = MPL_OK;
        if (form.direct) rc = launch_h2d_stack(a, ((a.n_teams + 7) / 8) * 8 * a.G, s);
        else if (a.rgs != 4) rc = launch_h2n_stack(a, ((a.n_teams + 7) / 8) * 8 * a.G, s);
        else if (pairs) hipLaunchKernelGGL(h2_pair_kernel<NP>(), dim3(((a.n_teams + 7) / 8) * 8 * a.G), dim3(512), H2_LDS_BYTES, s, a);
        else hipLaunchKernelGGL(h2_stack_kernel<NP>, dim3(((a.n_teams + 7) / 8) * 8 * a.G), dim3(512), H2_LDS_BYTES, s, a);
        if (rc == MPL_OK) rc = hip_check_launch();
    }
    if (hipEventRecord(ev, s) != hipSuccess) return MPL_E_LAUNCH;
    return rc;
}


}  // namespace mpl
