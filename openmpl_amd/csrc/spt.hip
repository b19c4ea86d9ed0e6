// Fused spatial stage: joint embedding + the whole SPT block stack + Spatial_norm + per-view glue,
// one launch, token matrices resident in LDS for all L+1 block applications.
//
// Reference (MPL/lib/models/multiview_mpl.py): Spatial_forward_features :349-414 (embedding :355-385,
// optional 3D position encoding :389-396, block loop with the last block applied twice :405-410,
// Spatial_norm :412) and the per-view part of forward :458-492 (confidence_in_FPT :465-467, ray
// embedding concat :469-471 / :486-489, 3D position embedding :474-483, flatten :491).
//
// Geometry (J = 17 joints, d = 32, H = 8 heads, head dim 4, MLP hidden 64):
//   * one 256-thread workgroup owns SEQ = 16 sequences (same view, 16 consecutive poses) = 272 token
//     rows = 17 MFMA row tiles of 16; at B*V = 4096 that is exactly 256 workgroups, one per CU.
//   * LDS: X[272][36] residual stream (39 kB) + Q[272][100] scratch (109 kB): q|k|v columns
//     0..95, attention output overwrites q in place, the MLP hidden layer (64 wide) aliases q|k.
//   * every Linear runs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32): A fragments come from LDS
//     with one ds_read_b128 per 16-deep k step (k permuted identically on both operands), B fragments
//     (the reference's [out][in] weights, 32 kB per block, L2 resident) are loaded straight into
//     registers once per phase -- they are shared by no other wave, so staging them in LDS buys nothing.
//   * LayerNorm is fused into A-fragment formation: the 32 values of a row sit in the 4 lanes
//     {i, i+16, i+32, i+48}, so mean/variance are two xor-shuffles.
//   * attention (17x17 scores, head dim 4) is VALU work: one thread per (row, head), scores in
//     registers, softmax without any cross-lane traffic; k/v rows are LDS broadcasts.
#include <stdlib.h>

#include "common.hpp"

namespace mpl {

constexpr int SJ = 17;          // joints
constexpr int SD = 32;          // embed_dim_ratio
constexpr int SH = 8;           // heads
constexpr int SEQ = 16;         // sequences per workgroup
constexpr int ROWS = SEQ * SJ;  // 272
constexpr int MT = ROWS / 16;   // 17 row tiles
constexpr int XS = 36;          // X row stride (floats)
constexpr int QS = 100;         // Q row stride (floats)
constexpr int NTHR = 512;        // 8 waves: two per SIMD
constexpr int NWAVE = NTHR / 64;
constexpr int SPT_LDS_BYTES = (ROWS * XS + ROWS * QS) * 4;  // 147968

struct SptParams {
    const float* poses[MPL_MAX_VIEWS];
    const float* rays[MPL_MAX_VIEWS];
    const float* centers[MPL_MAX_VIEWS];
    const mpl_spt_set* sets;
    const float *snorm_w, *snorm_b;
    const float *pos3d_embed, *pos3d_view, *pos3d_lin_w, *pos3d_lin_b;
    const float *ray_w, *ray_b, *cfpt_w, *cfpt_b;
    float* xs;
    int B, V, in_ch, n_apps;
    unsigned flags;
    int c3;  // channel count of the pos_3d_* tensors (d or 2d)
    int spw;  // sequences per workgroup (spt_kernel: 1..16, few sequences spread over the chip; spt3_kernel<SS>: SS)
    int abl;  // bench-only ablation mask (MPL_SPT_ABL): 1 no attention, 2 no GELU, 4 no MFMA phases, 8 no epilogue math
    unsigned* err_host;  // sticky error word of the device (common.hpp device_error_word): bit 1 = an operand left its fp16 window
    unsigned char sched[MPL_MAX_APPS];  // layer | weighted << 7
};

// Pointers fetched from device tables carry no address-space information; tell the compiler they are
// global so it emits global_load (vmcnt only) instead of flat_load.
typedef const __attribute__((address_space(1))) float* gfp;
__device__ __forceinline__ gfp G(const float* p) { return (gfp)p; }
__device__ __forceinline__ float4 ld4(gfp p) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 t = *reinterpret_cast<const __attribute__((address_space(1))) v4*>(p);
    return make_float4(t.x, t.y, t.z, t.w);
}

// LayerNorm'ed A fragments of row tile m for a K = 32 GEMM: a0 covers k = 4kq..4kq+3, a1 k = 16+4kq..
__device__ __forceinline__ void ln_frags(const float* X, int m, int li, int kq, const float4& g0, const float4& g1,
                                         const float4& b0, const float4& b1, float4& a0, float4& a1) {
    const float* xr = X + (m * 16 + li) * XS + 4 * kq;
    float4 x0 = ld4(xr), x1 = ld4(xr + 16);
    float s = ((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w));
    s = xor16_add(s);
    s = xor32_add(s);
    const float mean = s * (1.0f / 32.0f);
    x0.x -= mean; x0.y -= mean; x0.z -= mean; x0.w -= mean;
    x1.x -= mean; x1.y -= mean; x1.z -= mean; x1.w -= mean;
    float ss = ((x0.x * x0.x + x0.y * x0.y) + (x0.z * x0.z + x0.w * x0.w)) +
               ((x1.x * x1.x + x1.y * x1.y) + (x1.z * x1.z + x1.w * x1.w));
    ss = xor16_add(ss);
    ss = xor32_add(ss);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 32.0f) + 1e-6f);
    a0.x = x0.x * rstd * g0.x + b0.x; a0.y = x0.y * rstd * g0.y + b0.y;
    a0.z = x0.z * rstd * g0.z + b0.z; a0.w = x0.w * rstd * g0.w + b0.w;
    a1.x = x1.x * rstd * g1.x + b1.x; a1.y = x1.y * rstd * g1.y + b1.y;
    a1.z = x1.z * rstd * g1.z + b1.z; a1.w = x1.w * rstd * g1.w + b1.w;
}

// All MFMA B fragments and LayerNorm / bias vectors of one Block that this lane needs (128 + 30 registers).
// They are loaded straight from the reference's [out][in] tensors one phase ahead of their use, so the L2
// latency never sits on the critical path of a phase.
struct BlockFrags {
    float4 wq[6][2]; float bq[6];     // attn.qkv: 6 column tiles x (k 0..15 | k 16..31)
    float4 wp[2][2]; float bp[2];     // attn.proj
    float4 w1[4][2]; float b1[4];     // mlp.fc1
    float4 w2[2][4]; float b2[2];     // mlp.fc2 (K = 64: four 16-deep steps)
    float4 g1a, g1b, e1a, e1b;        // norm1 gamma/beta of this lane's 8 k columns
    float4 g2a, g2b, e2a, e2b;        // norm2
};

__device__ __forceinline__ void load_qkv_frags(const mpl_block_weights& bw, BlockFrags& F, int li, int kq) {
    F.g1a = ld4(G(bw.ln1_w) + 4 * kq); F.g1b = ld4(G(bw.ln1_w) + 16 + 4 * kq);
    F.e1a = ld4(G(bw.ln1_b) + 4 * kq); F.e1b = ld4(G(bw.ln1_b) + 16 + 4 * kq);
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        const gfp wr = G(bw.qkv_w) + (n * 16 + li) * SD + 4 * kq;
        F.wq[n][0] = ld4(wr);
        F.wq[n][1] = ld4(wr + 16);
        F.bq[n] = G(bw.qkv_b)[n * 16 + li];
    }
}

__device__ __forceinline__ void load_proj_frags(const mpl_block_weights& bw, BlockFrags& F, int li, int kq) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const gfp wr = G(bw.proj_w) + (n * 16 + li) * SD + 4 * kq;
        F.wp[n][0] = ld4(wr);
        F.wp[n][1] = ld4(wr + 16);
        F.bp[n] = G(bw.proj_b)[n * 16 + li];
    }
}

__device__ __forceinline__ void load_fc1_frags(const mpl_block_weights& bw, BlockFrags& F, int li, int kq) {
    F.g2a = ld4(G(bw.ln2_w) + 4 * kq); F.g2b = ld4(G(bw.ln2_w) + 16 + 4 * kq);
    F.e2a = ld4(G(bw.ln2_b) + 4 * kq); F.e2b = ld4(G(bw.ln2_b) + 16 + 4 * kq);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const gfp wr = G(bw.fc1_w) + (n * 16 + li) * SD + 4 * kq;
        F.w1[n][0] = ld4(wr);
        F.w1[n][1] = ld4(wr + 16);
        F.b1[n] = G(bw.fc1_b)[n * 16 + li];
    }
}

__device__ __forceinline__ void load_fc2_frags(const mpl_block_weights& bw, BlockFrags& F, int li, int kq) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const gfp wr = G(bw.fc2_w) + (n * 16 + li) * (2 * SD) + 4 * kq;
#pragma unroll
        for (int q = 0; q < 4; ++q) F.w2[n][q] = ld4(wr + 16 * q);
        F.b2[n] = G(bw.fc2_b)[n * 16 + li];
    }
}

// ---- the same fragments from a block STAGED in LDS (spt_kernel<true>: few sequences per workgroup).  A staged block is 32 1-KiB
// pieces in fragment order -- qkv: piece 2 n + h = W[16 n + li][16 h + 4 kq ..]; proj 12 + 2 n + h; fc1 16 + 2 n + h; fc2
// 24 + 4 n + q -- and, from float SPT_WB_VEC on, the vectors qkv_b[96] | proj_b[32] | fc1_b[64] | fc2_b[32] | ln1_w | ln1_b |
// ln2_w | ln2_b (32 each).
constexpr int SPT_WB_VEC = 8192;                 // floats
constexpr int SPT_WB_FLOATS = SPT_WB_VEC + 512;  // one staged block
constexpr int SPT_SMALL_ROWS = 144;              // token rows of the staged form: up to 8 sequences (136 rows) per workgroup
constexpr int SPT_SMALL_SPW = 8;
constexpr int SPT_SMALL_LDS_BYTES = (SPT_SMALL_ROWS * (XS + QS) + 2 * SPT_WB_FLOATS) * 4;   // 147968
// Staging is LDS-DMA with per-lane source addresses, pieces 0..31 weights, 32 / 33 the vectors; the waves w0 .. w0 + nw - 1 share them
// round robin.  What it costs is the rate at which the CU's address path accepts requests: ~60 cycles per piece in fragment order (16
// half-used lines; ~40 for a contiguous KiB), and a wave stands in its request until it is accepted -- 2100 cycles per wave and
// application when all eight waves request at the head of an application.  The requests are therefore made by the waves the attention
// phase leaves idle (17 nl x 8 (row, head) pairs: 136 threads at one sequence per workgroup).  Measured and not kept: ordinary 16-byte
// loads into registers at the head of the application, written to LDS three phases later (the loads queue up in the same address path:
// 3150 cycles); contiguous pieces from a packed copy (1330 cycles when everybody requests: not worth a second derived operand).
__device__ __forceinline__ const float* spt_uniform(const float* q) {
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ void stage_block(const mpl_block_weights& bwv, float* wb, int wave, int lane, int w0, int nw) {
    if (wave < w0 || wave >= w0 + nw) return;
    const int li = lane & 15, kq = lane >> 4;
    // The pointers came by vector loads: ALL of them into scalar registers first.  Left in vector registers the compiler puts a
    // vmcnt(0) in front of every use behind an opaque DMA statement -- i.e. waits for the previous piece's trip to memory, piece by piece.
    struct { const float *qkv_w, *proj_w, *fc1_w, *fc2_w, *qkv_b, *proj_b, *fc1_b, *fc2_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b; } bw;
    bw.qkv_w = spt_uniform(bwv.qkv_w); bw.proj_w = spt_uniform(bwv.proj_w); bw.fc1_w = spt_uniform(bwv.fc1_w); bw.fc2_w = spt_uniform(bwv.fc2_w);
    bw.qkv_b = spt_uniform(bwv.qkv_b); bw.proj_b = spt_uniform(bwv.proj_b); bw.fc1_b = spt_uniform(bwv.fc1_b); bw.fc2_b = spt_uniform(bwv.fc2_b);
    bw.ln1_w = spt_uniform(bwv.ln1_w); bw.ln1_b = spt_uniform(bwv.ln1_b); bw.ln2_w = spt_uniform(bwv.ln2_w); bw.ln2_b = spt_uniform(bwv.ln2_b);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)wb);
    for (int pc = wave - w0; pc < 34; pc += nw) {
        const float* src;
        bool on = true;
        if (pc < 12) src = bw.qkv_w + ((pc >> 1) * 16 + li) * SD + 16 * (pc & 1) + 4 * kq;
        else if (pc < 16) src = bw.proj_w + (((pc - 12) >> 1) * 16 + li) * SD + 16 * (pc & 1) + 4 * kq;
        else if (pc < 24) src = bw.fc1_w + (((pc - 16) >> 1) * 16 + li) * SD + 16 * (pc & 1) + 4 * kq;
        else if (pc < 32) src = bw.fc2_w + (((pc - 24) >> 2) * 16 + li) * (2 * SD) + 16 * (pc & 3) + 4 * kq;
        else if (pc == 32)
            src = lane < 24 ? bw.qkv_b + 4 * lane
                : lane < 32 ? bw.proj_b + 4 * (lane - 24)
                : lane < 48 ? bw.fc1_b + 4 * (lane - 32)
                : lane < 56 ? bw.fc2_b + 4 * (lane - 48) : bw.ln1_w + 4 * (lane - 56);
        else {
            src = lane < 8 ? bw.ln1_b + 4 * lane : lane < 16 ? bw.ln2_w + 4 * (lane - 8) : bw.ln2_b + 4 * (lane - 16);
            on = lane < 24;
        }
        if (on) dma16(src, lds0 + (unsigned)(pc * 1024));
    }
}
__device__ __forceinline__ float4 wb4(const float* wb, int piece, int lane) { return *reinterpret_cast<const float4*>(wb + piece * 256 + lane * 4); }
// Fragments are read ON DEMAND, tile by tile (a wave owns one or two output tiles of a phase; eight waves reading all fragments of a
// phase into registers were 96 KiB of LDS traffic per qkv phase).
__device__ __forceinline__ float4 wbv4(const float* wb, int off, int kq) { return *reinterpret_cast<const float4*>(wb + SPT_WB_VEC + off + 4 * kq); }

// "Touch" prefetched fragments: an empty asm that reads them makes hipcc place their s_waitcnt HERE.  Every phase
// first touches the fragments it is about to use (they were loaded at least one phase earlier, so the wait is
// free) and only then issues the next prefetch -- otherwise the compiler's vmcnt(0) in front of the first MFMA
// would also wait for the loads issued a moment ago and expose the full L2 latency every phase.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void touch(const float4& a) {
    const v4f v = {a.x, a.y, a.z, a.w};
    asm volatile("" ::"v"(v));
}
__device__ __forceinline__ void touch(float a) { asm volatile("" ::"v"(a)); }

// one 16x16 output tile of a K = 32 GEMM: two independent accumulator chains (k 0..15 / 16..31) so that
// consecutive MFMAs never wait on the 40-cycle dependent-accumulator latency
__device__ __forceinline__ f32x4 tile_k32(const float4& a0, const float4& a1, const float4& w0, const float4& w1) {
    f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    c0 = mfma16(a0.x, w0.x, c0); c1 = mfma16(a1.x, w1.x, c1);
    c0 = mfma16(a0.y, w0.y, c0); c1 = mfma16(a1.y, w1.y, c1);
    c0 = mfma16(a0.z, w0.z, c0); c1 = mfma16(a1.z, w1.z, c1);
    c0 = mfma16(a0.w, w0.w, c0); c1 = mfma16(a1.w, w1.w, c1);
    return c0 + c1;
}

// Row order of the token matrix X in LDS.  TM = false: row = sequence * 17 + joint (the fp32-MFMA kernel); TM = true:
// row = joint * 16 + sequence -- token-major: MFMA row tile j holds joint j of the 16 sequences, so a lane of a transposed
// accumulator tile is one (sequence, head) and the attention needs no cross-lane traffic (spt3_kernel).
template <bool TM, int SS = SEQ>
__device__ __forceinline__ void row_to_sj(int r, int& sq, int& j) {
    if (TM) { j = r / SS; sq = r % SS; }       // SS sequences per joint (a power of two): rows beyond 17 SS belong to no joint (j >= 17)
    else { sq = r / SJ; j = r - sq * SJ; }
}

// joint embedding (:355-396) of the workgroup's 16 sequences -> X
template <bool TM, int SS = SEQ>
__device__ __forceinline__ void spt_embed(const SptParams& p, const mpl_spt_set& set, float* X, int tid, int b0,
                                          const float* pose, const float* ray, const float* cen, int nseq = SEQ, int nrows = ROWS) {
    for (int idx = tid; idx < nrows * SD; idx += NTHR) {
        const int r = idx >> 5, c = idx & 31;
        int sq, j;
        row_to_sj<TM, SS>(r, sq, j);
        const int b = b0 + sq;
        float x = 0.f;
        if (b < p.B && sq < nseq && j < SJ) {
            const float* in = pose + ((size_t)b * SJ + j) * 3;
            const float* we = set.embed_w + c * p.in_ch;
            x = set.embed_b[c] + we[0] * in[0] + we[1] * in[1];
            if (p.in_ch == 3) x += we[2] * in[2];
            if (p.flags & MPL_F_CONF_ADD) x += set.conf_w[c] * in[2] + set.conf_b[c];
            if (p.flags & MPL_F_CONF_MULT) x *= set.conf_w[c] * in[2] + set.conf_b[c];
            x += set.pos_embed[j * SD + c];
            if (p.flags & MPL_F_POS3D_SPATIAL) {
                if (p.flags & MPL_F_POS3D_LEARN) {
                    x += p.pos3d_embed[j * p.c3 + c];
                } else {
                    const float* rr = ray + ((size_t)b * SJ + j) * 3;
                    const float* cc = cen + (size_t)b * 3;
                    const float vx = rr[0] - cc[0], vy = rr[1] - cc[1], vz = rr[2] - cc[2];
                    const float nrm = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);  // F.normalize eps
                    const float* wl = p.pos3d_lin_w + c * 3;
                    x += p.pos3d_lin_b[c] + wl[0] * (vx / nrm) + wl[1] * (vy / nrm) + wl[2] * (vz / nrm);
                }
            }
        }
        X[r * XS + c] = x;
    }
}

// Spatial_norm (:412) + per-view glue (:465-491) -> xs[b*V+v][...]
template <bool TM, int SS = SEQ>
__device__ __forceinline__ void spt_epilogue(const SptParams& p, const float* X, int tid, int view, int b0, const float* pose,
                                             const float* ray, const float* cen, int nseq = SEQ, int nrows = ROWS) {
    // ---------------- epilogue: Spatial_norm (:412) + per-view glue (:465-491) -> xs[b*V+v][...] ------------
    const bool to_rays = (p.flags & MPL_F_POS3D_TO_RAYS) && (p.flags & MPL_F_RAYS_TOKEN);   // feature concat (:469-471)
    const bool ray_tok = !(p.flags & MPL_F_POS3D_TO_RAYS) && (p.flags & MPL_F_RAYS_TOKEN);  // token concat (:486-489)
    const int cw = to_rays ? 2 * SD : SD;                 // channels per joint in the output row
    const int Df = SJ * SD * ((p.flags & MPL_F_RAYS_TOKEN) ? 2 : 1);
    for (int r = tid; r < nrows; r += NTHR) {
        int sq, j;
        row_to_sj<TM, SS>(r, sq, j);
        const int b = b0 + sq;
        if (b >= p.B || sq >= nseq || j >= SJ) continue;
        const float* xr = X + r * XS;
        float v[SD];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < SD; c += 4) {
            const float4 t = ld4(xr + c);
            v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
            s += (t.x + t.y) + (t.z + t.w);
        }
        const float mean = s * (1.0f / 32.0f);
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < SD; ++c) {
            v[c] -= mean;
            ss = fmaf(v[c], v[c], ss);
        }
        const float rstd = 1.0f / sqrtf(ss * (1.0f / 32.0f) + 1e-6f);
        const float conf = pose[((size_t)b * SJ + j) * 3 + 2];
        float dx = 0.f, dy = 0.f, dz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
        const bool need_dir = (p.flags & MPL_F_RAYS_TOKEN) ||
                              (!(p.flags & MPL_F_POS3D_SPATIAL) && !(p.flags & MPL_F_POS3D_LEARN));
        if (need_dir) {
            const float* rr = ray + ((size_t)b * SJ + j) * 3;
            const float* cc = cen + (size_t)b * 3;
            dx = rr[0] - cc[0]; dy = rr[1] - cc[1]; dz = rr[2] - cc[2];
            const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
            nx = dx / nrm; ny = dy / nrm; nz = dz / nrm;
        }
        // 3D position term for channel c of this joint (:474-483)
        auto pos3d = [=](int c) -> float {
            if (p.flags & MPL_F_POS3D_SPATIAL) return p.pos3d_view[j * p.c3 + c];
            if (p.flags & MPL_F_POS3D_LEARN) return p.pos3d_embed[j * p.c3 + c];
            const float* wl = p.pos3d_lin_w + c * 3;
            return p.pos3d_lin_b[c] + wl[0] * nx + wl[1] * ny + wl[2] * nz;
        };
        auto ray_emb = [=](int c) -> float {
            const float* wr = p.ray_w + c * 3;
            return p.ray_b[c] + wr[0] * dx + wr[1] * dy + wr[2] * dz;
        };
        float* orow = p.xs + ((size_t)b * p.V + view) * Df;
        float* o1 = orow + j * cw;
#pragma unroll
        for (int c = 0; c < SD; c += 4) {
            float t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y = v[c + q] * rstd * p.snorm_w[c + q] + p.snorm_b[c + q];
                if (p.flags & MPL_F_CONF_IN_FPT) y += p.cfpt_w[c + q] * conf + p.cfpt_b[c + q];
                t[q] = y + pos3d(c + q);
            }
            st4(o1 + c, float4{t[0], t[1], t[2], t[3]});
        }
        if (to_rays) {
#pragma unroll
            for (int c = 0; c < SD; c += 4) {
                float t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = ray_emb(c + q) + pos3d(SD + c + q);
                st4(o1 + SD + c, float4{t[0], t[1], t[2], t[3]});
            }
        } else if (ray_tok) {
            float* o2 = orow + (SJ + j) * SD;
#pragma unroll
            for (int c = 0; c < SD; c += 4) st4(o2 + c, float4{ray_emb(c), ray_emb(c + 1), ray_emb(c + 2), ray_emb(c + 3)});
        }
    }
}

// STAGED = false: the weights of a phase come straight from global memory into registers, requested one phase ahead (a phase
// over 17 row tiles is longer than the trip).  STAGED = true (at most SPT_SMALL_SPW sequences per workgroup): with 2-9 row tiles
// a phase is SHORTER than the trip to L2 / HBM (measured: 134 us per launch with the one-phase-ahead scheme at one sequence per
// workgroup, i.e. 2 us = one memory round trip per phase), so the whole block of the NEXT application is staged in LDS by LDS-DMA
// while the current one computes (32 KiB + vectors, two buffers in the LDS the missing rows leave free) and a phase reads its
// fragments from there.  The arithmetic of a row is the same instruction sequence in both forms.
template <bool STAGED>
__global__ __launch_bounds__(NTHR, 1) void spt_kernel(const SptParams p) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    constexpr int RX = STAGED ? SPT_SMALL_ROWS : ROWS;
    float* WB = smem;                                   // STAGED: two staged blocks in front (1-KiB aligned pieces)
    float* X = smem + (STAGED ? 2 * SPT_WB_FLOATS : 0);
    float* Q = X + RX * XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int view = blockIdx.x % p.V;
    const int b0 = (blockIdx.x / p.V) * p.spw;
    const mpl_spt_set set = p.sets[(p.flags & MPL_F_MULTI_SPT) ? view : 0];
    const float* pose = p.poses[view];
    const float* ray = p.rays[view];
    const float* cen = p.centers[view];
    // Few sequences (B V below the 16 x CU count a full launch needs) are SPREAD: p.spw = 1 .. 16 sequences per workgroup, rows
    // sequence-major, so the live rows are the first 17 nl and only their `mt` row tiles are walked (a single frame: one sequence
    // = 2 row tiles per workgroup instead of 17 mostly empty ones).  The arithmetic of a row does not depend on spw.
    const int nl = p.B - b0 < p.spw ? p.B - b0 : p.spw;
    const int rows_live = nl * SJ;
    const int mt = (rows_live + 15) >> 4;

    // weights of the first Block application: issue the loads before anything else
    BlockFrags F;
    mpl_block_weights bw, bw_next;
    if (p.n_apps > 0) {
        bw = set.blocks[p.sched[0] & 0x7f];
        if (STAGED) stage_block(bw, WB, wave, lane, 0, NWAVE);
        else load_qkv_frags(bw, F, li, kq);
        // STAGED: the pointers of an application are fetched one application ahead of the requests that need them
        if (STAGED && p.n_apps > 1) bw_next = set.blocks[p.sched[1] & 0x7f];
    }

    // ---------------- phase 0: joint embedding (:355-396) ----------------
    spt_embed<false>(p, set, X, tid, b0, pose, ray, cen, nl, mt * 16);
    // STAGED: nothing inside the application loop may come by a vector load from global memory -- the compiler's vmcnt(0) in front
    // of its use would wait for the block in flight.  The schedule bytes and the confidences of the live rows (the weighted
    // applications, :61-62) therefore wait in the free tails of the two vector regions.
    unsigned char* sched_l = reinterpret_cast<unsigned char*>(WB + SPT_WB_FLOATS + SPT_WB_VEC + 352);      // [MPL_MAX_APPS]
    float* conf_l = WB + SPT_WB_VEC + 352;                                                                  // [SPT_SMALL_ROWS]
    if (STAGED) {
        if (tid < MPL_MAX_APPS) sched_l[tid] = p.sched[tid];
        for (int r = tid; r < rows_live; r += NTHR) {
            const int sq = r / SJ;
            conf_l[r] = pose[((size_t)(b0 + sq) * SJ + (r - sq * SJ)) * 3 + 2];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the first block have landed
    }
    __syncthreads();

    // ---------------- block applications (:405-410) ----------------
    // Output tiles (16 rows x 16 columns) of every Linear are dealt to the 8 waves as contiguous ranges of the
    // row-major tile list; a wave walks its row tiles and tests each column tile against its range (static
    // indices keep every fragment in registers).
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tlast = (p.abl & 16) ? __builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](int k) {
        if (p.abl & 16) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            ph[k] += now - tlast;
            tlast = now;
        }
    };
    for (int app = 0; app < p.n_apps; ++app) {
        const bool weighted = ((STAGED ? sched_l[app] : p.sched[app]) & 0x80) != 0;
        const bool more = app + 1 < p.n_apps;
        if (!STAGED && more) bw_next = set.blocks[p.sched[app + 1] & 0x7f];   // pointers only; used two phases later
        const float* wb = WB + (app & 1) * SPT_WB_FLOATS;
        mpl_block_weights bw_after;
        // Prefetch schedule: every phase first touches its own fragments, then issues the loads of the NEXT phase
        // (proj weights during qkv, fc1 during proj, fc2 during fc1, the next application's qkv during fc2), so each
        // group has a whole phase to arrive and at most two groups are live at a time.
        if (!STAGED) {
#pragma unroll
            for (int n = 0; n < 6; ++n) { touch(F.wq[n][0]); touch(F.wq[n][1]); touch(F.bq[n]); }
            touch(F.g1a); touch(F.g1b); touch(F.e1a); touch(F.e1b);
            load_proj_frags(bw, F, li, kq);
        }

        // ---- QKV = LN1(X) . Wqkv^T + b : 17 x 6 tiles -> Q[:, 0:96]
        if (STAGED) {
            const int lo = (mt * 6 * wave) / NWAVE, hi = (mt * 6 * (wave + 1)) / NWAVE;
            const float4 g0 = wbv4(wb, 224, kq), g1 = wbv4(wb, 240, kq), e0 = wbv4(wb, 256, kq), e1 = wbv4(wb, 272, kq);
            float4 a0, a1;
            int m_have = -1;
            for (int u = lo; u < hi && !(p.abl & 4); ++u) {
                const int m = u / 6, n = u - 6 * m;
                const float4 w0 = wb4(wb, 2 * n, lane), w1 = wb4(wb, 2 * n + 1, lane);
                const float bq = wb[SPT_WB_VEC + n * 16 + li];
                if (m != m_have) { ln_frags(X, m, li, kq, g0, g1, e0, e1, a0, a1); m_have = m; }
                const f32x4 c = tile_k32(a0, a1, w0, w1);
                float* qd = Q + (m * 16 + 4 * kq) * QS + n * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) qd[r * QS] = c[r] + bq;
            }
        } else {
            const int lo = (mt * 6 * wave) / NWAVE, hi = (mt * 6 * (wave + 1)) / NWAVE;
            for (int m = lo / 6; m <= (hi - 1) / 6 && !(p.abl & 4); ++m) {
                float4 a0, a1;
                ln_frags(X, m, li, kq, F.g1a, F.g1b, F.e1a, F.e1b, a0, a1);
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    const int u = m * 6 + n;
                    if (u < lo || u >= hi) continue;
                    const f32x4 c = tile_k32(a0, a1, F.wq[n][0], F.wq[n][1]);
                    float* qd = Q + (m * 16 + 4 * kq) * QS + n * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) qd[r * QS] = c[r] + F.bq[n];
                }
            }
        }
        __syncthreads();
        stamp(0);

        // STAGED: the block of the next application is requested NOW, by the waves the attention leaves idle (stage_block).  Its readers
        // finished an application ago.
        if (STAGED && more) {
            const int busy = (rows_live * SH + 63) >> 6;                  // waves with attention work
            const int w0 = busy < NWAVE - 1 ? busy : 0;
            stage_block(bw_next, WB + ((app + 1) & 1) * SPT_WB_FLOATS, wave, lane, w0, NWAVE - w0);
            if (app + 2 < p.n_apps) bw_after = set.blocks[sched_l[app + 2] & 0x7f];
        }
        stamp(5);
        // ---- attention: thread per (row, head); 17 scores in registers (:55-64)
        for (int pr = tid; pr < rows_live * SH && !(p.abl & 1); pr += NTHR) {
            const int r = pr >> 3, h = pr & 7;
            const int sq = r / SJ;
            const float* kb = Q + (sq * SJ) * QS + SD + 4 * h;
            const float4 q = ld4(Q + r * QS + 4 * h);
            float sc[SJ];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < SJ; ++j) {
                const float4 k = ld4(kb + j * QS);
                sc[j] = 0.5f * (fmaf(q.x, k.x, q.y * k.y) + fmaf(q.z, k.z, q.w * k.w));  // hd^-0.5 = 0.5
                mx = fmaxf(mx, sc[j]);
            }
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < SJ; ++j) {
                sc[j] = __expf(sc[j] - mx);
                l += sc[j];
            }
            float inv = 1.0f / l;
            if (weighted) {  // attn * conf_weights.unsqueeze(1) after softmax (:61-62): scales query row r
                const int b = b0 + sq;
                if (STAGED) inv *= conf_l[r];
                else inv *= (b < p.B) ? pose[((size_t)b * SJ + (r - sq * SJ)) * 3 + 2] : 0.f;
            }
            float4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < SJ; ++j) {
                const float4 v = ld4(kb + j * QS + SD);
                const float pj = sc[j] * inv;
                o.x = fmaf(pj, v.x, o.x);
                o.y = fmaf(pj, v.y, o.y);
                o.z = fmaf(pj, v.z, o.z);
                o.w = fmaf(pj, v.w, o.w);
            }
            st4(Q + r * QS + 4 * h, o);  // overwrite q (only this thread ever reads it)
        }
        __syncthreads();
        stamp(1);

        // ---- X += attn_out . Wproj^T + b : 17 x 2 tiles
        if (!STAGED) {
#pragma unroll
            for (int n = 0; n < 2; ++n) { touch(F.wp[n][0]); touch(F.wp[n][1]); touch(F.bp[n]); }
            load_fc1_frags(bw, F, li, kq);
        }
        if (STAGED) {
            const int lo = (mt * 2 * wave) / NWAVE, hi = (mt * 2 * (wave + 1)) / NWAVE;
            for (int u = lo; u < hi && !(p.abl & 4); ++u) {
                const int m = u >> 1, n = u & 1;
                const float4 w0 = wb4(wb, 12 + 2 * n, lane), w1 = wb4(wb, 12 + 2 * n + 1, lane);
                const float bp = wb[SPT_WB_VEC + 96 + n * 16 + li];
                const float* ar = Q + (m * 16 + li) * QS + 4 * kq;
                const float4 a0 = ld4(ar), a1 = ld4(ar + 16);
                const f32x4 c = tile_k32(a0, a1, w0, w1);
                float* xd = X + (m * 16 + 4 * kq) * XS + n * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) xd[r * XS] += c[r] + bp;
            }
        } else {
            const int lo = (mt * 2 * wave) / NWAVE, hi = (mt * 2 * (wave + 1)) / NWAVE;
            for (int m = lo >> 1; m <= ((hi - 1) >> 1) && !(p.abl & 4); ++m) {
                const float* ar = Q + (m * 16 + li) * QS + 4 * kq;
                const float4 a0 = ld4(ar), a1 = ld4(ar + 16);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int u = 2 * m + n;
                    if (u < lo || u >= hi) continue;
                    const f32x4 c = tile_k32(a0, a1, F.wp[n][0], F.wp[n][1]);
                    float* xd = X + (m * 16 + 4 * kq) * XS + n * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) xd[r * XS] += c[r] + F.bp[n];
                }
            }
        }
        __syncthreads();
        stamp(2);

        // ---- Hid = gelu(LN2(X) . W1^T + b) : 17 x 4 tiles -> Q[:, 0:64]
        if (!STAGED) {
#pragma unroll
            for (int n = 0; n < 4; ++n) { touch(F.w1[n][0]); touch(F.w1[n][1]); touch(F.b1[n]); }
            touch(F.g2a); touch(F.g2b); touch(F.e2a); touch(F.e2b);
            load_fc2_frags(bw, F, li, kq);
        }
        if (STAGED) {
            const int lo = (mt * 4 * wave) / NWAVE, hi = (mt * 4 * (wave + 1)) / NWAVE;
            const float4 g0 = wbv4(wb, 288, kq), g1 = wbv4(wb, 304, kq), e0 = wbv4(wb, 320, kq), e1 = wbv4(wb, 336, kq);
            float4 a0, a1;
            int m_have = -1;
            for (int u = lo; u < hi && !(p.abl & 4); ++u) {
                const int m = u >> 2, n = u & 3;
                const float4 w0 = wb4(wb, 16 + 2 * n, lane), w1 = wb4(wb, 16 + 2 * n + 1, lane);
                const float b1 = wb[SPT_WB_VEC + 128 + n * 16 + li];
                if (m != m_have) { ln_frags(X, m, li, kq, g0, g1, e0, e1, a0, a1); m_have = m; }
                const f32x4 c = tile_k32(a0, a1, w0, w1);
                float* qd = Q + (m * 16 + 4 * kq) * QS + n * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) qd[r * QS] = (p.abl & 2) ? (c[r] + b1) : gelu_erf(c[r] + b1);
            }
        } else {
            const int lo = (mt * 4 * wave) / NWAVE, hi = (mt * 4 * (wave + 1)) / NWAVE;
            for (int m = lo >> 2; m <= ((hi - 1) >> 2) && !(p.abl & 4); ++m) {
                float4 a0, a1;
                ln_frags(X, m, li, kq, F.g2a, F.g2b, F.e2a, F.e2b, a0, a1);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int u = 4 * m + n;
                    if (u < lo || u >= hi) continue;
                    const f32x4 c = tile_k32(a0, a1, F.w1[n][0], F.w1[n][1]);
                    float* qd = Q + (m * 16 + 4 * kq) * QS + n * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        qd[r * QS] = (p.abl & 2) ? (c[r] + F.b1[n]) : gelu_erf(c[r] + F.b1[n]);
                }
            }
        }
        __syncthreads();
        stamp(3);

        // ---- X += Hid . W2^T + b : K = 64, 17 x 2 tiles
        if (!STAGED) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int q = 0; q < 4; ++q) touch(F.w2[n][q]);
                touch(F.b2[n]);
            }
            if (more) load_qkv_frags(bw_next, F, li, kq);   // qkv fragments are long dead: next application's weights
        }
        if (STAGED) {
            const int lo = (mt * 2 * wave) / NWAVE, hi = (mt * 2 * (wave + 1)) / NWAVE;
            for (int u = lo; u < hi && !(p.abl & 4); ++u) {
                const int m = u >> 1, n = u & 1;
                const float4 w0 = wb4(wb, 24 + 4 * n, lane), w1 = wb4(wb, 24 + 4 * n + 1, lane), w2 = wb4(wb, 24 + 4 * n + 2, lane),
                             w3 = wb4(wb, 24 + 4 * n + 3, lane);
                const float b2 = wb[SPT_WB_VEC + 192 + n * 16 + li];
                const float* ar = Q + (m * 16 + li) * QS + 4 * kq;
                const float4 a0 = ld4(ar), a1 = ld4(ar + 16), a2 = ld4(ar + 32), a3 = ld4(ar + 48);
                const f32x4 c = tile_k32(a0, a1, w0, w1) + tile_k32(a2, a3, w2, w3);
                float* xd = X + (m * 16 + 4 * kq) * XS + n * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) xd[r * XS] += c[r] + b2;
            }
        } else {
            const int lo = (mt * 2 * wave) / NWAVE, hi = (mt * 2 * (wave + 1)) / NWAVE;
            for (int m = lo >> 1; m <= ((hi - 1) >> 1) && !(p.abl & 4); ++m) {
                const float* ar = Q + (m * 16 + li) * QS + 4 * kq;
                const float4 a0 = ld4(ar), a1 = ld4(ar + 16), a2 = ld4(ar + 32), a3 = ld4(ar + 48);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int u = 2 * m + n;
                    if (u < lo || u >= hi) continue;
                    const f32x4 c = tile_k32(a0, a1, F.w2[n][0], F.w2[n][1]) + tile_k32(a2, a3, F.w2[n][2], F.w2[n][3]);
                    float* xd = X + (m * 16 + 4 * kq) * XS + n * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) xd[r * XS] += c[r] + F.b2[n];
                }
            }
        }
        if (STAGED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next block have landed
        __syncthreads();
        stamp(4);
        bw = bw_next;
        if (STAGED && app + 2 < p.n_apps) bw_next = bw_after;
    }

    if ((p.abl & 16) && lane == 0 && blockIdx.x < 32) {
        float* o = p.xs + (size_t)(blockIdx.x * NWAVE + wave) * 8;
        for (int k = 0; k < 6; ++k) o[k] = (float)ph[k];
        return;
    }
    if (p.abl & 16) return;
    spt_epilogue<false>(p, X, tid, view, b0, pose, ray, cen, nl, rows_live);
}

// =====================================================================================================================
// spt3_kernel -- the same stage with the Linear layers on the bf16 matrix cores (fp32 arithmetic from exactly split
// operands, as the round-2 fp32 engine did: x = hi + mid + lo in bf16, six partial products per product, fp32 accumulation).
//
//   * token-major rows (row = joint * 16 + sequence) and the W fragment as FIRST MFMA operand: lane (s, kq) of an
//     accumulator tile holds 4 consecutive columns of (joint j, sequence s) -- for the qkv tiles exactly the 4-dim vector
//     of ONE head (h = 4 hg + kq).  Wave (hg, part) computes q, k, v of head group hg for the joints j = part (mod 4):
//     q stays in registers, k / v go to LDS as K[j][h][s][4] (one ds_write_b128 per tile);
//   * attention: the lane keeps its (sequence, head) and its <= 5 query joints; every K / V row is read once
//     (contiguous 1-KiB wave reads) for all of them -- 34 ds_read_b128 per lane and block application instead of 145;
//   * proj / fc1 / fc2: A fragments are read from LDS (attention output, normalised X, GELU output), split in registers
//     into two fp16 parts (~20 VALU ops per fragment) and multiplied with weight fragments that the binding split once
//     (mpl_spt_pack: fp16 hi | lo in MFMA fragment order under exact power-of-two scales, LayerNorm gain / offset and the
//     biases folded in, prefetched one phase ahead): the arithmetic of h2_gemm.hip -- 816 fp16 MFMAs of 16 cycles per block
//     application (round 2: 1632 on three bf16 parts; round 1: 2176 fp32 MFMAs of 32).
// LDS: X[272][36] | K[17][8][16][4] | V[17][8][16][4] | ATT[272][36]; the MLP hidden HID[272][68] aliases K | V | ATT.
constexpr int ATS = 36;                       // ATT row stride (floats)
constexpr int HS = 68;                        // HID row stride (floats)
constexpr int KV_F = SJ * SH * SEQ * 4;       // 8704 floats each
constexpr int SPT3_RING_BYTES = (ROWS * XS + 2 * KV_F + ROWS * ATS) * 4;   // 147968: X | K | V | ATT
constexpr int SPT3_LDS_BYTES = 160 * 1024;          // + 15872 B: staged weights of the next phase | parameter vectors
// Weights and parameter vectors of a phase are staged in LDS while the phase BEFORE it runs (LDS-DMA for the packed
// weights, one float4 per thread for the 456 epilogue values of a block): a phase that starts with ~16 global
// loads per lane waits ~1.5 k cycles for L2 before its first MFMA, five times per block application (14 % of the kernel).
//   S_W   spare + 0      8 KiB   proj weights (4 KiB, staged during qkv + attention), then fc2 weights (staged during fc1)
//   S_PAR spare + 8 KiB  2 x 456 floats, double buffered by block application (staged during fc2 of the one before)
//   F1    K + 0          8 KiB   fc1 weights (staged during proj: K is dead after the attention); HID starts 12 KiB in
//   Q     ATT + 20 KiB   12 KiB  qkv weights of the NEXT application (staged during fc2; HID ends at ATT + 16.3 KiB)
constexpr int SPT3_HID_OFF = 3072;                  // floats: HID = K + 12 KiB
constexpr int SPT3_Q_OFF = 20480;                   // bytes into ATT
constexpr int SPT3_NPAR = 456;                      // floats of epilogue vectors per block: c[224] | sc[224] | scalars[8]
static_assert(SPT3_HID_OFF + ROWS * HS <= 2 * KV_F + ROWS * ATS, "HID does not fit its alias");
static_assert((SPT3_HID_OFF + ROWS * HS - 2 * KV_F) * 4 <= SPT3_Q_OFF, "HID reaches into the staged qkv weights");
static_assert(SPT3_Q_OFF + 12 * 1024 <= ROWS * ATS * 4, "staged qkv weights do not fit behind HID in ATT");
static_assert(SPT3_RING_BYTES + 8 * 1024 + 2 * SPT3_NPAR * 4 <= SPT3_LDS_BYTES, "spare LDS too small");
// packed block (mpl_spt_pack): fp16 hi | lo fragments of the four weight matrices (2 KiB per 16-column x 32-k unit), then
// the epilogue vectors
constexpr int SPT_PACK_QKV = 0, SPT_PACK_PROJ = 12 * 1024, SPT_PACK_FC1 = 16 * 1024, SPT_PACK_FC2 = 24 * 1024;
constexpr int SPT_PACK_VEC = 32 * 1024;
constexpr int SPT_PACK_BYTES = 48 * 1024;
constexpr int SPT_C_QKV = 0, SPT_C_PROJ = 96, SPT_C_FC1 = 128, SPT_C_FC2 = 192, SPT_NCOL = 224;
constexpr float SPT_SA = 1024.0f;                   // scale of a normalised LayerNorm input (|z| <= sqrt(32))
constexpr float SPT_QS = 0.5f * 1.4426950408889634f;   // hd^-0.5 log2 e, folded into the q columns (scores in the exp2 domain)

typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 8 fp32 -> hi / lo packed fp16 (RNE; the residual is exact in fp32; subnormal results are kept): h2_gemm.hip
__device__ __forceinline__ void spt_split2(const float (&x)[8], sf16x8& hi, sf16x8& lo) { ::mpl::split2_f16(x, hi, lo); }     // common.hpp
// largest power of two p with p * v <= 2^15 (v > 0, finite); 1 for v == 0
__device__ inline float spt_window_scale(float v) {
    if (!(v > 0.f) || !(v < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(32768.0f / v, &e);
    e = e - 1 < -100 ? -100 : (e - 1 > 100 ? 100 : e - 1);
    return ldexpf(1.0f, e);
}

// The D = 32 Linear layers of an SPT block as split-operand fp16 GEMMs (the arithmetic of h2_gemm.hip: x = hi + lo, three
// products, exact power-of-two scales): ONE workgroup packs a block.
//   * LayerNorm GEMMs (qkv, fc1): gamma is folded into W, beta and the bias into c_n = b_n + sum_k beta_k W_nk; the kernel
//     multiplies z = (x - mean) rstd 2^10;
//   * every column n has its own scale sw_n (max_k |W'_nk| sw_n in [2^13, 2^14)) that the epilogue multiplier sc_n takes out;
//   * the inputs of proj (attention output) and fc2 (GELU output) carry ONE static scale each from the data-free bound
//     |LN(x) . W'_n + c_n| <= sqrt(32) |W'_n|_2 + |c_n| of the producing columns (v columns of qkv; fc1), window 2^15;
//   * the q columns also carry hd^-0.5 log2 e (the scores are formed in the exp2 domain).
// Layout: fragments [16 units][hi | lo][64 lanes][8 fp16] (units: 6 qkv, 2 proj, 4 fc1, 4 fc2 = (n tile, k step)), then at
// SPT_PACK_VEC floats c[224] | sc[224] | {s_att, s_hid / 2, ...}: columns qkv 0..95, proj 96.., fc1 128.., fc2 192..
__global__ __launch_bounds__(256) void spt_pack_kernel(const float* __restrict__ qkv_w, const float* __restrict__ qkv_b,
                                                        const float* __restrict__ ln1_w, const float* __restrict__ ln1_b,
                                                        const float* __restrict__ proj_w, const float* __restrict__ proj_b,
                                                        const float* __restrict__ fc1_w, const float* __restrict__ fc1_b,
                                                        const float* __restrict__ ln2_w, const float* __restrict__ ln2_b,
                                                        const float* __restrict__ fc2_w, const float* __restrict__ fc2_b,
                                                        char* __restrict__ dst, int fold_q) {
    __shared__ float Wf[8192];                  // qkv' [96][32] | proj [32][32] | fc1' [64][32] | fc2 [32][64]
    __shared__ float cn[SPT_NCOL], sw[SPT_NCOL], bnd[SPT_NCOL], scal[2];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) {
        float w;
        if (i < 3072) w = qkv_w[i] * ln1_w[i & 31];
        else if (i < 4096) w = proj_w[i - 3072];
        else if (i < 6144) w = fc1_w[i - 4096] * ln2_w[i & 31];
        else w = fc2_w[i - 6144];
        Wf[i] = w;
    }
    __syncthreads();
    if (tid < SPT_NCOL) {
        const int n = tid;
        const float *wr, *raw, *beta = nullptr;
        int K = 32;
        float bias;
        if (n < 96) { wr = Wf + n * 32; raw = qkv_w + n * 32; beta = ln1_b; bias = qkv_b[n]; }
        else if (n < 128) { wr = Wf + 3072 + (n - 96) * 32; raw = proj_w + (n - 96) * 32; bias = proj_b[n - 96]; }
        else if (n < 192) { wr = Wf + 4096 + (n - 128) * 32; raw = fc1_w + (n - 128) * 32; beta = ln2_b; bias = fc1_b[n - 128]; }
        else { wr = Wf + 6144 + (n - 192) * 64; raw = fc2_w + (n - 192) * 64; bias = fc2_b[n - 192]; K = 64; }
        float amax = 0.f;
        double ss = 0.0, c = (double)bias;
        for (int k = 0; k < K; ++k) {
            amax = fmaxf(amax, fabsf(wr[k]));
            ss += (double)wr[k] * (double)wr[k];
            if (beta) c += (double)raw[k] * (double)beta[k];
        }
        float s = 1.0f;
        if (amax > 0.f && amax < 3.0e38f) {
            int e;
            (void)frexpf(amax, &e);             // amax = m 2^e, m in [0.5, 1): amax 2^(14 - e) in [2^13, 2^14)
            e = 14 - e;
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            s = ldexpf(1.0f, e);
        }
        cn[n] = (float)c;
        sw[n] = s;
        bnd[n] = beta ? (float)(sqrt(32.0) * sqrt(ss)) + fabsf((float)c) : 0.f;
    }
    __syncthreads();
    if (tid == 0) {
        float batt = 0.f, bhid = 0.f;
        for (int n = 64; n < 96; ++n) batt = fmaxf(batt, bnd[n]);                    // v columns of qkv
        for (int n = SPT_C_FC1; n < SPT_C_FC2; ++n) bhid = fmaxf(bhid, bnd[n]);
        scal[0] = spt_window_scale(batt);
        scal[1] = spt_window_scale(bhid);
    }
    __syncthreads();
    float* vec = reinterpret_cast<float*>(dst + SPT_PACK_VEC);
    if (tid < SPT_NCOL) {
        const int n = tid;
        float c = cn[n], sc;
        if (n < 96) sc = 1.0f / (SPT_SA * sw[n]);
        else if (n < 128) sc = 1.0f / (scal[0] * sw[n]);
        else if (n < 192) sc = 1.0f / (SPT_SA * sw[n]);
        else sc = 1.0f / (scal[1] * sw[n]);
        if (n < 32 && fold_q) { c *= SPT_QS; sc *= SPT_QS; }
        vec[n] = c;
        vec[SPT_NCOL + n] = sc;
    }
    if (tid < 8) vec[2 * SPT_NCOL + tid] = tid == 0 ? scal[0] : (tid == 1 ? 0.5f * scal[1] : 0.f);
    // fragment f of the packed block = 8 consecutive k of one weight row (scaled by its column scale), two parts
    sf16x8* frag = reinterpret_cast<sf16x8*>(dst);
    for (int idx = tid; idx < 16 * 64; idx += 256) {
        const int lane = idx & 63, u = idx >> 6, li = lane & 15, kq = lane >> 4;
        const float* src;
        int n;
        if (u < 6) { n = 16 * u + li; src = Wf + n * 32 + 8 * kq; }
        else if (u < 8) { n = 16 * (u - 6) + li; src = Wf + 3072 + n * 32 + 8 * kq; n += SPT_C_PROJ; }
        else if (u < 12) { n = 16 * (u - 8) + li; src = Wf + 4096 + n * 32 + 8 * kq; n += SPT_C_FC1; }
        else { n = 16 * ((u - 12) >> 1) + li; src = Wf + 6144 + n * 64 + 32 * ((u - 12) & 1) + 8 * kq; n += SPT_C_FC2; }   // [n][ks]
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = src[j] * sw[n];
        sf16x8 hi, lo;
        spt_split2(x, hi, lo);
        frag[(size_t)(u * 2) * 64 + lane] = hi;
        frag[(size_t)(u * 2 + 1) * 64 + lane] = lo;
    }
}

// fold_q: the q columns carry hd^-0.5 log2 e (the SPT kernel's exp2-domain attention); 0 for the D = 32 FPT blocks, whose
// attention kernel scales q itself
int launch_spt_pack(const mpl_block_weights* bw_host, unsigned short* dst, int fold_q, hipStream_t s) {
    const mpl_block_weights* b = bw_host;
    if (!b || !dst || !b->qkv_w || !b->proj_w || !b->fc1_w || !b->fc2_w || !b->qkv_b || !b->proj_b || !b->fc1_b || !b->fc2_b ||
        !b->ln1_w || !b->ln1_b || !b->ln2_w || !b->ln2_b)
        return MPL_E_INVALID;
    ProfScope prof(MPL_K_PACK, s);
    hipLaunchKernelGGL(spt_pack_kernel, dim3(1), dim3(256), 0, s, b->qkv_w, b->qkv_b, b->ln1_w, b->ln1_b, b->proj_w, b->proj_b, b->fc1_w,
                       b->fc1_b, b->ln2_w, b->ln2_b, b->fc2_w, b->fc2_b, reinterpret_cast<char*>(dst), fold_q);
    return hip_check_launch();
}

size_t spt_pack_bytes() { return SPT_PACK_BYTES; }

// acc(16 x 16, transposed) += the three significant part products of A (hi, lo) and W (hi, lo): lo.hi, hi.lo, hi.hi
__device__ __forceinline__ f32x4 mfma3(const sf16x8 (&w)[2], const sf16x8& ah, const sf16x8& al, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], al, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], ah, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], ah, c, 0, 0, 0);
    return c;
}

// SS = sequences per workgroup (16, 8, 4, 2 or 1): rows = joint * SS + sequence, 17 SS of them in MTS row tiles.  A launch of few
// sequences takes as few per workgroup as keep it within one workgroup per CU (launch_spt): the kernel's time is VALU work per ROW
// TILE, so 4 sequences per workgroup walk 5 tiles instead of 17.  The arithmetic of a row does not depend on SS (the matrix
// instructions treat rows independently; the attention of a (sequence, head, query joint) visits the keys in the same order).
template <int SS>
__global__ __launch_bounds__(NTHR, 1) void spt3_kernel(const SptParams p) {
    constexpr int RLIVE = SJ * SS;                 // live rows
    constexpr int MTS = (RLIVE + 15) / 16;         // row tiles
    constexpr int NT = (MTS + 3) / 4;              // row tiles of a wave in the qkv / attention phases (tiles part, part + 4, ...)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* Kb = smem + ROWS * XS;
    float* Vb = Kb + KV_F;
    float* ATT = Vb + KV_F;
    float* HID = Kb + SPT3_HID_OFF;                // alias (K, V, ATT are dead between proj and the next qkv)
    char* S_W = reinterpret_cast<char*>(ATT + ROWS * ATS);
    float* S_PAR = reinterpret_cast<float*>(S_W + 8 * 1024);
    char* R_F1 = reinterpret_cast<char*>(Kb);
    char* R_Q = reinterpret_cast<char*>(ATT) + SPT3_Q_OFF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;      // li = sequence (row in tile), kq = k quarter / column quad
    // every barrier of this kernel also publishes staged DMA pieces: the compiler does not see the LDS-DMA requests (inline
    // asm), so the wait for them is explicit
    auto phase_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const int hg = wave & 1, part = wave >> 1;     // head group, joint class
    const int view = blockIdx.x % p.V;
    const int b0 = (blockIdx.x / p.V) * SS;
    const int sl = li % SS;                        // this lane's sequence in every row tile (16 is a multiple of SS)
    const mpl_spt_set set = p.sets[(p.flags & MPL_F_MULTI_SPT) ? view : 0];
    const float* pose = p.poses[view];
    const float* ray = p.rays[view];
    const float* cen = p.centers[view];

    // stage n_pieces KiB of a packed block (section at byte_off) into LDS at dst: wave w brings pieces w, w + 8, ...;
    // the __syncthreads() that ends the current phase (it waits for vmcnt(0)) publishes them
    auto stage_w = [&](char* dst, const unsigned short* pack, int byte_off, int n_pieces) {
        const unsigned l0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
        for (int i = wave; i < n_pieces; i += NWAVE)
            ::mpl::dma16(reinterpret_cast<const float*>(reinterpret_cast<const char*>(pack) + byte_off + i * 1024) + lane * 4,
                         l0 + (unsigned)(i * 1024));
    };
    // the epilogue vectors of a block (c | sc | scalars, written by mpl_spt_pack behind the fragments), one float4 per thread
    auto load_par = [&](const mpl_block_weights& b) -> float4 {
        if (tid >= SPT3_NPAR / 4) return float4{0.f, 0.f, 0.f, 0.f};
        return ld4(G(reinterpret_cast<const float*>(reinterpret_cast<const char*>(b.qkv_w3) + SPT_PACK_VEC)) + 4 * tid);
    };
    auto store_par = [&](int app_of, const float4& v) {
        if (tid < SPT3_NPAR / 4) st4(S_PAR + (app_of & 1) * SPT3_NPAR + 4 * tid, v);
    };
    mpl_block_weights bw;
    if (p.n_apps > 0) {                            // application 0: its qkv weights and parameters, under the embedding
        bw = set.blocks[p.sched[0] & 0x7f];
        stage_w(R_Q, bw.qkv_w3, SPT_PACK_QKV, 12);
    }
    const float4 par0 = p.n_apps > 0 ? load_par(bw) : float4{0.f, 0.f, 0.f, 0.f};
    spt_embed<true, SS>(p, set, X, tid, b0, pose, ray, cen, SS, MTS * 16);
    store_par(0, par0);
    phase_sync();

    auto load_w = [&](const char* region, int unit, sf16x8 (&w)[2]) {      // from the staged section in LDS
        const sf16x8* g = reinterpret_cast<const sf16x8*>(region) + (size_t)unit * 2 * 64 + lane;
        w[0] = g[0];
        w[1] = g[64];
    };
    // normalised, split A fragment of row tile m (K = 32): lane (s, kq) holds k = 8 kq .. 8 kq + 7 of row 16 m + s;
    // z = (x - mean) rstd 2^10 (gamma / beta live in the packed weights / c)
    auto ln_frag = [&](int m, sf16x8& ah, sf16x8& al) {
        const float* xr = X + (m * 16 + li) * XS + 8 * kq;
        float4 x0 = ::mpl::ld4(xr), x1 = ::mpl::ld4(xr + 4);
        float sm = ((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w));
        sm = ::mpl::xor32_add(::mpl::xor16_add(sm));
        const float mean = sm * (1.0f / 32.0f);
        x0.x -= mean; x0.y -= mean; x0.z -= mean; x0.w -= mean;
        x1.x -= mean; x1.y -= mean; x1.z -= mean; x1.w -= mean;
        float ss = ((x0.x * x0.x + x0.y * x0.y) + (x0.z * x0.z + x0.w * x0.w)) +
                   ((x1.x * x1.x + x1.y * x1.y) + (x1.z * x1.z + x1.w * x1.w));
        ss = ::mpl::xor32_add(::mpl::xor16_add(ss));
        const float rs = __builtin_amdgcn_rsqf(ss * (1.0f / 32.0f) + 1e-6f) * SPT_SA;   // v_rsq_f32 (1 ulp)
        const float y[8] = {x0.x * rs, x0.y * rs, x0.z * rs, x0.w * rs, x1.x * rs, x1.y * rs, x1.z * rs, x1.w * rs};
        spt_split2(y, ah, al);
    };
    // plain A fragment: the producer already applied the static scale of the operand (attention output, GELU output)
    auto raw_frag = [&](const float* rowp, sf16x8& ah, sf16x8& al) {
        const float4 x0 = ::mpl::ld4(rowp), x1 = ::mpl::ld4(rowp + 4);
        const float y[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        spt_split2(y, ah, al);
    };

    for (int app = 0; app < p.n_apps; ++app) {
        const bool weighted = (p.sched[app] & 0x80) != 0;
        bw = set.blocks[p.sched[app] & 0x7f];
        const unsigned short* pack = bw.qkv_w3;
        const float* par = S_PAR + (app & 1) * SPT3_NPAR;
        stage_w(S_W, pack, SPT_PACK_PROJ, 4);      // proj weights: land during qkv + attention
        // ---------------- qkv: this wave's q, k, v tiles (head group hg) of its joints
        {
            sf16x8 wq[3][2];
            float4 bq[3], sq[3];                   // c_n and sc_n of this lane's q, k, v columns (q: times hd^-0.5 log2 e)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                load_w(R_Q, 2 * c + hg, wq[c]);
                bq[c] = ::mpl::ld4(par + SPT_C_QKV + 32 * c + 16 * hg + 4 * kq);
                sq[c] = ::mpl::ld4(par + SPT_NCOL + SPT_C_QKV + 32 * c + 16 * hg + 4 * kq);
            }
            float4 q[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                q[t] = float4{0.f, 0.f, 0.f, 0.f};
                if (part + 4 * t < MTS && !(p.abl & 8)) {
                    const int m = part + 4 * t;
                    const int j = (16 * m + li) / SS;                   // this lane's joint in row tile m (SS = 16: j = m)
                    const bool live = 16 * m + li < RLIVE;
                    sf16x8 ah, al;
                    ln_frag(m, ah, al);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 cq = mfma3(wq[0], ah, al, z), ck = mfma3(wq[1], ah, al, z), cv = mfma3(wq[2], ah, al, z);
                    q[t] = float4{fmaf(cq[0], sq[0].x, bq[0].x), fmaf(cq[1], sq[0].y, bq[0].y), fmaf(cq[2], sq[0].z, bq[0].z),
                                  fmaf(cq[3], sq[0].w, bq[0].w)};
                    const int h = 4 * hg + kq;
                    // K tile: keys in PAIRS, components interleaved -- [pair][plane][h][seq]{c_j, c_j+1, c'_j, c'_j+1} with plane 0
                    // = (x, y), plane 1 = (z, w) -- so that two scores come out of one packed multiply-add; the 17th key stays
                    // a plain [h][seq]{x, y, z, w} record behind the 8 pairs
                    const float kx = fmaf(ck[0], sq[1].x, bq[1].x), ky = fmaf(ck[1], sq[1].y, bq[1].y);
                    const float kz = fmaf(ck[2], sq[1].z, bq[1].z), kw = fmaf(ck[3], sq[1].w, bq[1].w);
                    if (live && j < SJ - 1) {
                        float* kp = Kb + ((((j >> 1) * 2) * SH + h) * SS + sl) * 4 + (j & 1);
                        kp[0] = kx;
                        kp[2] = ky;
                        kp[SH * SS * 4] = kz;
                        kp[SH * SS * 4 + 2] = kw;
                    } else if (live) {
                        st4(Kb + (SJ - 1) * SH * SS * 4 + (h * SS + sl) * 4, float4{kx, ky, kz, kw});
                    }
                    if (live)
                        st4(Vb + ((j * SH + h) * SS + sl) * 4, float4{fmaf(cv[0], sq[2].x, bq[2].x), fmaf(cv[1], sq[2].y, bq[2].y),
                                                                       fmaf(cv[2], sq[2].z, bq[2].z), fmaf(cv[3], sq[2].w, bq[2].w)});
                }
            }
            phase_sync();
            // ---------------- attention (:55-64): lane = (sequence li, head h), its <= 5 query joints against all 17 keys.
            // Scores in the exp2 domain (the q columns carry hd^-0.5 log2 e = 0.5 log2 e from their epilogue multiplier), the
            // probabilities stay unnormalised until the output row is complete; the output leaves with the static scale of the
            // proj operand (par[448], a power of two).
            if (!(p.abl & 1)) {
                const int h = 4 * hg + kq;
                float sc[NT][SJ];
#pragma unroll
                for (int jp = 0; jp < SJ / 2; ++jp) {
                    const float4 k01 = ::mpl::ld4(Kb + (((jp * 2) * SH + h) * SS + sl) * 4);        // x_j x_j+1 y_j y_j+1
                    const float4 k23 = ::mpl::ld4(Kb + (((jp * 2 + 1) * SH + h) * SS + sl) * 4);    // z_j z_j+1 w_j w_j+1
                    const f32x2 kx = {k01.x, k01.y}, ky = {k01.z, k01.w}, kz = {k23.x, k23.y}, kw = {k23.z, k23.w};
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f32x2 qx = {q[t].x, q[t].x}, qy = {q[t].y, q[t].y}, qz = {q[t].z, q[t].z}, qw = {q[t].w, q[t].w};
                        f32x2 s2 = qx * kx;
                        s2 = __builtin_elementwise_fma(qy, ky, s2);
                        s2 = __builtin_elementwise_fma(qz, kz, s2);
                        s2 = __builtin_elementwise_fma(qw, kw, s2);
                        sc[t][2 * jp] = s2[0];
                        sc[t][2 * jp + 1] = s2[1];
                    }
                }
                {
                    const float4 k = ::mpl::ld4(Kb + (SJ - 1) * SH * SS * 4 + (h * SS + sl) * 4);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        sc[t][SJ - 1] = fmaf(q[t].w, k.w, fmaf(q[t].z, k.z, fmaf(q[t].y, k.y, q[t].x * k.x)));
                }
                const float s_att = par[2 * SPT_NCOL];
                float inv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float mx = sc[t][0];
#pragma unroll
                    for (int j = 1; j < SJ; ++j) mx = fmaxf(mx, sc[t][j]);
                    float l = 0.f;
#pragma unroll
                    for (int j = 0; j < SJ; ++j) {
                        sc[t][j] = __builtin_amdgcn_exp2f(sc[t][j] - mx);
                        l += sc[t][j];
                    }
                    inv[t] = __builtin_amdgcn_rcpf(l) * s_att;              // v_rcp_f32 (1 ulp)
                    if (weighted) {  // attn * conf_weights.unsqueeze(1) after softmax (:61-62): scales the query row
                        const int b = b0 + sl, r = 16 * (part + 4 * t) + li;
                        inv[t] *= (b < p.B && r < RLIVE) ? pose[((size_t)b * SJ + r / SS) * 3 + 2] : 0.f;
                    }
                }
                float4 o[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) o[t] = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < SJ; ++j) {
                    const float4 v = ::mpl::ld4(Vb + ((j * SH + h) * SS + sl) * 4);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float pj = sc[t][j];
                        o[t].x = fmaf(pj, v.x, o[t].x);
                        o[t].y = fmaf(pj, v.y, o[t].y);
                        o[t].z = fmaf(pj, v.z, o[t].z);
                        o[t].w = fmaf(pj, v.w, o[t].w);
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    o[t] = float4{o[t].x * inv[t], o[t].y * inv[t], o[t].z * inv[t], o[t].w * inv[t]};
                    // the confidence weights are data (reference :61-62 multiplies the softmax rows by whatever `conf` it is
                    // given): only with them can the operand leave the window its static, data-free scale assumes.  That is
                    // reported, never absorbed: the row becomes NaN (so do the poses of its sequence) and the device error word
                    // gets bit 1 -- the next API call and check_device() raise, pointing at the native-fp32 engine, which has
                    // no window.  (Rounds 3-5 clamped to +-65000 here: plausible-looking poses from saturated operands.)
                    if (weighted) {
                        const float big = fmaxf(fmaxf(fabsf(o[t].x), fabsf(o[t].y)), fmaxf(fabsf(o[t].z), fabsf(o[t].w)));
                        if (!(big <= 65000.f)) {
                            const float qn = __builtin_nanf("");
                            o[t] = float4{qn, qn, qn, qn};
                            if (p.err_host) __hip_atomic_fetch_or(p.err_host, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (part + 4 * t < MTS) st4(ATT + ((part + 4 * t) * 16 + li) * ATS + 4 * h, o[t]);
            }
        }
        phase_sync();
        // ---------------- X += attn_out . Wproj^T + b : 17 x 2 tiles, dealt as contiguous ranges of the (m, n) list
        {
            stage_w(R_F1, pack, SPT_PACK_FC1, 8);       // fc1 weights into the dead K tile
            sf16x8 wp[2][2];
            float4 bp[2], sp[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                load_w(S_W, n, wp[n]);
                bp[n] = ::mpl::ld4(par + SPT_C_PROJ + 16 * n + 4 * kq);
                sp[n] = ::mpl::ld4(par + SPT_NCOL + SPT_C_PROJ + 16 * n + 4 * kq);
            }
            const int lo = (MTS * 2 * wave) / NWAVE, hi = (MTS * 2 * (wave + 1)) / NWAVE;
            for (int m = lo >> 1; m <= ((hi - 1) >> 1) && !(p.abl & 32); ++m) {
                sf16x8 ah, al;
                raw_frag(ATT + (m * 16 + li) * ATS + 8 * kq, ah, al);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int u = 2 * m + n;
                    if (u < lo || u >= hi) continue;
                    const f32x4 c = mfma3(wp[n], ah, al, f32x4{0.f, 0.f, 0.f, 0.f});
                    float* xd = X + (m * 16 + li) * XS + 16 * n + 4 * kq;
                    const float4 x = ::mpl::ld4(xd);
                    st4(xd, float4{x.x + fmaf(c[0], sp[n].x, bp[n].x), x.y + fmaf(c[1], sp[n].y, bp[n].y),
                                   x.z + fmaf(c[2], sp[n].z, bp[n].z), x.w + fmaf(c[3], sp[n].w, bp[n].w)});
                }
            }
        }
        phase_sync();
        // ---------------- Hid = gelu(LN2(X) . W1^T + b) : 17 x 4 tiles
        {
            stage_w(S_W, pack, SPT_PACK_FC2, 8);        // fc2 weights (the proj weights in S_W were read a phase ago)
            sf16x8 w1[4][2];
            float4 b1[4], s1[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                load_w(R_F1, n, w1[n]);
                b1[n] = ::mpl::ld4(par + SPT_C_FC1 + 16 * n + 4 * kq);
                s1[n] = ::mpl::ld4(par + SPT_NCOL + SPT_C_FC1 + 16 * n + 4 * kq);
            }
            const float hs = par[2 * SPT_NCOL + 1];     // half the static scale of the fc2 operand (a power of two)
            const int lo = (MTS * 4 * wave) / NWAVE, hi = (MTS * 4 * (wave + 1)) / NWAVE;
            for (int m = lo >> 2; m <= ((hi - 1) >> 2) && !(p.abl & 64); ++m) {
                sf16x8 ah, al;
                ln_frag(m, ah, al);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int u = 4 * m + n;
                    if (u < lo || u >= hi) continue;
                    const f32x4 c = mfma3(w1[n], ah, al, f32x4{0.f, 0.f, 0.f, 0.f});
                    st4(HID + (m * 16 + li) * HS + 16 * n + 4 * kq,
                        float4{gelu_as_scaled(fmaf(c[0], s1[n].x, b1[n].x), hs), gelu_as_scaled(fmaf(c[1], s1[n].y, b1[n].y), hs),
                               gelu_as_scaled(fmaf(c[2], s1[n].z, b1[n].z), hs), gelu_as_scaled(fmaf(c[3], s1[n].w, b1[n].w), hs)});
                }
            }
        }
        phase_sync();
        // ---------------- X += Hid . W2^T + b : K = 64 (two k steps), 17 x 2 tiles
        {
            // the next application's qkv weights (behind HID in ATT) and parameters travel under this phase
            float4 parn = float4{0.f, 0.f, 0.f, 0.f};
            if (app + 1 < p.n_apps) {
                const mpl_block_weights bn = set.blocks[p.sched[app + 1] & 0x7f];
                stage_w(R_Q, bn.qkv_w3, SPT_PACK_QKV, 12);
                parn = load_par(bn);
            }
            sf16x8 w2[2][2][2];
            float4 b2[2], s2[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                load_w(S_W, 2 * n, w2[n][0]);
                load_w(S_W, 2 * n + 1, w2[n][1]);
                b2[n] = ::mpl::ld4(par + SPT_C_FC2 + 16 * n + 4 * kq);
                s2[n] = ::mpl::ld4(par + SPT_NCOL + SPT_C_FC2 + 16 * n + 4 * kq);
            }
            const int lo = (MTS * 2 * wave) / NWAVE, hi = (MTS * 2 * (wave + 1)) / NWAVE;
            for (int m = lo >> 1; m <= ((hi - 1) >> 1) && !(p.abl & 128); ++m) {
                sf16x8 ah0, al0, ah1, al1;
                raw_frag(HID + (m * 16 + li) * HS + 8 * kq, ah0, al0);
                raw_frag(HID + (m * 16 + li) * HS + 32 + 8 * kq, ah1, al1);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int u = 2 * m + n;
                    if (u < lo || u >= hi) continue;
                    f32x4 c = mfma3(w2[n][0], ah0, al0, f32x4{0.f, 0.f, 0.f, 0.f});
                    c = mfma3(w2[n][1], ah1, al1, c);
                    float* xd = X + (m * 16 + li) * XS + 16 * n + 4 * kq;
                    const float4 x = ::mpl::ld4(xd);
                    st4(xd, float4{x.x + fmaf(c[0], s2[n].x, b2[n].x), x.y + fmaf(c[1], s2[n].y, b2[n].y),
                                   x.z + fmaf(c[2], s2[n].z, b2[n].z), x.w + fmaf(c[3], s2[n].w, b2[n].w)});
                }
            }
            if (app + 1 < p.n_apps) store_par(app + 1, parn);
        }
        phase_sync();
    }
    spt_epilogue<true, SS>(p, X, tid, view, b0, pose, ray, cen, SS, MTS * 16);
}

// ---------------------------------------------------------------------------------------------- D = 32 FPT blocks
// The FPT blocks of the keypoint-token variant (FPT_blocks_view_keypoint_tokens: 17 V tokens of width 32, :261-266, :436-437)
// have the SPT block's Linear shapes, so they run from the same packed operand (mpl_spt_pack format, q columns unscaled) with
// the same arithmetic.  Everything except the attention is ROW-LOCAL at this width: a wave takes a 16-row tile through a
// whole chain of GEMMs by itself -- weights in registers for all its tiles, the accumulator layout (lane = row i, 4 columns)
// turned into the next A fragment (lane = row i, 8 consecutive k) through a 16-row scratch tile of its own in LDS, no
// workgroup barrier anywhere:
//   d32_qkv_kernel:  qkv = LN1(x) . Wqkv^T + b                                         (-> token attention kernel)
//   d32_mlp_kernel:  x += att . Wproj^T + b;  x += fc2(gelu(fc1(LN2(x))))              (Block.forward :84-92, Mlp :31-37)
// Two launches per block application besides the attention instead of four GEMMs and two statistics passes.
__device__ __forceinline__ void d32_ln_split(const float4& a0, const float4& a1, sf16x8& ah, sf16x8& al) {
    float4 x0 = a0, x1 = a1;
    float sm = ((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w));
    sm = xor32_add(xor16_add(sm));
    const float mean = sm * (1.0f / 32.0f);
    x0.x -= mean; x0.y -= mean; x0.z -= mean; x0.w -= mean;
    x1.x -= mean; x1.y -= mean; x1.z -= mean; x1.w -= mean;
    float ss = ((x0.x * x0.x + x0.y * x0.y) + (x0.z * x0.z + x0.w * x0.w)) + ((x1.x * x1.x + x1.y * x1.y) + (x1.z * x1.z + x1.w * x1.w));
    ss = xor32_add(xor16_add(ss));
    const float rs = __builtin_amdgcn_rsqf(ss * (1.0f / 32.0f) + 1e-6f) * SPT_SA;
    const float y[8] = {x0.x * rs, x0.y * rs, x0.z * rs, x0.w * rs, x1.x * rs, x1.y * rs, x1.z * rs, x1.w * rs};
    spt_split2(y, ah, al);
}
__device__ __forceinline__ void d32_load_w(const char* pack, int unit, int lane, sf16x8 (&w)[2]) {
    const sf16x8* g = reinterpret_cast<const sf16x8*>(pack) + (size_t)unit * 2 * 64 + lane;
    w[0] = g[0];
    w[1] = g[64];
}

__global__ __launch_bounds__(512) void d32_qkv_kernel(const float* __restrict__ x, int M, const char* __restrict__ pack,
                                                       float* __restrict__ qkv) {
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    const int gw = blockIdx.x * 8 + (threadIdx.x >> 6), nw = gridDim.x * 8;
    const float* vec = reinterpret_cast<const float*>(pack + SPT_PACK_VEC);
    sf16x8 wq[6][2];
    float4 cq[6], sq[6];
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        d32_load_w(pack, n, lane, wq[n]);
        cq[n] = ld4(vec + SPT_C_QKV + 16 * n + 4 * kq);
        sq[n] = ld4(vec + SPT_NCOL + SPT_C_QKV + 16 * n + 4 * kq);
    }
    const int n_tiles = (M + 15) / 16;
    // the rows of the NEXT tile of this wave are requested before the current one is multiplied (a wave walks ~4 tiles; one
    // memory round trip per tile in the open was most of the kernel's time)
    auto rows_of = [&](int tile, float4& a0, float4& a1) {
        const int r = tile * 16 + li;
        const float* xr = x + (size_t)(r < M ? r : M - 1) * SD + 8 * kq;
        a0 = ld4(xr);
        a1 = ld4(xr + 4);
    };
    float4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
    if (gw < n_tiles) rows_of(gw, n0, n1);
    for (int tile = gw; tile < n_tiles; tile += nw) {
        const int row = tile * 16 + li;
        const bool ok = row < M;
        const float4 c0 = n0, c1 = n1;
        if (tile + nw < n_tiles) rows_of(tile + nw, n0, n1);
        sf16x8 ah, al;
        d32_ln_split(c0, c1, ah, al);
        float* o = qkv + (size_t)row * (3 * SD) + 4 * kq;
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const f32x4 c = mfma3(wq[n], ah, al, f32x4{0.f, 0.f, 0.f, 0.f});
            if (ok) st4(o + 16 * n, float4{fmaf(c[0], sq[n].x, cq[n].x), fmaf(c[1], sq[n].y, cq[n].y), fmaf(c[2], sq[n].z, cq[n].z),
                                           fmaf(c[3], sq[n].w, cq[n].w)});
        }
    }
}

__global__ __launch_bounds__(512) void d32_mlp_kernel(float* __restrict__ x, const float* __restrict__ att, int M,
                                                       const char* __restrict__ pack) {
    __shared__ __attribute__((aligned(16))) float scratch[8][16 * XS + 16 * HS];
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
    float* XT = scratch[wave];              // [16][36]: x after the attention half, in A-fragment order for norm2
    float* HT = XT + 16 * XS;               // [16][68]: the hidden layer (already times the static scale of the fc2 operand)
    const float* vec = reinterpret_cast<const float*>(pack + SPT_PACK_VEC);
    sf16x8 wp[2][2], w1[4][2], w2[2][2][2];
    float4 bp[2], sp[2], b1[4], s1[4], b2[2], s2[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        d32_load_w(pack, 6 + n, lane, wp[n]);
        d32_load_w(pack, 12 + 2 * n, lane, w2[n][0]);
        d32_load_w(pack, 12 + 2 * n + 1, lane, w2[n][1]);
        bp[n] = ld4(vec + SPT_C_PROJ + 16 * n + 4 * kq);
        sp[n] = ld4(vec + SPT_NCOL + SPT_C_PROJ + 16 * n + 4 * kq);
        b2[n] = ld4(vec + SPT_C_FC2 + 16 * n + 4 * kq);
        s2[n] = ld4(vec + SPT_NCOL + SPT_C_FC2 + 16 * n + 4 * kq);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        d32_load_w(pack, 8 + n, lane, w1[n]);
        b1[n] = ld4(vec + SPT_C_FC1 + 16 * n + 4 * kq);
        s1[n] = ld4(vec + SPT_NCOL + SPT_C_FC1 + 16 * n + 4 * kq);
    }
    const float s_att = vec[2 * SPT_NCOL], hs = vec[2 * SPT_NCOL + 1];
    const int n_tiles = (M + 15) / 16;
    // operands of the NEXT tile of this wave (attention rows as A fragment, x in accumulator layout) are requested up front
    auto rows_of = [&](int tile, float4& a0, float4& a1, float4 (&xo)[2]) {
        const int r = tile * 16 + li;
        const size_t rcl = (size_t)(r < M ? r : M - 1);
        a0 = ld4(att + rcl * SD + 8 * kq);
        a1 = ld4(att + rcl * SD + 8 * kq + 4);
        xo[0] = ld4(x + rcl * SD + 4 * kq);
        xo[1] = ld4(x + rcl * SD + 16 + 4 * kq);
    };
    float4 na0 = {0.f, 0.f, 0.f, 0.f}, na1 = na0, nx[2] = {na0, na0};
    if (gw < n_tiles) rows_of(gw, na0, na1, nx);
    for (int tile = gw; tile < n_tiles; tile += nw) {
        const int row = tile * 16 + li;
        const bool ok = row < M;
        // ---- x += att . Wproj^T + b
        float4 xn[2];
        {
            const float4 a0 = na0, a1 = na1;
            const float4 xc[2] = {nx[0], nx[1]};
            if (tile + nw < n_tiles) rows_of(tile + nw, na0, na1, nx);
            const float y[8] = {a0.x * s_att, a0.y * s_att, a0.z * s_att, a0.w * s_att, a1.x * s_att, a1.y * s_att, a1.z * s_att, a1.w * s_att};
            sf16x8 ah, al;
            spt_split2(y, ah, al);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 c = mfma3(wp[n], ah, al, f32x4{0.f, 0.f, 0.f, 0.f});
                const float4 xo = xc[n];
                xn[n] = float4{xo.x + fmaf(c[0], sp[n].x, bp[n].x), xo.y + fmaf(c[1], sp[n].y, bp[n].y), xo.z + fmaf(c[2], sp[n].z, bp[n].z),
                               xo.w + fmaf(c[3], sp[n].w, bp[n].w)};
                st4(XT + li * XS + 16 * n + 4 * kq, xn[n]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS stores, read back in another lane order
        // ---- hidden = gelu(LN2(x) . W1^T + b)
        {
            sf16x8 ah, al;
            d32_ln_split(ld4(XT + li * XS + 8 * kq), ld4(XT + li * XS + 8 * kq + 4), ah, al);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f32x4 c = mfma3(w1[n], ah, al, f32x4{0.f, 0.f, 0.f, 0.f});
                st4(HT + li * HS + 16 * n + 4 * kq,
                    float4{gelu_as_scaled(fmaf(c[0], s1[n].x, b1[n].x), hs), gelu_as_scaled(fmaf(c[1], s1[n].y, b1[n].y), hs),
                           gelu_as_scaled(fmaf(c[2], s1[n].z, b1[n].z), hs), gelu_as_scaled(fmaf(c[3], s1[n].w, b1[n].w), hs)});
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- x += hidden . W2^T + b   (K = 64: two k steps)
        {
            sf16x8 ah0, al0, ah1, al1;
            {
                const float4 h0 = ld4(HT + li * HS + 8 * kq), h1 = ld4(HT + li * HS + 8 * kq + 4);
                const float y[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                spt_split2(y, ah0, al0);
            }
            {
                const float4 h0 = ld4(HT + li * HS + 32 + 8 * kq), h1 = ld4(HT + li * HS + 32 + 8 * kq + 4);
                const float y[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                spt_split2(y, ah1, al1);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 c = mfma3(w2[n][0], ah0, al0, f32x4{0.f, 0.f, 0.f, 0.f});
                c = mfma3(w2[n][1], ah1, al1, c);
                if (ok) st4(x + (size_t)row * SD + 16 * n + 4 * kq,
                            float4{xn[n].x + fmaf(c[0], s2[n].x, b2[n].x), xn[n].y + fmaf(c[1], s2[n].y, b2[n].y),
                                   xn[n].z + fmaf(c[2], s2[n].z, b2[n].z), xn[n].w + fmaf(c[3], s2[n].w, b2[n].w)});
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the scratch tiles are rewritten by the next tile
    }
}

static int d32_grid(int M) {
    const int need = ((M + 15) / 16 + 7) / 8;
    return need < 512 ? (need > 0 ? need : 1) : 512;
}
int launch_d32_qkv(const float* x, int M, const unsigned short* pack, float* qkv, hipStream_t s) {
    if (!x || !pack || !qkv || M <= 0) return MPL_E_INVALID;
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL(d32_qkv_kernel, dim3(d32_grid(M)), dim3(512), 0, s, x, M, reinterpret_cast<const char*>(pack), qkv);
    return hip_check_launch();
}
int launch_d32_mlp(float* x, const float* att, int M, const unsigned short* pack, hipStream_t s) {
    if (!x || !pack || !att || M <= 0) return MPL_E_INVALID;
    ProfScope prof(MPL_K_GEMM, s);
    hipLaunchKernelGGL(d32_mlp_kernel, dim3(d32_grid(M)), dim3(512), 0, s, x, att, M, reinterpret_cast<const char*>(pack));
    return hip_check_launch();
}

int launch_spt(const mpl_config* cfg, const mpl_weights* w, const mpl_inputs* in, float* xs, int use_packed, hipStream_t s) {
    if (cfg->num_joints != SJ || cfg->dim != SD || cfg->heads != SH) return MPL_E_UNSUPPORTED;
    if (cfg->num_views < 1 || cfg->num_views > MPL_MAX_VIEWS || in->batch <= 0) return MPL_E_INVALID;
    if (cfg->in_chans != 2 && cfg->in_chans != 3) return MPL_E_INVALID;
    const unsigned f = cfg->flags;
    if ((f & MPL_F_POS3D_TO_RAYS) && !(f & MPL_F_RAYS_TOKEN)) return MPL_E_UNSUPPORTED;  // reference itself fails (:483)
    if ((f & MPL_F_POS3D_TO_RAYS) && (f & MPL_F_POS3D_SPATIAL)) return MPL_E_UNSUPPORTED;
    SptParams p;
    const bool needs_rays = (f & MPL_F_RAYS_TOKEN) || !(f & MPL_F_POS3D_LEARN);
    for (int v = 0; v < MPL_MAX_VIEWS; ++v) {
        const bool on = v < cfg->num_views;
        p.poses[v] = on ? in->poses[v] : nullptr;
        p.rays[v] = on ? in->rays[v] : nullptr;
        p.centers[v] = on ? in->centers[v] : nullptr;
        if (on && !p.poses[v]) return MPL_E_INVALID;
        if (on && needs_rays && (!p.rays[v] || !p.centers[v])) return MPL_E_INVALID;
    }
    p.sets = w->spt_sets;
    p.snorm_w = w->spatial_norm_w; p.snorm_b = w->spatial_norm_b;
    p.pos3d_embed = w->pos_3d_embed; p.pos3d_view = w->pos_3d_view_coding;
    p.pos3d_lin_w = w->pos_3d_linear_w; p.pos3d_lin_b = w->pos_3d_linear_b;
    p.ray_w = w->ray_embed_w; p.ray_b = w->ray_embed_b;
    p.cfpt_w = w->conf_fpt_w; p.cfpt_b = w->conf_fpt_b;
    p.xs = xs;
    p.B = in->batch; p.V = cfg->num_views; p.in_ch = cfg->in_chans;
    p.flags = f;
    p.c3 = (f & MPL_F_POS3D_TO_RAYS) ? 2 * SD : SD;
    // the phase ablations (garbage results) exist in laboratory builds only: tools/build_variants.sh -f spt.hip
    static const int abl = lab_getenv("MPL_SPT_ABL") ? atoi(lab_getenv("MPL_SPT_ABL")) : 0;
    p.abl = abl;
    {
        int dev = 0;
        p.err_host = hipGetDevice(&dev) == hipSuccess ? device_error_word(dev) : nullptr;
    }
    // schedule (:405-410): [blk(x,w)]; if last: blk(x); blk(x)
    int n = 0;
    if (!(f & MPL_F_NO_SPT)) {
        for (int l = 0; l < cfg->depth; ++l) {
            if (n + 3 > MPL_MAX_APPS) return MPL_E_UNSUPPORTED;
            if (f & MPL_F_CONF_ATTN_W) p.sched[n++] = (unsigned char)(l | 0x80);
            if (l == cfg->depth - 1) p.sched[n++] = (unsigned char)l;
            p.sched[n++] = (unsigned char)l;
        }
    }
    p.n_apps = n;
    // >64 KiB of dynamic LDS needs an explicit opt-in, once per device
    static std::atomic<bool> attr_set[64];   // set-once flags: a racing second hipFuncSetAttribute is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MPL_E_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)spt_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SPT_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)spt_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SPT_SMALL_LDS_BYTES) != hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    // Sequences per workgroup of the fp32-MFMA kernel: as few as keep the launch inside one wave of workgroups (one per CU), so
    // that a single frame or a few hundred sequences use the whole chip with 2-3 live row tiles per workgroup instead of a few
    // workgroups with 17; up to SPT_SMALL_SPW per workgroup run the staged form (spt_kernel<true>).
    static std::atomic<int> n_cus[64];
    int cus = n_cus[dev].load(std::memory_order_acquire);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return MPL_E_LAUNCH;
        n_cus[dev].store(cus, std::memory_order_release);
    }
    int spw = SEQ;
    for (int c = 1; c < SEQ; ++c)
        if ((long long)cfg->num_views * ((in->batch + c - 1) / c) <= cus) { spw = c; break; }
    // the packed-operand kernel takes 16, 8, 4, 2 or 1 sequences per workgroup (bitwise the same rows)
    int ss = 1;
    while (ss < spw) ss *= 2;
    p.spw = use_packed ? ss : spw;
    const int grid = cfg->num_views * ((in->batch + p.spw - 1) / p.spw);
    ProfScope prof(MPL_K_SPT, s);
    if (use_packed) {
        void (*k3)(const SptParams) = ss == 1 ? spt3_kernel<1> : ss == 2 ? spt3_kernel<2> : ss == 4 ? spt3_kernel<4> : ss == 8 ? spt3_kernel<8> : spt3_kernel<16>;
        const int ki = ss == 1 ? 0 : ss == 2 ? 1 : ss == 4 ? 2 : ss == 8 ? 3 : 4;
        static std::atomic<bool> attr3[64][5];
        if (!attr3[dev][ki].load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, SPT3_LDS_BYTES) != hipSuccess)
                return MPL_E_LAUNCH;
            attr3[dev][ki].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(k3, dim3(grid), dim3(NTHR), SPT3_LDS_BYTES, s, p);
    } else {
        if (p.spw <= SPT_SMALL_SPW) hipLaunchKernelGGL(spt_kernel<true>, dim3(grid), dim3(NTHR), SPT_SMALL_LDS_BYTES, s, p);
        else hipLaunchKernelGGL(spt_kernel<false>, dim3(grid), dim3(NTHR), SPT_LDS_BYTES, s, p);
    }
    return hip_check_launch();
}

}  // namespace mpl
