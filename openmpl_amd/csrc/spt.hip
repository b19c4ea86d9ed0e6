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
    int abl;  // bench-only ablation mask (MPL_SPT_ABL): 1 no attention, 2 no GELU, 4 no MFMA phases, 8 no epilogue math
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
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / 32.0f);
    x0.x -= mean; x0.y -= mean; x0.z -= mean; x0.w -= mean;
    x1.x -= mean; x1.y -= mean; x1.z -= mean; x1.w -= mean;
    float ss = ((x0.x * x0.x + x0.y * x0.y) + (x0.z * x0.z + x0.w * x0.w)) +
               ((x1.x * x1.x + x1.y * x1.y) + (x1.z * x1.z + x1.w * x1.w));
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
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

__global__ __launch_bounds__(NTHR, 1) void spt_kernel(const SptParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* Q = smem + ROWS * XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int view = blockIdx.x % p.V;
    const int b0 = (blockIdx.x / p.V) * SEQ;
    const mpl_spt_set set = p.sets[(p.flags & MPL_F_MULTI_SPT) ? view : 0];
    const float* pose = p.poses[view];
    const float* ray = p.rays[view];
    const float* cen = p.centers[view];

    // weights of the first Block application: issue the loads before anything else
    BlockFrags F;
    mpl_block_weights bw, bw_next;
    if (p.n_apps > 0) {
        bw = set.blocks[p.sched[0] & 0x7f];
        load_qkv_frags(bw, F, li, kq);
    }

    // ---------------- phase 0: joint embedding (:355-396) ----------------
    for (int idx = tid; idx < ROWS * SD; idx += NTHR) {
        const int r = idx >> 5, c = idx & 31;
        const int sq = r / SJ, j = r - sq * SJ;
        const int b = b0 + sq;
        float x = 0.f;
        if (b < p.B) {
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
        const bool weighted = (p.sched[app] & 0x80) != 0;
        const bool more = app + 1 < p.n_apps;
        if (more) bw_next = set.blocks[p.sched[app + 1] & 0x7f];   // pointers only; used two phases later
        // Prefetch schedule: every phase first touches its own fragments, then issues the loads of the NEXT phase
        // (proj weights during qkv, fc1 during proj, fc2 during fc1, the next application's qkv during fc2), so each
        // group has a whole phase to arrive and at most two groups are live at a time.
#pragma unroll
        for (int n = 0; n < 6; ++n) { touch(F.wq[n][0]); touch(F.wq[n][1]); touch(F.bq[n]); }
        touch(F.g1a); touch(F.g1b); touch(F.e1a); touch(F.e1b);
        load_proj_frags(bw, F, li, kq);

        // ---- QKV = LN1(X) . Wqkv^T + b : 17 x 6 tiles -> Q[:, 0:96]
        {
            const int lo = (MT * 6 * wave) / NWAVE, hi = (MT * 6 * (wave + 1)) / NWAVE;
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

        // ---- attention (:55-64): thread = (sequence, head, group of 4-5 query rows) -- exactly 16 x 8 x 4 = 512 tasks.
        // Every K and V row of the (sequence, head) is read from LDS ONCE per thread and used for all its query rows
        // (the scores of 5 rows x 17 keys stay in registers): 34 ds_read_b128 per thread instead of 34 per (row, head)
        // = 145 per thread.  The four row groups of a (sequence, head) sit in adjacent lanes and read the same
        // addresses (broadcast); a 16-lane ds_read_b128 group touches 4 heads x 16 B: conflict free.
        if (!(p.abl & 1)) {
            const int rgp = tid & 3, h = (tid >> 2) & 7, sq = tid >> 5;
            const int r0 = rgp == 0 ? 0 : 1 + 4 * rgp;          // rows 0..4 | 5..8 | 9..12 | 13..16 of the sequence
            const int nr = rgp == 0 ? 5 : 4;
            float* qb = Q + (sq * SJ) * QS + 4 * h;              // q rows (overwritten with the output), k at +SD, v at +2 SD
            // two passes of up to 3 and 2 rows (the scores of 3 rows x 17 keys fit the register budget next to the
            // prefetched weight fragments): K and V are read twice per thread, 68 reads
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                constexpr int NRW = 3;
                const int rb = r0 + 3 * half, cnt = half == 0 ? 3 : nr - 3;      // 3 | 2 (or 1) rows
                float4 q[NRW];
#pragma unroll
                for (int i = 0; i < NRW; ++i) q[i] = ld4(qb + (rb + (i < cnt ? i : 0)) * QS);
                float sc[NRW][SJ];
#pragma unroll
                for (int j = 0; j < SJ; ++j) {
                    if ((j & 3) == 0) asm volatile("" ::: "memory");   // keep the unrolled K / V loads from being hoisted en bloc
                    const float4 k = ld4(qb + j * QS + SD);
#pragma unroll
                    for (int i = 0; i < NRW; ++i)
                        sc[i][j] = 0.5f * (fmaf(q[i].x, k.x, q[i].y * k.y) + fmaf(q[i].z, k.z, q[i].w * k.w));  // hd^-0.5 = 0.5
                }
                float inv[NRW];
#pragma unroll
                for (int i = 0; i < NRW; ++i) {
                    float mx = sc[i][0];
#pragma unroll
                    for (int j = 1; j < SJ; ++j) mx = fmaxf(mx, sc[i][j]);
                    float l = 0.f;
#pragma unroll
                    for (int j = 0; j < SJ; ++j) {
                        sc[i][j] = __expf(sc[i][j] - mx);
                        l += sc[i][j];
                    }
                    inv[i] = 1.0f / l;
                    if (weighted) {  // attn * conf_weights.unsqueeze(1) after softmax (:61-62): scales query row r
                        const int b = b0 + sq;
                        inv[i] *= (b < p.B && i < cnt) ? pose[((size_t)b * SJ + (rb + i)) * 3 + 2] : 0.f;
                    }
                }
                float4 o[NRW];
#pragma unroll
                for (int i = 0; i < NRW; ++i) o[i] = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < SJ; ++j) {
                    if ((j & 3) == 0) asm volatile("" ::: "memory");
                    const float4 v = ld4(qb + j * QS + 2 * SD);
#pragma unroll
                    for (int i = 0; i < NRW; ++i) {
                        const float pj = sc[i][j] * inv[i];
                        o[i].x = fmaf(pj, v.x, o[i].x);
                        o[i].y = fmaf(pj, v.y, o[i].y);
                        o[i].z = fmaf(pj, v.z, o[i].z);
                        o[i].w = fmaf(pj, v.w, o[i].w);
                    }
                }
#pragma unroll
                for (int i = 0; i < NRW; ++i)
                    if (i < cnt) st4(qb + (rb + i) * QS, o[i]);   // overwrite q (only this thread ever reads these q rows)
            }
        }
        __syncthreads();
        stamp(1);

        // ---- X += attn_out . Wproj^T + b : 17 x 2 tiles
#pragma unroll
        for (int n = 0; n < 2; ++n) { touch(F.wp[n][0]); touch(F.wp[n][1]); touch(F.bp[n]); }
        load_fc1_frags(bw, F, li, kq);
        {
            const int lo = (MT * 2 * wave) / NWAVE, hi = (MT * 2 * (wave + 1)) / NWAVE;
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
#pragma unroll
        for (int n = 0; n < 4; ++n) { touch(F.w1[n][0]); touch(F.w1[n][1]); touch(F.b1[n]); }
        touch(F.g2a); touch(F.g2b); touch(F.e2a); touch(F.e2b);
        load_fc2_frags(bw, F, li, kq);
        {
            const int lo = (MT * 4 * wave) / NWAVE, hi = (MT * 4 * (wave + 1)) / NWAVE;
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
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) touch(F.w2[n][q]);
            touch(F.b2[n]);
        }
        if (more) load_qkv_frags(bw_next, F, li, kq);   // qkv fragments are long dead: next application's weights
        {
            const int lo = (MT * 2 * wave) / NWAVE, hi = (MT * 2 * (wave + 1)) / NWAVE;
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
        __syncthreads();
        stamp(4);
        bw = bw_next;
    }

    if ((p.abl & 16) && lane == 0 && blockIdx.x < 32) {
        float* o = p.xs + (size_t)(blockIdx.x * NWAVE + wave) * 8;
        for (int k = 0; k < 5; ++k) o[k] = (float)ph[k];
        return;
    }
    if (p.abl & 16) return;
    // ---------------- epilogue: Spatial_norm (:412) + per-view glue (:465-491) -> xs[b*V+v][...] ------------
    const bool to_rays = (p.flags & MPL_F_POS3D_TO_RAYS) && (p.flags & MPL_F_RAYS_TOKEN);   // feature concat (:469-471)
    const bool ray_tok = !(p.flags & MPL_F_POS3D_TO_RAYS) && (p.flags & MPL_F_RAYS_TOKEN);  // token concat (:486-489)
    const int cw = to_rays ? 2 * SD : SD;                 // channels per joint in the output row
    const int Df = SJ * SD * ((p.flags & MPL_F_RAYS_TOKEN) ? 2 : 1);
    for (int r = tid; r < ROWS; r += NTHR) {
        const int sq = r / SJ, j = r - sq * SJ;
        const int b = b0 + sq;
        if (b >= p.B) continue;
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

int launch_spt(const mpl_config* cfg, const mpl_weights* w, const mpl_inputs* in, float* xs, hipStream_t s) {
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
    static const int abl = getenv("MPL_SPT_ABL") ? atoi(getenv("MPL_SPT_ABL")) : 0;
    p.abl = abl;
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
        if (hipFuncSetAttribute((const void*)spt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SPT_LDS_BYTES) !=
            hipSuccess)
            return MPL_E_LAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int grid = cfg->num_views * ((in->batch + SEQ - 1) / SEQ);
    ProfScope prof(MPL_K_SPT, s);
    hipLaunchKernelGGL(spt_kernel, dim3(grid), dim3(NTHR), SPT_LDS_BYTES, s, p);
    return hip_check_launch();
}

}  // namespace mpl
