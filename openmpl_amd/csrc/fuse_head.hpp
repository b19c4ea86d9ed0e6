// The tail of the forward as device functions: strip ray features, View_norm, learned weighted mean over views, head LayerNorm +
// Linear (reference multiview_mpl.py :425-434, :439, :445, :521-523), used by fuse_head_kernel (fuse_head.hip).
// They are separate functions because round 6 also ran them INSIDE the persistent block-stack launch (the team that finished a
// row tile lifted that tile's poses itself: one launch fewer per forward).  Built, bitwise the separate launch on ten shapes, and
// measured in one process (profiles/r06_tail_ab.txt): headline 1.224-1.241 ms separate against 1.238-1.244 fused, depth 2
// 0.338 against 0.336, bf16 V = 8 0.451-0.474 against 0.468 -- nothing: the teams of a launch finish together, so the in-kernel
// tail (every workgroup stages the 111-kB head weight, two dependent round trips per pose) sits on the critical path where the
// 21-us launch sat, and inlined into the stack kernels it cost them 43-112 spilled VGPRs.  Removed again; the split stays.
#pragma once
#include "gemm_common.hpp"

namespace mpl {

constexpr int kMaxE = 1024;                   // J*d upper bound held in LDS
constexpr int FH_W_FLOATS = 28 * 1024;        // LDS floats reserved for the head weight (3J * J*d = 27 744 at J = 17, d = 32)
constexpr int FH_NF = 9;                      // features per lane: E <= 64 * 9 (J*d = 544), checked by the launchers

struct FhParams {
    const float *vn_w, *vn_b;                 // View_norm
    const float *wm_w, *wm_b;                 // weighted_mean Conv1d (V -> 1)
    const float *hl_w, *hl_b;                 // head[0] LayerNorm
    const float *hw, *hb;                     // head[1] Linear (n_out, E)
    int V, Df, E, d, strip_mode, n_out;       // strip_mode: 0 none, 1 feature concat [J][2d], 2 token concat
};

// one float of a row another workgroup may have written during this launch: past the L1 (sc1), as every hand-off read is
template <bool L2>
__device__ __forceinline__ float fh_ld(const float* p) {
    if constexpr (L2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// request the whole head weight (111 kB at E = 544, n_out = 51) into LDS at `hws`: 1-KiB pieces, round-robin over n_waves waves
// (the last piece would read past the weight: clamped lanes re-read the tail).  Lands behind an s_waitcnt vmcnt(0) + barrier.
__device__ __forceinline__ void fh_stage_weight(const FhParams& p, float* hws, int wave, int lane, int n_waves) {
    const int n_w = p.n_out * p.E;                           // floats
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)hws;
    for (int pc = wave; pc * 256 < n_w; pc += n_waves) {
        int off = pc * 256 + lane * 4;
        off = off + 4 <= n_w ? off : n_w - 4;                // n_w is a multiple of 4
        dma16(p.hw + off, lds0 + (unsigned)(pc * 1024));
    }
}

// Everything that is per pose, by ONE wave: weighted mean over views of the View_norm-ed rows (:439, :445; per view two-pass
// statistics in registers, four views at a time with all their loads in flight together), then either the fused (E) feature to
// y_out_row (non-default heads) or the head LayerNorm (eps 1e-5) into the LDS row y_row.  xb = the V rows of this pose.
template <bool L2>
__device__ __forceinline__ void fh_pose(const FhParams& p, const float* xb, int lane, float* y_row, float* y_out_row, bool poisoned) {
    constexpr int NF = FH_NF, VC = 4;
    const int E = p.E, V = p.V, d = p.d;
    auto src = [&](int f) { return p.strip_mode == 1 ? (f / d) * 2 * d + (f % d) : f; };
    float acc[NF], gam[NF], bet[NF], hlw[NF], hlb[NF];
    // EVERY parameter this wave needs is requested up front, together with the first rows: one memory round trip for the
    // whole per-pose part
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int f = lane + 64 * i;
        acc[i] = 0.f;
        gam[i] = f < E ? p.vn_w[f] : 0.f;
        bet[i] = f < E ? p.vn_b[f] : 0.f;
        hlw[i] = (!y_out_row && f < E) ? p.hl_w[f] : 0.f;
        hlb[i] = (!y_out_row && f < E) ? p.hl_b[f] : 0.f;
    }
    const float wv_l = lane < V ? p.wm_w[lane] : 0.f;        // view weight v in lane v (V <= 32), broadcast by readlane below
    const float wb = p.wm_b[0];
    for (int v0 = 0; v0 < V; v0 += VC) {
        float xv[VC][NF];
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            const float* xr = xb + (size_t)(v0 + u < V ? v0 + u : V - 1) * p.Df;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = lane + 64 * i;
                xv[u][i] = f < E ? fh_ld<L2>(xr + src(f)) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            if (v0 + u >= V) break;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NF; ++i) s += xv[u][i];
            const float mean = wave_sum(s) / (float)E;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const float t = (lane + 64 * i < E) ? xv[u][i] - mean : 0.f;
                ss += t * t;
            }
            const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)E + 1e-6f);
            const float wv = __shfl(wv_l, v0 + u, 64);
#pragma unroll
            for (int i = 0; i < NF; ++i) acc[i] = fmaf(wv, (xv[u][i] - mean) * rstd * gam[i] + bet[i], acc[i]);
        }
    }
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int f = lane + 64 * i;
        if (f < E) {
            acc[i] += wb;
            part += acc[i];
        }
    }
    if (y_out_row) {  // caller wants the fused (E) feature only (non-default heads): stop before head[0]
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = lane + 64 * i;
            if (f < E) y_out_row[f] = poisoned ? __builtin_nanf("") : acc[i];
        }
        return;
    }
    // head LayerNorm (eps 1e-5), two-pass in registers
    const float mean = wave_sum(part) / (float)E;
    float p2 = 0.f;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int f = lane + 64 * i;
        if (f < E) {
            const float t = acc[i] - mean;
            p2 += t * t;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(p2) / (float)E + 1e-5f);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int f = lane + 64 * i;
        if (f < E) y_row[f] = (acc[i] - mean) * rstd * hlw[i] + hlb[i];
    }
}

// Linear(E -> n_out) for np poses out of LDS (y[p] = the head-LayerNorm'ed feature of pose slot p, hws = the staged weight):
// thread = (pose slot, output), a complete dot product without any cross-lane reduction, read as float4.  Thread o walks its
// weight row rotated by 4 o floats (f = (i + 4 o) mod E): the 16 lanes a ds_read_b128 serves together then sit on 16 different
// 16-byte bank groups.  pose_of(p) = index of pose slot p in `out`.
template <typename PoseOf>
__device__ __forceinline__ void fh_linear(const FhParams& p, const float* hws, const float (*y)[kMaxE], int np, int tid, int n_threads,
                                          float* out, bool poisoned, PoseOf pose_of) {
    const int E = p.E, n_out = p.n_out;
    for (int t = tid; t < np * n_out; t += n_threads) {
        const int ps = t / n_out, o = t - ps * n_out;
        const float* wr = hws + o * E;
        const float* yr = y[ps];
        float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
        int f = (4 * o) % E;
        if ((E & 3) == 0) {
#pragma unroll 4
            for (int i = 0; i < E; i += 4) {
                const float4 a4 = ld4(yr + f), w4 = ld4(wr + f);
                sx = fmaf(a4.x, w4.x, sx);
                sy = fmaf(a4.y, w4.y, sy);
                sz = fmaf(a4.z, w4.z, sz);
                sw = fmaf(a4.w, w4.w, sw);
                f = f + 4 == E ? 0 : f + 4;
            }
        } else {
            for (int i = 0; i < E; ++i) {
                sx = fmaf(yr[f], wr[f], sx);
                f = f + 1 == E ? 0 : f + 1;
            }
        }
        out[(size_t)pose_of(ps) * n_out + o] = poisoned ? __builtin_nanf("") : ((sx + sy) + (sz + sw)) + p.hb[o];
    }
}

// host side: the parameters of the default tail from the boundary structs; false = this shape has no fused / LDS form
inline bool fh_params(const mpl_config* cfg, const mpl_weights* w, FhParams* p) {
    const int J = cfg->num_joints, d = cfg->dim, V = cfg->num_views, E = J * d;
    if (E > kMaxE || E > 64 * FH_NF || V > MPL_MAX_VIEWS || 3 * J * E > FH_W_FLOATS || (E & 1)) return false;
    int strip = 0;
    if (cfg->flags & MPL_F_POS3D_TO_RAYS) strip = 1;           // :430-434 (takes precedence, elif order)
    else if (cfg->flags & MPL_F_RAYS_TOKEN) strip = 2;         // :425-429
    *p = FhParams{w->view_norm_w, w->view_norm_b, w->wmean_w, w->wmean_b, w->head_ln_w, w->head_ln_b, w->head_w, w->head_b,
                  V, mpl_fpt_width(cfg), E, d, strip, 3 * J};
    return true;
}

}  // namespace mpl
