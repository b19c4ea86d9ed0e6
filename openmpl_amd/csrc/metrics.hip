// Output-side epilogue on the device: what the reference does on the host after every
// forward -- `output.clone().cpu().numpy()`, room de-normalisation, MPJPE-family reductions.
//
// Reference (MPL/lib/core/): function_mpl.py:476-488 (de-normalisation), evaluate.py:91-114 calc_mpjpe (absolute and
// root-relative, np.nansum inside the squared sum), evaluate.py:117-125 calc_distance_per_dim (np.nanmean),
// loss.py:39-57 MPJPE and :110-124 Weighted_MPJPE (mean Euclidean error + per-axis mean |error| on the raw tensors).
//
// One workgroup, thread = (joint j, batch slice s): every thread walks poses s, s+NS, ... accumulating in fp64, slices
// are folded in a fixed order through LDS (deterministic; B*J*3 floats is a few hundred kB at most).
#include "common.hpp"

namespace mpl {

constexpr int MJ = 17;  // joints handled per thread row (J <= 17)
constexpr int MS = 15;  // batch slices: 17 * 15 = 255 threads
constexpr int MACC = 12;

__global__ __launch_bounds__(256) void pose_metrics_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                            const float* __restrict__ wgt, int B, int J, float sx, float sy,
                                                            float sz, float ox, float oy, float oz, unsigned skip_mask,
                                                            const unsigned* __restrict__ dev_err, float* __restrict__ res) {
    __shared__ double acc[MS][MJ][MACC];
    const int tid = threadIdx.x;
    // A forward that lost a hand-off writes NaN poses, and nansum / nanmean semantics would score those as ZERO error: while
    // the device's error word is set (the failing forward ran before this kernel on the stream) every result is NaN instead.
    if (dev_err && __hip_atomic_load(dev_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
        const int n = 4 + 2 * (J + 1) + 3 * J + 3;
        for (int i = tid; i < n; i += 256) res[i] = __builtin_nanf("");
        return;
    }
    const int j = tid % MJ, s = tid / MJ;
    const float sc[3] = {sx, sy, sz}, of[3] = {ox, oy, oz};
    double a[MACC] = {};   // 0 loss, 1-3 |e| raw, 4 pjpe_abs, 5 pjpe_rel, 6-8 dist sum, 9-11 dist count
    if (j < J && s < MS) {
        for (int b = s; b < B; b += MS) {
            const float* o = out + ((size_t)b * J + j) * 3;
            const float* t = tgt + ((size_t)b * J + j) * 3;
            const float* o0 = out + (size_t)b * J * 3;
            const float* t0 = tgt + (size_t)b * J * 3;
            float n2 = 0.f, abs2 = 0.f, rel2 = 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float e = o[d] - t[d];                        // loss.py works on the raw tensors
                n2 += e * e;
                a[1 + d] += fabsf(e);
                const float od = o[d] * sc[d] + of[d], td = t[d] * sc[d] + of[d];      // function_mpl.py:480-488
                const float ed = od - td;
                if (!isnan(ed)) {                                   // np.nansum / np.nanmean skip NaN terms
                    abs2 += ed * ed;
                    a[6 + d] += fabsf(ed);
                    a[9 + d] += 1.0;
                }
                const float er = (od - (o0[d] * sc[d] + of[d])) - (td - (t0[d] * sc[d] + of[d]));   // evaluate.py:105-108
                if (!isnan(er)) rel2 += er * er;
            }
            const float nrm = sqrtf(n2);
            a[0] += wgt ? (double)(wgt[(size_t)b * J + j] * nrm) : (double)nrm;   // loss.py:57 / :124
            a[4] += sqrtf(abs2);
            a[5] += sqrtf(rel2);
        }
    }
    if (s < MS)
#pragma unroll
        for (int k = 0; k < MACC; ++k) acc[s][j][k] = a[k];
    __syncthreads();
    // result layout: [0] loss, [1..3] loss per axis, [4..4+J) pjpe_abs, [4+J] mpjpe_abs, then pjpe_rel, mpjpe_rel,
    // dist (J x 3), dist_mean (3)
    if (tid < J) {
        double t[MACC] = {};
        for (int q = 0; q < MS; ++q)
#pragma unroll
            for (int k = 0; k < MACC; ++k) t[k] += acc[q][tid][k];
#pragma unroll
        for (int k = 0; k < MACC; ++k) acc[0][tid][k] = t[k];
        res[4 + tid] = (float)(t[4] / B);
        res[4 + (J + 1) + tid] = (float)(t[5] / B);
#pragma unroll
        for (int d = 0; d < 3; ++d) res[4 + 2 * (J + 1) + tid * 3 + d] = (float)(t[6 + d] / t[9 + d]);
    }
    __syncthreads();
    if (tid == 0) {
        double loss = 0, ax[3] = {0, 0, 0}, ma = 0, mr = 0, dm[3] = {0, 0, 0};
        int kept = 0;          // evaluate.py:101-104, :110-113: joints in not_consider_kp are deleted from the MEAN only
        for (int q = 0; q < J; ++q) {
            loss += acc[0][q][0];
            if (!((skip_mask >> q) & 1u)) {
                ma += acc[0][q][4] / B;
                mr += acc[0][q][5] / B;
                ++kept;
            }
            for (int d = 0; d < 3; ++d) {
                ax[d] += acc[0][q][1 + d];
                dm[d] += acc[0][q][6 + d] / acc[0][q][9 + d];
            }
        }
        res[0] = (float)(loss / ((double)B * J));
        for (int d = 0; d < 3; ++d) {
            res[1 + d] = (float)(ax[d] / ((double)B * J));
            res[4 + 2 * (J + 1) + 3 * J + d] = (float)(dm[d] / J);
        }
        res[4 + J] = (float)(ma / kept);              // kept == 0: NaN, like numpy's mean of an empty array
        res[4 + (J + 1) + J] = (float)(mr / kept);
    }
}

int launch_pose_metrics(const float* out, const float* tgt, const float* wgt, int B, int J, const float* scale3,
                        const float* offset3, unsigned skip_mask, float* res, hipStream_t s) {
    if (!out || !tgt || !res || B <= 0 || J <= 0 || J > MJ) return MPL_E_INVALID;
    const float sx = scale3 ? scale3[0] : 1.f, sy = scale3 ? scale3[1] : 1.f, sz = scale3 ? scale3[2] : 1.f;
    const float ox = offset3 ? offset3[0] : 0.f, oy = offset3 ? offset3[1] : 0.f, oz = offset3 ? offset3[2] : 0.f;
    int dev = 0;
    const unsigned* dev_err = hipGetDevice(&dev) == hipSuccess ? device_error_word(dev) : nullptr;
    ProfScope prof(MPL_K_FUSE_HEAD, s);
    hipLaunchKernelGGL(pose_metrics_kernel, dim3(1), dim3(256), 0, s, out, tgt, wgt, B, J, sx, sy, sz, ox, oy, oz, skip_mask, dev_err, res);
    return hip_check_launch();
}

}  // namespace mpl
