// Input preparation on the device: the step right before the path, which the reference runs
// in numpy per sample and per view inside DataLoader workers.
//
// Reference (MPL/lib/dataset/joints_dataset_mpl.py): normalize_screen_coordinates :817-820 -- (X/w)*2 - [1, h/w];
// camera normalisation :615-623; create_3d_ray_coords :872-904 -- world point at unit depth
// R^T [(x-cx)/fx, (y-cy)/fy, 1] + t; cam_center = t :646; model input = [x, y, conf] :772.
// From raw detections (B,V,J,2)+(B,V,J) and one calibration per view it writes exactly the V x (B,J,3) poses,
// V x (B,J,3) rays and V x (B,1,3) centers that mpl_forward consumes (150 B/pose of raw input instead of 1680 B).
// Arithmetic is fp64 and rounded once, like the reference's float64 numpy followed by .float().
#include "common.hpp"

namespace mpl {

struct PrepParams {
    float* poses[MPL_MAX_VIEWS];
    float* rays[MPL_MAX_VIEWS];
    float* centers[MPL_MAX_VIEWS];
    const float* px;
    const float* conf;
    const double* cams;   // device (V,16): fx fy cx cy | R row-major (world->camera) | t (camera centre, world)
    int B, V, J;
    double w, h;
    int norm_in, norm_cam;
};

__global__ __launch_bounds__(256) void prepare_inputs_kernel(const PrepParams p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = p.B * p.V * p.J;
    if (idx >= total) return;
    const int j = idx % p.J, v = (idx / p.J) % p.V, b = idx / (p.J * p.V);
    const double* c = p.cams + v * 16;
    double fx = c[0], fy = c[1], cx = c[2], cy = c[3];
    double x = p.px[(size_t)idx * 2], y = p.px[(size_t)idx * 2 + 1];
    if (p.norm_in) {
        x = (x / p.w) * 2.0 - 1.0;
        y = (y / p.w) * 2.0 - p.h / p.w;
        if (p.norm_cam) {
            cx = (cx / p.w) * 2.0 - 1.0;
            cy = (cy / p.w) * 2.0 - p.h / p.w;
            fx = fx / p.w * 2.0;
            fy = fy / p.w * 2.0;
        }
    }
    const double u0 = (x - cx) / fx, u1 = (y - cy) / fy, u2 = 1.0;
    const size_t o = ((size_t)b * p.J + j) * 3;
    float* po = p.poses[v] + o;
    po[0] = (float)x;
    po[1] = (float)y;
    po[2] = p.conf ? p.conf[idx] : 1.0f;
    float* ro = p.rays[v] + o;
#pragma unroll
    for (int d = 0; d < 3; ++d) ro[d] = (float)(u0 * c[4 + d] + u1 * c[7 + d] + u2 * c[10 + d] + c[13 + d]);   // R^T u + t
    if (j == 0) {
        float* co = p.centers[v] + (size_t)b * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) co[d] = (float)c[13 + d];
    }
}

int launch_prepare_inputs(const float* px, const float* conf, const double* cams_dev, int B, int V, int J, float w, float h,
                          int norm_in, int norm_cam, float* const* poses, float* const* rays, float* const* centers,
                          hipStream_t s) {
    if (!px || !cams_dev || !poses || !rays || !centers || B <= 0 || V <= 0 || V > MPL_MAX_VIEWS || J <= 0 || w <= 0 || h <= 0)
        return MPL_E_INVALID;
    PrepParams p;
    for (int v = 0; v < MPL_MAX_VIEWS; ++v) {
        p.poses[v] = v < V ? poses[v] : nullptr;
        p.rays[v] = v < V ? rays[v] : nullptr;
        p.centers[v] = v < V ? centers[v] : nullptr;
        if (v < V && (!p.poses[v] || !p.rays[v] || !p.centers[v])) return MPL_E_INVALID;
    }
    p.px = px; p.conf = conf; p.cams = cams_dev;
    p.B = B; p.V = V; p.J = J; p.w = w; p.h = h; p.norm_in = norm_in; p.norm_cam = norm_cam;
    const int total = B * V * J;
    ProfScope prof(MPL_K_FUSE_HEAD, s);
    hipLaunchKernelGGL(prepare_inputs_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p);
    return hip_check_launch();
}

}  // namespace mpl
