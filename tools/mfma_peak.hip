// Standalone calibration: sustained fp32 MFMA rate of this MI355X (v_mfma_f32_16x16x4_f32 / 32x32x2),
// registers only.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
double run(F launch, double flops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return flops * 5 / (ms * 1e-3) / 1e12;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {
        int grid = 256 * wg_per_cu;
        double f9 = (double)grid * 4 * iters * 4 * 9 * 2048.0;
        printf("16x16x4 9acc  %d WG/CU: %.1f TF\n", wg_per_cu, run([&] { hipLaunchKernelGGL(k16<9>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); }, f9));
        double f2 = (double)grid * 4 * iters * 4 * 2 * 2048.0;
        printf("16x16x4 2acc  %d WG/CU: %.1f TF\n", wg_per_cu, run([&] { hipLaunchKernelGGL(k16<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); }, f2));
        double f32 = (double)grid * 4 * iters * 4 * 4 * 4096.0;
        printf("32x32x2 4acc  %d WG/CU: %.1f TF\n", wg_per_cu, run([&] { hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); }, f32));
    }
    return 0;
}
