// Issue rate of v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 on one MI355X by number of independent
// accumulators per wave and waves per SIMD (registers only).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_peak.hip -o build_tmp/mfma_bf16_peak && build_tmp/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int BIG>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    float s = 0;
    if (BIG) {
        f32x16 acc[NACC];
        for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) acc[n][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
        }
        for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
    } else {
        f32x4 acc[NACC];
        for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
        }
        for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int BIG>
void run(float* out, int wgs_per_cu) {
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)grid * 4 * iters * 8 * NACC;
    const double fl = n_mfma * (BIG ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2);
    printf("%s %d acc, %d wave/SIMD: %7.1f TF, %.1f cycles @2.4GHz per MFMA per SIMD\n", BIG ? "32x32x16" : "16x16x32", NACC, wgs_per_cu,
           fl / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 8.0 * NACC * wgs_per_cu));
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<1, 0>(out, 1); run<2, 0>(out, 1); run<3, 0>(out, 1); run<4, 0>(out, 1); run<6, 0>(out, 1); run<9, 0>(out, 1);
    run<1, 0>(out, 2); run<2, 0>(out, 2); run<4, 0>(out, 2); run<2, 0>(out, 4); run<4, 0>(out, 4);
    run<1, 1>(out, 1); run<2, 1>(out, 1); run<4, 1>(out, 1); run<2, 1>(out, 2);
    return 0;
}
