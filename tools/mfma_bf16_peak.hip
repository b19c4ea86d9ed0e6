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
// the split-operand kernel's MFMA pattern from registers only: NTW accumulators, 3 A parts, NTW x 3 B parts,
// six product passes per k-tile
template <int NTW>
__global__ __launch_bounds__(512) void kpat(float* out, int iters) {
    bf16x8 a[3], b[NTW][3];
    for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) a[p][i] = (__bf16)(float)(threadIdx.x + i + p);
    for (int n = 0; n < NTW; ++n) for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) b[n][p][i] = (__bf16)(float)(threadIdx.x * 3 + i + n + p);
    f32x4 acc[NTW];
    for (int n = 0; n < NTW; ++n) acc[n] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#define PASS(AP, BP) _Pragma("unroll") for (int n = 0; n < NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AP], b[n][BP], acc[n], 0, 0, 0);
        PASS(2, 0) PASS(0, 2) PASS(1, 1) PASS(1, 0) PASS(0, 1) PASS(0, 0)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
    }
    float s = 0;
    for (int n = 0; n < NTW; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
// same pattern, with the kernel's LDS fragment reads (NTW x 3 B parts + 2 A) feeding it every k-tile, no barrier
template <int NTW, int RD>
__global__ __launch_bounds__(512) void kpat_lds(float* out, int iters) {
    __shared__ __attribute__((aligned(1024))) char sm[36864];
    for (int i = threadIdx.x; i < 36864 / 4; i += blockDim.x) ((float*)sm)[i] = 0.001f * (i & 63);
    __syncthreads();
    const bf16x8* bs = (const bf16x8*)sm + (threadIdx.x & 63);
    bf16x8 a[3], b[NTW][3];
    for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) a[p][i] = (__bf16)(float)(threadIdx.x + i + p);
    f32x4 acc[NTW];
    for (int n = 0; n < NTW; ++n) acc[n] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[n][p] = bs[((n * 3 + p) % RD) * 64 + (it & 1) * 64 * 9];
        PASS(2, 0) PASS(0, 2) PASS(1, 1) PASS(1, 0) PASS(0, 1) PASS(0, 0)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
    }
    float s = 0;
    for (int n = 0; n < NTW; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NTW, int RD>
void runpat_lds(float* out, int waves_per_simd) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kpat_lds<NTW, RD>), dim3(grid), dim3(256 * waves_per_simd), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kpat_lds<NTW, RD>), dim3(grid), dim3(256 * waves_per_simd), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("pattern + LDS reads (%d distinct of %d b128 per k-tile) %d tiles, %d wave/SIMD: %.1f cycles per MFMA per SIMD\n", RD < NTW * 3 ? RD : NTW * 3, NTW * 3,
           NTW, waves_per_simd, ms * 1e-3 * 2.4e9 / (iters * 6.0 * NTW * waves_per_simd));
}
template <int NTW>
void runpat(float* out, int waves_per_simd) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kpat<NTW>), dim3(grid), dim3(256 * waves_per_simd), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kpat<NTW>), dim3(grid), dim3(256 * waves_per_simd), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel pattern %d tiles, %d wave/SIMD: %.1f cycles @2.4GHz per MFMA per SIMD\n", NTW, waves_per_simd,
           ms * 1e-3 * 2.4e9 / (iters * 6.0 * NTW * waves_per_simd));
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
    float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
    run<1, 0>(out, 1); run<2, 0>(out, 1); run<3, 0>(out, 1); run<4, 0>(out, 1); run<6, 0>(out, 1); run<9, 0>(out, 1);
    run<1, 0>(out, 2); run<2, 0>(out, 2); run<4, 0>(out, 2); run<2, 0>(out, 4); run<4, 0>(out, 4);
    run<1, 1>(out, 1); run<2, 1>(out, 1); run<4, 1>(out, 1); run<2, 1>(out, 2);
    runpat<5>(out, 1); runpat<4>(out, 1); runpat<5>(out, 2); runpat<4>(out, 2); runpat<2>(out, 2); runpat<3>(out, 2);
    runpat_lds<5, 15>(out, 1); runpat_lds<5, 15>(out, 2); runpat_lds<4, 12>(out, 2); runpat_lds<5, 1>(out, 2);
    return 0;
}
