// How long is a dependent v_mfma_f32_16x16x4_f32 on an otherwise idle MI355X, in s_memtime ticks and in 100-MHz ticks?
// (the single-frame engine, sm_stack.hip, runs ~36 of them per wave and step behind one ds_read_b128 each)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma4_chain.hip -o build_tmp/mfma4_chain && build_tmp/mfma4_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long* out, float* sink, int mode) {
    __shared__ float4 lds[64 * 17];
    for (int i = threadIdx.x; i < 64 * 17; i += blockDim.x) lds[i] = float4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {
#pragma unroll
        for (int i = 0; i < 256; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {      // the loop of sm_tile: one fragment read, four dependent MFMAs
            const float4 w = lds[(i % 17) * 64 + (threadIdx.x & 63)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.w, acc, 0, 0, 0);
        }
    }
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
int main() {
    unsigned long long *o, h[2];
    float* s;
    hipMalloc(&o, 16); hipMalloc(&s, 4096);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s, mode);
            hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
            printf("mode %d: %llu s_memtime ticks, %llu x 10 ns: %s -> %.1f ticks, %.1f ns each\n", mode, h[0], h[1], mode == 0 ? "256 dependent MFMAs" : "16 x (ds_read_b128 + 4 MFMAs)",
                   h[0] / (mode == 0 ? 256.0 : 64.0), h[1] * 10.0 / (mode == 0 ? 256.0 : 64.0));
        }
    return 0;
}
