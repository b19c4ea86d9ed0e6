// Does v_fma_mixlo_f16 / v_fma_mixhi_f16 give bit for bit the lo part of the two-term fp16 split, lo = fp16_rne(x - (float)fp16_rne(x))?
// (csrc/h2_phase.hpp split2: 8 v_cvt_f32_f16 + 8 v_sub_f32 + 4 v_cvt_pk_f16_f32 per 8 values -> 8 v_fma_mix*.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mix_probe.hip -o build_tmp/mix_probe && ./build_tmp/mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_ref(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = (_Float16)x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = (_Float16)(x[i] - (float)hi[i]);
}
__device__ __forceinline__ void split_mix(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = (_Float16)x[i];
    const u32x4 h = __builtin_bit_cast(u32x4, hi);
    u32x4 l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned d;
        // d.lo16 = fp16(h.lo16 * -1.0 + x[2p]);  d.hi16 = fp16(h.hi16 * -1.0 + x[2p+1])     (fma in fp32, one rounding to fp16)
        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(d) : "v"(h[p]), "v"(x[2 * p]), "v"(x[2 * p + 1]));
        l[p] = d;
    }
    lo = __builtin_bit_cast(f16x8, l);
}
__global__ void k(const float* x, u32x4* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = x[i * 8 + j];
    f16x8 h0, l0, h1, l1;
    split_ref(v, h0, l0);
    split_mix(v, h1, l1);
    out[i * 4 + 0] = __builtin_bit_cast(u32x4, h0);
    out[i * 4 + 1] = __builtin_bit_cast(u32x4, l0);
    out[i * 4 + 2] = __builtin_bit_cast(u32x4, h1);
    out[i * 4 + 3] = __builtin_bit_cast(u32x4, l1);
}
int main() {
    const int n = 1 << 18;
    std::vector<float> h(n * 8);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t r = (uint32_t)(s >> 16);
        float f;
        switch (i % 7) {
            case 0: { uint32_t b = r; memcpy(&f, &b, 4); if (!(fabsf(f) < 60000.f)) f = 0.5f; break; }        // random bit patterns inside the fp16 window
            case 1: f = ldexpf((float)(r & 0xffffff) / 16777216.0f - 0.5f, (int)(r >> 27) - 20); break;          // 2^-20 .. 2^11, random mantissa
            case 2: f = ldexpf(1.0f + (float)(r & 0x7ff) * (1.0f / 4096.0f), -14 - (int)(r >> 29)); break;      // around the fp16 subnormal boundary
            case 3: f = 65000.f * ((float)(r & 0xffff) / 65536.f - 0.5f) * 2.f; break;                           // up to the clamp
            case 4: f = (float)((int)(r & 0xfff) - 2048) + 0.5f * (float)(r >> 31); break;                       // ties
            case 5: f = ldexpf((float)(r & 0x7fffff) / 8388608.0f, -24 - (int)(r >> 28)); break;                 // below the smallest subnormal
            default: f = (r & 1) ? 0.f : -0.f;
        }
        h[i] = f;
    }
    float* dx; u32x4* dout;
    hipMalloc(&dx, h.size() * 4); hipMalloc(&dout, (size_t)n * 4 * 16);
    hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<uint32_t> o((size_t)n * 16);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int i = 0; i < n; ++i)
        for (int w = 0; w < 8; ++w)
            if (o[(size_t)i * 16 + w] != o[(size_t)i * 16 + 8 + w]) {
                if (bad < 5) printf("mismatch at %d word %d: ref %08x mix %08x (x = %g %g)\n", i, w, o[(size_t)i * 16 + w], o[(size_t)i * 16 + 8 + w],
                                    h[(size_t)i * 8 + 2 * (w & 3)], h[(size_t)i * 8 + 2 * (w & 3) + 1]);
                ++bad;
            }
    printf("mix_probe: %d x 8 values, %zu mismatching words\n", n, bad);
    return bad ? 1 : 0;
}
