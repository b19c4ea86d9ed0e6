// Hardware facts behind the fp16x2 ("h2") split-operand engine, measured on one MI355X:
//   A. does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs (or flush them)?  issue rate vs the bf16 form
//   B. the LDS-DMA ceiling of a CU: bytes per cycle a 512-thread workgroup (one per CU, all 256 CUs at once) can pull
//      from L2 with global_load_lds_dwordx4, by source pattern (one shared region = weights / a private region per
//      workgroup = activations) and by cache policy; the same bytes as plain global_load_dwordx4 into VGPRs
//   C. the same DMA stream with the engine's MFMA + ds_read_b128 pattern running beside it
// hipcc --offload-arch=gfx950 -O3 tools/h2_probe.hip -o build_tmp/h2_probe && build_tmp/h2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- A
__global__ void subnormal_kernel(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    // A[i][k]: row i = lane & 15, k = 8 (lane >> 4) + j.  All ones scaled: a = 2^-20 (subnormal in fp16: min normal 2^-14)
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.0f; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];                 // exact: 32 * 2^-20 * 2^10 = 2^-5 = 0.03125 ; flushed: 0
    f16x8 c;
    for (int j = 0; j < 8; ++j) c[j] = (_Float16)5.9604644775390625e-08f;   // 2^-24: smallest subnormal
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c, b, acc, 0, 0, 0);
    if (lane == 0) out[1] = acc[0];                 // exact: 32 * 2^-24 * 2^10 = 2^-9
    // conversion: does v_cvt_f16_f32 produce subnormals (or flush)?
    volatile float tiny = 3.0e-6f;
    if (lane == 0) out[2] = (float)(_Float16)tiny;  // nearest fp16 subnormal multiple of 2^-24 (5.96e-8): 50 * 2^-24 = 2.98e-6
}

template <int F16, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
    f16x8 a, b;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)(threadIdx.x + i); b[i] = (_Float16)(float)(i + 1); ab[i] = (__bf16)(float)(threadIdx.x + i); bb[i] = (__bf16)(float)(i + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n)
                acc[n] = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- B / C
constexpr int RING = 152 * 1024;      // LDS bytes the DMA cycles through
// MODE 0: LDS-DMA default policy, 1: LDS-DMA sc1, 2: LDS-DMA nt, 3: plain global_load_dwordx4 into VGPRs
// every wave issues `per_wave` 1-KiB pieces per round, `rounds` rounds, at most `depth` rounds in flight (vmcnt)
// MF: MFMAs (16x16x32 f16, 4 accumulators) + ds_read_b128 issued per round by every wave beside the DMA
template <int MODE, int PER_WAVE, int MF, int RD>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, size_t wg_stride, size_t region, int rounds,
                                                   float* out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const char* base = src + (size_t)blockIdx.x * wg_stride;
    const int nw = blockDim.x >> 6;
    const size_t round_bytes = (size_t)nw * PER_WAVE * 1024;
    size_t off = (size_t)wave * PER_WAVE * 1024;
    unsigned slot = (unsigned)(wave * PER_WAVE * 1024);
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(float)(lane + i); fb[i] = (_Float16)(float)(i); }
    u32x4 sink = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int p = 0; p < PER_WAVE; ++p) {
            const char* g = base + off + (size_t)p * 1024 + lane * 16;
            const unsigned dst = lds0 + slot + p * 1024;
            if (MODE == 3) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(g);
                sink[0] ^= v[0]; sink[1] ^= v[1]; sink[2] ^= v[2]; sink[3] ^= v[3];
            } else {
                unsigned keep;
                if (MODE == 0)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
                else if (MODE == 1)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
                else
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
            }
        }
        if (MF) {
            const f16x8* fr = reinterpret_cast<const f16x8*>(smem + ((r & 3) * 32 * 1024)) + lane;
#pragma unroll
            for (int q = 0; q < MF; ++q) {
                if (q < RD) fb = fr[(q * 8 + wave) * 64 % 2048];
                acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[q & 3], 0, 0, 0);
            }
        }
        off += round_bytes;
        if (off + round_bytes > region) off = (size_t)wave * PER_WAVE * 1024;
        slot += (unsigned)round_bytes;
        if (slot + round_bytes > RING) slot = (unsigned)(wave * PER_WAVE * 1024);
        if (MODE != 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER_WAVE > 60 ? 60 : 3 * PER_WAVE) : "memory");   // <= 4 rounds in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + (float)(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]);
}

template <int MODE, int PER_WAVE, int MF, int RD>
void run_dma(const char* name, const char* buf, size_t wg_stride, size_t region, float* out, int threads) {
    const int rounds = 4000, grid = 256;
    hipFuncSetAttribute((const void*)dma_kernel<MODE, PER_WAVE, MF, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, RING + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((dma_kernel<MODE, PER_WAVE, MF, RD>), dim3(grid), dim3(threads), RING + 4096, 0, buf, wg_stride, region, rounds, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * rounds * (threads / 64) * PER_WAVE * 1024;
    const double per_cu = bytes / grid / (ms * 1e-3);
    printf("%-58s %2d waves x %d KiB/round%s: %7.2f TB/s chip, %6.1f GB/s per CU = %5.1f B/cycle @2.4 GHz, %5.1f cycles per KiB",
           name, threads / 64, PER_WAVE, MF ? " + MFMA" : "", bytes / (ms * 1e-3) / 1e12, per_cu / 1e9, per_cu / 2.4e9, 1024.0 / (per_cu / 2.4e9));
    if (MF) printf(" | %d MFMA/wave/round -> %.1f cycles per MFMA per SIMD", MF, ms * 1e-3 * 2.4e9 / ((double)rounds * MF * (threads / 256)));
    printf("\n");
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
    hipMemset(out, 0, 64);
    hipLaunchKernelGGL(subnormal_kernel, dim3(1), dim3(64), 0, 0, out);
    float h[3]; hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
    printf("A. fp16 MFMA inputs 2^-20 (subnormal) x 2^10, 32 terms: %g (exact 0.03125, flushed 0)\n", h[0]);
    printf("   fp16 MFMA inputs 2^-24 (min subnormal) x 2^10, 32 terms: %g (exact 0.001953125)\n", h[1]);
    printf("   (float)(half)3.0e-6f = %g (subnormal kept: 2.98023e-06, flushed: 0)\n", h[2]);
    {
        const int iters = 4000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto rate = [&](auto kern, const char* nm, int wgs) {
            for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256 * wgs), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("   %s, 4 acc, %d wave/SIMD: %.0f TFLOP/s\n", nm, wgs, 256.0 * wgs * 4 * iters * 8 * 4 * 16384 / (ms * 1e-3) / 1e12);
        };
        rate(rate_kernel<1, 4>, "v_mfma_f32_16x16x32_f16 ", 2);
        rate(rate_kernel<0, 4>, "v_mfma_f32_16x16x32_bf16", 2);
        rate(rate_kernel<1, 4>, "v_mfma_f32_16x16x32_f16 ", 4);
    }
    // sources: 8 MiB shared region (weights of a layer: every CU reads the same bytes, L2 resident per XCD) and a private
    // 192 KiB region per workgroup (activations of a row tile: L2 / Infinity-Cache resident, 48 MiB in total)
    const size_t shared = 8u << 20, priv = 192u << 10;
    char* buf; hipMalloc(&buf, 256 * priv + shared);
    hipMemset(buf, 1, 256 * priv + shared);
    printf("B. LDS-DMA ceiling, 256 workgroups (one per CU), <= 4 rounds in flight per wave\n");
    run_dma<0, 4, 0, 0>("shared 8 MiB region, default policy", buf, 0, shared, out, 512);
    run_dma<0, 4, 0, 0>("shared 8 MiB region, default policy", buf, 0, shared, out, 256);
    run_dma<0, 2, 0, 0>("shared 8 MiB region, default policy", buf, 0, shared, out, 512);
    run_dma<0, 8, 0, 0>("shared 8 MiB region, default policy", buf, 0, shared, out, 512);
    run_dma<2, 4, 0, 0>("shared 8 MiB region, nt", buf, 0, shared, out, 512);
    run_dma<1, 4, 0, 0>("shared 8 MiB region, sc1", buf, 0, shared, out, 512);
    run_dma<0, 4, 0, 0>("private 192 KiB per workgroup, default policy", buf + shared, priv, priv, out, 512);
    run_dma<1, 4, 0, 0>("private 192 KiB per workgroup, sc1", buf + shared, priv, priv, out, 512);
    run_dma<3, 4, 0, 0>("shared 8 MiB region, global_load_dwordx4 -> VGPR", buf, 0, shared, out, 512);
    run_dma<3, 4, 0, 0>("private 192 KiB per workgroup, global_load_dwordx4 -> VGPR", buf + shared, priv, priv, out, 512);
    printf("C. the same stream beside the engine's matrix work (per wave and round: MF MFMAs, RD ds_read_b128)\n");
    run_dma<0, 4, 14, 0>("shared, default policy, MFMA only", buf, 0, shared, out, 512);
    run_dma<0, 4, 14, 11>("shared, default policy, MFMA + ds_read", buf, 0, shared, out, 512);
    run_dma<0, 3, 14, 11>("shared, default policy, MFMA + ds_read", buf, 0, shared, out, 512);
    run_dma<0, 2, 14, 11>("shared, default policy, MFMA + ds_read", buf, 0, shared, out, 512);
    run_dma<0, 0, 14, 11>("no DMA, MFMA + ds_read", buf, 0, shared, out, 512);
    return 0;
}
