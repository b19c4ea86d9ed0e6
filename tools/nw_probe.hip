// De-risking probe: the k loop of the fp16x2 stage (h2_gemm.hip, one row tile, one pass: A 8 KiB + W 18 KiB per stage, ring of 6,
// one barrier per two stages, double-buffered fragments) with 8 waves per workgroup as built (wave = row group x {5 | 4} column
// tiles, 2 waves per SIMD, 256 registers) against 12 waves (wave = row group x 3 column tiles, 3 waves per SIMD, 168 registers):
// does a third wave per SIMD overlap the MFMA rows of one wave with the fragment reads / DMA requests of the others?
// Dummy data, no epilogue, 256 workgroups.   hipcc --offload-arch=gfx950 -O3 tools/nw_probe.hip -o build_tmp/nw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int STAGE = 26624, NST = 6, A_B = 8192;

template <int NTW, int WC, int NA, bool MFMA_FIRST>
__device__ __forceinline__ float role(char* smem, const char* g, int wave, int lane, int slot0, int w_first, int iters) {
    const int rg = wave & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    f32x4 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 A0[2], A1[2], B0[NTW][2], B1[NTW][2];
    unsigned voW = (unsigned)(lane * 16 + w_first * 1024), voA = (unsigned)(lane * 16);
    const char* srcW = g + (size_t)blockIdx.x % 4 * (64 * 18432);
    const char* srcA = g + (1 << 22) + (size_t)blockIdx.x * 65536 + rg * 2048;
    auto request = [&](unsigned slot, int t) {
        const char* sw = srcW + (size_t)(t & 63) * 18432;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : "=&s"(keep) : "v"(voW), "s"(sw), "s"(lds0 + slot + A_B + w_first * 1024) : "memory");
        if (WC > 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(voW), "s"(sw) : "memory");
        if (WC > 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(voW), "s"(sw) : "memory");
        if (NA) {
            const char* sa = srcA + (size_t)(t & 7) * 8192;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(voA), "s"(sa), "s"(lds0 + slot + rg * 2048) : "memory");
        }
        asm volatile("s_mov_b32 m0, %0" : : "s"(keep));
    };
    auto reads = [&](unsigned slot, f16x8 (&a)[2], f16x8 (&b)[NTW][2]) {
        const f16x8* as = reinterpret_cast<const f16x8*>(smem + slot + rg * 2048) + lane;
        a[0] = as[0]; a[1] = as[64];
        const f16x8* bs = reinterpret_cast<const f16x8*>(smem + slot + A_B) + slot0 * 128 + lane;
#pragma unroll
        for (int n = 0; n < NTW; ++n) { b[n][0] = bs[n * 128]; b[n][1] = bs[n * 128 + 64]; }
    };
    auto rows = [&](const f16x8 (&a)[2], const f16x8 (&b)[NTW][2]) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n][0], a[1], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n][1], a[0], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n][0], a[0], acc[n], 0, 0, 0);
    };
    for (int t = 0; t < NST - 1; ++t) request(t * STAGE, t);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WC + NA)) : "memory");
    __builtin_amdgcn_s_barrier();
    reads(0, A0, B0);
    unsigned slot = 0;
    auto stage = [&](int t, bool sync, const f16x8 (&ac)[2], f16x8 (&an)[2], const f16x8 (&bc)[NTW][2], f16x8 (&bn)[NTW][2]) {
        const unsigned sn = slot + STAGE == NST * STAGE ? 0u : slot + STAGE;
        const unsigned sp = slot == 0 ? (unsigned)((NST - 1) * STAGE) : slot - STAGE;
        if (sync) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WC + NA)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!MFMA_FIRST) { reads(sn, an, bn); request(sp, t + NST - 1); __builtin_amdgcn_sched_barrier(0); }
        rows(ac, bc);
        __builtin_amdgcn_sched_barrier(0);
        if (MFMA_FIRST) { reads(sn, an, bn); request(sp, t + NST - 1); }
        __builtin_amdgcn_sched_barrier(0);
        slot = sn;
    };
    for (int t = 0; t < iters; t += 2) {
        stage(t, true, A0, A1, B0, B1);
        stage(t + 1, false, A1, A0, B1, B0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < NTW; ++n) s += acc[n][0] + acc[n][3];
    return s;
}

template <int NW, int VAR>
__global__ __launch_bounds__(64 * NW, 1) void probe(const char* g, float* out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NST * STAGE / 4; i += 64 * NW) ((float*)smem)[i] = 0.f;
    __syncthreads();
    float s;
    if (NW == 8) {          // as built: waves 0..3 five tiles + 2 A + 2 W pieces, waves 4, 5 four tiles + 3 W, waves 6, 7 four tiles + 2 W
        if (wave < 4) s = role<5, 2, 2, true>(smem, g, wave, lane, 0, 10 + 2 * wave, iters);
        else if (wave < 6) s = role<4, 3, 0, false>(smem, g, wave, lane, 5, 3 * (wave - 4), iters);
        else s = role<4, 2, 0, false>(smem, g, wave, lane, 5, 6 + 2 * (wave - 6), iters);
    } else {                // 12 waves: three tiles each; waves 0..3: 2 A + 1 W piece, the other eight 2, 2, 2, 1, 2, 2, 2, 1 W pieces
        const int third = wave >> 2;
        const int wf = wave < 4 ? 14 + wave : (wave < 8 ? 2 * (wave - 4) : 7 + 2 * (wave - 8));
        if (wave < 4) s = role<3, 1, 2, true>(smem, g, wave, lane, 0, wf, iters);
        else if ((wave & 3) == 3) s = role<3, 1, 0, (VAR & 1) != 0>(smem, g, wave, lane, 3 * third, wf, iters);
        else if (wave < 8) s = role<3, 2, 0, (VAR & 1) != 0>(smem, g, wave, lane, 3, wf, iters);
        else s = role<3, 2, 0, false>(smem, g, wave, lane, 6, wf, iters);
    }
    out[blockIdx.x * 64 * NW + tid] = s;
}

template <int NW, int VAR>
void run(const char* g, float* out, const char* name) {
    const int iters = 3000, grid = 256;
    hipFuncSetAttribute((const void*)probe<NW, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NW, VAR>), dim3(grid), dim3(64 * NW), NST * STAGE, 0, g, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %2d waves: %.3f ms for %d stages = %.0f ns per stage (%.0f cycles at 2.0 GHz); MFMA pipe time 432 cycles\n", name, NW, ms, iters,
           ms * 1e6 / iters, ms * 1e6 / iters * 2.0);
}
int main() {
    char* g; float* out;
    hipMalloc(&g, (1 << 22) + 256 * 65536 + (1 << 20)); hipMemset(g, 0, (1 << 22) + 256 * 65536 + (1 << 20));
    hipMalloc(&out, 256 * 1024 * 4);
    for (int r = 0; r < 2; ++r) {
        run<8, 0>(g, out, "8 waves (as built)");
        run<12, 0>(g, out, "12 waves, thirds 1, 2 load first");
        run<12, 1>(g, out, "12 waves, third 1 multiplies first");
    }
    return 0;
}
