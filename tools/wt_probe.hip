// Round-6 probe for the review's proposal: the k loop of the fp16x2 stage (A 8 KiB + W 18 KiB per stage, ring of 6, one barrier per
// two stages, double-buffered fragments, waves 0..3 multiply first / waves 4..7 load first: the structure of h2_phase.hpp) with
//   (a) the wave tile as built: ONE row group x {5 | 4} column-tile slots   (12 / 10 ds_read_b128 per 15 / 12 MFMAs: 88 KiB per stage)
//   (b) TWO row groups x {3, 2, 2, 2} slots per row half                    (10 / 8 ds_read_b128 per 18 / 12 MFMAs: 68 KiB per stage),
//       slots dealt so that the two waves of a SIMD hold 3 + 2 or 2 + 2 slots (30 or 24 MFMAs per SIMD and stage, 27 on average)
//   (c) = (b) with every wave also doing the LayerNorm conversion of its TWO row groups' A fragment (the raw-x operand of qkv / fc1:
//       ~35 VALU per row group and k-tile), against (a') = (a) with the conversion of its ONE row group.
// Dummy data (zeros: no DVFS effect, the comparison is cycles), no epilogue, 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/wt_probe.hip -o build_tmp/wt_probe && ./build_tmp/wt_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int STAGE = 26624, NST = 6, A_B = 8192;

__device__ __forceinline__ void conv(f16x8& hi, f16x8& lo, float a, float b) {     // normalise + split 8 values (the finish_a of h2_phase)
    float z[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        z[i] = fmaf((float)hi[i], a, b);
        z[i] = __builtin_amdgcn_fmed3f(z[i], -65000.f, 65000.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = (_Float16)z[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = (_Float16)(z[i] - (float)hi[i]);
}

// RG = row groups per wave (1 | 2), NTW = column-tile slots, WC = W pieces this wave requests per stage, NA = A pieces
template <int RG, int NTW, int WC, int NA, bool MFMA_FIRST, bool CONV>
__device__ __forceinline__ float role(char* smem, const char* g, int rg0, int lane, int slot0, int w_first, int a_first, int iters, float ca, float cb) {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    f32x4 acc[RG][NTW];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 A0[RG][2], A1[RG][2], B0[NTW][2], B1[NTW][2];
    unsigned voW = (unsigned)(lane * 16 + w_first * 1024), voA = (unsigned)(lane * 16 + a_first * 1024);
    const char* srcW = g + (size_t)blockIdx.x % 4 * (64 * 18432);
    const char* srcA = g + (1 << 22) + (size_t)blockIdx.x * 65536;
    auto request = [&](unsigned slot, int t) {
        const char* sw = srcW + (size_t)(t & 63) * 18432;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
        if (WC > 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voW), "s"(sw), "s"(lds0 + slot + A_B + w_first * 1024) : "memory");
        if (WC > 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(voW), "s"(sw) : "memory");
        if (WC > 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(voW), "s"(sw) : "memory");
        if (WC > 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" : : "v"(voW), "s"(sw) : "memory");
        if (NA > 0) {
            const char* sa = srcA + (size_t)(t & 7) * 8192;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voA), "s"(sa), "s"(lds0 + slot + a_first * 1024) : "memory");
            if (NA > 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(voA), "s"(sa) : "memory");
        }
        asm volatile("s_mov_b32 m0, %0" : : "s"(keep));
    };
    auto reads = [&](unsigned slot, f16x8 (&a)[RG][2], f16x8 (&b)[NTW][2]) {
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const f16x8* as = reinterpret_cast<const f16x8*>(smem + slot + (rg0 + r) * 2048) + lane;
            a[r][0] = as[0]; a[r][1] = as[64];
        }
        const f16x8* bs = reinterpret_cast<const f16x8*>(smem + slot + A_B) + slot0 * 128 + lane;
#pragma unroll
        for (int n = 0; n < NTW; ++n) { b[n][0] = bs[n * 128]; b[n][1] = bs[n * 128 + 64]; }
    };
    auto rows = [&](const f16x8 (&a)[RG][2], const f16x8 (&b)[NTW][2]) {
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n][0], a[r][1], acc[r][n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n][1], a[r][0], acc[r][n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n][0], a[r][0], acc[r][n], 0, 0, 0);
    };
    constexpr int PW = WC + NA;
    for (int t = 0; t < NST - 1; ++t) request(t * STAGE, t);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
    __builtin_amdgcn_s_barrier();
    reads(0, A0, B0);
    unsigned slot = 0;
    auto stage = [&](int t, bool sync, const f16x8 (&ac)[RG][2], f16x8 (&an)[RG][2], const f16x8 (&bc)[NTW][2], f16x8 (&bn)[NTW][2]) {
        const unsigned sn = slot + STAGE == NST * STAGE ? 0u : slot + STAGE;
        const unsigned sp = slot == 0 ? (unsigned)((NST - 1) * STAGE) : slot - STAGE;
        if (sync) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!MFMA_FIRST) { reads(sn, an, bn); request(sp, t + NST - 1); __builtin_amdgcn_sched_barrier(0); }
        rows(ac, bc);
        __builtin_amdgcn_sched_barrier(0);
        if (MFMA_FIRST) { reads(sn, an, bn); request(sp, t + NST - 1); }
        if (CONV) {
#pragma unroll
            for (int r = 0; r < RG; ++r) conv(an[r][0], an[r][1], ca, cb);
        }
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
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int n = 0; n < NTW; ++n) s += acc[r][n][0] + acc[r][n][3];
    return s;
}

template <int VAR, bool CONV>
__global__ __launch_bounds__(512, 1) void probe(const char* g, float* out, int iters, float ca, float cb) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NST * STAGE / 4; i += 512) ((float*)smem)[i] = 0.f;
    __syncthreads();
    float s;
    if (VAR == 0) {         // as built: waves 0..3 five slots + 2 A + 2 W pieces, waves 4, 5 four slots + 3 W, waves 6, 7 four slots + 2 W
        if (wave < 4) s = role<1, 5, 2, 2, true, CONV>(smem, g, wave & 3, lane, 0, 10 + 2 * wave, 2 * wave, iters, ca, cb);
        else if (wave < 6) s = role<1, 4, 3, 0, false, CONV>(smem, g, wave & 3, lane, 5, 3 * (wave - 4), 0, iters, ca, cb);
        else s = role<1, 4, 2, 0, false, CONV>(smem, g, wave & 3, lane, 5, 6 + 2 * (wave - 6), 0, iters, ca, cb);
    } else {                // two row groups per wave.  Row half h = wave >> 2 (row groups 2h, 2h + 1), slots per wave {3,2,2,2} in half 0 and
                            // {2,3,2,2} in half 1: waves w and w + 4 share a SIMD -> 5, 5, 4, 4 slots per SIMD.  26 pieces: waves 0..3 two A
                            // + two / one W, waves 4..7 three / four W (3 + 3 + 3 + 3 + 4 + 3 + 4 + 3 = 26)
        const int h = wave >> 2, q = wave & 3;
        const int rg0 = 2 * h;
        const int big = h == 0 ? 0 : 1;                 // which q holds three slots
        const int s0 = q == 0 ? 0 : (big == 0 ? 3 + 2 * (q - 1) : (q == 1 ? 2 : (q == 2 ? 5 : 7)));
        if (wave < 4) {
            const int wf = wave;                        // W pieces 0..3 (one each), A pieces 2 wave, 2 wave + 1
            if (q == big) s = role<2, 3, 1, 2, true, CONV>(smem, g, rg0, lane, s0, wf, 2 * wave, iters, ca, cb);
            else s = role<2, 2, 1, 2, true, CONV>(smem, g, rg0, lane, s0, wf, 2 * wave, iters, ca, cb);
        } else {
            const int wf = 4 + (wave == 4 ? 0 : (wave == 5 ? 4 : (wave == 6 ? 7 : 11)));      // 4, 3, 4, 3 W pieces -> 4..17
            if (wave == 5) s = role<2, 3, 3, 0, false, CONV>(smem, g, rg0, lane, s0, wf, 0, iters, ca, cb);          // q == big: three slots
            else if (wave == 7) s = role<2, 2, 3, 0, false, CONV>(smem, g, rg0, lane, s0, wf, 0, iters, ca, cb);
            else s = role<2, 2, 4, 0, false, CONV>(smem, g, rg0, lane, s0, wf, 0, iters, ca, cb);                    // waves 4, 6: four W pieces
        }
    }
    out[blockIdx.x * 512 + tid] = s;
}

template <int VAR, bool CONV>
void run(const char* g, float* out, const char* name, int grid) {
    const int iters = 3000;
    hipFuncSetAttribute((const void*)probe<VAR, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<VAR, CONV>), dim3(grid), dim3(512), NST * STAGE, 0, g, out, iters, 1.0f, 0.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %3d workgroups: %7.1f ns per stage\n", name, grid, ms * 1e6 / iters);
}
int main() {
    char* g; float* out;
    hipMalloc(&g, (1 << 22) + 256 * 65536 + (1 << 20)); hipMemset(g, 0, (1 << 22) + 256 * 65536 + (1 << 20));
    hipMalloc(&out, 256 * 512 * 4);
    for (int grid : {256, 32})
        for (int r = 0; r < 2; ++r) {
            run<0, false>(g, out, "(a)  1 row group x {5|4} slots (as built), packed A", grid);
            run<1, false>(g, out, "(b)  2 row groups x {3,2,2,2} slots, packed A", grid);
            run<0, true>(g, out, "(a') as built + LayerNorm conversion of 1 row group per wave and stage", grid);
            run<1, true>(g, out, "(c)  2 row groups + LayerNorm conversion of 2 row groups per wave and stage", grid);
        }
    return 0;
}
