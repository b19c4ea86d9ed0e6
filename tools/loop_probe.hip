// Bottom-up probe of the GEMM inner structure on one MI355X: how fast does a wave (or 2-3 waves per SIMD) run
//   P0: 36 MFMA blocks with loop-invariant operands (registers only)
//   P1: + 10 ds_read_b128 fragment reads per block, double buffered (LDS resident, no barrier)
//   P2: + one s_barrier per 2 blocks
//   P3: + DMA global_load_lds of 26 KiB per 2 blocks into a 3-stage ring (counted vmcnt)
// hipcc --offload-arch=gfx950 -O3 tools/loop_probe.hip -o /tmp/loop_probe && /tmp/loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

__device__ __forceinline__ void dma16(const float* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

struct Frag { float4 a; float4 b[9]; };
__device__ __forceinline__ void load_frag(Frag& f, const float* as, const float* bs, int cc) {
    f.a = *(const float4*)(as + cc);
#pragma unroll
    for (int n = 0; n < 9; ++n) f.b[n] = *(const float4*)(bs + n * 16 * 32 + cc);
}
__device__ __forceinline__ void block36(const Frag& c, f32x4 (&acc)[9]) {
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[n] = MF(c.a.x, c.b[n].x, acc[n]);
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[n] = MF(c.a.y, c.b[n].y, acc[n]);
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[n] = MF(c.a.z, c.b[n].z, acc[n]);
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[n] = MF(c.a.w, c.b[n].w, acc[n]);
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void probe(const float* g, float* out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4, swz = (li >> 1) & 7;
    constexpr int STAGE = 26624;
    for (int i = tid; i < 3 * STAGE / 4; i += 64 * WAVES) ((float*)smem)[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[9];
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[n] = f32x4{0, 0, 0, 0};
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const float* src = g + (size_t)blockIdx.x * 8192 + wave * 256 + lane * 4;
    const int rg = wave & 3;
    Frag f[2];
    load_frag(f[0], (const float*)smem + (rg * 16 + li) * 32, (const float*)(smem + 8192) + li * 32, (kq ^ swz) << 2);
    if (MODE >= 3) {  // prologue: 2 stages in flight
        for (int st = 0; st < 2; ++st)
            for (int p = wave; p < 26; p += WAVES) dma16(src + p * 16384, lds0 + st * STAGE + p * 1024);
    }
    const int np = (26 - wave + WAVES - 1) / WAVES;
    for (int t = 0; t < iters; ++t) {
        const char* st = smem + (t % 3) * STAGE;
        const float* as = (const float*)st + (rg * 16 + li) * 32;
        const float* bs = (const float*)(st + 8192) + li * 32;
        if (MODE >= 3) {
            if (np == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (np == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (np == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (np == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (MODE >= 2) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        if (MODE == 4 || MODE == 5) {   // P1 / P2 with the fragment reads interleaved into the MFMA stream (1 read : 3 MFMA)
            if (MODE == 5) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
            load_frag(f[1], as, bs, ((4 + kq) ^ swz) << 2);
            block36(f[0], acc);
#pragma unroll
            for (int i = 0; i < 10; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_frag(f[0], as, bs, (kq ^ swz) << 2);
            block36(f[1], acc);
#pragma unroll
            for (int i = 0; i < 10; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (MODE >= 1) {
            if (MODE >= 2) load_frag(f[0], as, bs, (kq ^ swz) << 2);
            load_frag(f[1], as, bs, ((4 + kq) ^ swz) << 2);
            __builtin_amdgcn_sched_barrier(0);
            block36(f[0], acc);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 3) for (int p = wave; p < 26; p += WAVES) dma16(src + p * 16384 + (t & 7) * 64, lds0 + ((t + 2) % 3) * STAGE + p * 1024);
            if (MODE == 1) load_frag(f[0], as, bs, (kq ^ swz) << 2);
            __builtin_amdgcn_sched_barrier(0);
            block36(f[1], acc);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            block36(f[0], acc);
            block36(f[0], acc);
        }
    }
    float s = 0;
#pragma unroll
    for (int n = 0; n < 9; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 64 * WAVES + tid] = s;
}

template <int MODE, int WAVES>
void run(const float* g, float* out, const char* name) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)probe<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 26624);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(grid), dim3(64 * WAVES), 3 * 26624, 0, g, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(grid), dim3(64 * WAVES), 3 * 26624, 0, g, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * WAVES * iters * 72 * 2048.0;
    printf("%-34s waves/WG=%2d: %7.1f TF  (%.0f cycles @2.4GHz per 72-MFMA interval per wave)\n", name, WAVES, fl / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / iters);
}
int main() {
    float *g, *out; hipMalloc(&g, 256 * 8192 * 4 + (1 << 22)); hipMemset(g, 0, 256 * 8192 * 4 + (1 << 22)); hipMalloc(&out, 256 * 1024 * 4);
    run<0, 4>(g, out, "P0 regs only");
    run<1, 4>(g, out, "P1 +ds_read frags");
    run<2, 4>(g, out, "P2 +barrier");
    run<3, 4>(g, out, "P3 +DMA ring");
    run<4, 4>(g, out, "P4 ds_read interleaved");
    run<5, 4>(g, out, "P5 interleaved + barrier (PF)");
    run<4, 8>(g, out, "P4 ds_read interleaved");
    run<5, 8>(g, out, "P5 interleaved + barrier (PF)");
    run<0, 8>(g, out, "P0 regs only");
    run<1, 8>(g, out, "P1 +ds_read frags");
    run<2, 8>(g, out, "P2 +barrier");
    run<3, 8>(g, out, "P3 +DMA ring");
    run<1, 12>(g, out, "P1 +ds_read frags");
    run<2, 12>(g, out, "P2 +barrier");
    run<3, 12>(g, out, "P3 +DMA ring");
    return 0;
}
