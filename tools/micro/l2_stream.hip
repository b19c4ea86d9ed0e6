// Micro-benchmark: how fast can ONE compute unit stream a weight-like operand out of L2 (the W stream of the block stack)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_stream tools/micro/l2_stream.hip && /tmp/l2_stream
// Every workgroup (one per CU, `waves` waves) walks the same `bytes`-long buffer `reps` times, 1 KiB per wave instruction, as
//   mode 0: global_load_dwordx4 into registers          mode 1: global_load_lds_dwordx4 (LDS-DMA)
//   mode 2: global_load_dword (256 B per instruction)   mode 3: global_load_dwordx4 nt
// with `depth` instructions in flight per wave.  Reported: GB/s per CU and B/clk at the measured shader clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const char* buf, size_t bytes, int reps, unsigned* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    u32x4 acc = {0, 0, 0, 0};
    const size_t step = MODE == 2 ? 256 : 1024;
    const size_t n = bytes / step;                     // wave instructions per pass
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * DEPTH * 1024;
    for (int r = 0; r < reps; ++r) {
        for (size_t i = wave; i + (size_t)(DEPTH - 1) * nw < n; i += (size_t)DEPTH * nw) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const char* p = buf + (i + (size_t)d * nw) * step;
                if (MODE == 0) v[d] = *reinterpret_cast<const u32x4*>(p + lane * 16);
                else if (MODE == 3) v[d] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + lane * 16));
                else if (MODE == 2) { v[d] = u32x4{*reinterpret_cast<const unsigned*>(p + lane * 4), 0, 0, 0}; }
                else {
                    const unsigned long long pu = (unsigned long long)p;
                    const unsigned long long ps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pu >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pu);
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + d * 1024);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane * 16), "s"(ps), "s"(dst) : "memory");
                    v[d] = u32x4{0, 0, 0, 0};
                }
            }
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if (MODE == 1) acc.x ^= *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* buf, size_t bytes, int reps, int waves, int grid, unsigned* out, unsigned long long* clk) {
    hipFuncSetAttribute((const void*)stream_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(grid), dim3(waves * 64), 64 * 1024, 0, buf, bytes, reps, out, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    cyc /= grid; real /= grid;
    const double per_cu = (double)bytes * reps;
    printf("%-34s depth %2d waves %d grid %3d buf %6.1f MiB: %7.3f ms  %6.1f GB/s per CU  %5.1f B per memtime-tick  %5.1f B/clk at 2.4 GHz  (aggregate %.2f TB/s)\n",
           name, DEPTH, waves, grid, bytes / 1048576.0, ms, per_cu / ms / 1e6, per_cu / cyc, per_cu / (real / 100e6 * 2.4e9),
           per_cu * grid / ms / 1e9);
}

int main(int argc, char** argv) {
    const size_t big = 256u << 20;
    char* buf;
    unsigned* out;
    unsigned long long* clk;
    hipMalloc(&buf, big);
    hipMemset(buf, 1, big);
    hipMalloc(&out, 64);
    hipMalloc(&clk, 4096 * 16);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    for (int grid : {cus, cus / 2, 8}) {
        for (size_t mb : {1, 16}) {       // 1 MiB: L2-resident after the first pass; 16 MiB: beyond the 4 MiB L2, inside the 256 MiB MALL
            const size_t bytes = mb << 20;
            const int reps = (int)((size_t)64 << 20) / (int)bytes;
            run<0, 4>("dwordx4 -> registers", buf, bytes, reps, 8, grid, out, clk);
            run<0, 8>("dwordx4 -> registers", buf, bytes, reps, 8, grid, out, clk);
            run<0, 8>("dwordx4 -> registers", buf, bytes, reps, 2, grid, out, clk);
            run<0, 16>("dwordx4 -> registers", buf, bytes, reps, 2, grid, out, clk);
            run<0, 8>("dwordx4 -> registers", buf, bytes, reps, 4, grid, out, clk);
            run<1, 4>("LDS-DMA dwordx4", buf, bytes, reps, 8, grid, out, clk);
            run<1, 8>("LDS-DMA dwordx4", buf, bytes, reps, 8, grid, out, clk);
            run<1, 8>("LDS-DMA dwordx4", buf, bytes, reps, 4, grid, out, clk);
            run<2, 8>("dword -> registers", buf, bytes, reps, 8, grid, out, clk);
            run<3, 8>("dwordx4 nt -> registers", buf, bytes, reps, 8, grid, out, clk);
        }
    }
    return 0;
}
