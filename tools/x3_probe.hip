// Feasibility probe for fp32-equivalent GEMM on the bf16 matrix cores (3-way bf16 operand split, 6 products):
// inner loop of a 64 x 136 x K tile per workgroup, A fp32 in LDS (split in registers), W pre-split into
// 3 bf16 parts in fragment order, DMA-staged 2-stage ring.  Reports fp32-algorithmic TFLOP/s.
// hipcc --offload-arch=gfx950 -O3 tools/x3_probe.hip -o build_tmp/x3_probe && build_tmp/x3_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ void dma16(const void* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
// 8 fp32 -> three bf16x8 parts (hi, mid, lo): x = hi + mid + lo to ~2^-24 relative
__device__ __forceinline__ void split3(const float4& p, const float4& q, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r = x[i] - (float)h;
        const __bf16 m = (__bf16)r;
        const float r2 = r - (float)m;
        hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2;
    }
}

constexpr int STAGE = 8192 + 27 * 1024;   // A 64 x 128 B | W 3 parts x 9 tiles x 1 KiB
constexpr int PIECES = 35;


template <int MODE, int NPROD, int NTW, typename Issue>
__device__ __forceinline__ void k_loop(f32x4 (&acc)[2][5], const char* smem, int iters, Issue& issue, int rp, int t0, int li,
                                       int kq, int swz, int lane) {
    for (int t = 0; t < iters; ++t) {
        if (MODE >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE >= 1) issue(t + 1);
        const char* st = smem + (t & 1) * STAGE;
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float* as = (const float*)st + (rp * 32 + r * 16 + li) * 32;
            const float4 p = *(const float4*)(as + (((2 * kq) ^ swz) << 2)), q = *(const float4*)(as + (((2 * kq + 1) ^ swz) << 2));
            split3(p, q, ah[r], am[r], al[r]);
        }
        const bf16x8* bs = (const bf16x8*)(st + 8192) + lane;
        bf16x8 bh[NTW], bm[NTW], bl[NTW];
#pragma unroll
        for (int n = 0; n < NTW; ++n) { bh[n] = bs[((t0 + n) * 3 + 0) * 64]; bm[n] = bs[((t0 + n) * 3 + 1) * 64]; bl[n] = bs[((t0 + n) * 3 + 2) * 64]; }
#define ALL(A_, B_)                                                          \
    _Pragma("unroll") for (int r = 0; r < 2; ++r)                            \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n) acc[r][n] = MF(A_[r], B_[n], acc[r][n]);
        ALL(al, bh) ALL(ah, bl) ALL(am, bm)
        if (NPROD == 9) { ALL(al, bm) ALL(am, bl) ALL(al, bl) }
        ALL(am, bh) ALL(ah, bm) ALL(ah, bh)
    }
}

template <int MODE, int NPROD>
__global__ __launch_bounds__(256, 2) void probe(const char* g, float* out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4, swz = (li >> 1) & 7;
    const int rp = wave & 1, ch = wave >> 1;           // 32-row pair, column half (5 / 4 tiles)
    for (int i = tid; i < 2 * STAGE / 4; i += 256) ((float*)smem)[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[2][5];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int n = 0; n < 5; ++n) acc[r][n] = f32x4{0, 0, 0, 0};
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // A: 64 row tiles x (64 rows x 544 fp32), shared by the column tiles; W: 8 column tiles x 17 k-tiles x 27 KiB
    // fragment-ordered bf16 parts (1.8 MB per 4 column tiles: L2 resident, as in the real GEMM)
    const int rt = blockIdx.x & 63, ct = (blockIdx.x >> 6) & 7;
    const char* srcA = g + (size_t)rt * 64 * 2176 + (size_t)(lane >> 3) * 2176 + (lane & 7) * 16;
    const char* srcW = g + (size_t)64 * 64 * 2176 + (size_t)ct * 17 * 27 * 1024 + lane * 16;
    auto issue = [&](int t) {
        const int kt = t % 17;
        for (int p = wave; p < PIECES; p += 4) {
            if (p < 8) dma16(srcA + (size_t)p * 8 * 2176 + kt * 128, lds0 + (t & 1) * STAGE + p * 1024);
            else dma16(srcW + (size_t)(kt * 27 + (p - 8)) * 1024, lds0 + (t & 1) * STAGE + p * 1024);
        }
    };
    if (MODE >= 1) issue(0);
    if (ch) k_loop<MODE, NPROD, 4>(acc, smem, iters, issue, rp, 5, li, kq, swz, lane);
    else k_loop<MODE, NPROD, 5>(acc, smem, iters, issue, rp, 0, li, kq, swz, lane);
    float s = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int n = 0; n < 5; ++n) s += acc[r][n][0] + acc[r][n][1] + acc[r][n][2] + acc[r][n][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NPROD>
void run(const char* g, float* out, const char* name, int grid) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<MODE, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, NPROD>), dim3(grid), dim3(256), 2 * STAGE, 0, g, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NPROD>), dim3(grid), dim3(256), 2 * STAGE, 0, g, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * iters * 64.0 * 136.0 * 32.0 * 2.0;    // fp32-algorithmic
    printf("%-30s grid %4d: %7.1f TF fp32-equivalent  (%.0f cycles @2.4GHz per k-tile of 32 per resident WG)\n", name, grid,
           fl / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / iters);
}
int main() {
    char* g; float* out;
    hipMalloc(&g, (size_t)64 << 20); hipMemset(g, 0, (size_t)64 << 20); hipMalloc(&out, 512 * 256 * 4);
    run<0, 6>(g, out, "x3 6-prod, no DMA", 256);
    run<1, 6>(g, out, "x3 6-prod, DMA ring", 256);
    run<0, 6>(g, out, "x3 6-prod, no DMA", 512);
    run<1, 6>(g, out, "x3 6-prod, DMA ring", 512);
    run<0, 9>(g, out, "x3 9-prod, no DMA", 512);
    run<1, 9>(g, out, "x3 9-prod, DMA ring", 512);
    return 0;
}
