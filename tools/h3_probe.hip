// De-risking probe for a k loop WITHOUT the per-stage barrier and WITHOUT LDS-DMA of the weights (the h2 stage costs ~1000
// cycles for 27 MFMAs per SIMD, additive: skeleton 190 + barrier 75-210 + MFMA 417 + DMA 200 + fragment reads 114 + ...):
//   * wave w owns column tile w (16 columns) for ALL 4 row groups (waves 0..3 also a quarter of the half tile): its W
//     fragments are private -> plain global_load_dwordx4 straight into VGPRs, prefetched PF stages ahead, no LDS, no barrier;
//   * the A operand (64 rows x 32 k, hi | lo = 8 KiB per k-tile) goes through an LDS ring of RA slots written by waves 0..3
//     (global load -> convert -> 2 ds_write_b128) and read by every wave (8 ds_read_b128 per k-tile);
//   * ONE barrier per KB k-tiles publishes KB slots of A.
// Dummy data, 256 workgroups, K = 17 k-tiles x NPASS passes, repeated; prints cycles per stage (= per k-tile and pass).
// hipcc --offload-arch=gfx950 -O3 tools/h3_probe.hip -o build_tmp/h3_probe && build_tmp/h3_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int RA = 8;                       // A ring slots (8 KiB each)
template <int NPASS, int KB, int PF>
__global__ __launch_bounds__(512, 2) void h3_loop(const char* __restrict__ W, const float* __restrict__ X, float* out, int reps) {
    __shared__ __attribute__((aligned(1024))) char lds[RA * 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int KT = 17;
    const bool extra = wave < 4;            // waves 0..3: + the half tile for row group `wave`
    f32x4 acc[NPASS][5];
    for (int p = 0; p < NPASS; ++p) for (int r = 0; r < 5; ++r) acc[p][r] = f32x4{0, 0, 0, 0};
    // W fragment stream of this wave: [stage][slot wave][2 parts][lane] (+ slot 8), 18 KiB per stage as in the real operand
    const char* wb = W + (size_t)(blockIdx.x & 3) * (KT * NPASS * 18432) + wave * 2048 + lane * 16;
    const float* xr = X + (size_t)blockIdx.x * 64 * 544 + (size_t)((wave & 3) * 16 + (lane & 15)) * 544 + 4 * (lane >> 4);
    // queue of PF + 1 fragment sets, index 0 = the stage being multiplied; rotated by register moves after every stage
    f16x8 wq[PF + 1][2], we[PF + 1][2];
    auto load_w = [&](int s) {                  // into the LAST queue entry
        const char* p = wb + (size_t)(s % (KT * NPASS)) * 18432;
        wq[PF][0] = *reinterpret_cast<const f16x8*>(p);
        wq[PF][1] = *reinterpret_cast<const f16x8*>(p + 1024);
        if (extra) {
            we[PF][0] = *reinterpret_cast<const f16x8*>(p + 16384);
            we[PF][1] = *reinterpret_cast<const f16x8*>(p + 17408);
        }
    };
    auto rotate = [&]() {
#pragma unroll
        for (int i = 0; i < PF; ++i) { wq[i][0] = wq[i + 1][0]; wq[i][1] = wq[i + 1][1]; we[i][0] = we[i + 1][0]; we[i][1] = we[i + 1][1]; }
    };
    // A staging by waves 0..3: k-tile kt -> slot kt % RA (row group = wave)
    float4 ra0, ra1;
    auto load_a = [&](int kt) {
        if (wave < 4) {
            ra0 = *reinterpret_cast<const float4*>(xr + (kt % KT) * 32);
            ra1 = *reinterpret_cast<const float4*>(xr + (kt % KT) * 32 + 16);
        }
    };
    auto store_a = [&](int kt) {
        if (wave < 4) {
            float z[8] = {ra0.x, ra0.y, ra0.z, ra0.w, ra1.x, ra1.y, ra1.z, ra1.w};
            f16x8 hi, lo;
            for (int j = 0; j < 8; ++j) { z[j] = fmaf(z[j], 3.0f, 0.5f); hi[j] = (_Float16)z[j]; }
            for (int j = 0; j < 8; ++j) lo[j] = (_Float16)(z[j] - (float)hi[j]);
            char* p = lds + (kt % RA) * 8192 + wave * 2048 + lane * 16;
            *reinterpret_cast<f16x8*>(p) = hi;
            *reinterpret_cast<f16x8*>(p + 1024) = lo;
        }
    };
    f16x8 af[4][2], ae[2];
    auto read_a = [&](int kt) {
        const char* p = lds + (kt % RA) * 8192 + lane * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            af[r][0] = *reinterpret_cast<const f16x8*>(p + r * 2048);
            af[r][1] = *reinterpret_cast<const f16x8*>(p + r * 2048 + 1024);
        }
        if (extra) {
            ae[0] = *reinterpret_cast<const f16x8*>(p + wave * 2048);
            ae[1] = *reinterpret_cast<const f16x8*>(p + wave * 2048 + 1024);
        }
    };
    const int n_kt = KT * reps;
    // prologue: A of k-tiles 0 .. 2 KB - 1 staged, published; W of stages 0 .. PF - 1 in flight
    for (int kt = 0; kt < 2 * KB; ++kt) { load_a(kt); store_a(kt); }
#pragma unroll
    for (int s = 0; s < PF; ++s) { load_w(s); rotate(); }      // entries 0 .. PF - 1 = stages 0 .. PF - 1
    load_a(2 * KB);
    __syncthreads();
    int stage = 0;
    for (int kt = 0; kt < n_kt; ++kt) {
        if (kt % KB == 0 && kt > 0) __syncthreads();         // publishes the A slots written during the last KB k-tiles
        read_a(kt);
        // A of k-tile kt + 2 KB: loaded one k-tile ago, written now (its slot was read KB .. 2 KB k-tiles ago, behind a barrier)
        store_a(kt + 2 * KB);
        load_a(kt + 2 * KB + 1);
#pragma unroll
        for (int p = 0; p < NPASS; ++p, ++stage) {
            load_w(stage + PF);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[p][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[0][0], af[r][1], acc[p][r], 0, 0, 0);
            if (extra) acc[p][4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(we[0][0], ae[1], acc[p][4], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[p][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[0][1], af[r][0], acc[p][r], 0, 0, 0);
            if (extra) acc[p][4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(we[0][1], ae[0], acc[p][4], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[p][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[0][0], af[r][0], acc[p][r], 0, 0, 0);
            if (extra) acc[p][4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(we[0][0], ae[0], acc[p][4], 0, 0, 0);
            rotate();
        }
    }
    float s = 0;
    for (int p = 0; p < NPASS; ++p) for (int r = 0; r < 5; ++r) s += acc[p][r][0] + acc[p][r][1] + acc[p][r][2] + acc[p][r][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int NPASS, int KB, int PF>
void run(const char* W, const float* X, float* out, int grid) {
    const int reps = 40;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((h3_loop<NPASS, KB, PF>), dim3(grid), dim3(512), 0, 0, W, X, out, reps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double stages = 17.0 * reps * NPASS;
    printf("NPASS %d, barrier every %d k-tiles, W prefetch %d stages, %3d workgroups: %6.0f cycles per stage @2.4 GHz (27 MFMAs per SIMD = 432)\n",
           NPASS, KB, PF, grid, ms * 1e-3 * 2.4e9 / stages);
}

int main() {
    char* W; float *X, *out;
    (void)hipMalloc(&W, 4 * 51 * 18432 + 65536);
    (void)hipMalloc(&X, (size_t)256 * 64 * 544 * 4 + 65536);
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemset(W, 0x11, 4 * 51 * 18432 + 65536);
    (void)hipMemset(X, 0, (size_t)256 * 64 * 544 * 4 + 65536);
    for (int grid : {32, 256}) {
        run<1, 1, 2>(W, X, out, grid);
        run<1, 2, 2>(W, X, out, grid);
        run<1, 2, 3>(W, X, out, grid);
        run<1, 4, 3>(W, X, out, grid);
        run<3, 1, 2>(W, X, out, grid);
        run<3, 2, 3>(W, X, out, grid);
        run<2, 2, 3>(W, X, out, grid);
    }
    return 0;
}
