// Does feeding MFMA B operands from AGPRs (ds_read_b128 straight into a[...]) avoid the LDS-return / MFMA contention
// measured by tools/mfma_bf16_peak.hip (16.8 -> 21.6 cycles per MFMA with two waves per SIMD)?
// Pattern per iteration: 15 ds_read_b128 (5 tiles x 3 parts) + 30 MFMAs (6 product passes over 5 accumulators).
// hipcc --offload-arch=gfx950 -O3 tools/agpr_probe.hip -o build_tmp/agpr_probe && build_tmp/agpr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define RD_A(i) "ds_read_b128 a[" #i "*4:" #i "*4+3], %[addr] offset:" #i "*1024\n\t"
#define RD_V(i) "ds_read_b128 v[" #i "*4+100:" #i "*4+103], %[addr] offset:" #i "*1024\n\t"

#define VCLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219"
template <int MODE>   // 2: double-buffered VGPR sets, 0: B in VGPRs v[100..159], 1: B in AGPRs a[0..59]
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(1024))) char sm[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((float*)sm)[i] = 0.001f * (i & 63);
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm + (threadIdx.x & 63) * 16;
    bf16x8 a0, a1, a2;
    for (int i = 0; i < 8; ++i) { a0[i] = (__bf16)(float)(threadIdx.x + i); a1[i] = (__bf16)(float)(i + 1); a2[i] = (__bf16)(float)(i + 2); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            asm volatile(
                "ds_read_b128 a[0:3], %[addr]\n\t ds_read_b128 a[4:7], %[addr] offset:1024\n\t ds_read_b128 a[8:11], %[addr] offset:2048\n\t"
                "ds_read_b128 a[12:15], %[addr] offset:3072\n\t ds_read_b128 a[16:19], %[addr] offset:4096\n\t ds_read_b128 a[20:23], %[addr] offset:5120\n\t"
                "ds_read_b128 a[24:27], %[addr] offset:6144\n\t ds_read_b128 a[28:31], %[addr] offset:7168\n\t ds_read_b128 a[32:35], %[addr] offset:8192\n\t"
                "ds_read_b128 a[36:39], %[addr] offset:9216\n\t ds_read_b128 a[40:43], %[addr] offset:10240\n\t ds_read_b128 a[44:47], %[addr] offset:11264\n\t"
                "ds_read_b128 a[48:51], %[addr] offset:12288\n\t ds_read_b128 a[52:55], %[addr] offset:13312\n\t ds_read_b128 a[56:59], %[addr] offset:14336\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
#define M5(A, P) "v_mfma_f32_16x16x32_bf16 %[c0], %[" #A "], a[" #P "+0:" #P "+3], %[c0]\n\t" \
                 "v_mfma_f32_16x16x32_bf16 %[c1], %[" #A "], a[" #P "+12:" #P "+15], %[c1]\n\t" \
                 "v_mfma_f32_16x16x32_bf16 %[c2], %[" #A "], a[" #P "+24:" #P "+27], %[c2]\n\t" \
                 "v_mfma_f32_16x16x32_bf16 %[c3], %[" #A "], a[" #P "+36:" #P "+39], %[c3]\n\t" \
                 "v_mfma_f32_16x16x32_bf16 %[c4], %[" #A "], a[" #P "+48:" #P "+51], %[c4]\n\t"
                M5(a2, 0) M5(a0, 8) M5(a1, 4) M5(a1, 0) M5(a0, 4) M5(a0, 0)
#undef M5
                : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [c4] "+v"(c4)
                : [addr] "v"(addr), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2)
                : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15",
                  "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31",
                  "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47",
                  "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59");
        } else if (MODE == 2) {
#define RDS(B) "ds_read_b128 v[" #B "+0:" #B "+3], %[addr]\n\t ds_read_b128 v[" #B "+4:" #B "+7], %[addr] offset:1024\n\t ds_read_b128 v[" #B "+8:" #B "+11], %[addr] offset:2048\n\t" \
               "ds_read_b128 v[" #B "+12:" #B "+15], %[addr] offset:3072\n\t ds_read_b128 v[" #B "+16:" #B "+19], %[addr] offset:4096\n\t ds_read_b128 v[" #B "+20:" #B "+23], %[addr] offset:5120\n\t" \
               "ds_read_b128 v[" #B "+24:" #B "+27], %[addr] offset:6144\n\t ds_read_b128 v[" #B "+28:" #B "+31], %[addr] offset:7168\n\t ds_read_b128 v[" #B "+32:" #B "+35], %[addr] offset:8192\n\t" \
               "ds_read_b128 v[" #B "+36:" #B "+39], %[addr] offset:9216\n\t ds_read_b128 v[" #B "+40:" #B "+43], %[addr] offset:10240\n\t ds_read_b128 v[" #B "+44:" #B "+47], %[addr] offset:11264\n\t" \
               "ds_read_b128 v[" #B "+48:" #B "+51], %[addr] offset:12288\n\t ds_read_b128 v[" #B "+52:" #B "+55], %[addr] offset:13312\n\t ds_read_b128 v[" #B "+56:" #B "+59], %[addr] offset:14336\n\t"
#define M5B(A, B, P) "v_mfma_f32_16x16x32_bf16 %[c0], %[" #A "], v[" #B "+" #P "+0:" #B "+" #P "+3], %[c0]\n\t" \
                     "v_mfma_f32_16x16x32_bf16 %[c1], %[" #A "], v[" #B "+" #P "+12:" #B "+" #P "+15], %[c1]\n\t" \
                     "v_mfma_f32_16x16x32_bf16 %[c2], %[" #A "], v[" #B "+" #P "+24:" #B "+" #P "+27], %[c2]\n\t" \
                     "v_mfma_f32_16x16x32_bf16 %[c3], %[" #A "], v[" #B "+" #P "+36:" #B "+" #P "+39], %[c3]\n\t" \
                     "v_mfma_f32_16x16x32_bf16 %[c4], %[" #A "], v[" #B "+" #P "+48:" #B "+" #P "+51], %[c4]\n\t"
#define ALL30(B) M5B(a2, B, 0) M5B(a0, B, 8) M5B(a1, B, 4) M5B(a1, B, 0) M5B(a0, B, 4) M5B(a0, B, 0)
            // two k-tiles per loop trip: reads of set Y are in flight while the 30 MFMAs of set X run (15 reads outstanding)
            asm volatile(
                RDS(160) "s_waitcnt lgkmcnt(15)\n\t" ALL30(100)
                RDS(100) "s_waitcnt lgkmcnt(15)\n\t" ALL30(160)
                : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [c4] "+v"(c4)
                : [addr] "v"(addr), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2)
                : "memory", VCLOB);
        } else {
            asm volatile(
                "ds_read_b128 v[100:103], %[addr]\n\t ds_read_b128 v[104:107], %[addr] offset:1024\n\t ds_read_b128 v[108:111], %[addr] offset:2048\n\t"
                "ds_read_b128 v[112:115], %[addr] offset:3072\n\t ds_read_b128 v[116:119], %[addr] offset:4096\n\t ds_read_b128 v[120:123], %[addr] offset:5120\n\t"
                "ds_read_b128 v[124:127], %[addr] offset:6144\n\t ds_read_b128 v[128:131], %[addr] offset:7168\n\t ds_read_b128 v[132:135], %[addr] offset:8192\n\t"
                "ds_read_b128 v[136:139], %[addr] offset:9216\n\t ds_read_b128 v[140:143], %[addr] offset:10240\n\t ds_read_b128 v[144:147], %[addr] offset:11264\n\t"
                "ds_read_b128 v[148:151], %[addr] offset:12288\n\t ds_read_b128 v[152:155], %[addr] offset:13312\n\t ds_read_b128 v[156:159], %[addr] offset:14336\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
#define M5(A, P) "v_mfma_f32_16x16x32_bf16 %[c0], %[" #A "], v[100+" #P "+0:100+" #P "+3], %[c0]\n\t" \
                 "v_mfma_f32_16x16x32_bf16 %[c1], %[" #A "], v[100+" #P "+12:100+" #P "+15], %[c1]\n\t" \
                 "v_mfma_f32_16x16x32_bf16 %[c2], %[" #A "], v[100+" #P "+24:100+" #P "+27], %[c2]\n\t" \
                 "v_mfma_f32_16x16x32_bf16 %[c3], %[" #A "], v[100+" #P "+36:100+" #P "+39], %[c3]\n\t" \
                 "v_mfma_f32_16x16x32_bf16 %[c4], %[" #A "], v[100+" #P "+48:100+" #P "+51], %[c4]\n\t"
                M5(a2, 0) M5(a0, 8) M5(a1, 4) M5(a1, 0) M5(a0, 4) M5(a0, 0)
#undef M5
                : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [c4] "+v"(c4)
                : [addr] "v"(addr), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2)
                : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
                  "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126",
                  "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140",
                  "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154",
                  "v155", "v156", "v157", "v158", "v159");
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0];
}
template <int MODE>
void run(float* out, int wps) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256 * wps), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256 * wps), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B operands in %s, %d wave/SIMD: %.1f cycles @2.4GHz per MFMA per SIMD\n", MODE == 2 ? "VGPRs, prefetched one k-tile ahead" : MODE ? "AGPRs" : "VGPRs", wps,
           ms * 1e-3 * 2.4e9 / (iters * (MODE == 2 ? 60.0 : 30.0) * wps));
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0>(out, 1); run<1>(out, 1); run<2>(out, 1); run<0>(out, 2); run<1>(out, 2); run<2>(out, 2);
    return 0;
}
