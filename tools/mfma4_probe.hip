// Operand / result layout of v_mfma_f32_4x4x1_16B_f32 (16 blocks of a 4 x 4 x 1 product) on gfx950, found by one-hot inputs:
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4_probe.hip -o /tmp/mfma4_probe && /tmp/mfma4_probe
// Expected (and what token_attention_long_m4_kernel relies on): lane l = 4 b + x supplies A_b[i = x][0] and B_b[0][j = x];
// result register r of lane l = 4 b + x is D_b[i = r][j = x].
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float *a, *b, *d, ha[64], hb[64], hd[256];
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    int bad = 0;
    for (int la = 0; la < 64; la += 5)
        for (int lb = 0; lb < 64; ++lb) {
            for (int i = 0; i < 64; ++i) { ha[i] = 0.f; hb[i] = 0.f; }
            ha[la] = 2.f; hb[lb] = 3.f;
            hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
            hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const float want = (la / 4 == lb / 4 && l == 4 * (lb / 4) + (lb & 3) && r == (la & 3)) ? 6.f : 0.f;
                    if (hd[l * 4 + r] != want) {
                        if (bad < 12) printf("A lane %d, B lane %d: D[lane %d][reg %d] = %g (expected %g)\n", la, lb, l, r, hd[l * 4 + r], want);
                        ++bad;
                    }
                }
        }
    printf(bad ? "layout DIFFERS from the assumption (%d mismatches)\n" : "layout as assumed: A lane 4b+i, B lane 4b+j, D[reg i][lane 4b+j]\n", bad);
    return bad != 0;
}
