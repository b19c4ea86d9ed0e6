// Shared device helpers for the gfx950 kernels (wave64, fp32 MFMA 16x16x4).
#pragma once
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "../../include/mpl_hip.h"

namespace mpl {

// Environment switches of measurement scripts (geometry overrides, phase timing, alternative kernels).  They exist in laboratory
// builds only (-DMPL_LAB, tools/build_variants.sh): the product library never changes behaviour because a variable happens to be set.
inline const char* lab_getenv(const char* name) {
#ifdef MPL_LAB
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

#ifndef MPL_LAB
#ifdef MPL_SPLIT_MIX
#error "MPL_SPLIT_MIX is a laboratory switch: build with -DMPL_LAB"
#endif
#endif
#ifndef MPL_SPLIT_MIX
#define MPL_SPLIT_MIX 1     // 0: the lo part of the two-term fp16 split by convert / subtract / convert (rounds 3-5); bitwise the same
#endif
// x = hi + lo in fp16: hi = fp16(x), lo = fp16(x - hi) (RNE; the residual is exact in fp32; subnormal results are kept) -- the
// operand split of the fp16x2 engines (h2_phase.hpp, spt.hip).  8 values per lane = one MFMA fragment per part.
// Round 6: the lo part is ONE v_fma_mix{lo,hi}_f16 per value (fp16(hi * -1.0 + x): hi read as fp16, the fma in fp32, one rounding
// to fp16) instead of v_cvt_f32_f16 + v_sub_f32 + half a v_cvt_pk_f16_f32: 20 VALU per split of 8 values instead of 32-36 --
// the LayerNorm GEMMs of the FPT stack split one A fragment per k-tile and wave, the SPT kernel is VALU-bound.  Bit for bit the
// old result (tools/micro/mix_probe.hip: 2 M values incl. subnormals, ties and the edges of the window; every byte-exact
// operand test and golden).
typedef _Float16 mpl_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 mpl_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_f16(const float (&x)[8], mpl_f16x8& hi, mpl_f16x8& lo) {
#if MPL_SPLIT_MIX
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const mpl_f16x2 hp = {(_Float16)x[2 * p], (_Float16)x[2 * p + 1]};       // v_cvt_pk_f16_f32 (RNE)
        h[p] = __builtin_bit_cast(unsigned, hp);
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(d) : "v"(h[p]), "v"(x[2 * p]), "v"(x[2 * p + 1]));
        l[p] = d;
    }
    hi = __builtin_bit_cast(mpl_f16x8, h);
    lo = __builtin_bit_cast(mpl_f16x8, l);
#else
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = (_Float16)x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = (_Float16)(x[i] - (float)hi[i]);
#endif
}


typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// D(16x16) += A(16x4) * B(4x16), exact fp32 (v_mfma_f32_16x16x4_f32, 32 cycles/SIMD).
// Operand layout: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// result reg r of lane l is D[row = 4*(l>>4) + r][col = l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// K-permuted 16-deep step: lane (i, kq) holds a = X[i][kb + 4*kq + 0..3], b = W[j][kb + 4*kq + 0..3].
// Step q contracts k = {kb + 4*kq' + q : kq' = 0..3}; the four steps cover kb..kb+15 exactly once.
__device__ __forceinline__ f32x4 mfma16_k16(const float4& a, const float4& b, f32x4 c) {
    c = mfma16(a.x, b.x, c);
    c = mfma16(a.y, b.y, c);
    c = mfma16(a.z, b.z, c);
    c = mfma16(a.w, b.w, c);
    return c;
}

// v[l] + v[l ^ 16] and v[l] + v[l ^ 32] in every lane by the gfx950 row / half swaps (one VALU instruction, no trip through
// the LDS crossbar that __shfl_xor takes as ds_bpermute_b32: ~100 cycles each in a dependent chain of four per LayerNorm
// fragment).  a = b = v; the swap exchanges the odd rows of a with the even rows of b (the upper half of a with the lower half
// of b), after which a + b is the pair sum in both partners -- the same two addends as v + shfl(v), so bitwise the same.
__device__ __forceinline__ float xor16_add(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float xor32_add(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact-erf GELU == torch.nn.GELU() default (multiview_mpl.py:22 act_layer=nn.GELU)
// One 1-KiB DMA piece: lane l's 16 bytes at g land at LDS byte address lds_dst + 16 l (lds_dst wave-uniform).
// M0 is saved/restored inside the statement (the compiler does not preserve it around asm).
__device__ __forceinline__ void dma16(const float* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_dst)
        : "memory");
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU with erf from Abramowitz & Stegun 7.1.26 (|error| of erf <= 1.5e-7 in exact arithmetic, 6e-7 in fp32; the GELU
// value is as close to the fp64 one as with an fp32 libm erf: 4.7e-7 vs 4.4e-7 max abs over [-6, 6]) on the hardware
// reciprocal and exp2: ~16 instructions instead of the ~40 of erff -- the MLP activation was a third of the SPT fc1 phase and
// 16 k of the 76 k cycles of an FPT fc1 phase with erff.
// gelu_as(x) * 2 hs for a power-of-two hs (hs = 1/2: the plain GELU): the static scale of a split operand rides on the 0.5
__device__ __forceinline__ float gelu_as_scaled(float x, float hs);
__device__ __forceinline__ float gelu_as(float x) { return gelu_as_scaled(x, 0.5f); }
__device__ __forceinline__ float gelu_as_scaled(float x, float hs) {
    const float z = x * 0.70710678118654752440f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * az * az);
    // 0.5 x (1 + sign(x) (1 - p t e)) = h + |h| - |h| (p t e),  h = x / 2: no copysign, no 1 + erf
    const float h = hs * x, ah = fabsf(h);
    return fmaf(-ah, (pl * t) * e, h + ah);
}


__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

// ---- optional per-launch event bracketing (mpl_profile_start/stop) ----------------------------
struct ProfScope {
    ProfScope(int kind, hipStream_t s);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

inline int hip_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fprintf(stderr, "mpl_hip: kernel launch failed: %s\n", hipGetErrorString(e));
    return e == hipSuccess ? MPL_OK : MPL_E_LAUNCH;
}

// ---- kernel launchers (defined in the .hip files) -------------------------------------------
// LayerNorm statistics buffer: 2 * M * ln_stat_slices(K) floats (per-slice {mean, M2}, see ln_gemm.hip)
inline int ln_stat_slices(int K) { return (K % 136 == 0) ? K / 136 : 1; }
int launch_row_stats(const float* x, int M, int K, int ldx, float* stats, hipStream_t s);
// stats_out (optional, residual epilogue only, N % 136 == 0): the epilogue also emits the LayerNorm partials of
// the rows it produced, so the next LN-GEMM needs no statistics pass.
int launch_ln_gemm(const float* A, int lda, const float* stats, const float* ln_w, const float* ln_b, float eps,
                   const float* W, const float* bias, const float* R, int ldr, float* C, int ldc, int M, int N, int K,
                   int epi, float* stats_out, hipStream_t s);
bool qkv_attention_fusable(int n_tok, int dim, int heads);
int launch_ln_qkv_attention(const float* x, int M, int D, const float* stats, const float* ln_w, const float* ln_b,
                            float eps, const float* W, const float* bias, int n_tok, int heads, float* att,
                            hipStream_t s);
// ---- device-side failures (a persistent kernel whose wait for a partner workgroup ran out) -------------------------
// One sticky word per device in pinned, device-visible host memory: kernels set it (system scope), the host reads it
// without any synchronisation at the start of the next API call and fails that call (MPL_E_DEVICE) until it is cleared.
unsigned* device_error_word(int dev);      // device-usable address; nullptr if the allocation failed
int device_error_pending(int dev);         // 1 when a kernel reported a failure since the last clear
void device_error_clear(int dev);
// test hook (mpl_x3_spin_limit): the phase before which one workgroup of the NEXT persistent launch deserts its team; taking it
// clears it (one-shot, shared by both engines)
void set_fault_injection(int phase);
int take_fault_injection();
// 0 when `s` is not being captured into a graph; MPL_E_UNSUPPORTED when it is: the persistent launches are serialised per
// device through a process-global event, which a capture would turn into a captured event (later eager launches on other
// streams would then depend on a graph-internal node) and a graph replay would bypass altogether
int refuse_stream_capture(hipStream_t s);
void h2_set_spin_log2(int log2_polls);     // test hook (mpl_x3_spin_limit)
void h2_set_row_tiles(int rt);             // A/B switch of the block stack: 0 by shape, 1 / 2 row tiles per stage
void h2_set_narrow(int mode);              // row-narrow teams: 0 by shape, 1 never, 2 / 3 = 32- / 16-row workgroups where legal
// arrival counters of a stack call: one per (row tile, 16-row group) -- the row-narrow teams synchronise per sub-tile -- then the error word
constexpr int H2_CTR_PER_TILE = 4;
constexpr int H2_XCC_WORDS = 256;          // behind the arrival counters: one word per team, the set of XCDs its workgroups run on
inline int h2_err_index(int n_tiles) { return H2_CTR_PER_TILE * n_tiles + H2_XCC_WORDS; }      // the error word of the call (the last word)
int h2_stack_form_code(int M, int D, int n_tok, int np, int cus);      // h2_gemm.hip: MPL_FORM_* of a team launch of this shape
void h2_set_direct_w(int on);              // A/B switch: 0 = the 16-row teams run the ring form (h2n_gemm.hip) instead of the direct-W form
void h2_set_write_through(int always);     // A/B switch: 1 = write-through hand-off stores whatever the placement of a team
// The persistent block-stack kernels (h2_stack_kernel and its pair forms, sm_stack_kernel) need every workgroup resident: the library
// serialises its own launches of them per device, whatever stream they are on -- each launch waits for the event recorded
// behind the previous one (api.hip).  The event exists from the first call (waiting on a never-recorded event is a no-op).
hipEvent_t stack_chain_event(int dev);
std::mutex& stack_chain_mutex(int dev);
// fp32 GEMMs on the fp16 matrix cores from operands split in TWO fp16 parts (three partial products), h2_gemm.hip: the
// default engine of the FPT block stack
bool h2_shape_ok(int N, int K);
size_t h2_operand_bytes(int N, int K, int np = 2);        // packed weights + {c, sc, sw, bound, so}[N] + meta[8]; 0 = no layout.  np = 1:
                                                          // the bf16 operand of b1_gemm.hip (pairs of k-tiles, trailer {c, s}[N])
const float* h2_out_scale(const unsigned short* op, int N, int K);   // so[N] inside an operand: static per-column scales of its outputs
size_t h2_act_bytes(int M, int K, int rpt, int np = 2);
int h2_rows_per_tile(int n_tok);
bool h2_attention_fusable(int n_tok, int dim, int heads);
void h2_set_debug_buffer(unsigned long long* p);
int launch_pack_h2(const float* W, int N, int K, const float* ln_w, const float* ln_b, const float* bias, const float* in_scale,
                   unsigned short* dst, hipStream_t s);
int launch_h2_entry(const float* X, int M, int K, int ldx, float* stats, unsigned* counters, int n_counters,
                    const unsigned short* const* ops, int n_apps, hipStream_t s);
int launch_h2_pack_rows(const float* X, int M, int K, int ldx, int rpt, unsigned short* dst, float* sc, hipStream_t s);
int launch_h2_gemm(const float* X, const unsigned short* A2, const float* a_inv, const unsigned short* W2, bool ln, const float* stats,
                   float eps, const float* R, int ldr, float* C, int ldc, unsigned short* C2, float* stats_out,
                   int M, int N, int K, int rpt, int epi, hipStream_t s);
int launch_h2_qkv_attention(const float* X, const unsigned short* W2, const float* stats, float eps, int M, int D, int n_tok,
                            int heads, unsigned short* att2, hipStream_t s);
int launch_h2_stack(float* x, int M, int D, int n_tok, int heads, const unsigned short* const* ops, int n_apps,
                    unsigned short* att2, unsigned short* hid2, float* stats, unsigned* counters, float eps, int stop_after,
                    hipStream_t s);
// bf16 GEMMs (operands rounded to bf16, fp32 accumulation) on the same stage as the h2 engine, b1_gemm.hip: BASELINE configs[2].
// A stage carries a PAIR of k-tiles; every A operand arrives packed (x too: the residual epilogues keep a bf16 copy x16 of
// the fp32 rows); a LayerNorm in front of a Linear is folded (gain into W, mean / rstd applied by the epilogue).
int launch_pack_b1(const float* W, int N, int K, const float* ln_w, const float* ln_b, const float* bias, unsigned short* dst,
                   hipStream_t s);
// entry of a bf16 stack, one launch: x -> x16 (packed, zero padded), LayerNorm slice partials, zeroed counters + error word
int launch_b1_entry(const float* X, int M, int K, int ldx, int rpt, unsigned short* x16, float* stats, unsigned* counters,
                    int n_counters, hipStream_t s);
int launch_b1_gemm(const unsigned short* A1, const unsigned short* W1, bool ln, const float* stats, float eps, const float* R, int ldr,
                   float* C, int ldc, unsigned short* C1, float* stats_out, int M, int N, int K, int rpt, int epi, hipStream_t s);
int launch_b1_qkv_attention(const unsigned short* x16, const unsigned short* W1, const float* stats, float eps, int M, int D, int n_tok,
                            int heads, unsigned short* att1, hipStream_t s);
int launch_b1_stack(float* x, unsigned short* x16, int M, int D, int n_tok, int heads, const unsigned short* const* ops, int n_apps,
                    unsigned short* att1, unsigned short* hid1, float* stats, unsigned* counters, float eps, int stop_after,
                    hipStream_t s);
// Block stack for up to 80 token rows (sm_stack.hip; beyond one sequence: groups of sequences of at most 16 rows side by side): every GEMM on the whole chip (one 16-column tile per workgroup, weights
// read in place), activations handed over as {value, tag} pairs, exact fp32 on the matrix cores
bool sm_stack_ok(int M, int D, int n_tok, int H, int n_apps, int n_blocks, int cus);      // cus: every workgroup must be resident
bool sm_stack_enabled();
void sm_stack_disable(int off);            // A/B switch (mpl_x3_stack_mode bit 3)
size_t sm_stack_ws_bytes(int M, int D);
int sm_stack_max_rows();
int launch_sm_stack(float* x, int n_seq, int n_tok, int D, int H, const mpl_block_weights* blocks, const uint8_t* schedule, int n_apps,
                    void* ws, size_t ws_bytes, const unsigned** err_ws, int spin_log2, hipStream_t s);
int launch_token_attention(const float* qkv, int n_seq, int n_tok, int dim, int heads, float* out, hipStream_t s);
// use_packed: every SPT block carries the split operand of mpl_spt_pack in qkv_w3 (spt3_kernel: Linear layers on the bf16
// matrix cores); else the fp32-MFMA kernel reads the nn.Linear weights in place
int launch_spt(const mpl_config* cfg, const mpl_weights* w, const mpl_inputs* in, float* xs, int use_packed, hipStream_t s);
size_t spt_pack_bytes();
int launch_spt_pack(const mpl_block_weights* bw_host, unsigned short* dst, int fold_q, hipStream_t s);
int launch_d32_qkv(const float* x, int M, const unsigned short* pack, float* qkv, hipStream_t s);
int launch_d32_mlp(float* x, const float* att, int M, const unsigned short* pack, hipStream_t s);
// y_out != nullptr: stop after the Conv1d weighted mean and write the (B, J*d) feature instead of running head[0..1]
// err_ws (optional): the error word of this call's block stack (see device_error_word); set -> the output is NaN
int launch_fuse_head(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* out, float* y_out,
                     const unsigned* err_ws, hipStream_t s);
int launch_layernorm_rows(const float* x, int M, int K, int ldx, const float* g, const float* b, float eps, float* y,
                          int ldy, hipStream_t s);
int launch_view_norm(const mpl_config* cfg, const mpl_weights* w, const float* x, int batch, float* xn, hipStream_t s);
int launch_linear_act(const float* xa, int Ka, int lda, const float* xb, int Kb, int ldb, int M, const float* W, int ldw,
                      const float* bias, int N, const float* bn_w, const float* bn_b, const float* bn_mean,
                      const float* bn_var, float bn_eps, int relu, float* y, int ldy, hipStream_t s);

int launch_pose_metrics(const float* out, const float* tgt, const float* wgt, int B, int J, const float* scale3,
                        const float* offset3, unsigned skip_mask, float* res, hipStream_t s);

int launch_prepare_inputs(const float* px, const float* conf, const double* cams_dev, int B, int V, int J, float w, float h,
                          int norm_in, int norm_cam, float* const* poses, float* const* rays, float* const* centers,
                          hipStream_t s);

}  // namespace mpl
