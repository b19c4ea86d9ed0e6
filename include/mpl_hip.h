/*
 * mpl_hip.h -- C ABI of libmpl_hip.so: the MI355X (gfx950) implementation of OpenMPL's
 * multi-view pose-lifting forward pass.
 *
 * The reference has no FFI layer: its "operator API" for this path is the Python
 * nn.Module contract of MPL/lib/models/multiview_mpl.py.  This
 * header is the boundary a binding for that contract calls: plain pointers and sizes,
 * no torch types.  All `const float*` below are DEVICE pointers to fp32 data in the
 * reference's own parameter layouts (nn.Linear weight = [out][in] row-major, i.e. the
 * tensors of the reference state_dict are consumed as they are).  The one exception is
 * optional DERIVED data: the packed fp16x2 / bf16 copies of the FPT Linear weights that
 * mpl_pack_h2 / mpl_pack_bf16 build from those tensors; whoever hands them over
 * must rebuild them when the source parameters change (the Python binding keys them on
 * the parameters' storage addresses and versions).
 * `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); every call only
 * enqueues work on it and never synchronises.  Every function returns 0 on success or a
 * negative MPL_E_* code (mpl_hip_error_string() explains it); nothing falls back to a CPU
 * path.
 *
 * Reference interface each entry point replaces (file:line in /root/reference/MPL/lib/models/):
 *   mpl_forward            MultiView_MPL.forward              multiview_mpl.py:450-525
 *   mpl_spt_tokens         Spatial_forward_features + per-view glue   :349-414, :458-499
 *   mpl_block_stack        the `for blk in self.blocks` loop of forward_features  :420-423
 *                          (Block :84-92, Attention :53-67, Mlp :31-37)
 *   mpl_ln_linear          nn.LayerNorm + nn.Linear (+GELU | +residual) pairs inside Block
 *   mpl_token_attention    Attention.forward minus the two Linear layers   :55-64
 *   mpl_fuse_head          forward_features tail :425-446 + default head :283-286,:521-523
 */
#ifndef MPL_HIP_H_
#define MPL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPL_HIP_ABI_VERSION 13
#define MPL_MAX_VIEWS 32
#define MPL_MAX_APPS 64 /* max Block applications in one stack schedule */

/* error codes */
#define MPL_OK 0
#define MPL_E_INVALID (-1)     /* bad argument / shape */
#define MPL_E_UNSUPPORTED (-2) /* flag combination or size not implemented in HIP */
#define MPL_E_WORKSPACE (-3)   /* workspace too small */
#define MPL_E_LAUNCH (-4)      /* hip runtime reported an error at launch */
#define MPL_E_DEVICE (-5)      /* a kernel of an EARLIER call on this device reported a failure (see mpl_device_error) */

/* flag bits of mpl_config.flags == constructor kwargs of MultiView_MPL (multiview_mpl.py:98-117) */
#define MPL_F_MULTI_SPT (1u << 0)       /* multiple_spatial_blocks */
#define MPL_F_CONF_ADD (1u << 1)        /* add_confidence_input */
#define MPL_F_CONF_MULT (1u << 2)       /* mult_confidence_emb */
#define MPL_F_CONF_ATTN_W (1u << 3)     /* confidence_as_attention_uncertainty_weight */
#define MPL_F_POS3D_LEARN (1u << 4)     /* pose_3d_emb_learnable */
#define MPL_F_POS3D_SPATIAL (1u << 5)   /* add_3D_pos_encoding_in_Spatial */
#define MPL_F_RAYS_TOKEN (1u << 6)      /* input_rays_as_token */
#define MPL_F_POS3D_TO_RAYS (1u << 7)   /* add_3D_pos_encoding_to_rays */
#define MPL_F_NO_SPT (1u << 8)          /* no_transformer_spt */
#define MPL_F_NO_FPT (1u << 9)          /* no_transformer_fpt */
#define MPL_F_CONF_IN_FPT (1u << 10)    /* confidence_in_FPT */
#define MPL_F_KPTOK (1u << 11)          /* FPT_blocks_view_keypoint_tokens: FPT blocks of width d over 17*V tokens */
/* not a constructor kwarg: engine selection of the FPT block stack.  Stacks of up to 80 token rows (a single frame, a few
 * persons; groups of sequences of at most 16 rows, as many as the compute units hold) normally run the small-batch engine (csrc/sm_stack.hip: exact fp32 MFMA, the whole chip per GEMM), larger ones the
 * team kernels (fp16x2 operands): two fp32 engines that agree to ~1e-7 but not bit for bit.  With this flag the team kernels
 * run for EVERY batch size, so that a pose carries the same bits whatever batch or shard it arrives in. */
#define MPL_F_NO_SMALL_STACK (1u << 12)

/* epilogues of mpl_ln_linear */
#define MPL_EPI_BIAS 0          /* y = a W^T + b                       (attn.qkv) */
#define MPL_EPI_BIAS_GELU 1     /* y = gelu_erf(a W^T + b)             (mlp.fc1 + nn.GELU) */
#define MPL_EPI_BIAS_RESIDUAL 2 /* y = r + a W^T + b                   (attn.proj / mlp.fc2 + residual) */

/* Parameters of one Block (multiview_mpl.py:70-92), device pointers, reference layouts:
 * norm{1,2}.{weight,bias} (D); attn.qkv (3D,D),(3D); attn.proj (D,D),(D);
 * mlp.fc1 (2D,D),(2D); mlp.fc2 (D,2D),(D). */
typedef struct mpl_block_weights {
    const float *ln1_w, *ln1_b;
    const float *qkv_w, *qkv_b;
    const float *proj_w, *proj_b;
    const float *ln2_w, *ln2_b;
    const float *fc1_w, *fc1_b;
    const float *fc2_w, *fc2_b;
    /* Optional packed bf16 operands (mpl_pack_bf16) of the four Linear layers: qkv built with norm1 folded, fc1 with norm2
     * folded, proj / fc2 plain.  When all four are non-NULL in every block of a stack whose shape the engine supports (D a
     * multiple of 544, n_tok <= 32), the stack's GEMMs run on the bf16 matrix cores with bf16 operands and fp32
     * accumulation (activations handed from GEMM to GEMM as bf16; LayerNorm statistics, softmax, GELU, the residual
     * stream and the output stay fp32) -- BASELINE.json configs[2] "bf16".  The fp32 tensors remain the source of truth. */
    const uint16_t *qkv_w16, *proj_w16, *fc1_w16, *fc2_w16;
    /* qkv_w3: the row-local split operand of a d = 32 block (mpl_spt_pack for the SPT blocks of an mpl_spt_set, mpl_d32_pack for
     * the keypoint-token FPT blocks); proj_w3 / fc1_w3 / fc2_w3 are unused and must be NULL (they carried the three-part bf16
     * operands of the round-2 fp32 engine, removed in round 5). */
    const uint16_t *qkv_w3, *proj_w3, *fc1_w3, *fc2_w3;
    /* Optional fp16x2 operands (mpl_pack_h2) of the four Linear layers, same folding as above: the DEFAULT fp32 engine of the
     * FPT block stack (csrc/h2_gemm.hip).  Every fp32 operand is split in two fp16 parts under an exact power-of-two scale
     * (per weight column; static per activation column, from a data-free bound), each product is accumulated in fp32
     * from three partial products ("3xTF32" on the fp16 matrix cores): as accurate as an fp32 GEMM, half the matrix
     * instructions of the round-2 three-part engine.  Used when all four are non-NULL (and the *_w16 are NULL) in every block; shapes must satisfy mpl_pack_h2_bytes() != 0.  qkv_h2 / fc1_h2 come from mpl_pack_h2 (norm1 /
     * norm2 folded); proj_h2 / fc2_h2 from mpl_pack_h2_scaled against the static output scales of the layer that produces their
     * input: in_scale = mpl_pack_h2_out_scale(qkv_h2, 3D, D) + 2D (the v columns) for proj, mpl_pack_h2_out_scale(fc1_h2, 2D, D)
     * for fc2.  The stack verifies that pairing on the device (fingerprints inside the operands) and poisons the call's output
     * (NaN poses, MPL_E_DEVICE) on a mismatch. */
    const uint16_t *qkv_h2, *proj_h2, *fc1_h2, *fc2_h2;
} mpl_block_weights;

/* Per-view (or shared) spatial parameter set, multiview_mpl.py:159-195, :236-249. */
typedef struct mpl_spt_set {
    const float *embed_w, *embed_b;  /* Spatial_patch_to_embedding[.v]  (d,in_chans),(d) */
    const float *conf_w, *conf_b;    /* confidence_to_embedding[.v]     (d,1),(d) or NULL */
    const float *pos_embed;          /* Spatial_pos_embed[.v]           (J,d) */
    const mpl_block_weights *blocks; /* DEVICE array [depth]            Spatial_blocks[.v] */
} mpl_spt_set;

typedef struct mpl_config {
    int32_t num_joints; /* NETWORK.NUM_JOINTS (17) */
    int32_t dim;        /* NETWORK.DIM = embed_dim_ratio (32) */
    int32_t depth;      /* NETWORK.TRANSFORMER_DEPTH */
    int32_t heads;      /* NETWORK.TRANSFORMER_HEADS */
    int32_t num_views;  /* V, a constructor constant (multiview_mpl.py:534-546) */
    int32_t in_chans;   /* 2, or 3 with confidence_input_as_third (:159-168) */
    uint32_t flags;     /* MPL_F_* */
    int32_t reserved;
} mpl_config;

typedef struct mpl_weights {
    const mpl_spt_set *spt_sets;       /* DEVICE array: [V] if MPL_F_MULTI_SPT else [1] */
    const float *spatial_norm_w, *spatial_norm_b;          /* Spatial_norm (d) */
    const float *pos_3d_embed;                             /* (J, d|2d) */
    const float *pos_3d_view_coding;                       /* (J, d|2d) */
    const float *pos_3d_linear_w, *pos_3d_linear_b;        /* (d|2d,3),(d|2d) */
    const float *ray_embed_w, *ray_embed_b;                /* ray_to_embedding (d,3),(d) or NULL */
    const float *conf_fpt_w, *conf_fpt_b;                  /* confidence_to_embedding_FPT (d,1),(d) or NULL */
    const mpl_block_weights *fpt_blocks;                   /* HOST array [depth]: blocks.{l} */
    const float *view_norm_w, *view_norm_b;                /* View_norm (J*d) */
    const float *wmean_w, *wmean_b;                        /* weighted_mean Conv1d (1,V,1),(1) */
    const float *head_ln_w, *head_ln_b;                    /* head.0 (J*d) */
    const float *head_w, *head_b;                          /* head.1 (3J, J*d),(3J) */
    uint32_t spt_packed;  /* != 0: EVERY block of every spt_set carries the operand of mpl_spt_pack in its qkv_w3 field: the
                           * SPT Linear layers run as fp32 arithmetic on the fp16 matrix cores (spt3_kernel); 0: the fp32
                           * matrix instructions read the nn.Linear weights in place */
    uint32_t reserved;
} mpl_weights;

typedef struct mpl_inputs {
    int32_t batch;
    int32_t reserved;
    const float *poses[MPL_MAX_VIEWS];   /* V x (B,J,3) contiguous: x_norm, y_norm, conf (function_mpl.py:350) */
    const float *rays[MPL_MAX_VIEWS];    /* V x (B,J,3) or NULL when unused by the flags */
    const float *centers[MPL_MAX_VIEWS]; /* V x (B,1,3) or NULL */
} mpl_inputs;

int mpl_hip_abi_version(void);
const char *mpl_hip_error_string(int code);

/* FPT token width D_f = J*d (x2 with MPL_F_RAYS_TOKEN), multiview_mpl.py:140-142. */
int mpl_fpt_width(const mpl_config *cfg);

/* Bytes of scratch mpl_forward needs for `batch` poses (activations only; 256-B aligned). */
size_t mpl_forward_workspace_bytes(const mpl_config *cfg, int batch);

/* Whole forward: V x (B,J,3) keypoints -> out (B,J,3).  MultiView_MPL.forward :450-525. */
int mpl_forward(const mpl_config *cfg, const mpl_weights *w, const mpl_inputs *in, float *out,
                void *workspace, size_t workspace_bytes, void *stream);

/* Stage 1: embedding + SPT stack + Spatial_norm + per-view glue -> xs (B*V, D_f) row-major,
 * row = b*V + v.  Spatial_forward_features :349-414 and forward :458-499. */
int mpl_spt_tokens(const mpl_config *cfg, const mpl_weights *w, const mpl_inputs *in, float *xs, void *stream);

/* Stage 2: a stack of Blocks applied in place on x (n_seq*n_tok, D).  `schedule[i]` = layer
 * index of the i-th application (the reference applies the last layer twice, :420-423).
 * `blocks` is a HOST array.  Workspace: mpl_block_stack_workspace_bytes(). */
size_t mpl_block_stack_workspace_bytes(int n_seq, int n_tok, int dim);
int mpl_block_stack(float *x, int n_seq, int n_tok, int dim, int heads, const mpl_block_weights *blocks,
                    const uint8_t *schedule, int n_apps, void *workspace, size_t workspace_bytes, void *stream);
/* the same with engine flags (MPL_F_NO_SMALL_STACK; 0 = mpl_block_stack) */
int mpl_block_stack_ex(float *x, int n_seq, int n_tok, int dim, int heads, const mpl_block_weights *blocks,
                       const uint8_t *schedule, int n_apps, void *workspace, size_t workspace_bytes, unsigned flags, void *stream);

/* y[M,N] = epilogue( LN(x)[M,K] . W[N,K]^T + bias ); ln_w == NULL skips the LayerNorm.
 * `stats` is scratch for 2*M*max(1, K/136) floats (per-slice LayerNorm partials) when ln_w != NULL.
 * residual may alias y. */
int mpl_ln_linear(const float *x, int M, int K, const float *ln_w, const float *ln_b, float eps, const float *W,
                  const float *bias, int N, int epilogue, const float *residual, float *y, float *stats,
                  void *stream);

/* Split operand of ONE SPT block (d = 32: qkv 96x32, proj 32x32, fc1 64x32, fc2 32x64): the four weights as two fp16 parts
 * each (hi | lo, MFMA fragment order) under one exact power-of-two scale per output column, with norm1 / norm2 folded into
 * the qkv / fc1 weights; behind them the epilogue vectors c_n (bias + folded LayerNorm offset), the multipliers sc_n and the
 * static scales of the proj / fc2 inputs (the arithmetic of mpl_pack_h2).  mpl_spt_pack_bytes() = 48 KiB (34.6 KiB used).
 * `block` is a HOST struct whose pointers are device addresses -- EVERY weight, bias and LayerNorm vector of the block must
 * be set; the result goes into the qkv_w3 field of the block's entry in the DEVICE array of mpl_spt_set and is derived data:
 * repack when any tensor of the block changes. */
size_t mpl_spt_pack_bytes(void);
int mpl_spt_pack(const mpl_block_weights *block, uint16_t *dst, void *stream);
/* The same operand for a D = 32 FPT block (keypoint-token variant, multiview_mpl.py:261-266): identical layout, the q columns
 * NOT pre-scaled (the token attention kernel scales q itself).  A block whose qkv_w3 field carries it (proj_w3 / fc1_w3 /
 * fc2_w3 NULL) runs in mpl_block_stack as d32_qkv_kernel -> token attention -> d32_mlp_kernel: two row-local launches per
 * block application instead of four GEMMs and two LayerNorm-statistics passes. */
int mpl_d32_pack(const mpl_block_weights *block, uint16_t *dst, void *stream);

/* Packed bf16 operand of an nn.Linear layer (W[N][K], bias[N]) for the bf16 engine (csrc/b1_gemm.hip): ONE bf16 per weight
 * (round to nearest even of gamma o W) in MFMA fragment order, two 32-deep k-tiles per 2-KiB fragment slot (an odd k-tile count
 * is padded with a zero k-tile), followed by fp32 vectors of N entries: with ln_w / ln_b != NULL the LayerNorm in front of the
 * layer is folded in, LN(x).W^T + b = rstd (x16.(gamma o W)^T - mean s) + c with s_n = sum_k of the ROUNDED gamma_k W_nk and
 * c_n = b_n + sum_k beta_k W_nk (else c = bias, s = 0).  mpl_pack_bf16_bytes() is the size of `dst` in bytes, or 0 when the
 * shape is not supported (N must be a multiple of 136, K of 544). */
size_t mpl_pack_bf16_bytes(int N, int K);
int mpl_pack_bf16(const float *W, const float *bias, const float *ln_w, const float *ln_b, int N, int K, uint16_t *dst,
                  void *stream);

/* Diagnostics (tools/chain_phase.py): when non-NULL, every packed-operand GEMM launch writes five shader-clock stamps per
 * wave (entry, k-loop start, k-loop end, stores issued, stores drained) at device_buffer[(block * 8 + wave) * 8 ..];
 * the buffer must hold 64 bytes per wave of the largest launch.  NULL (the default) switches it off.  The stamps are
 * compiled only into a library built with -DH2_DBG=1 (MPL_HIPCC_FLAGS); in the product build the call is a no-op. */
int mpl_x3_debug_buffer(void *device_buffer);
/* Diagnostics / A-B: bit 0: 0 (default) = a block stack on packed operands is ONE persistent launch (row-tile chains of
 * workgroups, h2_stack_kernel); 1 = one launch per GEMM (results agree to <= 4 ulp, each mode is bitwise
 * deterministic).  Bits 1-2: 0 = the stack picks its stage by the shape of the launch (fp16x2: teams that own two or more row
 * tiles walk PAIRS of tiles, h2_stack2_kernel; bf16: always one tile at a time), 1 / 2 = force the one- / two-tile stage
 * (bitwise the same poses; bf16: h2_stackp_kernel, pairs in every phase).  Bit 3: no
 * small-batch engine (sm_stack.hip: stacks of up to 80 token rows run every GEMM on the whole chip, the activations handed
 * over as {value, tag} pairs, exact fp32 MFMA on the nn.Linear tensors in place; two fp32 engines, <= 1e-6 apart).  Bit 4: the 16-row teams in
 * the ring form (h2_stackn_kernel) instead of the direct-W form (h2_stackd_kernel).  Bits 5-6: row-narrow teams: 0 = by the
 * shape of the launch, 1 = never, 2 / 3 = 32- / 16-row workgroups wherever legal.  Bit 7: write-through hand-off stores also
 * for teams that sit on one XCD.  Every one of these forms yields bitwise the same poses.  Bits 8.. = stop after that many
 * GEMM phases (tools/chain_phase.py). */
int mpl_x3_stack_mode(int one_launch_per_gemm);

/* Which kernel form mpl_block_stack(_ex) takes for a stack of this shape on the current device, by the library's own rule
 * (csrc/h2_phase.hpp h2_stack_form, csrc/api.hip block_stack_impl): for benchmarks that name the kernel they time and tests that
 * pin the rule.  operand_parts: 2 = the blocks carry mpl_pack_h2 operands, 1 = mpl_pack_bf16, 0 = neither; `flags` as
 * mpl_config.flags (MPL_F_NO_SMALL_STACK).  The small-batch engine additionally needs the nn.Linear tensors of the blocks
 * (the query assumes they are there).  Negative = MPL_E_*.  Every team form yields bitwise the same poses. */
enum {
    MPL_FORM_UNPACKED = 0,        /* no packed operands: one launch per GEMM on the fp32-MFMA engine (ln_gemm.hip / the D = 32 path) */
    MPL_FORM_SMALL = 1,           /* sm_stack_kernel: up to 80 token rows, every GEMM on the whole chip */
    MPL_FORM_TEAMS = 2,           /* h2_stack_kernel<NP>: one 64-row tile per team step */
    MPL_FORM_PAIRS = 3,           /* h2_stack2_kernel<2> (A/B: h2_stackp_kernel<1>): pairs of row tiles */
    MPL_FORM_ROWS32 = 4,          /* h2_stackn_kernel<2>: 32-row teams */
    MPL_FORM_ROWS16 = 5,          /* h2_stackn_kernel<2>: 16-row teams, ring form (A/B, or A operand too large for LDS) */
    MPL_FORM_ROWS16_DIRECT = 6,   /* h2_stackd_kernel<2>: 16-row teams, direct-W form */
    MPL_FORM_PER_GEMM = 7         /* h2_gemm_kernel: one launch per GEMM (mpl_x3_stack_mode bit 0) */
};
int mpl_block_stack_form(int n_seq, int n_tok, int D, int heads, int n_apps, int operand_parts, unsigned flags);
/* The same question with everything the launch rule looks at: n_blocks = distinct blocks the schedule indexes (mpl_block_stack_form
 * assumes the reference's schedule: n_apps - 1), raw_tensors != 0 = every nn.Linear / LayerNorm tensor of those blocks is present
 * (the small-batch engine reads them in place; a caller that hands over packed operands only gets the team kernels).  Both queries
 * and mpl_block_stack(_ex) itself go through ONE predicate (csrc/api.hip small_engine_taken + h2_phase.hpp h2_stack_form). */
int mpl_block_stack_form_ex(int n_seq, int n_tok, int D, int heads, int n_apps, int n_blocks, int raw_tensors, int operand_parts,
                            unsigned flags);
/* MPL_FORM_* of the calling thread's most recent successful mpl_block_stack(_ex) / mpl_forward: the form that was LAUNCHED (incl.
 * the fall-through when the small-batch engine's occupancy query refuses); negative before the first one. */
int mpl_block_stack_last_form(void);

/* softmax(q k^T * hd^-0.5) v per (sequence, head) on a packed qkv (n_seq*n_tok, 3*dim). Attention :55-64. */
int mpl_token_attention(const float *qkv, int n_seq, int n_tok, int dim, int heads, float *out, void *stream);

/* Stage 3: strip ray features, View_norm, Conv1d weighted mean over views, head LN + Linear.
 * x (B*V, D_f) -> out (B, 3J).  forward_features :425-446, head :521-523. */
int mpl_fuse_head(const mpl_config *cfg, const mpl_weights *w, const float *x, int batch, float *out, void *stream);

/* ---- output-side variants (constructor flags linear_weighted_mean, deep_head, head_kadkhod; :277-317, :506-519).
 * The default tail stays fused in mpl_fuse_head / mpl_forward; these building blocks let the binding compose the
 * other tails exactly as the reference does. */

/* strip ray features + View_norm + Conv1d weighted mean (:425-446) WITHOUT the head: x (B*V, D_f) -> y (B, J*d). */
int mpl_view_fuse(const mpl_config *cfg, const mpl_weights *w, const float *x, int batch, float *y, void *stream);

/* strip ray features + View_norm only (:425-439): x (B*V, D_f) -> xn (B, V*J*d), the input of the
 * linear_weighted_mean Linear (:441-443). */
int mpl_view_norm(const mpl_config *cfg, const mpl_weights *w, const float *x, int batch, float *xn, void *stream);

/* y[M,K] = LayerNorm(x[M,K]) (head[0] of every head variant, eps 1e-5). */
int mpl_layernorm(const float *x, int M, int K, const float *gamma, const float *beta, float eps, float *y,
                  void *stream);

/* y[M,N] = act( bn( [xa | xb] . W^T + bias ) ): nn.Linear on the concatenation of xa (M,Ka) and xb (M,Kb; may be
 * NULL/0) -- torch.cat([x_prev, x], dim=1) of head_kadkhod :511-513 without materialising it -- followed by an
 * optional BatchNorm1d in eval mode (running statistics, bn_w == NULL skips it) and an optional ReLU.
 * W is (N, Ka+Kb) row-major. */
int mpl_linear(const float *xa, int Ka, const float *xb, int Kb, int M, const float *W, const float *bias, int N,
               const float *bn_w, const float *bn_b, const float *bn_mean, const float *bn_var, float bn_eps, int relu,
               float *y, void *stream);

/* ---- input preparation (the step right before the path): raw per-view detections ->
 * the tensors mpl_forward consumes.  Replaces, per sample and view, normalize_screen_coordinates
 * (joints_dataset_mpl.py:817-820), the camera normalisation of __getitem__ (:615-623), create_3d_ray_coords (:872-904),
 * cam_center (:646) and the [x, y, conf] concat (:772).
 *   joints_px (B,V,J,2) pixel coordinates, conf (B,V,J) or NULL (-> 1), device;
 *   cams_dev device (V,16) float64: fx fy cx cy | R row-major (world->camera) | t (camera centre in world coords);
 *   img_w/img_h = NETWORK.IMAGE_SIZE; normalize_inputs = DATASET.INPUTS_NORMALIZED, normalize_cameras =
 *   DATASET.NORMALIZE_CAMERAS;  poses/rays/centers: host arrays of V device pointers to (B,J,3),(B,J,3),(B,1,3). */
int mpl_prepare_inputs(const float *joints_px, const float *conf, const double *cams_dev, int batch, int views,
                       int joints, float img_w, float img_h, int normalize_inputs, int normalize_cameras,
                       float *const *poses, float *const *rays, float *const *centers, void *stream);

/* ---- output-side epilogue (the step right after the path): what validate() does on the host
 * with `output.clone().cpu().numpy()` -- room de-normalisation x*scale+offset (function_mpl.py:476-488, host float[3]
 * arrays, NULL = identity) and the MPJPE family: loss.py:39-57 / :110-124 (mean Euclidean error, optional (B,J)
 * weights, per-axis mean |error|, on the RAW tensors) and evaluate.py:91-125 (per-joint absolute and root-relative
 * PJPE with np.nansum semantics, per-axis distances with np.nanmean semantics, on the de-normalised tensors).
 * result (device, mpl_pose_metrics_size(J) floats): [0] loss, [1..3] loss per axis, [4..4+J) pjpe_abs, [4+J] mpjpe_abs,
 * [5+J..5+2J) pjpe_rel, [5+2J] mpjpe_rel, then dist (J x 3), dist_mean (3). */
int mpl_pose_metrics_size(int joints);
int mpl_pose_metrics(const float *output, const float *target, const float *weight, int batch, int joints,
                     const float *scale3, const float *offset3, float *result, void *stream);
/* The same with config.NOT_CONSIDER_SOME_KP_IN_EVAL (evaluate.py:101-104, :110-113): bit j of not_consider_mask set = joint j is
 * deleted from the two MEANS over joints (mpjpe_abs, mpjpe_rel); the per-joint errors are reported unchanged. */
int mpl_pose_metrics_ex(const float *output, const float *target, const float *weight, int batch, int joints,
                        const float *scale3, const float *offset3, uint32_t not_consider_mask, float *result, void *stream);

/* ---- fp16x2 split-operand engine (csrc/h2_gemm.hip): "fp32" precision of MultiView_MPL (the default).
 * mpl_pack_h2: derived operand of one nn.Linear (W (N,K) row-major, bias (N)), optionally with the LayerNorm in front of it
 * folded in (ln_w, ln_b of length K, or both NULL).  dst: mpl_pack_h2_bytes(N, K) bytes (0 = the shape has no layout:
 * N % 136 == 0, K % 544 == 0 required; with a LayerNorm folded in also K <= 1088: K in {544, 1088}).
 * mpl_ln_linear_h2: y[M,N] = epi(LN?(x) W^T + b) from that operand -- the unit-test entry of one GEMM (the forward runs all
 * GEMMs of a stack in ONE launch, mpl_block_stack); has_ln: x is normalised with eps (stats: 2 * M * K/136 floats of
 * scratch); else x is packed with its measured amax.  workspace: mpl_ln_linear_h2_workspace_bytes(M, K). */
size_t mpl_pack_h2_bytes(int N, int K);
int mpl_pack_h2(const float *W, const float *bias, const float *ln_w, const float *ln_b, int N, int K, uint16_t *dst,
                void *stream);
/* Static activation scales.  A LayerNorm operand stores, per output column n, the power of two so_n that brings the data-free
 * bound of |out_n| (sqrt(K) |gamma o W_n|_2 + |bias_n + beta . W_n|) to the top of the fp16 window; the GEMM epilogue that hands
 * column n on as a packed operand (attention output: a convex combination of v rows; GELU output: |gelu(t)| <= |t|) multiplies
 * by so_n.  mpl_pack_h2_out_scale returns the DEVICE address of so[N] inside such an operand.  The Linear that consumes those
 * columns (proj after qkv's v columns, fc2 after fc1: multiview_mpl.py:65, :35) is packed with mpl_pack_h2_scaled: W_nk is stored
 * as W_nk / in_scale_k (exact), so operands are equilibrated per channel and one outlier channel costs no other channel
 * resolution.  in_scale: K powers of two on the device. */
int mpl_pack_h2_scaled(const float *W, const float *bias, const float *in_scale, int N, int K, uint16_t *dst, void *stream);
const float *mpl_pack_h2_out_scale(const uint16_t *operand, int N, int K);
size_t mpl_ln_linear_h2_workspace_bytes(int M, int K);
int mpl_ln_linear_h2(const float *x, int M, int K, int has_ln, float eps, const uint16_t *W2, int N, int epilogue,
                     const float *residual, float *y, float *stats, void *workspace, size_t workspace_bytes, void *stream);

/* ---- device-side failures.  The block stack runs as ONE persistent launch whose workgroups hand operands to each
 * other (h2_phase.hpp); it needs all its workgroups resident, i.e. the device to itself for the duration of the launch
 * (single tenant: launches of this library on different streams of one process are serialised by the library, other
 * processes on the same GPU are not).  A workgroup whose wait for a partner runs out (~10 s) never goes on with stale
 * operands: it sets a sticky per-device error word, the poses of that call are NaN, and EVERY later call on the device
 * returns MPL_E_DEVICE until mpl_device_error_clear() -- the reference's convention for a failed forward is a Python
 * exception (SURVEY.md 8b "Error convention"), which is what the binding turns this into.
 * mpl_device_error(dev): the word, 0 = no failure (dev < 0: the current device); no synchronisation, reads pinned memory.  Bits:
 *   1 = lost hand-off / operands packed against other scales (above);
 *   2 = the split-operand SPT engine met a confidence-weighted attention row (confidence_as_attention_uncertainty_weight,
 *       reference multiview_mpl.py:61-62: the softmax rows are multiplied by the caller's `conf`, which is DATA) beyond the
 *       fp16 window its static scales assume: that sequence's poses are NaN -- never a saturated, plausible-looking pose --
 *       and the caller is pointed at the native-fp32 engine (MPL precision "fp32_mfma"), which has no window.
 * mpl_x3_spin_limit(v): test hook.  v & 0xff = log2 of the polls before a wait counts as lost (default 23 ~ 10 s);
 * v >> 8 = fault injection: when > 0, one workgroup of the next launches deserts its team before that GEMM phase, so
 * that the failure path can be exercised deterministically (tests/test_failures_gpu.py). */
int mpl_device_error(int device);
int mpl_device_error_clear(int device);
int mpl_x3_spin_limit(int log2_polls);

/* Measurement aid (bench.py roofline leg): between start and stop every kernel launched by this
 * library on ANY stream is bracketed by a hipEvent pair recorded on that same stream.  stop()
 * synchronises the events and returns, per kernel kind, the summed device time (ms) and the launch
 * count.  Process-global and not thread safe; never enabled on the product path. */
#define MPL_K_SPT 0
#define MPL_K_ROW_STATS 1
#define MPL_K_GEMM 2
#define MPL_K_ATTENTION 3
#define MPL_K_FUSE_HEAD 4
#define MPL_K_PACK 5 /* derived-operand builders: mpl_pack_h2*, mpl_spt_pack, mpl_d32_pack, mpl_pack_bf16 */
#define MPL_K_COUNT 6
int mpl_profile_start(void);
int mpl_profile_stop(float *kind_ms, int *kind_launches, int n_kinds);

#ifdef __cplusplus
}
#endif
#endif /* MPL_HIP_H_ */
