"""Golden-case table shared by the fixture generator and the parity tests.

Each case = (name, flags, batch, weight seed, input seed).  Weights are NOT stored
in the fixtures (except for the micro case): they are regenerated bit-identically
from ``openmpl_amd.detrng`` on any machine.
"""
BASE = dict(num_joints=17, embed_dim_ratio=32, num_heads=8)

CHOSEN = dict(pose_3d_emb_learnable=True)                       # h36m.yaml:85-100 / generator :180
FULL = dict(pose_3d_emb_learnable=True, confidence_input_as_third=True, input_rays_as_token=True,
            multiple_spatial_blocks=True, add_3D_pos_encoding_to_rays=True)   # hm_0_...yaml:71-81


def _c(name, extra, V, B, L, wseed=11, iseed=7, **kw):
    f = dict(BASE, depth=L, num_views=V)
    f.update(extra)
    f.update(kw)
    return dict(name=name, flags=f, batch=B, wseed=wseed, iseed=iseed)


CASES = [
    # BASELINE.json configs[0]: single-frame V=2 plumbing case
    _c("chosen_v2_b1_l12", CHOSEN, 2, 1, 12),
    _c("full_v2_b1_l12", FULL, 2, 1, 12),
    # configs[1] shape at small batch (same weights as the B=1024 bench workload)
    _c("chosen_v4_b8_l12", CHOSEN, 4, 8, 12),
    _c("full_v4_b8_l12", FULL, 4, 8, 12),
    _c("chosen_v4_b8_l2", CHOSEN, 4, 8, 2),
    _c("full_v4_b8_l2", FULL, 4, 8, 2),
    # configs[2] CMU shape (yaml depth 2)
    _c("chosen_v8_b4_l2", CHOSEN, 8, 4, 2),
    _c("full_v8_b4_l2", FULL, 8, 4, 2),
    # configs[4] large-view stress
    _c("chosen_v31_b2_l12", CHOSEN, 31, 2, 12),
    _c("full_v31_b2_l2", FULL, 31, 2, 2),
    # ragged batch (not a multiple of any tile) + 5 views (the reference's CMU default, :534-535)
    _c("chosen_v5_b19_l2", CHOSEN, 5, 19, 2),
    _c("full_v5_b19_l2", FULL, 5, 19, 2),
    # second tier (SURVEY.md 8a "Minimum flag coverage")
    _c("chosen_conf3rd_v3_b3_l2", dict(CHOSEN, confidence_input_as_third=True), 3, 3, 2),
    _c("multi_spt_v3_b3_l2", dict(CHOSEN, multiple_spatial_blocks=True), 3, 3, 2),
    _c("no_spt_v3_b3_l2", dict(CHOSEN, no_transformer_spt=True), 3, 3, 2),
    _c("no_fpt_v3_b3_l2", dict(CHOSEN, no_transformer_fpt=True), 3, 3, 2),
    _c("kptok_v3_b3_l2", dict(CHOSEN, FPT_blocks_view_keypoint_tokens=True), 3, 3, 2),
    # third tier
    _c("conf_add_v3_b3_l2", dict(CHOSEN, add_confidence_input=True), 3, 3, 2),
    _c("conf_mult_v3_b3_l2", dict(CHOSEN, mult_confidence_emb=True), 3, 3, 2),
    _c("conf_attnw_v3_b3_l2", dict(CHOSEN, confidence_as_attention_uncertainty_weight=True), 3, 3, 2),
    _c("conf_fpt_v3_b3_l2", dict(CHOSEN, confidence_in_FPT=True), 3, 3, 2),
    _c("linear_wmean_v3_b3_l2", dict(CHOSEN, linear_weighted_mean=True), 3, 3, 2),
    _c("deep_head_v3_b3_l2", dict(CHOSEN, deep_head=True, hidden_dim=64), 3, 3, 2),
    _c("kadkhod_v3_b3_l2", dict(CHOSEN, head_kadkhod=True, hidden_dim=64), 3, 3, 2),
    # the same heads at the reference's default width, TRANSFORMER_OUTPUT_HEAD_HIDDEN_DIM = 1024 (config.py:98)
    _c("deep_head_h1024_v3_b5_l2", dict(CHOSEN, deep_head=True, hidden_dim=1024), 3, 5, 2),
    _c("kadkhod_h1024_v3_b5_l2", dict(CHOSEN, head_kadkhod=True, hidden_dim=1024), 3, 5, 2),
    _c("linear_wmean_v4_b5_l2", dict(CHOSEN, linear_weighted_mean=True), 4, 5, 2),
    _c("geo3d_v3_b3_l2", dict(pose_3d_emb_learnable=False), 3, 3, 2),
    _c("inspatial_learn_v3_b3_l2", dict(pose_3d_emb_learnable=True, add_3D_pos_encoding_in_Spatial=True), 3, 3, 2),
    _c("inspatial_geo_v3_b3_l2", dict(pose_3d_emb_learnable=False, add_3D_pos_encoding_in_Spatial=True), 3, 3, 2),
    _c("raytoken_v3_b3_l2", dict(CHOSEN, input_rays_as_token=True), 3, 3, 2),
    _c("concat_conf_noop_v3_b3_l2", dict(CHOSEN, concat_confidence_emb=True, add_confidence_input=True), 3, 3, 2),
]

# One micro model committed whole, weights included (<50 kB): d=8, H=2, L=1, V=2.
MICRO = dict(name="micro_d8_h2_l1_v2", batch=3, wseed=3, iseed=5,
             flags=dict(num_joints=17, embed_dim_ratio=8, num_heads=2, depth=1, num_views=2,
                        pose_3d_emb_learnable=True))

BY_NAME = {c["name"]: c for c in CASES}
BY_NAME[MICRO["name"]] = MICRO
