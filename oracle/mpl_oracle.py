"""CPU oracle for OpenMPL's multi-view pose-lifting forward pass.

TEST INFRASTRUCTURE ONLY.  Nothing under ``openmpl_amd/`` imports this file; it is
used by ``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s
``cpu_baseline`` leg as the *checker* / reported CPU baseline, never as the product
path.

What it is: a from-scratch functional restatement (plain ``torch`` tensor ops on
CPU, any float dtype) of ``/root/reference/MPL/lib/models/multiview_mpl.py`` that
consumes a reference-layout ``state_dict``.  Each function cites the reference
lines it follows.  It exists because the reference's Python cannot travel to the
GPU box.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned by ``tests/golden/*.npz`` --
input/output/intermediate vectors produced by importing the *reference itself*
in the build container (``tests/golden/make_golden.py``, committed) -- and
``tests/test_oracle_golden.py`` checks the oracle against every one of them.

All tensors are (B, N, C) row-major like the reference.  ``flags`` is a plain
dict with the constructor keyword names of ``MultiView_MPL.__init__``
(multiview_mpl.py:95-117).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

DEFAULT_FLAGS = dict(
    num_joints=17, in_chans=2, embed_dim_ratio=32, depth=4, num_heads=8, num_views=5,
    add_confidence_input=False, mult_confidence_emb=False, concat_confidence_emb=False,
    confidence_input_as_third=False, pose_3d_emb_learnable=False, linear_weighted_mean=False,
    add_3D_pos_encoding_in_Spatial=False, input_rays_as_token=False,
    add_3D_pos_encoding_to_rays=False, confidence_as_attention_uncertainty_weight=False,
    multiple_spatial_blocks=False, no_transformer_spt=False, no_transformer_fpt=False,
    confidence_in_FPT=False, deep_head=False, head_kadkhod=False, hidden_dim=1024,
    FPT_blocks_view_keypoint_tokens=False,
)

LN_EPS_BLOCK = 1e-6   # norm_layer = partial(nn.LayerNorm, eps=1e-6), multiview_mpl.py:139
LN_EPS_HEAD = 1e-5    # head[0] = nn.LayerNorm(embed_dim) default eps, multiview_mpl.py:284


def resolve_flags(flags: Optional[dict]) -> dict:
    f = dict(DEFAULT_FLAGS)
    if flags:
        unknown = set(flags) - set(f)
        if unknown:
            raise KeyError("unknown flags: %s" % sorted(unknown))
        f.update(flags)
    # multiview_mpl.py:173-176 -- concat_confidence_emb switches all three conf flags off
    if f["concat_confidence_emb"]:
        f["add_confidence_input"] = False
        f["mult_confidence_emb"] = False
        f["concat_confidence_emb"] = False
    return f


_BF16_PREFIXES = ()   # set by forward(fpt_matmul_bf16=True): Linear layers whose operands are rounded to bf16


def _lin(x, sd, prefix):
    w = sd[prefix + ".weight"]
    return F.linear(x, w, sd[prefix + ".bias"])


def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def attention(x, sd, prefix, num_heads, row_weights=None):
    """Attention.forward, multiview_mpl.py:53-67."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = _lin(x, sd, prefix + ".qkv").reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)  # :55
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q @ k.transpose(-2, -1)) * (hd ** -0.5)          # :58 scale after the product
    att = att.softmax(dim=-1)                                # :59
    if row_weights is not None:                              # :61-62, (B,N,1) -> (B,1,N,1)
        att = att * row_weights.unsqueeze(1)
    y = (att @ v).transpose(1, 2).reshape(B, N, C)           # :64
    return _lin(y, sd, prefix + ".proj")                     # :65


def mlp(x, sd, prefix):
    """Mlp.forward, multiview_mpl.py:31-37 (nn.GELU() = exact erf form)."""
    return _lin(F.gelu(_lin(x, sd, prefix + ".fc1")), sd, prefix + ".fc2")


def _bf(t):
    """Round to bf16 (nearest even) and back: the operand rounding of the bf16 engine."""
    return t.to(torch.bfloat16).to(t.dtype)


def _ln_linear_bf16(x, sd, ln_prefix, lin_prefix, eps):
    """LayerNorm + Linear as openmpl_amd's bf16 engine computes it (csrc/b1_gemm.hip on csrc/h2_phase.hpp, NP = 1): the LayerNorm is folded
    into the GEMM, LN(x).W^T + b = rstd (x.(gamma o W)^T - mean s) + c, with the operands x and gamma o W rounded to bf16,
    s summed over the rounded weights, products and sums exact (here: in the evaluation dtype)."""
    W, b = sd[lin_prefix + ".weight"], sd[lin_prefix + ".bias"]
    g, e = sd[ln_prefix + ".weight"], sd[ln_prefix + ".bias"]
    mu = x.mean(-1, keepdim=True)
    rs = 1.0 / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + eps)
    Wp = _bf((W.float() * g.float()[None, :]).to(W.dtype))       # the kernel multiplies in fp32, then rounds to bf16
    s = Wp.sum(1)
    c = W @ e + b
    return rs * (_bf(x) @ Wp.T - mu * s) + c


def block_bf16(x, sd, prefix, num_heads):
    """Block.forward with the FPT GEMMs as the bf16 engine runs them: statistics, softmax, GELU and the residual stream
    in full precision; the four Linear layers see bf16 operands (activations rounded when handed to the next GEMM)."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = _ln_linear_bf16(x, sd, prefix + ".norm1", prefix + ".attn.qkv", LN_EPS_BLOCK)
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    att = ((qkv[0] @ qkv[1].transpose(-2, -1)) * (hd ** -0.5)).softmax(dim=-1)
    y = (att @ qkv[2]).transpose(1, 2).reshape(B, N, C)
    x = x + _bf(y) @ _bf(sd[prefix + ".attn.proj.weight"]).T + sd[prefix + ".attn.proj.bias"]
    h = F.gelu(_ln_linear_bf16(x, sd, prefix + ".norm2", prefix + ".mlp.fc1", LN_EPS_BLOCK))
    return x + _bf(h) @ _bf(sd[prefix + ".mlp.fc2.weight"]).T + sd[prefix + ".mlp.fc2.bias"]


def block(x, sd, prefix, num_heads, row_weights=None):
    """Block.forward, multiview_mpl.py:84-92 (DropPath/Dropout are identity in eval)."""
    if _BF16_PREFIXES and prefix.startswith(_BF16_PREFIXES) and row_weights is None:
        return block_bf16(x, sd, prefix, num_heads)
    x = x + attention(_ln(x, sd, prefix + ".norm1", LN_EPS_BLOCK), sd, prefix + ".attn", num_heads, row_weights)
    x = x + mlp(_ln(x, sd, prefix + ".norm2", LN_EPS_BLOCK), sd, prefix + ".mlp")
    return x


def block_schedule(depth: int, weighted: bool = False):
    """Order of Block applications of a depth-`depth` stack.

    multiview_mpl.py:405-410 / :420-423: ``for ix, blk: [x=blk(x,w)]; if last: x=blk(x); x=blk(x)``
    i.e. the last block runs twice; with confidence-as-attention-weight every block
    first runs once with row weights.  Returns [(layer, use_row_weights), ...].
    """
    out = []
    for ix in range(depth):
        if weighted:
            out.append((ix, True))
        if ix == depth - 1:
            out.append((ix, False))
        out.append((ix, False))
    return out


def spatial_features(pose, ray, center, view, sd, f):
    """Spatial_forward_features, multiview_mpl.py:349-414."""
    multi = f["multiple_spatial_blocks"]
    sfx = (".%d" % view) if multi else ""
    row_w = pose[:, :, 2:3] if f["confidence_as_attention_uncertainty_weight"] else None   # :352-353
    nin = 3 if f["confidence_input_as_third"] else 2                                        # :359-364
    x = _lin(pose[:, :, 0:nin], sd, "Spatial_patch_to_embedding" + sfx)
    if f["add_confidence_input"]:                                                           # :371-373
        x = x + _lin(pose[:, :, 2:3], sd, "confidence_to_embedding" + sfx)
    if f["mult_confidence_emb"]:                                                            # :374-376
        x = x * _lin(pose[:, :, 2:3], sd, "confidence_to_embedding" + sfx)
    x = x + sd["Spatial_pos_embed" + sfx]                                                   # :382-385
    if f["add_3D_pos_encoding_in_Spatial"] and ray is not None and center is not None:      # :389-396
        if f["pose_3d_emb_learnable"]:
            x = x + sd["pos_3d_embed"]
        else:
            x = x + _lin(F.normalize(ray - center, dim=2, p=2), sd, "pos_3d_linear")
    if not f["no_transformer_spt"]:                                                         # :251-252
        base = "Spatial_blocks" + sfx
        for layer, weighted in block_schedule(f["depth"], row_w is not None):               # :405-410
            x = block(x, sd, "%s.%d" % (base, layer), f["num_heads"], row_w if weighted else None)
    return _ln(x, sd, "Spatial_norm", LN_EPS_BLOCK)                                         # :412


def view_token(pose, ray, center, view, sd, f, taps=None):
    """Per-view body of MultiView_MPL.forward, multiview_mpl.py:458-492."""
    b = pose.shape[0]
    x = spatial_features(pose, ray, center, view, sd, f)
    if taps is not None and view == 0:
        taps["spt_view0"] = x
    if f["confidence_in_FPT"]:                                                              # :465-467
        x = x + _lin(pose[:, :, 2:3], sd, "confidence_to_embedding_FPT")
    if f["add_3D_pos_encoding_to_rays"] and f["input_rays_as_token"]:                       # :469-471
        x = torch.cat([x, _lin(ray - center, sd, "ray_to_embedding")], dim=2)
    if not f["add_3D_pos_encoding_in_Spatial"]:                                             # :474-481
        if f["pose_3d_emb_learnable"]:
            pe = sd["pos_3d_embed"]
        else:
            pe = _lin(F.normalize(ray - center, dim=2, p=2), sd, "pos_3d_linear")
    else:
        pe = sd["pos_3d_view_coding"]
    x = x + pe                                                                              # :483
    if (not f["add_3D_pos_encoding_to_rays"]) and f["input_rays_as_token"]:                 # :486-489
        x = torch.cat([x, _lin(ray - center, sd, "ray_to_embedding")], dim=1)
    return x.reshape(b, -1)                                                                 # :491


def fusion_features(xs, sd, f):
    """forward_features, multiview_mpl.py:416-447."""
    b = xs.shape[0]
    V, J, d = f["num_views"], f["num_joints"], f["embed_dim_ratio"]
    x = xs
    if not f["no_transformer_fpt"]:
        for layer, _ in block_schedule(f["depth"]):                                         # :420-423
            x = block(x, sd, "blocks.%d" % layer, f["num_heads"])
    if f["input_rays_as_token"] and not f["add_3D_pos_encoding_to_rays"]:                   # :425-429
        x = x.reshape(b, V, 2, J, d)[:, :, 0].reshape(b, V, -1)
    elif f["add_3D_pos_encoding_to_rays"]:                                                  # :430-434
        x = x.reshape(b, V, J, 2 * d)[:, :, :, :d].reshape(b, V, -1)
    if f["FPT_blocks_view_keypoint_tokens"]:                                                # :436-437
        x = x.reshape(b, V, -1)
    x = _ln(x, sd, "View_norm", LN_EPS_BLOCK)                                               # :439
    if f["linear_weighted_mean"]:                                                           # :441-443
        x = _lin(x.reshape(b, -1), sd, "weighted_mean")
    else:                                                                                   # :445 Conv1d(V,1,1)
        w = sd["weighted_mean.weight"].reshape(1, V, 1)
        x = (x * w).sum(dim=1) + sd["weighted_mean.bias"]
    return x.reshape(b, 1, -1)                                                              # :446


def _bn_eval(x, sd, prefix, eps=1e-5):
    return (x - sd[prefix + ".running_mean"]) / torch.sqrt(sd[prefix + ".running_var"] + eps) \
        * sd[prefix + ".weight"] + sd[prefix + ".bias"]


def _lbr(x, sd, lin, bn):
    return F.relu(_bn_eval(_lin(x, sd, lin), sd, bn))


def head(x, sd, f):
    """Output heads, multiview_mpl.py:283-317 (construction) and :506-525 (use)."""
    b = x.shape[0]
    if f["head_kadkhod"]:                                                                   # :301-317, :506-516
        x = x.reshape(b, -1)

        def stage(inp, s, first):
            if first:
                h = _ln(inp, sd, "head.%d.0.0" % s, LN_EPS_HEAD)
                h = _lbr(h, sd, "head.%d.0.1" % s, "head.%d.0.2" % s)
            else:
                h = _lbr(inp, sd, "head.%d.0.0" % s, "head.%d.0.1" % s)
            h = _lbr(h, sd, "head.%d.1.0" % s, "head.%d.1.1" % s)
            h = _lbr(h, sd, "head.%d.2.0" % s, "head.%d.2.1" % s)
            return _lin(h, sd, "head.%d.3" % s)

        x1 = stage(x, 0, True)
        x2 = stage(torch.cat([x1, x], dim=1), 1, False)
        x3 = stage(torch.cat([x2, x], dim=1), 2, False)
        return x3.reshape(b, -1, 3), [x1.reshape(b, -1, 3), x2.reshape(b, -1, 3)]
    if f["deep_head"]:                                                                      # :287-300, :517-519
        h = _ln(x.reshape(b, -1), sd, "head.0", LN_EPS_HEAD)
        h = _lbr(h, sd, "head.1", "head.2")
        h = _lbr(h, sd, "head.4", "head.5")
        h = _lbr(h, sd, "head.7", "head.8")
        return _lin(h, sd, "head.10").reshape(b, -1, 3)
    y = _lin(_ln(x, sd, "head.0", LN_EPS_HEAD), sd, "head.1")                               # :283-286, :521
    return y.reshape(b, -1, 3)                                                              # :523


def forward(sd: Dict[str, torch.Tensor], flags: Optional[dict], poses: Sequence[torch.Tensor],
            rays: Sequence[torch.Tensor], centers: Sequence[torch.Tensor],
            dtype: torch.dtype = torch.float32, taps: Optional[dict] = None, fpt_matmul_bf16: bool = False):
    """MultiView_MPL.forward, multiview_mpl.py:450-525.

    ``sd`` may carry the ``features.`` prefix of MultiView_MPL_G (multiview_mpl.py:552).
    Returns (B,J,3) -- or ((B,J,3), [x1,x2]) for the kadkhod head.

    ``fpt_matmul_bf16`` emulates MultiView_MPL.set_matmul_precision("bf16") with the engine's own rounding points
    (block_bf16): the four Linear layers of every FPT block see bf16-rounded operands, everything else is unchanged.
    """
    global _BF16_PREFIXES
    _BF16_PREFIXES = ("blocks.",) if fpt_matmul_bf16 else ()
    try:
        return _forward(sd, flags, poses, rays, centers, dtype, taps)
    finally:
        _BF16_PREFIXES = ()


def _forward(sd, flags, poses, rays, centers, dtype, taps):
    f = resolve_flags(flags)
    sd = {(k[len("features."):] if k.startswith("features.") else k): v.to(dtype) if v.is_floating_point() else v
          for k, v in sd.items()}
    poses = [torch.as_tensor(p).to(dtype) for p in poses]
    rays = [torch.as_tensor(r).to(dtype) for r in rays]
    centers = [torch.as_tensor(c).to(dtype) for c in centers]
    b = poses[0].shape[0]
    with torch.no_grad():
        xs = [view_token(poses[i], rays[i], centers[i], i, sd, f, taps) for i in range(len(poses))]
        xs = torch.cat(xs, dim=1)                                                           # :495
        if f["FPT_blocks_view_keypoint_tokens"]:                                            # :496-499
            xs = xs.reshape(b, len(poses) * f["num_joints"], -1)
        else:
            xs = xs.reshape(b, len(poses), -1)
        if taps is not None:
            taps["fpt_in"] = xs
        x = fusion_features(xs, sd, f)                                                      # :505
        if taps is not None:
            taps["fused"] = x
        return head(x, sd, f)


# --------------------------------------------------------------------------- helpers
def param_shapes(flags: Optional[dict]) -> Dict[str, tuple]:
    """Shape map of the reference state_dict for ``flags`` (multiview_mpl.py:134-317).

    Restated independently so that tests can check the boundary module's
    ``state_dict()`` names/shapes without importing the reference.
    """
    f = resolve_flags(flags)
    J, d, L, V = f["num_joints"], f["embed_dim_ratio"], f["depth"], f["num_views"]
    multi = f["multiple_spatial_blocks"]
    S: Dict[str, tuple] = {}

    def lin(name, o, i):
        S[name + ".weight"] = (o, i)
        S[name + ".bias"] = (o,)

    def ln(name, n):
        S[name + ".weight"] = (n,)
        S[name + ".bias"] = (n,)

    def bn(name, n):
        ln(name, n)
        S[name + ".running_mean"] = (n,)
        S[name + ".running_var"] = (n,)
        S[name + ".num_batches_tracked"] = ()

    def blk(name, D):
        ln(name + ".norm1", D)
        lin(name + ".attn.qkv", 3 * D, D)
        lin(name + ".attn.proj", D, D)
        ln(name + ".norm2", D)
        lin(name + ".mlp.fc1", int(D * 2.0), D)      # mlp_ratio is always 2.0 (:96, never plumbed)
        lin(name + ".mlp.fc2", D, int(D * 2.0))

    D_f = d * J * (2 if f["input_rays_as_token"] else 1)                                     # :140-142
    nin = 3 if f["confidence_input_as_third"] else 2
    views = range(V) if multi else [None]
    for v in views:
        sfx = "" if v is None else ".%d" % v
        lin("Spatial_patch_to_embedding" + sfx, d, nin)                                      # :159-168
        if f["add_confidence_input"] or f["mult_confidence_emb"]:
            lin("confidence_to_embedding" + sfx, d, 1)                                       # :180-184
        S["Spatial_pos_embed" + sfx] = (1, J, d)                                             # :192-195
        if not f["no_transformer_spt"]:
            for l in range(L):                                                               # :236-249
                blk("Spatial_blocks%s.%d" % (sfx, l), d)
    if f["add_3D_pos_encoding_to_rays"]:                                                     # :209-215
        lin("pos_3d_linear", d if f["add_3D_pos_encoding_in_Spatial"] else 2 * d, 3)
        S["pos_3d_embed"] = (1, J, 2 * d)
        S["pos_3d_view_coding"] = (1, J, 2 * d)
    else:                                                                                    # :216-219
        lin("pos_3d_linear", d, 3)
        S["pos_3d_embed"] = (1, J, d)
        S["pos_3d_view_coding"] = (1, J, d)
    if f["input_rays_as_token"]:
        lin("ray_to_embedding", d, 3)                                                        # :224-225
    if f["confidence_in_FPT"]:
        lin("confidence_to_embedding_FPT", d, 1)                                             # :228-229
    if not f["no_transformer_fpt"]:
        Db = d if f["FPT_blocks_view_keypoint_tokens"] else D_f                              # :255-266
        for l in range(L):
            blk("blocks.%d" % l, Db)
    ln("Spatial_norm", d)                                                                    # :271
    E = d * J                                                                                # :272-274
    ln("View_norm", E)
    if f["linear_weighted_mean"]:
        lin("weighted_mean", E, V * E)                                                       # :279
    else:
        S["weighted_mean.weight"] = (1, V, 1)                                                # :281
        S["weighted_mean.bias"] = (1,)
    Hd, O = f["hidden_dim"], J * 3
    if f["head_kadkhod"]:                                                                    # :301-317
        for s in range(3):
            if s == 0:
                ln("head.0.0.0", E); lin("head.0.0.1", Hd, E); bn("head.0.0.2", Hd)
            else:
                lin("head.%d.0.0" % s, Hd, O + E); bn("head.%d.0.1" % s, Hd)
            lin("head.%d.1.0" % s, Hd, Hd); bn("head.%d.1.1" % s, Hd)
            lin("head.%d.2.0" % s, Hd, Hd); bn("head.%d.2.1" % s, Hd)
            lin("head.%d.3" % s, O, Hd)
    elif f["deep_head"]:                                                                     # :287-300
        ln("head.0", E); lin("head.1", Hd, E); bn("head.2", Hd)
        lin("head.4", Hd, Hd); bn("head.5", Hd)
        lin("head.7", Hd, Hd); bn("head.8", Hd)
        lin("head.10", O, Hd)
    else:                                                                                    # :283-286
        ln("head.0", E); lin("head.1", O, E)
    return S


def flop_count(flags: Optional[dict], num_views: Optional[int] = None) -> float:
    """Matmul-only FLOPs (2*MAC) per pose: block = 16*N*D^2 + 4*N^2*D."""
    f = resolve_flags(flags)
    V = num_views or f["num_views"]
    J, d, L = f["num_joints"], f["embed_dim_ratio"], f["depth"]

    def blk(N, D):
        return 16.0 * N * D * D + 4.0 * N * N * D

    n_spt = 0 if f["no_transformer_spt"] else len(block_schedule(L, f["confidence_as_attention_uncertainty_weight"]))
    n_fpt = 0 if f["no_transformer_fpt"] else len(block_schedule(L))
    D_f = d * J * (2 if f["input_rays_as_token"] else 1)
    total = V * n_spt * blk(J, d)
    if f["FPT_blocks_view_keypoint_tokens"]:
        total += n_fpt * blk(J * V, d)
    else:
        total += n_fpt * blk(V, D_f)
    nin = 3 if f["confidence_input_as_third"] else 2
    total += V * J * 2.0 * nin * d + 2.0 * (d * J) * (J * 3)
    if f["input_rays_as_token"]:
        total += V * J * 2.0 * 3 * d
    return total


def mpjpe(a: torch.Tensor, b: torch.Tensor) -> float:
    """mean_{b,j} ||a-b||_2  (loss.py:57 / evaluate.py:100 applied between two outputs)."""
    return float(torch.linalg.norm(a.double() - b.double(), dim=-1).mean())


def rel_errors(out: torch.Tensor, ref: torch.Tensor):
    """(max-scaled, norm-wise) relative errors, SURVEY.md section 8c 'Tolerance definition'."""
    o, r = out.double(), ref.double()
    mx = float((o - r).abs().max() / r.abs().max().clamp_min(1e-30))
    nw = float(torch.linalg.norm(o - r) / torch.linalg.norm(r).clamp_min(1e-30))
    return mx, nw
