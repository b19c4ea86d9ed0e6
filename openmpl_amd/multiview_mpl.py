"""Drop-in model module for OpenMPL's ``MPL/lib/models/multiview_mpl.py`` on MI355X.

Same Python contract as the reference:

  * ``get_multiview_mpl_net(cfg, is_train, **kwargs)``             reference :649-654
  * ``MultiView_MPL_G(cfg).forward(x, centers=None, rays=None)``   reference :528-585
  * ``MultiView_MPL(**flags).forward(poses, rays=None, centers=None)``  reference :94-525
  * identical parameter names/shapes, so ``state_dict()`` / ``load_state_dict()`` /
    ``.parameters()`` / ``DataParallel`` work on reference checkpoints unchanged.

The modules below are *parameter containers* only: all arithmetic of the forward pass runs in
the hand-written gfx950 kernels of ``libmpl_hip.so`` through the C ABI (``cabi.py``).  There is no
eager / CPU fallback -- a missing library, a CPU tensor or an unsupported flag combination raises.
The forward is inference-only (the reference's validate() path, function_mpl.py:331-350).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from functools import partial
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import cabi

logger = logging.getLogger(__name__)

# Walking ~300 parameters through nn.Module.__getattr__ costs ~0.6 ms per list, which a forward would pay three times -- more
# than the kernels of a single frame take.  The lists are cached per module and invalidated by a process-wide generation
# counter that torch bumps whenever ANY module registers a parameter, buffer or submodule (assignment of a new Parameter,
# load into a fresh module, ...); in-place updates keep the objects and are caught by data_ptr / _version as before.
_STRUCT_GEN = [0]


def _bump_struct_gen(*_a, **_k):
    _STRUCT_GEN[0] += 1
    return None


_HOOKS_OK = True
try:
    from torch.nn.modules import module as _tnm
    _tnm.register_module_parameter_registration_hook(_bump_struct_gen)
    _tnm.register_module_buffer_registration_hook(_bump_struct_gen)
    _tnm.register_module_module_registration_hook(_bump_struct_gen)
except Exception:                       # an older torch without the global hooks: no caching
    _HOOKS_OK = False


class _Container(nn.Module):
    """Holds parameters under the reference's attribute names; not callable on its own."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("%s is a parameter container; the fused HIP forward of MultiView_MPL owns the "
                           "arithmetic" % type(self).__name__)


class Mlp(_Container):
    """Parameters of reference Mlp (:21-37): fc1, fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(_Container):
    """Parameters of reference Attention (:40-67): qkv, proj."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class Block(_Container):
    """Parameters of reference Block (:70-92): norm1, attn, norm2, mlp."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = nn.Identity()      # DropPath has no parameters and is identity in eval (:79)
        self.drop_path_rate = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _ver(t: torch.Tensor) -> int:
    """Version counter of a tensor; inference tensors (created under torch.inference_mode) do not track one -- and cannot be
    updated in place outside that mode either, so 0 is the right key for them."""
    try:
        return t._version
    except RuntimeError:
        return 0


# ---------------------------------------------------------------------------------------------- the forward as ONE torch operator
# north_star: "hand-written HIP ... exposed as a torch extension".  The library is reached through ctypes (cabi.py); this
# registration makes the same call a dispatcher-visible operator, openmpl_amd::forward, without another build: torch.profiler
# shows it, torch.library.opcheck can test it, and torch.compile(fullgraph=True) of a caller traces through it as ONE opaque
# node with the meta function below.  The module is not a tensor: it travels as an integer handle into a registry of live
# modules (weak references).  The parameters are captured state of the operator, like the weights of a compiled graph.
import weakref as _weakref

_LIVE_MODULES = _weakref.WeakValueDictionary()
_NEXT_HANDLE = [1]


def _module_of(handle: int) -> "MultiView_MPL":
    m = _LIVE_MODULES.get(int(handle))
    if m is None:
        raise RuntimeError("openmpl_amd::forward: module handle %d is not alive" % handle)
    return m


@torch.library.custom_op("openmpl_amd::forward", mutates_args=(), device_types="cuda")
def _forward_op(handle: int, poses: List[torch.Tensor], rays: List[torch.Tensor], centers: List[torch.Tensor]) -> List[torch.Tensor]:
    """[poses (B,17,3)] of MultiView_MPL.forward (reference :450-525); with head_kadkhod [x3, x1, x2].  rays / centers: V tensors
    or an empty list (= None)."""
    m = _module_of(handle)
    out = m._forward_impl(list(poses), list(rays) or None, list(centers) or None)
    if isinstance(out, tuple):
        return [out[0]] + list(out[1])
    return [out]


@_forward_op.register_fake
def _forward_op_fake(handle, poses, rays, centers):
    m = _module_of(handle)
    B = poses[0].shape[0]
    n = 3 if m.head_kadkhod else 1
    return [poses[0].new_empty((B, m.num_joints, 3), dtype=torch.float32) for _ in range(n)]


_EXT_LIFT = [None]        # torch.ops.openmpl_amd.lift once the extension is loaded (openmpl_amd/torch_ext.py)


def _unbind(handle: int):
    try:
        if _EXT_LIFT[0] is not None:
            torch.ops.openmpl_amd.unbind(int(handle))
        from . import torch_ext
        torch_ext._FAKE_SHAPES.pop(int(handle), None)
    except Exception:          # interpreter shutdown: the extension may already be gone
        pass


def _release_binding(ent):
    if ent is not None and ent.get("bind") is not None:
        _unbind(ent.pop("bind"))


class MultiView_MPL(nn.Module):
    """Constructor surface and parameter layout of the reference ``MultiView_MPL`` (:94-317)."""

    def __init__(self, num_joints=17, in_chans=2, embed_dim_ratio=32, depth=4, num_heads=8, mlp_ratio=2.0,
                 qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2,
                 norm_layer=None, num_views=5,
                 add_confidence_input=False, mult_confidence_emb=False, concat_confidence_emb=False,
                 confidence_input_as_third=False, pose_3d_emb_learnable=False, linear_weighted_mean=False,
                 pos_embedding_type="learnable", add_3D_pos_encoding_in_Spatial=False, input_rays_as_token=False,
                 add_3D_pos_encoding_to_rays=False, confidence_as_attention_uncertainty_weight=False,
                 multiple_spatial_blocks=False, no_transformer_spt=False, no_transformer_fpt=False,
                 confidence_in_FPT=False, deep_head=False, head_kadkhod=False, hidden_dim=1024,
                 FPT_blocks_view_keypoint_tokens=False):
        super().__init__()
        if qk_scale is not None or not qkv_bias or float(mlp_ratio) != 2.0 or in_chans != 2:
            raise NotImplementedError("HIP path implements the reference's fixed qkv_bias=True, qk_scale=None, "
                                      "mlp_ratio=2.0, in_chans=2 (multiview_mpl.py:552-580 never overrides them)")
        self.num_joints = num_joints
        self.num_views = num_views
        self.embed_dim_ratio = embed_dim_ratio
        self.depth = depth
        self.num_heads = num_heads
        self.drop_rate = float(drop_rate)
        self.attn_drop_rate = float(attn_drop_rate)
        self.drop_path_rate = float(drop_path_rate)
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)                      # :139
        embed_dim = embed_dim_ratio * num_joints                                        # :140
        if input_rays_as_token:
            embed_dim = embed_dim_ratio * 2 * num_joints                                # :142
        out_dim = num_joints * 3

        self.deep_head = deep_head
        self.head_kadkhod = head_kadkhod
        self.hidden_dim = hidden_dim
        self.confidence_input_as_third = confidence_input_as_third
        self.multiple_spatial_blocks = multiple_spatial_blocks
        self.no_transformer_spt = no_transformer_spt
        self.no_transformer_fpt = no_transformer_fpt
        self.FPT_blocks_view_keypoint_tokens = FPT_blocks_view_keypoint_tokens

        n_in = in_chans + 1 if confidence_input_as_third else in_chans                  # :159-168
        if multiple_spatial_blocks:
            self.Spatial_patch_to_embedding = nn.ModuleList(
                [nn.Linear(n_in, embed_dim_ratio) for _ in range(num_views)])
        else:
            self.Spatial_patch_to_embedding = nn.Linear(n_in, embed_dim_ratio)

        self.add_confidence_input = add_confidence_input
        self.mult_confidence_emb = mult_confidence_emb
        self.concat_confidence_emb = concat_confidence_emb
        if self.concat_confidence_emb:                                                  # :173-176
            self.add_confidence_input = False
            self.mult_confidence_emb = False
            self.concat_confidence_emb = False
        self.confidence_to_embedding = None
        if self.add_confidence_input or self.mult_confidence_emb:                       # :179-184
            if multiple_spatial_blocks:
                self.confidence_to_embedding = nn.ModuleList(
                    [nn.Linear(1, embed_dim_ratio) for _ in range(num_views)])
            else:
                self.confidence_to_embedding = nn.Linear(1, embed_dim_ratio)
        self.confidence_as_attention_uncertainty_weight = confidence_as_attention_uncertainty_weight

        if multiple_spatial_blocks:                                                     # :192-195
            self.Spatial_pos_embed = nn.ParameterList(
                [nn.Parameter(torch.zeros(1, num_joints, embed_dim_ratio)) for _ in range(num_views)])
        else:
            self.Spatial_pos_embed = nn.Parameter(torch.zeros(1, num_joints, embed_dim_ratio))
        self.pos_drop = nn.Dropout(p=drop_rate)

        self.pose_3d_emb_learnable = pose_3d_emb_learnable
        self.add_3D_pos_encoding_in_Spatial = add_3D_pos_encoding_in_Spatial
        self.add_3D_pos_encoding_to_rays = add_3D_pos_encoding_to_rays
        if add_3D_pos_encoding_to_rays:                                                 # :209-215
            self.pos_3d_linear = nn.Linear(3, embed_dim_ratio if add_3D_pos_encoding_in_Spatial
                                           else embed_dim_ratio * 2)
            self.pos_3d_embed = nn.Parameter(torch.zeros(1, num_joints, embed_dim_ratio * 2))
            self.pos_3d_view_coding = nn.Parameter(torch.zeros(1, num_joints, embed_dim_ratio * 2))
        else:                                                                           # :216-219
            self.pos_3d_linear = nn.Linear(3, embed_dim_ratio)
            self.pos_3d_embed = nn.Parameter(torch.zeros(1, num_joints, embed_dim_ratio))
            self.pos_3d_view_coding = nn.Parameter(torch.zeros(1, num_joints, embed_dim_ratio))

        self.input_rays_as_token = input_rays_as_token
        if input_rays_as_token:                                                         # :224-225
            self.ray_to_embedding = nn.Linear(3, embed_dim_ratio)
        self.confidence_in_FPT = confidence_in_FPT
        if confidence_in_FPT:                                                           # :228-229
            self.confidence_to_embedding_FPT = nn.Linear(1, embed_dim_ratio)

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]              # :233
        mk = lambda dim, i: Block(dim=dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                  qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i],
                                  norm_layer=norm_layer)
        if multiple_spatial_blocks:                                                     # :236-249
            self.Spatial_blocks = nn.ModuleList(
                [nn.ModuleList([mk(embed_dim_ratio, i) for i in range(depth)]) for _ in range(num_views)])
        else:
            self.Spatial_blocks = nn.ModuleList([mk(embed_dim_ratio, i) for i in range(depth)])
        if no_transformer_spt:                                                          # :251-252
            self.Spatial_blocks = nn.ModuleList([])

        fpt_dim = embed_dim_ratio if FPT_blocks_view_keypoint_tokens else embed_dim     # :255-266
        self.blocks = nn.ModuleList([mk(fpt_dim, i) for i in range(depth)])
        if no_transformer_fpt:                                                          # :268-269
            self.blocks = nn.ModuleList([])

        self.Spatial_norm = norm_layer(embed_dim_ratio)                                 # :271
        if input_rays_as_token:
            embed_dim = embed_dim // 2
        self.View_norm = norm_layer(embed_dim)                                          # :274
        self.linear_weighted_mean = linear_weighted_mean
        if linear_weighted_mean:                                                        # :277-281
            self.weighted_mean = nn.Linear(num_views * embed_dim, embed_dim)
        else:
            self.weighted_mean = nn.Conv1d(in_channels=num_views, out_channels=1, kernel_size=1)

        self.head = nn.Sequential(nn.LayerNorm(embed_dim), nn.Linear(embed_dim, out_dim))  # :283-286
        if deep_head:                                                                   # :287-300
            h = hidden_dim
            self.head = nn.Sequential(
                nn.LayerNorm(embed_dim), nn.Linear(embed_dim, h), nn.BatchNorm1d(h), nn.ReLU(),
                nn.Linear(h, h), nn.BatchNorm1d(h), nn.ReLU(),
                nn.Linear(h, h), nn.BatchNorm1d(h), nn.ReLU(), nn.Linear(h, out_dim))
        if head_kadkhod:                                                                # :301-317
            h = hidden_dim

            def lbr(i, o):
                return nn.Sequential(nn.Linear(i, o), nn.BatchNorm1d(o), nn.ReLU(True))

            first = nn.Sequential(nn.Sequential(nn.LayerNorm(embed_dim), nn.Linear(embed_dim, h), nn.BatchNorm1d(h),
                                                nn.ReLU(True)), lbr(h, h), lbr(h, h), nn.Linear(h, out_dim))
            rest = [nn.Sequential(lbr(out_dim + embed_dim, h), lbr(h, h), lbr(h, h), nn.Linear(h, out_dim))
                    for _ in range(2)]
            self.head = nn.ModuleList([first] + rest)

        self._unsupported = self._find_unsupported()
        self.__dict__["_op_handle"] = 0                     # plain attributes: not part of state_dict / repr; see _handle()
        self.__dict__["_use_torch_op"] = "auto"
        self.__dict__["_small_batch_engine"] = "auto"
        self._hip_cache = {}
        self._dp_replica = False
        self._dp_src = None
        self.matmul_precision = "fp32"
        self.__dict__["_fast_bind"] = {}                    # device index -> (binding handle, _STRUCT_GEN, identity probes)

    # ------------------------------------------------------------------ support matrix
    def _find_unsupported(self) -> Optional[str]:
        """Flag combinations the HIP path does not implement yet (they raise, they never fall back)."""
        if self.num_joints != 17 or self.embed_dim_ratio != 32 or self.num_heads != 8:
            return "HIP kernels are specialised for NUM_JOINTS=17, DIM=32, HEADS=8 (every shipped yaml)"
        if self.num_views > cabi.MPL_MAX_VIEWS:
            return "num_views > %d" % cabi.MPL_MAX_VIEWS
        if self.FPT_blocks_view_keypoint_tokens and self.input_rays_as_token:
            return "FPT_blocks_view_keypoint_tokens with input_rays_as_token (the reference itself fails: LN(32) on 64)"
        if self.FPT_blocks_view_keypoint_tokens and self.num_joints * self.num_views * (self.embed_dim_ratio // self.num_heads) * 8 > 64 * 1024:
            return "joints x views grid too long for the LDS-resident K/V of one head"
        if self.add_3D_pos_encoding_to_rays and not self.input_rays_as_token:
            return "add_3D_pos_encoding_to_rays without input_rays_as_token (the reference itself fails, :483)"
        if self.add_3D_pos_encoding_to_rays and self.add_3D_pos_encoding_in_Spatial:
            return "add_3D_pos_encoding_to_rays together with add_3D_pos_encoding_in_Spatial"
        return None

    def set_matmul_precision(self, precision: str):
        """Arithmetic of the FPT block GEMMs (everything else is fp32 always):
        "fp32" (default) -- fp32 in, fp32 out, fp32 accumulation; where the FPT width is a multiple of 136 (every
            view-token model) the products are formed on the fp16 matrix cores from operands split into two fp16 terms under
            exact power-of-two scales, three partial products per product (csrc/h2_gemm.hip: as accurate as an fp32 GEMM,
            1/5 of its matrix-pipe time on gfx950); other widths (KPTOK, D = 32) use the native fp32 MFMA kernels;
        "fp32_mfma" -- native fp32 matrix instructions (v_mfma_f32_16x16x4_f32) everywhere;
        "bf16" -- the same stage with ONE bf16 per operand element (csrc/b1_gemm.hip): operands rounded to bf16 (activations
            when a GEMM epilogue hands them to the next GEMM, weights with the LayerNorm gain folded in), exact products, fp32
            accumulation; statistics, softmax, GELU and the residual stream stay fp32: BASELINE.json configs[2].
        The packed weight copies are derived data, rebuilt whenever a parameter's storage or version changes."""
        if precision not in ("fp32", "fp32_mfma", "bf16"):
            raise ValueError("matmul precision must be 'fp32', 'fp32_mfma' or 'bf16'")
        if precision == "bf16" and not self._x3_supported():
            raise NotImplementedError("the %s engine covers the view-token FPT blocks (widths 544 / 1088, up to 32 views)" % precision)
        self.matmul_precision = precision
        self._drop_caches()
        return self

    def _x3_supported(self) -> bool:
        """The packed-operand engines (fp16x2 / bf16) need every FPT Linear shape to have a packed layout (out features a
        multiple of 136, in features of 544: widths 544 and 1088, i.e. every view-token model at DIM 32;
        mpl_pack_bf16_bytes / mpl_pack_h2_bytes decide) and fuse the attention for up to 32 tokens per sequence."""
        c = self.__dict__.get("_x3_ok")
        if c is not None and c[0] == _STRUCT_GEN[0] and _HOOKS_OK:
            return c[1]
        ok = False
        if not (self.no_transformer_fpt or len(self.blocks) == 0 or self.FPT_blocks_view_keypoint_tokens or self.num_views > 32):
            lib = cabi.load()
            b = self.blocks[0]
            ok = all(lib.mpl_pack_bf16_bytes(int(t.shape[0]), int(t.shape[1])) > 0 and
                     lib.mpl_pack_h2_bytes(int(t.shape[0]), int(t.shape[1])) > 0
                     for t in (b.attn.qkv.weight, b.attn.proj.weight, b.mlp.fc1.weight, b.mlp.fc2.weight))
        self.__dict__["_x3_ok"] = (_STRUCT_GEN[0], ok)
        return ok

    def _tensor_lists(self):
        """(every tensor handed to the library, the tensors of the FPT blocks, the tensors of the SPT blocks) of THIS module,
        cached until some module of the process registers a parameter / buffer / submodule (see _STRUCT_GEN).  Replicas of
        DataParallel are fresh objects with plain tensor attributes on every forward: never cached."""
        c = self.__dict__.get("_tl_cache")
        if c is not None and c[0] == _STRUCT_GEN[0] and _HOOKS_OK and not self._dp_replica:
            # torch.func.functional_call / stateless reparametrisation write module._parameters[...] directly (no hook): a few
            # identity probes (first / last FPT block, the head, the embedding) catch a swapped tensor set and rebuild the lists
            if all(a is b for a, b in zip(c[2], self._tl_probe())):
                return c[1]
        fpt = [t for b in self.blocks for t in self._block_ptrs(b)]
        stacks = self.Spatial_blocks if self.multiple_spatial_blocks else [self.Spatial_blocks]
        spt = [t for st in stacks for b in st for t in self._block_ptrs(b)]
        tl = (self._param_list(), fpt, spt)
        if not self._dp_replica:
            self.__dict__["_tl_cache"] = (_STRUCT_GEN[0], tl, self._tl_probe())
        return tl

    def _tl_probe(self):
        """A handful of live parameter objects (identity, not value) that any wholesale swap of the parameters would change."""
        out = [self.Spatial_norm.weight, self.View_norm.bias, self.weighted_mean.weight]
        if len(self.blocks) > 0:
            out += [self.blocks[0].attn.qkv.weight, self.blocks[-1].mlp.fc2.weight]
        return out

    # ------------------------------------------------------------------ nn.Module plumbing
    def __getstate__(self):
        """copy.deepcopy / pickle / torch.save of the MODULE: parameters, buffers and flags travel; everything derived from them
        -- structs of addresses, packed operands, their events (torch.cuda.Event cannot be pickled: a deepcopy of a model that
        had run a forward raised TypeError before round 6), extension bindings, the operator handle -- is rebuilt by the copy on
        its first forward."""
        st = self.__dict__.copy()
        st["_hip_cache"] = {}
        st["_fast_bind"] = {}
        st["_op_handle"] = 0
        st["_dp_src"] = None
        st["_dp_replica"] = False
        st.pop("_tl_cache", None)
        st.pop("_x3_ok", None)
        return st

    def _apply(self, fn, *a, **k):
        # .to() / .cuda() / .half(): under torch.__future__.set_overwrite_module_params_on_conversion the parameters are REPLACED
        # without any registration hook firing, so every cache that holds tensors goes, not only the packed operands
        self._drop_caches()
        self.__dict__.pop("_tl_cache", None)
        self.__dict__.pop("_x3_ok", None)
        return super()._apply(fn, *a, **k)

    def _drop_caches(self):
        """Forget every struct of addresses, derived operand and extension binding of this module (all devices)."""
        for ent in list(self.__dict__.get("_hip_cache", {}).values()):
            _release_binding(ent)
        self._hip_cache = {}
        self.__dict__["_fast_bind"] = {}

    def _replicate_for_data_parallel(self):
        """DataParallel replicas (valid_mpl.py:177-178) are shallow copies that torch re-creates on EVERY forward with freshly
        broadcast parameter storage.  The packed operands are derived from the parameter VALUES, which are the source module's:
        a replica shares the source's per-device cache (the shallow copy of __dict__ already aliases the dict) and keys the
        derived copies on the SOURCE parameters' storage + version, so a device packs once and every later forward only
        rebuilds the small struct of addresses (_marshal).  Round 3 re-packed everything per forward (1.2 ms of GPU time)."""
        r = super()._replicate_for_data_parallel()
        r._dp_replica = True
        r.__dict__["_fast_bind"] = {}               # replicas take the ctypes route (fresh parameter tensors every forward)
        # a plain attribute: nn.Module.__setattr__ would register the source module as a CHILD of its own replica
        r.__dict__["_dp_src"] = self._dp_src if self._dp_replica else self
        return r

    # ------------------------------------------------------------------ C-ABI argument marshalling
    def _config(self) -> cabi.Config:
        f = 0
        f |= cabi.F_MULTI_SPT if self.multiple_spatial_blocks else 0
        f |= cabi.F_CONF_ADD if self.add_confidence_input else 0
        f |= cabi.F_CONF_MULT if self.mult_confidence_emb else 0
        f |= cabi.F_CONF_ATTN_W if self.confidence_as_attention_uncertainty_weight else 0
        f |= cabi.F_POS3D_LEARN if self.pose_3d_emb_learnable else 0
        f |= cabi.F_POS3D_SPATIAL if self.add_3D_pos_encoding_in_Spatial else 0
        f |= cabi.F_RAYS_TOKEN if self.input_rays_as_token else 0
        f |= cabi.F_POS3D_TO_RAYS if self.add_3D_pos_encoding_to_rays else 0
        f |= cabi.F_NO_SPT if self.no_transformer_spt else 0
        f |= cabi.F_NO_FPT if self.no_transformer_fpt else 0
        f |= cabi.F_CONF_IN_FPT if self.confidence_in_FPT else 0
        f |= cabi.F_KPTOK if self.FPT_blocks_view_keypoint_tokens else 0
        return cabi.Config(self.num_joints, self.embed_dim_ratio, self.depth, self.num_heads, self.num_views,
                           3 if self.confidence_input_as_third else 2, f, 0)

    @staticmethod
    def _block_ptrs(blk: Block) -> List[torch.Tensor]:
        return [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight,
                blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                blk.mlp.fc2.weight, blk.mlp.fc2.bias]

    def _param_list(self) -> List[torch.Tensor]:
        """Every tensor whose address is handed to the library, in a fixed order."""
        out: List[torch.Tensor] = []
        n_sets = self.num_views if self.multiple_spatial_blocks else 1
        for s in range(n_sets):
            emb = self.Spatial_patch_to_embedding[s] if self.multiple_spatial_blocks else self.Spatial_patch_to_embedding
            out += [emb.weight, emb.bias]
            if self.confidence_to_embedding is not None:
                ce = self.confidence_to_embedding[s] if self.multiple_spatial_blocks else self.confidence_to_embedding
                out += [ce.weight, ce.bias]
            out.append(self.Spatial_pos_embed[s] if self.multiple_spatial_blocks else self.Spatial_pos_embed)
            if not self.no_transformer_spt:
                blks = self.Spatial_blocks[s] if self.multiple_spatial_blocks else self.Spatial_blocks
                for b in blks:
                    out += self._block_ptrs(b)
        for b in self.blocks:
            out += self._block_ptrs(b)
        out += [self.Spatial_norm.weight, self.Spatial_norm.bias, self.pos_3d_embed, self.pos_3d_view_coding,
                self.pos_3d_linear.weight, self.pos_3d_linear.bias, self.View_norm.weight, self.View_norm.bias,
                self.weighted_mean.weight, self.weighted_mean.bias]
        if self.input_rays_as_token:
            out += [self.ray_to_embedding.weight, self.ray_to_embedding.bias]
        if self.confidence_in_FPT:
            out += [self.confidence_to_embedding_FPT.weight, self.confidence_to_embedding_FPT.bias]
        # heads: parameters and (BatchNorm) running statistics, in module order
        out += [t for t in self.head.parameters()]
        out += [b for n, b in self.head.named_buffers() if n.endswith("running_mean") or n.endswith("running_var")]
        return out

    def _derived_key(self, h2, bf16, spt3, d32):
        """What the packed operands were built from: engine selection + storage and version of every SOURCE tensor they fold
        (norm1 / norm2, weights and biases of each block).  A replica looks at the module it was replicated from."""
        src = self._dp_src if self._dp_replica else self
        _, fpt, spt = src._tensor_lists()
        ts = (fpt if (h2 or bf16 or d32) else []) + (spt if spt3 else [])
        return (self.matmul_precision, h2, bf16, spt3, d32) + tuple([(t.data_ptr(), _ver(t)) for t in ts])

    def _marshal(self, device: torch.device):
        """Build (and cache per device) the mpl_weights struct.  Parameters are consumed in place, so the struct stays valid
        under in-place updates; it is rebuilt whenever any storage address changes.  The derived (packed) operands are kept
        as long as the tensors they were built from keep their storage and version (_derived_key)."""
        plist = self._tensor_lists()[0]
        x3ok = self._x3_supported()
        bf16 = self.matmul_precision == "bf16" and x3ok
        h2 = self.matmul_precision == "fp32" and x3ok
        # the SPT Linear layers also run from split operands (fp32 arithmetic on the fp16 matrix cores) unless the native
        # fp32 matrix instructions were asked for
        spt3 = self.matmul_precision != "fp32_mfma" and not self.no_transformer_spt
        # keypoint-token FPT blocks (width 32 = the SPT block's shapes) run from the same kind of split operand (mpl_d32_pack)
        d32 = self.matmul_precision == "fp32" and self.FPT_blocks_view_keypoint_tokens \
            and not self.no_transformer_fpt and len(self.blocks) > 0 and tuple(self.blocks[0].attn.qkv.weight.shape) == (96, 32)
        dkey = self._derived_key(h2, bf16, spt3, d32)
        key = tuple([t.data_ptr() for t in plist]) + dkey
        ent = self._hip_cache.get(device.index)
        if ent is not None and ent["key"] == key:
            return ent
        for t in plist:
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("MultiView_MPL (HIP): every parameter must be a contiguous float32 tensor on %s "
                                   "(found %s %s)" % (device, t.device, t.dtype))
        lib = cabi.load()
        st = torch.cuda.current_stream(device).cuda_stream
        # derived operands of this device: reused when only addresses changed (every DataParallel forward), else rebuilt
        derived = ent["derived"] if (ent is not None and ent["dkey"] == dkey) else None
        fresh = derived is None
        if fresh:
            derived = dict(spt={}, fpt={}, d32={})
        n_sets = self.num_views if self.multiple_spatial_blocks else 1
        L = 0 if self.no_transformer_spt else self.depth
        # device blob: [n_sets x mpl_spt_set][n_sets x L x mpl_block_weights]
        set_sz, blk_sz = C.sizeof(cabi.SptSet), C.sizeof(cabi.BlockWeights)
        blob = torch.empty(n_sets * set_sz + max(1, n_sets * L) * blk_sz, dtype=torch.uint8, device=device)
        base = blob.data_ptr()
        sets = (cabi.SptSet * n_sets)()
        blks = (cabi.BlockWeights * max(1, n_sets * L))()
        for s in range(n_sets):
            multi = self.multiple_spatial_blocks
            emb = self.Spatial_patch_to_embedding[s] if multi else self.Spatial_patch_to_embedding
            ce = None
            if self.confidence_to_embedding is not None:
                ce = self.confidence_to_embedding[s] if multi else self.confidence_to_embedding
            pe = self.Spatial_pos_embed[s] if multi else self.Spatial_pos_embed
            sets[s] = cabi.SptSet(_ptr(emb.weight), _ptr(emb.bias), _ptr(ce.weight if ce else None),
                                  _ptr(ce.bias if ce else None), _ptr(pe), base + n_sets * set_sz + s * L * blk_sz)
            if L:
                stack = self.Spatial_blocks[s] if multi else self.Spatial_blocks
                for l, b in enumerate(stack):
                    bwl = cabi.BlockWeights(*[_ptr(t) for t in self._block_ptrs(b)])
                    if spt3:
                        pk = derived["spt"].get((s, l))
                        if pk is None:
                            pk = torch.empty(lib.mpl_spt_pack_bytes(), dtype=torch.uint8, device=device)
                            cabi.check(lib.mpl_spt_pack(C.byref(bwl), pk.data_ptr(), st), "mpl_spt_pack")
                            derived["spt"][(s, l)] = pk
                        bwl.qkv_w3 = pk.data_ptr()
                    blks[s * L + l] = bwl
        host = bytes(sets) + bytes(blks)
        blob.copy_(torch.frombuffer(bytearray(host), dtype=torch.uint8))
        fpt = (cabi.BlockWeights * max(1, len(self.blocks)))()
        for l, b in enumerate(self.blocks):
            ptrs = [_ptr(t) for t in self._block_ptrs(b)]
            if bf16 or h2:
                ops = derived["fpt"].get(l)
                if ops is None:
                    ops = self._pack_block(lib, b, device, st, bf16)
                    derived["fpt"][l] = ops
                if h2:
                    ptrs += [0] * 8           # behind the *_w16 and *_w3 fields
                ptrs += [c3.data_ptr() for c3 in ops]
            fpt[l] = cabi.BlockWeights(*ptrs)
            if d32:
                pk = derived["d32"].get(l)
                if pk is None:
                    pk = torch.empty(lib.mpl_spt_pack_bytes(), dtype=torch.uint8, device=device)
                    cabi.check(lib.mpl_d32_pack(C.byref(fpt[l]), pk.data_ptr(), st), "mpl_d32_pack")
                    derived["d32"][l] = pk
                fpt[l].qkv_w3 = pk.data_ptr()
        w = cabi.Weights()
        w.spt_sets = base
        w.spt_packed = 1 if (spt3 and L > 0) else 0
        w.spatial_norm_w, w.spatial_norm_b = _ptr(self.Spatial_norm.weight), _ptr(self.Spatial_norm.bias)
        w.pos_3d_embed, w.pos_3d_view_coding = _ptr(self.pos_3d_embed), _ptr(self.pos_3d_view_coding)
        w.pos_3d_linear_w, w.pos_3d_linear_b = _ptr(self.pos_3d_linear.weight), _ptr(self.pos_3d_linear.bias)
        if self.input_rays_as_token:
            w.ray_embed_w, w.ray_embed_b = _ptr(self.ray_to_embedding.weight), _ptr(self.ray_to_embedding.bias)
        if self.confidence_in_FPT:
            w.conf_fpt_w = _ptr(self.confidence_to_embedding_FPT.weight)
            w.conf_fpt_b = _ptr(self.confidence_to_embedding_FPT.bias)
        w.fpt_blocks = C.cast(fpt, C.POINTER(cabi.BlockWeights))
        w.view_norm_w, w.view_norm_b = _ptr(self.View_norm.weight), _ptr(self.View_norm.bias)
        w.wmean_w, w.wmean_b = _ptr(self.weighted_mean.weight), _ptr(self.weighted_mean.bias)
        if not (self.deep_head or self.head_kadkhod):
            w.head_ln_w, w.head_ln_b = _ptr(self.head[0].weight), _ptr(self.head[0].bias)
            w.head_w, w.head_b = _ptr(self.head[1].weight), _ptr(self.head[1].bias)
        new = dict(key=key, dkey=dkey, derived=derived, weights=w, keep=(blob, fpt, sets, blks), cfg=self._config(), fpt_blocks=fpt,
                   packed_now=fresh)
        # the struct blob and the derived copies were enqueued on the stream current NOW: a later forward on another
        # stream must not read them before that work has finished
        new["ready"] = torch.cuda.Event()
        new["ready"].record(torch.cuda.current_stream(device))
        new["ready_stream"] = st
        if fresh:
            # the packing kernels ran on THIS stream: the event and the stream stay with the derived operands, which outlive the
            # struct when only addresses change (DataParallel replicas) -- a later forward on another stream waits for it too
            derived["ready"], derived["stream"] = new["ready"], st
        _release_binding(ent)                  # the extension binding of the entry this one replaces
        self._hip_cache[device.index] = new
        return new

    def _bind(self, ent, dev) -> int:
        """Register a marshalled entry with the torch extension (csrc/torch_ext.cpp openmpl_amd::bind): the structs as bytes, the
        parameter tensors they point into (the extension compares their data_ptr / _version on every call: a changed tensor makes
        lift() answer "stale" and the forward comes back here), and what must stay alive (device blob, derived operands)."""
        from . import torch_ext
        ext = torch_ext.ops()
        plist, fpt, spt = self._tensor_lists()
        h2, bf16, spt3, d32 = ent["dkey"][1:5]
        folded = set(id(t) for t in ((fpt if (h2 or bf16 or d32) else []) + (spt if spt3 else [])))
        versioned = [1 if id(t) in folded else 0 for t in plist]
        dv = ent["derived"]
        keep = [ent["keep"][0]] + list(dv["spt"].values()) + list(dv["d32"].values()) + [t for ops in dv["fpt"].values() for t in ops]
        cur = torch.cuda.current_stream(dev)
        if dv.get("stream", cur.cuda_stream) != cur.cuda_stream:       # operands packed on another stream: the binding's own event
            cur.wait_event(dv["ready"])                               # (recorded on the current stream in bind) must cover them
        if ent["ready_stream"] != cur.cuda_stream:
            cur.wait_event(ent["ready"])
        needs_rays = bool(self.input_rays_as_token or not self.pose_3d_emb_learnable)
        h = int(ext.bind(torch_ext.struct_bytes(ent["cfg"]), torch_ext.struct_bytes(ent["weights"]),
                         torch_ext.struct_bytes(ent["fpt_blocks"]), list(plist), versioned, keep, dev.index, needs_rays))
        torch_ext._FAKE_SHAPES[h] = self.num_joints
        ent["bind"] = h
        _weakref.finalize(self, _unbind, h)       # a module that dies without being cleared must not pin its parameters in C++
        return h

    def _forward_fast(self, poses, rays, centers):
        """The default tail through the extension operator openmpl_amd::lift: ONE dispatcher call per forward.  Returns None when
        this call has to take the general route (which also owns every error message): replicas, non-default tails, train mode,
        unsupported flag sets, inputs that are not on a GPU."""
        t0 = poses[0] if len(poses) else None
        if t0 is None or not t0.is_cuda or self.training:
            return None
        flags = 0 if self._small_engine_allowed() else cabi.F_NO_SMALL_STACK
        f = self._fast_bind.get(t0.device.index)
        if f is not None and f[1] == _STRUCT_GEN[0] and all(a is b for a, b in zip(f[2], self._tl_probe())):
            out = _EXT_LIFT[0](f[0], poses, rays or (), centers or (), flags)
            if out.dim() != 0:
                return out
        # no binding for this device yet, or a parameter moved / changed: marshal (the slow, complete check) and bind again
        if self._unsupported or self.linear_weighted_mean or self.deep_head or self.head_kadkhod or not _HOOKS_OK:
            return None
        dev = self.Spatial_norm.weight.device
        if dev != t0.device:
            return None
        from . import torch_ext
        if _EXT_LIFT[0] is None:
            _EXT_LIFT[0] = torch_ext.ops().lift
        with torch.cuda.device(dev):
            ent = self._marshal(dev)
            h = ent.get("bind")
            if h is None:
                h = self._bind(ent, dev)
        self._fast_bind[dev.index] = (h, _STRUCT_GEN[0], self._tl_probe())
        out = _EXT_LIFT[0](h, poses, rays or (), centers or (), flags)
        if out.dim() == 0:
            raise RuntimeError("openmpl_amd::lift reports a stale binding right after it was made (a parameter is being modified "
                               "concurrently with the forward)")
        return out

    @staticmethod
    def _pack_block(lib, b, device, st, bf16):
        """The four packed Linear operands {qkv (norm1 folded), proj, fc1 (norm2 folded), fc2} of one FPT block."""
        h2 = not bf16
        nbytes, pack = (lib.mpl_pack_bf16_bytes, lib.mpl_pack_bf16) if bf16 else (lib.mpl_pack_h2_bytes, lib.mpl_pack_h2)
        ops = []
        in_scale = 0        # h2: static scales (device vector) of the columns the NEXT plain Linear consumes
        for lin, ln in ((b.attn.qkv, b.norm1), (b.attn.proj, None), (b.mlp.fc1, b.norm2), (b.mlp.fc2, None)):
            n, k = lin.weight.shape
            c3 = torch.empty(nbytes(n, k), dtype=torch.uint8, device=device)
            if h2 and ln is None:
                # proj / fc2: packed against the per-column static scales their producer (qkv's v columns / fc1) applies
                cabi.check(lib.mpl_pack_h2_scaled(lin.weight.data_ptr(), lin.bias.data_ptr(), in_scale, n, k,
                                                  c3.data_ptr(), st), "pack operand")
            else:
                cabi.check(pack(lin.weight.data_ptr(), lin.bias.data_ptr(), _ptr(ln.weight if ln else None),
                                _ptr(ln.bias if ln else None), n, k, c3.data_ptr(), st), "pack operand")
                if h2:
                    so = lib.mpl_pack_h2_out_scale(c3.data_ptr(), n, k)
                    in_scale = so + 4 * (n - k) if lin is b.attn.qkv else so      # qkv: the last third of the columns (v)
            ops.append(c3)
        return ops

    def _check_inputs(self, poses, rays, centers):
        if len(poses) != self.num_views:
            # the reference fails the same way inside Conv1d(V,1,1) (SURVEY.md 8a "Flag validity")
            raise RuntimeError("expected %d views (num_views is a constructor constant), got %d"
                               % (self.num_views, len(poses)))
        dev = self.Spatial_norm.weight.device
        if dev.type != "cuda":
            raise RuntimeError("MultiView_MPL (HIP) has no CPU path: move the model to a GPU first (.cuda())")
        B = poses[0].shape[0]
        needs_rays = self.input_rays_as_token or not self.pose_3d_emb_learnable

        def prep(lst, shape, name, required):
            if lst is None:
                if required:
                    raise RuntimeError("%s are required by this flag set" % name)
                return [None] * self.num_views
            if len(lst) != self.num_views:
                raise RuntimeError("expected %d %s tensors, got %d" % (self.num_views, name, len(lst)))
            out = []
            for t in lst:
                if t.device != dev:
                    raise RuntimeError("%s tensor on %s but the model is on %s" % (name, t.device, dev))
                if tuple(t.shape) != shape:
                    raise RuntimeError("%s tensor has shape %s, expected %s" % (name, tuple(t.shape), shape))
                if t.dtype != torch.float32:
                    raise RuntimeError("%s tensor must be float32 (got %s)" % (name, t.dtype))
                out.append(t if t.is_contiguous() else t.contiguous())
            return out

        J = self.num_joints
        poses = prep(poses, (B, J, 3), "pose", True)
        rays = prep(rays, (B, J, 3), "ray", needs_rays)
        centers = prep(centers, (B, 1, 3), "center", needs_rays)
        return dev, B, poses, rays, centers

    # ------------------------------------------------------------------ forward (reference :450-525)
    def use_torch_op(self, mode="auto"):
        """How forward() reaches the kernels -- every route is the same C-ABI call on the same stream, bitwise the same poses:
        "auto" (default) -- the C++ operator openmpl_amd::lift of the torch extension (csrc/torch_ext.cpp: one dispatcher call,
            the parameter check, the allocations and mpl_forward in C++; visible to torch.profiler) for the default tail;
            replicas of DataParallel, the non-default tails (linear_weighted_mean / deep_head / head_kadkhod) and torch.compile
            traces take the routes below;
        True -- the Python-defined operator openmpl_amd::forward (torch.library.custom_op: every flag set, traceable by
            torch.compile(fullgraph=True) as one node; costs the dispatcher's Python round trip);
        False -- the direct ctypes call (no dispatcher at all)."""
        if mode not in (True, False, "auto"):
            raise ValueError("use_torch_op: True, False or 'auto'")
        self.__dict__["_use_torch_op"] = mode
        return self

    def set_small_batch_engine(self, mode="auto"):
        """Engine of the FPT block stack for up to 80 token rows (B x V <= 80 on 256 compute units: a single frame, a few persons), fp32 precision:
        True / "auto" -- the small-batch engine (csrc/sm_stack.hip: every GEMM on the whole chip, exact fp32 MFMA), 2x lower
        latency; False -- the team kernels of the large batches for EVERY batch size.  The two fp32 engines agree to ~1e-7 but
        not bit for bit, so with "auto" a pose of a batch of <= 80 / V poses does not carry the bits it would carry inside a
        larger batch.  Code that needs results independent of how a batch is split -- the last ragged batch of a validate()
        loop against a rerun, shards against the unsharded batch -- asks for False; openmpl_amd.dist.ShardedLifter and
        DataParallel replicas always run with False for that reason (a shard must equal the single-process result bitwise)."""
        if mode not in (True, False, "auto"):
            raise ValueError("set_small_batch_engine: True, False or 'auto'")
        self.__dict__["_small_batch_engine"] = mode
        return self

    @torch.compiler.assume_constant_result      # a torch.compile trace calls it ONCE, eagerly (the registry is a weak dictionary)
    def _handle(self) -> int:
        """Integer under which THIS object is registered for openmpl_amd::forward.  The handle is a plain __dict__ entry, so
        copy.deepcopy, pickle / torch.save of the module and DataParallel's shallow replicas all inherit the number of the
        module they were made from; it is therefore checked against the registry on every use and re-issued when it does not
        resolve to this very object (a copy must run its OWN weights through the operator route, never the original's)."""
        h = self.__dict__.get("_op_handle", 0)
        if not h or _LIVE_MODULES.get(h) is not self:
            h = _NEXT_HANDLE[0]
            _NEXT_HANDLE[0] += 1
            self.__dict__["_op_handle"] = h
            _LIVE_MODULES[h] = self
        return h

    def _small_engine_allowed(self) -> bool:
        return self._small_batch_engine in (True, "auto") and not self._dp_replica

    def forward(self, poses: Sequence[torch.Tensor], rays=None, centers=None):
        mode = self._use_torch_op
        if mode == "auto" and not self._dp_replica and not torch.compiler.is_compiling():
            out = self._forward_fast(poses, rays, centers)           # the C++ operator openmpl_amd::lift (csrc/torch_ext.cpp)
            if out is not None:
                return out
        if mode is True or (mode == "auto" and not self._dp_replica and
                            (torch.compiler.is_compiling() or torch.autograd.profiler._is_profiler_enabled)):
            out = torch.ops.openmpl_amd.forward(self._handle(), list(poses), list(rays) if rays is not None else [],
                                                list(centers) if centers is not None else [])
            return (out[0], [out[1], out[2]]) if self.head_kadkhod else out[0]
        return self._forward_impl(poses, rays, centers)

    def _forward_impl(self, poses: Sequence[torch.Tensor], rays=None, centers=None):
        if self._unsupported:
            raise NotImplementedError("MultiView_MPL (HIP): unsupported configuration: " + self._unsupported)
        if self.training:
            # train-mode semantics (Dropout, DropPath, BatchNorm batch statistics in the deep / kadkhod heads) are not
            # implemented, with or without autograd: only the eval forward of validate() exists here
            raise RuntimeError("MultiView_MPL (HIP) implements the inference forward only (eval-mode semantics): call "
                               ".eval() first (training loop is out of scope, SURVEY.md section 2 row 3)")
        lib = cabi.load()
        dev, B, poses, rays, centers = self._check_inputs(poses, rays, centers)
        with torch.cuda.device(dev):
            ent = self._marshal(dev)
            cfg = ent["cfg"]
            # engine selection of the block stack is a per-call decision (set_small_batch_engine; replicas never take the small engine)
            cfg.flags = (cfg.flags & ~cabi.F_NO_SMALL_STACK) | (0 if self._small_engine_allowed() else cabi.F_NO_SMALL_STACK)
            inp = cabi.Inputs()
            inp.batch = B
            for v in range(self.num_views):
                inp.poses[v] = poses[v].data_ptr()
                inp.rays[v] = _ptr(rays[v])
                inp.centers[v] = _ptr(centers[v])
            stream = torch.cuda.current_stream(dev).cuda_stream
            if stream != ent["ready_stream"]:
                torch.cuda.current_stream(dev).wait_event(ent["ready"])
            dv = ent["derived"]
            if dv.get("stream", stream) != stream:
                torch.cuda.current_stream(dev).wait_event(dv["ready"])
            if self.linear_weighted_mean or self.deep_head or self.head_kadkhod:
                return self._forward_staged(lib, ent, inp, B, dev, stream)
            ws_bytes = lib.mpl_forward_workspace_bytes(C.byref(cfg), B)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            out = torch.empty((B, self.num_joints, 3), dtype=torch.float32, device=dev)
            rc = lib.mpl_forward(C.byref(cfg), C.byref(ent["weights"]), C.byref(inp), out.data_ptr(), ws.data_ptr(),
                                 ws_bytes, stream)
            cabi.check(rc, "mpl_forward")
            # ws / out are allocated and consumed on the current stream, so the caching allocator's
            # stream-ordered reuse keeps them alive for the enqueued kernels without a record_stream.
        return out


    # ------------------------------------------------------------------ non-default tails (reference :441-443, :506-519)
    def _forward_staged(self, lib, ent, inp, B, dev, stream):
        """linear_weighted_mean / deep_head / head_kadkhod: the same SPT and FPT kernels as mpl_forward, then the
        tail composed from the C-ABI building blocks (mpl_view_norm / mpl_view_fuse / mpl_layernorm / mpl_linear)."""
        cfg, w = ent["cfg"], ent["weights"]
        J, V, E = self.num_joints, self.num_views, self.num_joints * self.embed_dim_ratio
        Df = lib.mpl_fpt_width(C.byref(cfg))
        f32 = dict(dtype=torch.float32, device=dev)
        xs = torch.empty((B * V, Df), **f32)
        cabi.check(lib.mpl_spt_tokens(C.byref(cfg), C.byref(w), C.byref(inp), xs.data_ptr(), stream), "mpl_spt_tokens")
        if len(self.blocks) > 0:
            kp = self.FPT_blocks_view_keypoint_tokens
            n_tok, dim = (V * J, self.embed_dim_ratio) if kp else (V, Df)
            order = [l for l in range(self.depth)] + [self.depth - 1]          # last block twice (:420-423)
            sched = (C.c_uint8 * len(order))(*order)
            ws_bytes = lib.mpl_block_stack_workspace_bytes(B, n_tok, dim)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            cabi.check(lib.mpl_block_stack_ex(xs.data_ptr(), B, n_tok, dim, self.num_heads, ent["fpt_blocks"], sched,
                                              len(order), ws.data_ptr(), ws_bytes, cfg.flags & cabi.F_NO_SMALL_STACK, stream),
                       "mpl_block_stack")
        y = torch.empty((B, E), **f32)
        if self.linear_weighted_mean:                                           # :441-443
            xn = torch.empty((B, V * E), **f32)
            cabi.check(lib.mpl_view_norm(C.byref(cfg), C.byref(w), xs.data_ptr(), B, xn.data_ptr(), stream), "mpl_view_norm")
            self._lin(lib, stream, xn, None, self.weighted_mean, None, False, y)
        else:                                                                   # :445
            cabi.check(lib.mpl_view_fuse(C.byref(cfg), C.byref(w), xs.data_ptr(), B, y.data_ptr(), stream), "mpl_view_fuse")

        def ln(x, mod):
            o = torch.empty_like(x)
            cabi.check(lib.mpl_layernorm(x.data_ptr(), x.shape[0], x.shape[1], mod.weight.data_ptr(), mod.bias.data_ptr(),
                                         float(mod.eps), o.data_ptr(), stream), "mpl_layernorm")
            return o

        def lin(xa, xb, mod, bn=None, relu=False):
            o = torch.empty((B, mod.out_features), **f32)
            self._lin(lib, stream, xa, xb, mod, bn, relu, o)
            return o

        if self.head_kadkhod:                                                   # :506-516
            def stage(seq, prev):
                first = seq[0]
                if prev is None:                                                # Sequential(LN, Linear, BN, ReLU)
                    h = lin(ln(y, first[0]), None, first[1], first[2], True)
                else:                                                           # Sequential(Linear, BN, ReLU) on cat([prev, x])
                    h = lin(prev, y, first[0], first[1], True)
                h = lin(h, None, seq[1][0], seq[1][1], True)
                h = lin(h, None, seq[2][0], seq[2][1], True)
                return lin(h, None, seq[3])
            x1 = stage(self.head[0], None)
            x2 = stage(self.head[1], x1)
            x3 = stage(self.head[2], x2)
            return x3.view(B, -1, 3), [x1.view(B, -1, 3), x2.view(B, -1, 3)]
        if self.deep_head:                                                      # :517-519
            hd = self.head
            h = lin(ln(y, hd[0]), None, hd[1], hd[2], True)
            h = lin(h, None, hd[4], hd[5], True)
            h = lin(h, None, hd[7], hd[8], True)
            return lin(h, None, hd[10]).view(B, -1, 3)
        return lin(ln(y, self.head[0]), None, self.head[1]).view(B, -1, 3)     # default head after linear_weighted_mean

    @staticmethod
    def _lin(lib, stream, xa, xb, mod, bn, relu, out):
        ka = xa.shape[1]
        kb = 0 if xb is None else xb.shape[1]
        if ka + kb != mod.in_features:
            raise RuntimeError("linear layer expects %d inputs, got %d" % (mod.in_features, ka + kb))
        cabi.check(lib.mpl_linear(xa.data_ptr(), ka, None if xb is None else xb.data_ptr(), kb, xa.shape[0],
                                  mod.weight.data_ptr(), mod.bias.data_ptr(), mod.out_features,
                                  None if bn is None else bn.weight.data_ptr(), None if bn is None else bn.bias.data_ptr(),
                                  None if bn is None else bn.running_mean.data_ptr(),
                                  None if bn is None else bn.running_var.data_ptr(),
                                  1e-5 if bn is None else float(bn.eps), 1 if relu else 0, out.data_ptr(), stream),
                   "mpl_linear")


class MultiView_MPL_G(nn.Module):
    """cfg -> kwargs wrapper, same mapping as the reference (:528-585)."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        ds, net = cfg.DATASET, cfg.NETWORK
        # :534-546
        if ds.TEST_DATASET.startswith("multiview_cmu_panoptic") or \
                ds.TEST_DATASET.startswith("multiview_amass_cmu_panoptic_mpl"):
            num_views = 5
        else:
            num_views = 4
        if ds.TRAIN_VIEWS is not None:
            num_views = len(ds.TRAIN_VIEWS)
            if ds.USE_HELPER_CAMERAS:
                assert ds.TRAIN_VIEWS_HELPER is not None
                num_views += len(ds.TRAIN_VIEWS_HELPER)
        if ds.TRAIN_ON_ALL_CAMERAS and ds.TEST_ON_ALL_CAMERAS:
            num_views = ds.N_VIEWS_TRAIN_TEST_ALL
        self.init_weights_from = net.INIT_WEIGHTS_FROM
        # :552-580
        self.features = MultiView_MPL(
            num_joints=net.NUM_JOINTS, embed_dim_ratio=net.DIM, depth=net.TRANSFORMER_DEPTH,
            num_heads=net.TRANSFORMER_HEADS, drop_rate=net.TRANSFORMER_DROP_RATE,
            attn_drop_rate=net.TRANSFORMER_ATTN_DROP_RATE, drop_path_rate=net.TRANSFORMER_DROP_PATH_RATE,
            num_views=num_views,
            add_confidence_input=net.TRANSFORMER_ADD_CONFIDENCE_INPUT,
            mult_confidence_emb=net.TRANSFORMER_MULT_CONFIDENCE_EMB,
            concat_confidence_emb=net.TRANSFORMER_CONCAT_CONFIDENCE_EMB,
            confidence_input_as_third=net.TRANSFORMER_CONFIDENCE_INPUT_AS_THIRD,
            pose_3d_emb_learnable=net.POSE_3D_EMB_LEARNABLE,
            linear_weighted_mean=net.TRANSFORMER_LINEAR_WEIGHTED_MEAN,
            add_3D_pos_encoding_in_Spatial=net.TRANSFORMER_ADD_3D_POS_ENCODING_IN_SPATIAL,
            input_rays_as_token=net.TRANSFORMER_INPUT_RAYS_AS_TOKEN,
            add_3D_pos_encoding_to_rays=net.TRANSFORMER_ADD_3D_POS_ENCODING_TO_RAYS,
            confidence_as_attention_uncertainty_weight=net.TRANSFORMER_CONF_ATTENTION_UNCERTAINTY_WEIGHT,
            multiple_spatial_blocks=net.TRANSFORMER_MULTIPLE_SPATIAL_BLOCKS,
            no_transformer_spt=net.TRANSFORMER_NO_SPT, no_transformer_fpt=net.TRANSFORMER_NO_FPT,
            confidence_in_FPT=net.TRANSFORMER_CONFIDENCE_IN_FPT,
            deep_head=net.TRANSFORMER_OUTPUT_HEAD_DEEP, head_kadkhod=net.TRANSFORMER_OUTPUT_HEAD_KADKHOD,
            hidden_dim=net.TRANSFORMER_OUTPUT_HEAD_HIDDEN_DIM,
            FPT_blocks_view_keypoint_tokens=net.TRANSFORMER_FPT_BLOCKS_VIEW_KEYPOINT_TOKENS)

    def forward(self, x, centers=None, rays=None):                                     # :583-585
        return self.features(x, rays=rays, centers=centers)

    # engine switches of the HIP path (not part of the reference's surface), forwarded to the model inside
    def set_matmul_precision(self, precision: str):
        self.features.set_matmul_precision(precision)
        return self

    def set_small_batch_engine(self, mode="auto"):
        self.features.set_small_batch_engine(mode)
        return self

    def use_torch_op(self, mode="auto"):
        self.features.use_torch_op(mode)
        return self

    def init_weights(self, pretrained=""):
        """Reference :587-646.  A checkpoint path containing a dataset name is loaded non-strictly;
        otherwise the effective initialisation is PyTorch's defaults."""
        if os.path.isfile(pretrained):
            names = ("multiview_h36m", "multiview_amass_h36m", "multiview_cmu_panoptic",
                     "multiview_amass_cmu_panoptic_mpl")
            if any(n in pretrained for n in names):
                logger.info("=> loading Pretrained model %s", pretrained)
                self.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=False)
            else:
                raise RuntimeError("COCO-pretrained initialisation targets modules this model does not have "
                                   "(reference :597 self.features.mlp_head)")
        else:
            logger.info("=> init weights: PyTorch defaults (the reference's scratch branch touches no module of "
                        "this model, :635-646)")


def get_multiview_mpl_net(cfg, is_train, **kwargs):
    """Factory with the reference's name and signature (:649-654)."""
    model = MultiView_MPL_G(cfg, **kwargs)
    if is_train and cfg.NETWORK.INIT_WEIGHTS:
        model.init_weights(cfg.NETWORK.PRETRAINED)
    return model
