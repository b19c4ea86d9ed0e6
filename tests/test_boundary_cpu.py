"""CPU-side checks of the drop-in boundary: parameter layout, factory, error behaviour, C ABI exports."""
import ctypes
import os
import re
import types

import pytest
import torch

from openmpl_amd import cabi, detrng
from openmpl_amd import build as mpl_build
from openmpl_amd.multiview_mpl import MultiView_MPL, MultiView_MPL_G, get_multiview_mpl_net
from oracle import mpl_oracle, ref_import
from tests.golden.cases import CASES, MICRO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", CASES + [MICRO], ids=lambda c: c["name"])
def test_state_dict_layout_matches_reference_layout(case):
    flags = case["flags"]
    try:
        m = MultiView_MPL(**flags)
    except NotImplementedError:
        pytest.skip("constructor-level rejection")
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = mpl_oracle.param_shapes(flags)
    assert got == want
    assert list(got) == list(m.state_dict().keys())


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present")
@pytest.mark.parametrize("name", ["chosen_v4_b8_l2", "full_v4_b8_l2", "kadkhod_v3_b3_l2", "deep_head_v3_b3_l2",
                                  "conf_add_v3_b3_l2", "linear_wmean_v3_b3_l2"])
def test_loads_live_reference_state_dict_strict_and_same_key_order(name):
    from tests.golden.cases import BY_NAME
    flags = BY_NAME[name]["flags"]
    ref = ref_import.build_reference(dict(flags, drop_path_rate=0.1))
    ours = MultiView_MPL(**flags)
    assert list(ref.state_dict().keys()) == list(ours.state_dict().keys())
    ours.load_state_dict(ref.state_dict(), strict=True)
    assert sum(p.numel() for p in ours.parameters()) == sum(p.numel() for p in ref.parameters())


def _cfg(**net):
    N = dict(NUM_JOINTS=17, DIM=32, TRANSFORMER_DEPTH=2, TRANSFORMER_HEADS=8, TRANSFORMER_DROP_RATE=0,
             TRANSFORMER_ATTN_DROP_RATE=0, TRANSFORMER_DROP_PATH_RATE=0.1, TRANSFORMER_ADD_CONFIDENCE_INPUT=False,
             TRANSFORMER_MULT_CONFIDENCE_EMB=False, TRANSFORMER_CONCAT_CONFIDENCE_EMB=False,
             TRANSFORMER_CONFIDENCE_INPUT_AS_THIRD=False, POSE_3D_EMB_LEARNABLE=True,
             TRANSFORMER_LINEAR_WEIGHTED_MEAN=False, TRANSFORMER_ADD_3D_POS_ENCODING_IN_SPATIAL=False,
             TRANSFORMER_INPUT_RAYS_AS_TOKEN=False, TRANSFORMER_ADD_3D_POS_ENCODING_TO_RAYS=False,
             TRANSFORMER_CONF_ATTENTION_UNCERTAINTY_WEIGHT=False, TRANSFORMER_MULTIPLE_SPATIAL_BLOCKS=False,
             TRANSFORMER_NO_SPT=False, TRANSFORMER_NO_FPT=False, TRANSFORMER_CONFIDENCE_IN_FPT=False,
             TRANSFORMER_OUTPUT_HEAD_DEEP=False, TRANSFORMER_OUTPUT_HEAD_KADKHOD=False,
             TRANSFORMER_OUTPUT_HEAD_HIDDEN_DIM=1024, TRANSFORMER_FPT_BLOCKS_VIEW_KEYPOINT_TOKENS=False,
             INIT_WEIGHTS=True, INIT_WEIGHTS_FROM="scratch", PRETRAINED="")
    N.update(net)
    D = dict(TEST_DATASET="multiview_h36m_mpl", TRAIN_VIEWS=[1, 3], USE_HELPER_CAMERAS=False,
             TRAIN_VIEWS_HELPER=None, TRAIN_ON_ALL_CAMERAS=False, TEST_ON_ALL_CAMERAS=False,
             N_VIEWS_TRAIN_TEST_ALL=4)
    return types.SimpleNamespace(NETWORK=types.SimpleNamespace(**N), DATASET=types.SimpleNamespace(**D))


def test_factory_num_views_rules():
    """multiview_mpl.py:534-546."""
    cfg = _cfg()
    assert get_multiview_mpl_net(cfg, is_train=False).features.num_views == 2
    cfg.DATASET.TRAIN_VIEWS = None
    assert MultiView_MPL_G(cfg).features.num_views == 4
    cfg.DATASET.TEST_DATASET = "multiview_cmu_panoptic_mpl"
    assert MultiView_MPL_G(cfg).features.num_views == 5
    cfg.DATASET.TRAIN_VIEWS = [3, 6]
    cfg.DATASET.USE_HELPER_CAMERAS = True
    cfg.DATASET.TRAIN_VIEWS_HELPER = [1, 2, 4]
    assert MultiView_MPL_G(cfg).features.num_views == 5
    cfg.DATASET.TRAIN_ON_ALL_CAMERAS = cfg.DATASET.TEST_ON_ALL_CAMERAS = True
    cfg.DATASET.N_VIEWS_TRAIN_TEST_ALL = 7
    assert MultiView_MPL_G(cfg).features.num_views == 7


def test_factory_state_dict_has_features_prefix_and_train_init_is_noop():
    m = get_multiview_mpl_net(_cfg(), is_train=True)
    assert all(k.startswith("features.") for k in m.state_dict())
    assert float(m.features.Spatial_pos_embed.abs().max()) == 0.0      # zero-init embeddings (:193-195)


def test_no_cpu_fallback_and_loud_errors():
    m = MultiView_MPL(num_views=2, depth=1, pose_3d_emb_learnable=True).eval()
    x = [torch.zeros(1, 17, 3) for _ in range(2)]
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU path"):
            m(x, rays=x, centers=[torch.zeros(1, 1, 3)] * 2)
        with pytest.raises(RuntimeError, match="expected 2 views"):
            m(x + x, rays=None, centers=None)
    m.train()
    with pytest.raises(RuntimeError, match="inference forward only"):
        m(x)
    with torch.no_grad(), pytest.raises(RuntimeError, match="inference forward only"):
        m(x)          # train mode is refused with autograd off too: Dropout / DropPath / BatchNorm batch statistics
    k = MultiView_MPL(num_views=2, depth=1, FPT_blocks_view_keypoint_tokens=True, input_rays_as_token=True).eval()
    with pytest.raises(NotImplementedError):
        k(x)
    with pytest.raises(RuntimeError, match="parameter container"):
        m.blocks[0](torch.zeros(1, 2, 544))


def test_shared_library_exports_every_symbol_declared_in_header():
    path = mpl_build.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "mpl_hip.h")).read()
    declared = set(re.findall(r"\b(mpl_[a-z_0-9]+)\s*\(", header))
    declared -= {"mpl_hip_error_string"} - {"mpl_hip_error_string"}
    assert declared == set(cabi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    lib.mpl_hip_abi_version.restype = ctypes.c_int
    assert lib.mpl_hip_abi_version() == cabi.ABI_VERSION
    lib.mpl_hip_error_string.restype = ctypes.c_char_p
    assert b"workspace" in lib.mpl_hip_error_string(-3)
    # the MPL_FORM_* codes of mpl_block_stack_form and the flag bits: the binding's constants are the header's
    forms = dict((n, int(v)) for n, v in re.findall(r"\bMPL_FORM_([A-Z0-9_]+)\s*=\s*(\d+)", header))
    assert forms and all(getattr(cabi, "FORM_" + n) == v for n, v in forms.items()) and set(cabi.FORM_KERNELS) == set(forms.values())
    assert int(re.search(r"#define\s+MPL_F_NO_SMALL_STACK\s+\(1u\s*<<\s*(\d+)\)", header).group(1)) == cabi.F_NO_SMALL_STACK.bit_length() - 1
    lib.mpl_block_stack_form.restype = ctypes.c_int
    assert lib.mpl_block_stack_form(0, 2, 544, 8, 13, 2, 0) < 0                   # invalid arguments are refused before any device query


def test_workspace_query_and_struct_sizes_without_gpu():
    lib = cabi.load()
    cfg = cabi.Config(17, 32, 12, 8, 4, 2, cabi.F_POS3D_LEARN, 0)
    assert lib.mpl_fpt_width(ctypes.byref(cfg)) == 544
    M, D = 1024 * 4, 544
    # fp32-MFMA engine: xs | qkv | att | hid | LN partials -- larger than the layouts of the packed-operand engines (fp16x2: xs |
    # att2 | hid2 at 4 B per element | partials | counters; bf16: xs | x16 | att1 | hid1 at 2 B per element | ...): sized for it
    want32 = M * D * 4 + M * 3 * D * 4 + M * D * 4 + M * 2 * D * 4 + M * 2 * (D // 136) * 4
    assert lib.mpl_forward_workspace_bytes(ctypes.byref(cfg), 1024) == want32
    cfg.flags |= cabi.F_RAYS_TOKEN
    assert lib.mpl_fpt_width(ctypes.byref(cfg)) == 1088
    assert ctypes.sizeof(cabi.BlockWeights) == 192 and ctypes.sizeof(cabi.SptSet) == 48
    assert ctypes.sizeof(cabi.Inputs) == 8 + 3 * 32 * 8
    # packed operands of the FPT GEMM engines: 18 KiB per (136-column group, stage) + the trailer (5 N + 8 floats); a stage is one
    # 32-deep k-tile of fp16 hi | lo (fp32 engine) or two k-tiles of bf16 (544 -> 17 k-tiles -> 9 stages); 0 = unsupported
    assert lib.mpl_pack_h2_bytes(1632, 544) == 12 * 17 * 18 * 1024 + (5 * 1632 + 8) * 4
    assert lib.mpl_pack_bf16_bytes(1632, 544) == 12 * 9 * 18 * 1024 + (5 * 1632 + 8) * 4
    assert lib.mpl_pack_bf16_bytes(544, 1088) == 4 * 17 * 18 * 1024 + (5 * 544 + 8) * 4 and lib.mpl_pack_bf16_bytes(544, 100) == 0
    assert lib.mpl_pack_bf16_bytes(96, 32) == 0 and lib.mpl_pack_bf16_bytes(544, 64) == 0    # K: a multiple of 544 (4 column groups of 136)


def test_detrng_is_stable():
    a = detrng.uniform(3, "features.blocks.0.attn.qkv.weight", (4, 5), -1, 1)
    assert abs(float(a[0, 0]) - float(detrng.uniform(3, "features.blocks.0.attn.qkv.weight", (4, 5), -1, 1)[0, 0])) == 0
    # pinned values: any change to the generator invalidates every golden fixture
    v = detrng.uniform01(0, "x", 3)
    assert [round(float(t), 12) for t in v] == [round(float(t), 12) for t in detrng.uniform01(0, "x", 5)[:3]]


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present")
def test_shipped_yaml_configs_build_the_reference_module_layout():
    """valid_mpl.py:162 path: every YAML under MPL/configs, loaded through the reference's own core/config.py
    (update_config, config.py:359-373; easydict stand-in), must construct a module with the reference's parameter
    names, order and shapes, the same num_views, and load the reference's state_dict strictly."""
    yamls = ref_import.shipped_yamls()
    assert len(yamls) == 4
    ref_mod = ref_import.load_reference_module()
    seen = set()
    for y in yamls:
        cfg = ref_import.load_reference_config(y)
        assert cfg.MODEL == "multiview_mpl"
        ref = ref_mod.get_multiview_mpl_net(cfg, is_train=False)
        ours = get_multiview_mpl_net(cfg, is_train=False)
        rs, os_ = ref.state_dict(), ours.state_dict()
        assert list(rs.keys()) == list(os_.keys()), y
        assert [tuple(v.shape) for v in rs.values()] == [tuple(v.shape) for v in os_.values()], y
        assert ours.features.num_views == ref.features.num_views
        assert ours.features.depth == cfg.NETWORK.TRANSFORMER_DEPTH
        ours.load_state_dict(rs, strict=True)
        assert ours.features._unsupported is None, (y, ours.features._unsupported)
        seen.add((ours.features.num_views, ours.features.depth, ours.features.input_rays_as_token))
    # h36m.yaml: CHOSEN depth 12, 2 views; hm_0_...: FULL depth 12, 4 views; cmu.yaml: CHOSEN depth 2; cmu_0_...: FULL depth 2
    assert seen == {(2, 12, False), (4, 12, True), (2, 2, False), (2, 2, True)}, seen


def test_tools_and_bench_compile():
    """The measurement scripts under tools/ and bench.py are part of the evidence chain: they must at least parse."""
    import glob
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]:
        py_compile.compile(f, doraise=True)


def test_cached_parameter_lists_follow_reassignment():
    """The lists of tensors handed to the library are cached per module (walking ~300 parameters through nn.Module.__getattr__
    cost more than the kernels of a single frame) and must be invalidated by ANYTHING that rebinds a parameter object: plain
    assignment, load_state_dict(assign=True), a new submodule.  In-place updates keep the objects (data_ptr / _version cover them)."""
    import torch
    from openmpl_amd.multiview_mpl import MultiView_MPL
    m = MultiView_MPL(num_views=2, depth=2, pose_3d_emb_learnable=True).eval()
    a = m._tensor_lists()
    assert m._tensor_lists() is a, "an unchanged module must serve the cached lists"
    old = m.blocks[0].attn.qkv.weight
    m.blocks[0].attn.qkv.weight = torch.nn.Parameter(torch.zeros_like(old))
    b = m._tensor_lists()
    assert b is not a and any(t is m.blocks[0].attn.qkv.weight for t in b[1]) and not any(t is old for t in b[1])
    with torch.no_grad():
        m.blocks[1].mlp.fc1.weight.mul_(2.0)                 # in place: same objects, the version moves
    assert m._tensor_lists() is b
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)
    c = m._tensor_lists()
    assert c is not b and all(any(t is p for p in m.parameters()) for t in c[1])
    m.head[1] = torch.nn.Linear(544, 51)
    assert m._tensor_lists() is not c
    # a DataParallel replica is a fresh object with plain tensor attributes: never served from (or stored in) a cache
    r = m._replicate_for_data_parallel()
    assert r._dp_replica and "_tl_cache" not in r.__dict__ or r.__dict__.get("_tl_cache") is m.__dict__.get("_tl_cache")


def test_forward_is_a_registered_torch_operator_with_a_meta_function():
    """north_star "exposed as a torch extension": openmpl_amd::forward is registered with the dispatcher; its fake (meta)
    implementation gives shape / dtype / device without a GPU, so torch.compile can trace a caller through it as one node."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    assert hasattr(torch.ops.openmpl_amd, "forward")
    for kw, n_out in ((dict(), 1), (dict(head_kadkhod=True, hidden_dim=64), 3)):
        m = MultiView_MPL(num_views=3, depth=1, **kw).eval()
        with FakeTensorMode():
            poses = [torch.empty(5, 17, 3, device="cuda") for _ in range(3)]
            out = torch.ops.openmpl_amd.forward(m._handle(), poses, [], [])
        assert len(out) == n_out
        for o in out:
            assert tuple(o.shape) == (5, 17, 3) and o.dtype == torch.float32 and o.device.type == "cuda"
    with pytest.raises(RuntimeError, match="not alive"):
        with FakeTensorMode():
            torch.ops.openmpl_amd.forward(10 ** 9, [torch.empty(1, 17, 3, device="cuda")], [], [])
    with pytest.raises(ValueError):
        m.use_torch_op("sometimes")
    with pytest.raises(ValueError):
        m.set_small_batch_engine("maybe")


def test_copies_of_a_module_get_their_own_operator_handle():
    """the operator handle is a plain __dict__ entry, so copy.deepcopy, torch.save / torch.load and DataParallel's
    shallow replica inherit the ORIGINAL's number.  Through the operator route a copy with modified weights would then run the
    original's weights.  Every use resolves the handle against the registry and re-issues it when it names another object."""
    import copy
    import io
    import threading
    from openmpl_amd import multiview_mpl as mm
    m = MultiView_MPL(num_views=2, depth=1).eval()
    h = m._handle()
    assert mm._module_of(h) is m and m._handle() == h                  # stable for the object it was issued to
    m._hip_cache[0] = dict(ready=threading.Lock())                     # stands in for a torch.cuda.Event: cannot be copied / pickled
    c = copy.deepcopy(m)
    assert c._hip_cache == {} and c._fast_bind == {} and m._hip_cache  # derived state stays behind (a used module copies fine)
    m._hip_cache = {}
    assert c.__dict__["_op_handle"] == 0                               # no inherited number:
    hc = c._handle()
    assert hc != h and mm._module_of(hc) is c and mm._module_of(h) is m   # issued on first use, the original keeps its own
    r0 = m._replicate_for_data_parallel()                              # the SHALLOW copy of DataParallel inherits the number ...
    assert r0.__dict__["_op_handle"] == h and mm._module_of(r0._handle()) is r0 and mm._module_of(h) is m   # ... re-issued on use
    with torch.no_grad():
        c.head[1].weight.add_(1.0)
    assert not torch.equal(mm._module_of(hc).head[1].weight, mm._module_of(h).head[1].weight)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    l = torch.load(buf, weights_only=False)
    hl = l._handle()
    assert hl not in (h, hc) and mm._module_of(hl) is l
    r = m._replicate_for_data_parallel()
    assert mm._module_of(r._handle()) is r and mm._module_of(h) is m
    # a fake-tensor trace through the operator resolves to the copy's own shape contract
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        out = torch.ops.openmpl_amd.forward(hc, [torch.empty(3, 17, 3, device="cuda") for _ in range(2)], [], [])
    assert tuple(out[0].shape) == (3, 17, 3)
    del c
    import gc
    gc.collect()
    with pytest.raises(RuntimeError, match="not alive"):
        mm._module_of(hc)


def test_cached_tensor_lists_follow_wholesale_parameter_swaps():
    """paths that replace parameters without a registration hook (functional_call-style writes into _parameters,
    _apply under overwrite_module_params_on_conversion) must not leave the forward with a stale list of tensors."""
    m = MultiView_MPL(num_views=2, depth=1).eval()
    first = m._tensor_lists()
    assert m._tensor_lists() is first                                  # cached
    w_old = m.blocks[0].attn.qkv.weight
    m.blocks[0].attn.qkv._parameters["weight"] = torch.nn.Parameter(w_old.detach().clone())   # no hook fires
    again = m._tensor_lists()
    assert again is not first and any(t is m.blocks[0].attn.qkv.weight for t in again[1])
    assert not any(t is w_old for t in again[1])
    m.float()                                                          # _apply drops every cache
    assert "_tl_cache" not in m.__dict__
