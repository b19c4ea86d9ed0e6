"""The torch extension (csrc/torch_ext.cpp -> lib/mpl_torch_ext.so, openmpl_amd/torch_ext.py): TORCH_LIBRARY operators
openmpl_amd::bind / lift / unbind over the C ABI -- the default host route of MultiView_MPL.forward since round 6.

CPU part: it builds, loads, registers its schemas and a fake (meta) implementation, and validates what it is handed.
GPU part: the extension route is bitwise the ctypes route and the Python-operator route; a parameter that moves or changes is
noticed by the extension itself (data_ptr / _version of the module's own TensorImpls); bindings die with their module."""
import gc

import pytest
import torch

from openmpl_amd import cabi, detrng, torch_ext
from openmpl_amd.multiview_mpl import MultiView_MPL

DEV = "cuda:0"
FLAGS = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=2, num_views=4, pose_3d_emb_learnable=True)


def test_extension_builds_loads_and_registers_its_operators():
    o = torch_ext.ops()
    for name in ("bind", "unbind", "lift", "live_bindings", "set_entry_points"):
        assert hasattr(o, name), name
    assert "Tensor[] poses" in str(o.lift.default._schema) and str(o.lift.default._schema).endswith("-> Tensor")
    assert o.live_bindings() >= 0
    # the fake implementation: output shape / dtype / device without a GPU (what torch.compile and FakeTensorMode need)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        poses = [torch.empty(5, 17, 3, device="cuda") for _ in range(3)]
        out = o.lift(12345, poses, [], [], 0)
    assert tuple(out.shape) == (5, 17, 3) and out.dtype == torch.float32 and out.device.type == "cuda"
    # bind validates the struct bytes it is handed (no GPU needed to be refused)
    cfg = torch_ext.struct_bytes(cabi.Config(17, 32, 2, 8, 4, 2, 0, 0))
    w = torch_ext.struct_bytes(cabi.Weights())
    with pytest.raises(RuntimeError, match="bytes"):
        o.bind(cfg[:-1].clone(), w, torch.zeros(0, dtype=torch.uint8), [], [], [], 0, False)
    with pytest.raises(RuntimeError, match="whole mpl_block_weights"):
        o.bind(cfg, w, torch.zeros(7, dtype=torch.uint8), [], [], [], 0, False)
    with pytest.raises(RuntimeError, match="one `versioned` flag"):
        o.bind(cfg, w, torch.zeros(0, dtype=torch.uint8), [], [1], [], 0, False)
    with pytest.raises(RuntimeError, match="contiguous float32 tensor on cuda"):
        o.bind(cfg, w, torch.zeros(0, dtype=torch.uint8), [torch.zeros(3)], [0], [], 0, False)
    o.unbind(10 ** 9)                                                   # unknown handle: a no-op, not an error


def test_struct_layouts_of_the_binding_match_the_header():
    """The extension memcpy's the ctypes structs of cabi.py into the C structs of include/mpl_hip.h: the sizes must agree (bind
    checks them against sizeof on the C++ side; here the Python side is pinned)."""
    import ctypes as C
    assert C.sizeof(cabi.Config) == 32 and C.sizeof(cabi.BlockWeights) == 24 * 8
    assert C.sizeof(cabi.Weights) == 20 * 8 + 8 and C.sizeof(cabi.Inputs) == 8 + 3 * 32 * 8


def _model(**more):
    m = MultiView_MPL(**dict(FLAGS, **more))
    detrng.fill_module_(m, seed=31)
    return m.to(DEV).eval()


def _inputs(B, seed=1, V=4):
    p, r, c = detrng.make_inputs(B, V, seed=seed)
    mk = lambda lst: [torch.from_numpy(x).to(DEV) for x in lst]
    return mk(p), mk(r), mk(c)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 8, 64, 300])
def test_extension_route_is_bitwise_the_other_routes(B):
    m = _model()
    P, R, C = _inputs(B)
    with torch.no_grad():
        m.use_torch_op(False)
        direct = m(P, rays=R, centers=C)
        m.use_torch_op("auto")
        ext = m(P, rays=R, centers=C)
        assert m._fast_bind, "the default route did not go through openmpl_amd::lift"
        ext2 = m(tuple(P), rays=None, centers=None)                    # rays unused by CHOSEN; tuples are fine
        m.use_torch_op(True)
        pyop = m(P, rays=R, centers=C)
        m.use_torch_op("auto")
        strided = [x.expand(2, *x.shape)[0] if i % 2 else x.clone().transpose(0, 1).contiguous().transpose(0, 1) for i, x in enumerate(P)]
        ext3 = m(strided, rays=R, centers=C)                           # non-contiguous views are made contiguous in C++
    assert torch.equal(direct, ext) and torch.equal(direct, ext2) and torch.equal(direct, pyop) and torch.equal(direct, ext3)
    torch.library.opcheck(torch.ops.openmpl_amd.lift, (m._fast_bind[0][0], P, R, C, 0),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    with torch.no_grad(), torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        m(P, rays=R, centers=C)
    assert any("openmpl_amd::lift" in e.key for e in prof.key_averages())


@pytest.mark.gpu
def test_extension_notices_changed_and_moved_parameters():
    """What the 300-element data_ptr() tuple of the ctypes route caught per forward, the extension catches in C++ on the module's
    own TensorImpls: in-place updates of folded tensors (version), `p.data = ...` (moved storage), load_state_dict, .to()."""
    m = _model()
    ref = _model()
    P, R, C = _inputs(16, seed=2)
    lib_calls = lambda mod: mod(P, rays=R, centers=C)
    with torch.no_grad():
        base = lib_calls(m)
        h0 = m._fast_bind[0][0]
        # (1) in-place update of a tensor that is folded into a packed operand
        for mod in (m, ref):
            mod.blocks[1].mlp.fc1.weight.mul_(1.5)
        ref._drop_caches()
        a = lib_calls(m)
        assert not torch.equal(a, base) and torch.equal(a, lib_calls(ref))
        assert m._fast_bind[0][0] != h0, "a changed weight must end the old binding"
        # (2) moved storage without any hook or version change
        h1 = m._fast_bind[0][0]
        m.head[1].weight.data = m.head[1].weight.data.clone() * 2.0
        ref.head[1].weight.mul_(2.0)
        ref._drop_caches()
        b = lib_calls(m)
        assert torch.equal(b, lib_calls(ref)) and not torch.equal(b, a) and m._fast_bind[0][0] != h1
        # (3) an in-place update of a tensor the kernels read in place (not folded): same binding, new values
        h2 = m._fast_bind[0][0]
        m.head[1].bias.add_(0.5)
        c = lib_calls(m)
        assert torch.allclose(c, b + 0.5, atol=1e-5) and m._fast_bind[0][0] == h2
        # (4) load_state_dict (copies in place) and a device round trip
        sd = {k: v.clone() for k, v in ref.state_dict().items()}
        m.load_state_dict(sd)
        ref._drop_caches()
        assert torch.equal(lib_calls(m), lib_calls(ref))
        m.cpu()
        assert not m._fast_bind
        m.to(DEV)
        assert torch.equal(lib_calls(m), lib_calls(ref))


@pytest.mark.gpu
def test_extension_raises_the_boundary_errors_and_leaves_other_flag_sets_to_the_general_route():
    m = _model()
    P, R, C = _inputs(4, seed=3)
    with torch.no_grad():
        m(P, rays=R, centers=C)                                        # bound
        with pytest.raises(RuntimeError, match="expected 4 views"):
            m(P[:3], rays=R, centers=C)
        with pytest.raises(RuntimeError, match="float32"):
            m([x.double() for x in P], rays=R, centers=C)
        with pytest.raises(RuntimeError, match="shape"):
            m([x[:, :16] for x in P], rays=R, centers=C)
        with pytest.raises(RuntimeError, match="model is on"):
            m([x.cpu() for x in P], rays=R, centers=C)
        with pytest.raises(RuntimeError, match="expected 4 ray tensors"):
            m(P, rays=R[:2], centers=C)
        g = _model(pose_3d_emb_learnable=False)                        # geometric 3D encoding: rays are required
        with pytest.raises(RuntimeError, match="required"):
            g(P)
        assert torch.isfinite(g(P, rays=R, centers=C)).all()
        with pytest.raises(RuntimeError, match="required"):
            g(P)                                                       # ... also once the binding exists
        k = _model(head_kadkhod=True, hidden_dim=64)
        out = k(P, rays=R, centers=C)
        assert isinstance(out, tuple) and not k._fast_bind             # non-default tails: the general route
        m.train()
        with pytest.raises(RuntimeError, match="inference forward only"):
            m(P, rays=R, centers=C)
        m.eval()
        e = m([x[:0] for x in P], rays=[x[:0] for x in R], centers=[x[:0] for x in C])
        assert tuple(e.shape) == (0, 17, 3)


@pytest.mark.gpu
def test_inference_mode_and_inference_tensors():
    """torch.inference_mode(): the forward works inside it, and a model whose parameters BECAME inference tensors (moved to the
    device inside the mode: they track no version counter) still binds -- such tensors cannot be updated in place outside the mode,
    so the extension keys them by address only."""
    m = MultiView_MPL(**FLAGS)
    detrng.fill_module_(m, seed=31)
    P, R, C = _inputs(8, seed=6)
    with torch.inference_mode():
        mi = m.to(DEV).eval()
        out = mi(P, rays=R, centers=C)
        again = mi(P, rays=R, centers=C)
    with torch.no_grad():
        ref = _model()(P, rays=R, centers=C)
    assert torch.equal(out, again) and torch.equal(out, ref) and mi._fast_bind


@pytest.mark.gpu
def test_bindings_die_with_their_module_and_streams_are_honoured():
    o = torch_ext.ops()
    gc.collect()
    n0 = o.live_bindings()
    m = _model()
    P, R, C = _inputs(32, seed=4)
    with torch.no_grad():
        want = m(P, rays=R, centers=C)
        assert o.live_bindings() == n0 + 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            got = m(P, rays=R, centers=C)                              # another stream than the one that marshalled
        side.synchronize()
        assert torch.equal(got, want)
        m.set_matmul_precision("fp32_mfma")
        assert o.live_bindings() == n0                                 # dropping the caches releases the binding
        m(P, rays=R, centers=C)
        assert o.live_bindings() == n0 + 1
    del m
    gc.collect()
    assert o.live_bindings() == n0
