"""Arithmetic robustness of the default engine through the WHOLE forward (round-3 review, Weak #1 / Next #2).

The fp16x2 engine (openmpl_amd/csrc/h2_gemm.hip) carries the attention output and the GELU output between the GEMMs of a
block as hi + lo fp16 pairs under STATIC, data-free scales (one per COLUMN, from the column bounds
sqrt(K) |gamma o W_n|_2 + |c_n| the binding computes when it packs the weights; the consumer's weights are packed against them).  The unit tests scale a plain operand
by its measured amax; these tests put the static scales themselves under stress: whole layers far from the usual
magnitudes, single outlier channels (the case the bound is weakest for), degenerate inputs.  Every case is checked
against the fp64 oracle with the 1e-4 contract, and the error of the native fp32 matrix instructions ("fp32_mfma")
on the same inputs is printed beside it.  Reference ops: multiview_mpl.py:53-67 (Attention), :84-92 (Block).
"""
import pytest
import torch

from oracle import mpl_oracle
from tests.test_gpu_parity import DEV, TOL, _model
from tests.util import golden_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _team_kernels_only():
    """The tests are about the fp16x2 team kernels: the small-batch engine (sm_stack.hip, exact fp32 MFMA, up to 80 token rows)
    is switched off for their duration, whatever the size of a fixture."""
    from openmpl_amd import cabi
    lib = cabi.load()
    cabi.check(lib.mpl_x3_stack_mode(8), "stack mode")
    yield
    cabi.check(lib.mpl_x3_stack_mode(0), "stack mode")


def _errors(m, g, P, R, Cn, what, tol=TOL, ratio=None):
    """HIP "fp32" (fp16x2) and "fp32_mfma" against the fp64 oracle on the model's CURRENT weights."""
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cpu = lambda lst: [x.cpu() for x in lst]
    ref = mpl_oracle.forward(sd, g["flags"], cpu(P), cpu(R), cpu(Cn), dtype=torch.float64)
    ref32 = mpl_oracle.forward(sd, g["flags"], cpu(P), cpu(R), cpu(Cn), dtype=torch.float32)
    errs = {}
    for prec in ("fp32", "fp32_mfma"):
        m.set_matmul_precision(prec)
        with torch.no_grad():
            out = m(P, rays=R, centers=Cn)
        assert torch.isfinite(out).all(), what + ": non-finite poses with " + prec
        errs[prec] = mpl_oracle.rel_errors(out.cpu(), ref)
    m.set_matmul_precision("fp32")
    e32 = mpl_oracle.rel_errors(ref32, ref)
    print("%s: fp16x2 %.2e/%.2e | fp32 MFMA %.2e/%.2e | reference CPU fp32 %.2e/%.2e (max-scaled/norm-wise vs fp64)"
          % ((what,) + errs["fp32"] + errs["fp32_mfma"] + e32))
    assert errs["fp32"][0] <= tol and errs["fp32"][1] <= tol, "%s: %.3e %.3e" % ((what,) + errs["fp32"])
    if ratio is not None:
        # the split-operand engine may not be worse than genuine fp32 arithmetic by more than `ratio`
        worst32 = max(errs["fp32_mfma"][1], e32[1], 2e-7)
        assert errs["fp32"][1] <= ratio * worst32, "%s: fp16x2 %.2e vs fp32 %.2e" % (what, errs["fp32"][1], worst32)
    return errs


SETS = [dict(qkv=64.0, proj=1 / 64.0, fc1=1e-3, fc2=1e3, g1=30.0, g2=0.02),
        dict(qkv=1 / 64.0, proj=64.0, fc1=1e3, fc2=1e-3, g1=1 / 30.0, g2=50.0)]


@pytest.mark.parametrize("name", ["chosen_v4_b8_l2", "full_v4_b8_l2"])
@pytest.mark.parametrize("scales", SETS)
def test_fpt_split_operands_follow_the_weight_magnitudes(name, scales):
    """The FPT analogue of test_spt_split_operands_follow_the_weight_magnitudes: whole layers of the fusion blocks far from
    the usual magnitudes (the products qkv x proj and fc1 x fc2 kept near one so that the residual stream stays finite)."""
    m, g = _model(name)
    P, R, Cn = golden_inputs(g, DEV)
    with torch.no_grad():
        for blk in m.blocks:
            blk.attn.qkv.weight.mul_(scales["qkv"]); blk.attn.qkv.bias.mul_(scales["qkv"])
            blk.attn.proj.weight.mul_(scales["proj"])
            blk.mlp.fc1.weight.mul_(scales["fc1"]); blk.mlp.fc1.bias.mul_(scales["fc1"])
            blk.mlp.fc2.weight.mul_(scales["fc2"])
            blk.norm1.weight.mul_(scales["g1"]); blk.norm1.bias.mul_(scales["g1"])
            blk.norm2.weight.mul_(scales["g2"]); blk.norm2.bias.mul_(scales["g2"])
    _errors(m, g, P, R, Cn, "%s FPT weights at unusual magnitudes" % name, ratio=4.0)


def _v_column(m, blk, col):
    """Index into qkv's 3 D output features of v column `col` (qkv column = s D + h hd + e, multiview_mpl.py:55)."""
    D = blk.attn.qkv.weight.shape[1]
    return 2 * D + col


@pytest.mark.parametrize("name", ["chosen_v4_b8_l2", "full_v4_b8_l2"])
@pytest.mark.parametrize("case", ["v_bias_1e3", "v_bias_1e5", "fc1_row_1e4", "fc1_row_1e6", "v_row_1e4", "proj_zero_column",
                                  "everything"])
def test_fpt_outlier_channels(name, case):
    """One channel far outside the rest -- exactly where a per-layer static scale would be weakest: the outlier would set the
    fp16 window of every other channel of its layer.  The scales are per column and the weights of the consumer absorb them
    (W_mk / so_k), so the operand pair is equilibrated per channel."""
    m, g = _model(name)
    P, R, Cn = golden_inputs(g, DEV)
    with torch.no_grad():
        for li, blk in enumerate(m.blocks):
            D = blk.attn.qkv.weight.shape[1]
            if case in ("v_bias_1e3", "v_bias_1e5", "everything"):
                # the attention output of that channel is the bias itself (softmax rows sum to one): keep proj's column for it
                # small enough that the residual stream survives
                c = (37 + 151 * li) % D
                val = 1e5 if case == "v_bias_1e5" else 1e3
                blk.attn.qkv.bias[_v_column(m, blk, c)] = val
                blk.attn.proj.weight[:, c] *= 1.0 / val
            if case in ("fc1_row_1e4", "fc1_row_1e6", "everything"):
                r = (5 + 97 * li) % (2 * D)
                f = 1e6 if case == "fc1_row_1e6" else 1e4
                blk.mlp.fc1.weight[r] *= f
                blk.mlp.fc1.bias[r] *= f
                blk.mlp.fc2.weight[:, r] *= 1.0 / f
            if case in ("v_row_1e4", "everything"):
                c = (300 + 211 * li) % D
                blk.attn.qkv.weight[_v_column(m, blk, c)] *= 1e4
                blk.attn.proj.weight[:, c] *= 1e-4
            if case in ("proj_zero_column", "everything"):
                blk.attn.proj.weight[(11 + 7 * li) % D] = 0.0            # an all-zero OUTPUT column of proj (row of the weight)
                blk.attn.proj.weight[:, (13 + 5 * li) % D] = 0.0         # and an attention channel nobody reads
    _errors(m, g, P, R, Cn, "%s outlier %s" % (name, case), ratio=4.0)


@pytest.mark.parametrize("name", ["chosen_v4_b8_l2", "full_v4_b8_l2"])
@pytest.mark.parametrize("case", ["poses_1e3", "conf_zero", "poses_zero"])
def test_degenerate_inputs(name, case):
    """Poses at +-1e3 (far outside the normalised screen), confidence zero everywhere (every joint clipped,
    joints_dataset_mpl.py:710-715), all-zero poses."""
    m, g = _model(name)
    P, R, Cn = golden_inputs(g, DEV)
    P = [p.clone() for p in P]
    for v, p in enumerate(P):
        if case == "poses_1e3":
            p[..., 0:2] = p[..., 0:2].sign() * 1e3
            p[0, 0, 0], p[0, 1, 1] = -1e3, 1e3
        elif case == "conf_zero":
            p[..., 2] = 0.0
        else:
            p.zero_()
    _errors(m, g, P, R, Cn, "%s %s" % (name, case), ratio=4.0)


@pytest.mark.parametrize("name", ["chosen_v4_b8_l2", "full_v4_b8_l2"])
@pytest.mark.parametrize("case", ["v_bias_1e3", "v_bias_1e5", "fc1_row_1e4", "v_row_1e4", "everything"])
def test_spt_outlier_channels(name, case):
    """The same outlier channels in the SPATIAL blocks (width 32, `mpl_spt_pack`: the four Linear layers of a block run from
    two-part fp16 operands inside spt3_kernel, their attention / GELU outputs under static scales from data-free bounds)."""
    m, g = _model(name)
    P, R, Cn = golden_inputs(g, DEV)
    stacks = m.Spatial_blocks if m.multiple_spatial_blocks else [m.Spatial_blocks]
    with torch.no_grad():
        for st in stacks:
            for li, blk in enumerate(st):
                D = 32
                if case in ("v_bias_1e3", "v_bias_1e5", "everything"):
                    c = (5 + 11 * li) % D
                    val = 1e5 if case == "v_bias_1e5" else 1e3
                    blk.attn.qkv.bias[2 * D + c] = val
                    blk.attn.proj.weight[:, c] *= 1.0 / val
                if case in ("fc1_row_1e4", "everything"):
                    r = (3 + 7 * li) % (2 * D)
                    blk.mlp.fc1.weight[r] *= 1e4
                    blk.mlp.fc1.bias[r] *= 1e4
                    blk.mlp.fc2.weight[:, r] *= 1e-4
                if case in ("v_row_1e4", "everything"):
                    c = (20 + 13 * li) % D
                    blk.attn.qkv.weight[2 * D + c] *= 1e4
                    blk.attn.proj.weight[:, c] *= 1e-4
    _errors(m, g, P, R, Cn, "%s SPT outlier %s" % (name, case), ratio=4.0)
