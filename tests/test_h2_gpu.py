"""GPU tests of the fp16x2 split-operand engine (openmpl_amd/csrc/h2_gemm.hip), the default "fp32" arithmetic of the FPT
block stack (reference ops: multiview_mpl.py:84-92 Block, :53-67 Attention, :31-37 Mlp): unit GEMMs through the C ABI
against fp64 torch, the packed operand byte for byte against oracle/split_oracle.py, range robustness, and the engine
inside the whole forward (goldens, launch modes, the older engines)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from openmpl_amd import cabi, detrng
from oracle import mpl_oracle, split_oracle
from tests.test_gpu_parity import DEV, _assert_close, _big_inputs, _fp64_linear, _model, _stream

pytestmark = pytest.mark.gpu


def _h2_pack(lib, Wd, bd, gd, bed, N, K, ln):
    nbytes = lib.mpl_pack_h2_bytes(N, K)
    assert nbytes == (N // 136) * (K // 32) * 18 * 1024 + (5 * N + 8) * 4
    W2 = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    cabi.check(lib.mpl_pack_h2(Wd.data_ptr(), bd.data_ptr(), gd.data_ptr() if ln else None, bed.data_ptr() if ln else None,
                               N, K, W2.data_ptr(), _stream()), "mpl_pack_h2")
    return W2


def _h2_linear(lib, Ad, Wd, bd, gd, bed, Rd, M, K, N, epi, ln):
    W2 = _h2_pack(lib, Wd, bd, gd, bed, N, K, ln)
    so = torch.zeros(M + 64, max(1, K // 136), 2, device=DEV)
    wsb = lib.mpl_ln_linear_h2_workspace_bytes(M, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    Y = torch.full((M, N), float("nan"), device=DEV)
    cabi.check(lib.mpl_ln_linear_h2(Ad.data_ptr(), M, K, 1 if ln else 0, 1e-6, W2.data_ptr(), N, epi,
                                    Rd.data_ptr() if epi == cabi.EPI_BIAS_RESIDUAL else None, Y.data_ptr(),
                                    so.data_ptr() if ln else None, ws.data_ptr(), wsb, _stream()), "mpl_ln_linear_h2")
    return Y


@pytest.mark.parametrize("M,K,N,epi,ln", [
    (4096, 544, 1632, cabi.EPI_BIAS, True),
    (4096, 544, 544, cabi.EPI_BIAS_RESIDUAL, False),
    (1000, 544, 1088, cabi.EPI_BIAS_GELU, True),
    (77, 1088, 544, cabi.EPI_BIAS_RESIDUAL, False),
    (3, 544, 544, cabi.EPI_BIAS, False),
    (130, 1088, 3264, cabi.EPI_BIAS, True),
    (640, 2176 // 2, 1088, cabi.EPI_BIAS_RESIDUAL, True),       # LayerNorm in front of a one-pass (residual) GEMM
    (640, 1632, 1088, cabi.EPI_BIAS_RESIDUAL, False),
    (300, 2176, 1088, cabi.EPI_BIAS_RESIDUAL, False),            # fc2 of the FULL flag set
    (8192, 544, 1088, cabi.EPI_BIAS_GELU, True),
    (200, 544, 136, cabi.EPI_BIAS, True),                        # a single column group (one pass)
])
def test_h2_linear_is_fp32_accurate(M, K, N, epi, ln):
    """mpl_pack_h2 + mpl_ln_linear_h2 against an fp64 evaluation: the fp16x2 GEMM must be as accurate as fp32 arithmetic --
    its error may not exceed the native fp32 MFMA kernel's by more than rounding noise."""
    lib = cabi.load()
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g) * 1.7 + 0.3
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gam, bet = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    ref = _fp64_linear(A, W, b, R, gam, bet, epi, ln)
    Ad, Wd, bd, Rd, gd, bed = (t.to(DEV) for t in (A, W, b, R, gam, bet))
    Y = _h2_linear(lib, Ad, Wd, bd, gd, bed, Rd, M, K, N, epi, ln)
    torch.cuda.synchronize()
    assert torch.isfinite(Y).all()
    e_h2 = mpl_oracle.rel_errors(Y.cpu(), ref)
    Y2 = torch.full((M, N), float("nan"), device=DEV)
    so = torch.zeros(M, max(1, K // 136), 2, device=DEV)
    cabi.check(lib.mpl_ln_linear(Ad.data_ptr(), M, K, gd.data_ptr() if ln else None, bed.data_ptr() if ln else None, 1e-6,
                                 Wd.data_ptr(), bd.data_ptr(), N, epi, Rd.data_ptr() if epi == cabi.EPI_BIAS_RESIDUAL else None,
                                 Y2.data_ptr(), so.data_ptr() if ln else None, _stream()), "mpl_ln_linear")
    torch.cuda.synchronize()
    e_mfma = mpl_oracle.rel_errors(Y2.cpu(), ref)
    print("M=%d K=%d N=%d: h2 %.2e/%.2e  fp32 MFMA %.2e/%.2e" % ((M, K, N) + e_h2 + e_mfma))
    assert e_h2[0] <= 3e-6 and e_h2[1] <= 1e-6
    assert e_h2[1] <= 1.5 * e_mfma[1] + 1e-8, "the fp16x2 GEMM is less accurate than the fp32 MFMA GEMM"
    # a second call is bitwise the first (fixed k order, fixed product order)
    assert torch.equal(Y, _h2_linear(lib, Ad, Wd, bd, gd, bed, Rd, M, K, N, epi, ln))


@pytest.mark.parametrize("shift,scale", [(0.0, 1.0), (5.0, 1.0), (-40.0, 2.0), (0.0, 1e-20), (0.0, 1e15), (100.0, 1e-2)])
def test_h2_layernorm_input_is_robust_to_offset_and_scale(shift, scale):
    """The LayerNorm GEMMs normalise their input BEFORE it is split ((x - mean) rstd 2^10, always inside the fp16 window), so
    neither the offset nor the magnitude of the rows costs anything beyond what fp32 statistics cost any implementation."""
    lib = cabi.load()
    M, K, N = 256, 544, 1632
    g = torch.Generator().manual_seed(11)
    A = (torch.randn(M, K, generator=g) + shift) * scale
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    gam, bet = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    a = A.double()
    a = (a - a.mean(-1, keepdim=True)) / torch.sqrt(a.var(-1, unbiased=False, keepdim=True) + 1e-6) * gam.double() + bet.double()
    ref = a @ W.double().T + b.double()
    Ad, Wd, bd, gd, bed = (t.to(DEV) for t in (A, W, b, gam, bet))
    Y = _h2_linear(lib, Ad, Wd, bd, gd, bed, None, M, K, N, cabi.EPI_BIAS, True)
    torch.cuda.synchronize()
    assert torch.isfinite(Y).all()
    mx, nw = mpl_oracle.rel_errors(Y.cpu(), ref)
    # the fp32 row statistics lose |mean| / sigma of relative precision (as in any fp32 LayerNorm, the reference's included)
    amp = (1.0 + abs(shift) ** 2) ** 0.5
    print("shift %g scale %g: %.2e / %.2e" % (shift, scale, mx, nw))
    assert mx <= 3e-6 * amp and nw <= 1e-6 * amp and mx < 1e-4


@pytest.mark.parametrize("scale", [1e-30, 1e-8, 1.0, 3e4, 1e20])
def test_h2_plain_operand_is_robust_to_scale(scale):
    """A GEMM without LayerNorm (unit entry): the A operand is packed under its measured amax, the weights under their column
    maxima -- any magnitude that fp32 itself can carry works, including all-zero weight columns and all-zero rows."""
    lib = cabi.load()
    M, K, N = 192, 544, 544
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g) * scale
    A[7] = 0.0
    W = torch.randn(N, K, generator=g) * K ** -0.5 / max(scale, 1e-20) ** 0.5
    W[11] = 0.0
    W[12] *= 1e-6
    b = torch.randn(N, generator=g) * float((A.double() @ W.double().T).abs().max())
    ref = A.double() @ W.double().T + b.double()
    Ad, Wd, bd = (t.to(DEV) for t in (A, W, b))
    Y = _h2_linear(lib, Ad, Wd, bd, None, None, None, M, K, N, cabi.EPI_BIAS, False)
    torch.cuda.synchronize()
    assert torch.isfinite(Y).all()
    mx, nw = mpl_oracle.rel_errors(Y.cpu(), ref)
    print("scale %g: %.2e / %.2e" % (scale, mx, nw))
    assert mx <= 3e-6 and nw <= 1e-6
    assert torch.equal(Y[:, 11].cpu(), b[11].expand(M)) and torch.equal(Y[7].cpu(), b)


def test_h2_operand_bytes_match_the_definition():
    """mpl_pack_h2 against oracle/split_oracle.py: both fp16 parts in MFMA fragment order byte for byte, then the trailer."""
    lib = cabi.load()
    g = torch.Generator().manual_seed(3)
    for N, K, ln in ((272, 544, False), (544, 544, True), (1632, 544, True), (544, 1088, False)):
        W = torch.randn(N, K, generator=g) * K ** -0.5
        W[0, 0], W[1, 1], W[2, 2] = 0.0, 1.0 + 2 ** -23, -255.99998
        W[3] = 0.0
        bias = torch.randn(N, generator=g)
        gam, bet = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
        Wd, bd, gd, bed = (t.to(DEV) for t in (W, bias, gam, bet))
        dst = _h2_pack(lib, Wd, bd, gd, bed, N, K, ln)
        torch.cuda.synchronize()
        raw = dst.cpu().numpy()
        nfr = (N // 136) * (K // 32) * 18 * 1024
        got = raw[:nfr].view(np.uint16)
        want = split_oracle.h2_operand(W.numpy(), gam.numpy() if ln else None).reshape(-1)
        assert got.shape == want.shape
        assert np.array_equal(got, want), "N=%d K=%d: %d of %d fp16 words differ" % (N, K, int((got != want).sum()), want.size)
        c, sc, sw, bound, oscale, meta = split_oracle.h2_trailer(W.numpy(), bias.numpy(), gam.numpy() if ln else None, bet.numpy() if ln else None)
        vec = raw[nfr:].view(np.float32)
        assert np.array_equal(vec[:N], c) and np.array_equal(vec[N:2 * N], sc) and np.array_equal(vec[2 * N:3 * N], sw)
        assert np.allclose(vec[3 * N:4 * N], bound, rtol=2e-6, atol=0)
        # the static per-column scales: identical unless a bound sits within rounding of a power of two
        assert (vec[4 * N:5 * N] != oscale).sum() <= 1 and np.all(np.frexp(vec[4 * N:5 * N])[0] == 0.5)
        if ln:
            assert np.array_equal(vec[5 * N:5 * N + 4], meta[:4]), (vec[5 * N:5 * N + 8], meta)
            if np.array_equal(vec[4 * N:5 * N], oscale):
                assert np.array_equal(vec[5 * N + 6:5 * N + 8], meta[6:8])
            # a consumer of these columns (K2 = N inputs), packed against the scales: W / so, byte for byte, and the fingerprint
            if N % 544 == 0:
                N2 = 136
                Wc = torch.randn(N2, N, generator=g) * N ** -0.5
                bc = torch.randn(N2, generator=g)
                so_dev = lib.mpl_pack_h2_out_scale(dst.data_ptr(), N, K)
                assert so_dev == dst.data_ptr() + nfr + 4 * N * 4
                dst2 = torch.zeros(lib.mpl_pack_h2_bytes(N2, N), dtype=torch.uint8, device=DEV)
                Wcd, bcd = Wc.to(DEV), bc.to(DEV)
                cabi.check(lib.mpl_pack_h2_scaled(Wcd.data_ptr(), bcd.data_ptr(), so_dev, N2, N, dst2.data_ptr(), _stream()),
                           "mpl_pack_h2_scaled")
                torch.cuda.synchronize()
                raw2 = dst2.cpu().numpy()
                nfr2 = (N2 // 136) * (N // 32) * 18 * 1024
                ins = vec[4 * N:5 * N].copy()
                want2 = split_oracle.h2_operand(Wc.numpy(), None, ins).reshape(-1)
                assert np.array_equal(raw2[:nfr2].view(np.uint16), want2)
                t2 = split_oracle.h2_trailer(Wc.numpy(), bc.numpy(), None, None, ins)
                vec2 = raw2[nfr2:].view(np.float32)
                assert np.array_equal(vec2[N2:2 * N2], t2[1]) and vec2[5 * N2 + 6] == t2[5][6] == vec[5 * N + 6]


def test_h2_shapes_are_validated():
    lib = cabi.load()
    for n, k in ((100, 544), (544, 40), (136, 32), (0, 544), (544, -544), (544, 272), (544, 136), (544, 9248)):
        assert lib.mpl_pack_h2_bytes(n, k) == 0
    x = torch.zeros(64, 544, device=DEV)
    assert lib.mpl_pack_h2(x.data_ptr(), x.data_ptr(), None, None, 100, 544, x.data_ptr(), _stream()) != 0
    big = torch.zeros(lib.mpl_pack_h2_bytes(544, 544), dtype=torch.uint8, device=DEV)
    W = torch.zeros(544, 544, device=DEV)
    assert lib.mpl_pack_h2(W.data_ptr(), x.data_ptr(), x.data_ptr(), None, 544, 544, big.data_ptr(), _stream()) != 0    # half a LayerNorm
    assert lib.mpl_pack_h2_bytes(136, 2176) > 0              # fc2 of the FULL flag set (plain operand: any multiple of 544)
    # a folded LayerNorm combines at most 8 slice partials per row: K = 1632 packs as a plain operand only
    W3 = torch.zeros(136, 1632, device=DEV)
    g3 = torch.ones(1632, device=DEV)
    big3 = torch.zeros(lib.mpl_pack_h2_bytes(136, 1632), dtype=torch.uint8, device=DEV)
    assert lib.mpl_pack_h2(W3.data_ptr(), x.data_ptr(), None, None, 136, 1632, big3.data_ptr(), _stream()) == 0
    assert lib.mpl_pack_h2(W3.data_ptr(), x.data_ptr(), g3.data_ptr(), g3.data_ptr(), 136, 1632, big3.data_ptr(), _stream()) != 0
    assert lib.mpl_ln_linear_h2(x.data_ptr(), 64, 544, 0, 1e-6, big.data_ptr(), 544, 0, None, W.data_ptr(), None,
                                W.data_ptr(), 16, _stream()) == -3


def test_h2_is_the_default_engine_and_the_native_fp32_engine_agrees():
    """"fp32" packs fp16x2 operands; "fp32_mfma" gives the same poses to rounding; "bf16" packs bf16 operands into the *_w16
    fields; every engine is bitwise independent of the batch size."""
    m, g = _model("chosen_v4_b8_l12")
    P, R, Cn = _big_inputs(512, 4, 5)
    outs = {}
    for prec in ("fp32", "bf16", "fp32_mfma"):
        m.set_matmul_precision(prec)
        with torch.no_grad():
            full = m(P, rays=R, centers=Cn)
            part = m([x[100:400].contiguous() for x in P], rays=[x[100:400].contiguous() for x in R],
                     centers=[x[100:400].contiguous() for x in Cn])
        assert torch.equal(full[100:400], part), prec + ": batch slice changed results"
        outs[prec] = full
        blk = m._hip_cache[0]["fpt_blocks"][0]
        assert bool(blk.qkv_h2) == (prec == "fp32") and bool(blk.qkv_w16) == (prec == "bf16") and not blk.proj_w3
    mx, nw = mpl_oracle.rel_errors(outs["fp32"].cpu(), outs["fp32_mfma"].cpu())
    assert 0 < mx < 5e-6 and nw < 5e-6, (mx, nw)
    mx, nw = mpl_oracle.rel_errors(outs["fp32"].cpu(), outs["bf16"].cpu())
    assert 1e-5 < mx < 5e-2, (mx, nw)
    m.set_matmul_precision("fp32")
    with pytest.raises(ValueError):
        m.set_matmul_precision("fp32x3")            # the round-2 three-part engine is gone (round 5)


def _kinds(fn):
    """Kernel launches of the library (by kind) while fn() runs."""
    cabi.profile_start()
    out = fn()
    torch.cuda.synchronize()
    return out, {k: n for k, (ms, n) in cabi.profile_stop().items()}


def test_data_parallel_replicas_pack_once_per_device():
    """torch.nn.parallel.replicate (what DataParallel.forward does on every call, valid_mpl.py:177-178) hands each replica
    fresh parameter tensors.  The replica shares the source module's per-device cache and keys the packed operands on the
    SOURCE parameters' storage + version: the first forward on a device packs, every later one launches no packing kernel
    (round 3 re-packed everything on every forward: 1.2 ms of GPU time), and an in-place update of a source parameter re-packs."""
    from torch.nn.parallel import replicate
    m, g = _model("chosen_v4_b8_l2")
    P, R, Cn = _big_inputs(64, 4, 3)
    with torch.no_grad():
        want = m(P, rays=R, centers=Cn)
        rep = replicate(m, [0], detach=True)[0]
        assert rep is not m and rep._dp_replica and rep._dp_src is m and rep._hip_cache is m._hip_cache
        got, k1 = _kinds(lambda: rep(P, rays=R, centers=Cn))
        assert torch.equal(got, want) and k1["pack"] == 0, k1          # same device, same storage: the source's operands serve
        # what a second GPU sees: parameters in FRESH storage on every forward (cloned here, broadcast there)
        def fresh_replica():
            r = replicate(m, [0], detach=True)[0]
            for mod in r.modules():                 # the copies are plain tensor attributes (Parameters where storage is shared)
                for k, p in list(mod._former_parameters.items()):
                    mod._parameters.pop(k, None)
                    object.__setattr__(mod, k, p.detach().clone())
            return r
        outs = []
        for it in range(3):
            r = fresh_replica()
            o, k = _kinds(lambda: r(P, rays=R, centers=Cn))
            outs.append(o)
            assert k["pack"] == 0, "forward %d of a replica packed operands again: %s" % (it, k)
            assert k["gemm"] == 1 and k["spt"] == 1
            blk = r._hip_cache[0]["fpt_blocks"][0]
            assert bool(blk.qkv_h2) and r._hip_cache[0]["weights"].spt_packed
        assert all(torch.equal(o, want) for o in outs)
        # an optimiser step / load_state_dict on the SOURCE invalidates the derived operands of every replica
        m.blocks[0].attn.proj.weight.mul_(1.5)
        r = fresh_replica()
        o, k = _kinds(lambda: r(P, rays=R, centers=Cn))
        assert k["pack"] > 0 and not torch.equal(o, want)
        assert torch.equal(o, m(P, rays=R, centers=Cn))
        # replicas honour the other engines too (they fell back to the fp32 matrix instructions in round 3)
        m.set_matmul_precision("bf16")
        ob = m(P, rays=R, centers=Cn)
        r = fresh_replica()
        assert torch.equal(r(P, rays=R, centers=Cn), ob)
        assert bool(r._hip_cache[0]["fpt_blocks"][0].qkv_w16)
        m.set_matmul_precision("fp32")


def test_data_parallel_module_end_to_end():
    """nn.DataParallel over every visible GPU with CPU inputs (scatter does the H2D copy), twice: the second forward packs
    nothing; the result is bitwise the direct one."""
    m, g = _model("chosen_v4_b8_l2")
    p, r, c = detrng.make_inputs(96, 4, seed=3)
    P, R, Cn = ([torch.from_numpy(x) for x in lst] for lst in (p, r, c))
    with torch.no_grad():
        want = m([x.to(DEV) for x in P], rays=[x.to(DEV) for x in R], centers=[x.to(DEV) for x in Cn])
        dp = torch.nn.DataParallel(m).eval()
        first = dp(P, rays=R, centers=Cn)
        second, k = _kinds(lambda: dp(P, rays=R, centers=Cn))
    assert torch.equal(first, want) and torch.equal(second, want)
    assert k["pack"] == 0, k


def test_shipped_call_shape_b256_v2():
    """configs/h36m/mpl_amass/h36m.yaml: TEST.BATCH_SIZE 256, two views, depth 12 -- the shape validate() actually calls with."""
    m, g = _model("chosen_v2_b1_l12")
    P, R, Cn = _big_inputs(256, 2, 77)
    with torch.no_grad():
        out = m(P, rays=R, centers=Cn)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = mpl_oracle.forward(sd, g["flags"], [x.cpu() for x in P], [x.cpu() for x in R], [x.cpu() for x in Cn], dtype=torch.float64)
    _assert_close(out, ref, "B=256 V=2 vs fp64 oracle")


@pytest.mark.parametrize("name,B", [("chosen_v2_b1_l12", 256), ("chosen_v4_b8_l12", 256), ("chosen_v2_b1_l12", 32), ("chosen_v8_b4_l2", 61),
                                    ("chosen_v4_b8_l2", 100), ("full_v4_b8_l2", 64), ("chosen_v4_b8_l2", 9), ("chosen_v5_b19_l2", 40)])
def test_row_narrow_teams_are_bitwise_the_whole_tile_teams(name, B):
    """Launches that would leave compute units idle split every 64-row tile into sub-tiles of 16 or 32 rows, one team each
    (h2_stackn_kernel; VERDICT r4 item 2: the reference's shipped call shape, 256 frames x 2 views, is 8 tiles = 32 workgroups).
    The waves of a sub-tile run the full-tile code on their row groups, so the poses must be bitwise those of the whole-tile teams:
    forced both ways and both widths (mpl_x3_stack_mode bits 5, 6), the 16-row teams in their direct-W form (h2_stackd_kernel: the
    default) and in the ring form (bit 4) -- ragged tiles, the LDS attention of 8 views, width 1088, and
    five views (60-row tiles: no narrow form exists, the switch must be a no-op) included."""
    lib = cabi.load()
    m, g = _model(name)
    V = g["flags"]["num_views"]
    P, R, Cn = _big_inputs(B, V, 321)
    outs = {}
    try:
        for tag, bits in (("whole", 1 << 5), ("rows32", 2 << 5), ("rows16", 3 << 5), ("rows16ring", (3 << 5) | 16), ("auto", 0)):
            cabi.check(lib.mpl_x3_stack_mode(bits | 8), "stack mode")           # bit 3: the team kernels also at <= 80 rows
            with torch.no_grad():
                outs[tag] = m(P, rays=R, centers=Cn)
            torch.cuda.synchronize()
    finally:
        cabi.check(lib.mpl_x3_stack_mode(0), "stack mode")
    assert torch.isfinite(outs["whole"]).all()
    for tag in ("rows32", "rows16", "rows16ring", "auto"):
        assert torch.equal(outs["whole"], outs[tag]), "%s changed results: max |d| = %.3e" % (tag, float((outs["whole"] - outs[tag]).abs().max()))
    if B <= 100:
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        ref = mpl_oracle.forward(sd, g["flags"], [x.cpu() for x in P], [x.cpu() for x in R], [x.cpu() for x in Cn], dtype=torch.float64)
        _assert_close(outs["rows16"], ref, name + " 16-row teams vs fp64 oracle")


def test_the_library_reports_the_form_of_a_stack_launch():
    """`mpl_block_stack_form` is the launch rule itself (h2_phase.hpp h2_stack_form is what h2_launch_stack follows): the shapes the
    documentation names land on the kernels it names, the A/B switches move them, and the per-call flag keeps small batches on the
    team kernels.  (256 compute units assumed: MI355X.)"""
    lib = cabi.load()
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the expected forms are those of a 256-CU device")
    form = lambda B, V, parts=2, flags=0, D=544: lib.mpl_block_stack_form(B, V, D, 8, 13, parts, flags)
    try:
        cabi.check(lib.mpl_x3_stack_mode(0), "stack mode")
        assert form(1, 2) == cabi.FORM_SMALL and form(8, 4) == cabi.FORM_SMALL           # one group of sequences; two groups of 16 rows
        assert form(2, 12) == cabi.FORM_SMALL and form(3, 12) == cabi.FORM_SMALL          # 24 rows = 12 + 12; 36 rows = 3 groups, two column tiles per workgroup
        assert form(32, 2) == cabi.FORM_SMALL and form(20, 4) == cabi.FORM_SMALL          # 64 / 80 rows: five groups of 51 workgroups
        assert form(48, 2) != cabi.FORM_SMALL and form(8, 12) != cabi.FORM_SMALL          # six / eight groups do not fit 256 compute units: the team kernels
        assert form(1, 2, flags=cabi.F_NO_SMALL_STACK) == cabi.FORM_ROWS16_DIRECT         # batch-invariant bits: team kernels
        assert form(1, 2, parts=1) == cabi.FORM_TEAMS                                     # an explicit bf16 request keeps its engine
        assert form(256, 2) == cabi.FORM_ROWS16_DIRECT                                    # the shipped call shape: 8 tiles x 4 sub-tiles x 4
        assert form(256, 4) == cabi.FORM_ROWS16_DIRECT and form(512, 2) == cabi.FORM_ROWS16_DIRECT      # 16 tiles
        assert form(640, 2) == cabi.FORM_ROWS32 and form(512, 4) == cabi.FORM_ROWS32      # 17 .. 32 tiles
        assert form(1024, 4) == cabi.FORM_TEAMS                                           # headline: 64 tiles = 64 teams
        assert form(1024, 8) == cabi.FORM_PAIRS and form(1024, 8, parts=1) == cabi.FORM_TEAMS           # 128 tiles; bf16: one tile at a time
        assert form(100, 5) == cabi.FORM_TEAMS                                            # 60-row tiles: no narrow form
        assert form(64, 2, D=1088) == cabi.FORM_ROWS16                                    # K = 2176: the A operand does not fit, ring form
        assert form(256, 2, parts=0) == cabi.FORM_UNPACKED and form(256, 2, D=512) == cabi.FORM_UNPACKED
        assert form(0, 2) == cabi.E_INVALID if hasattr(cabi, "E_INVALID") else form(0, 2) < 0
        cabi.check(lib.mpl_x3_stack_mode(16), "stack mode")
        assert form(256, 2) == cabi.FORM_ROWS16
        cabi.check(lib.mpl_x3_stack_mode(1 << 5), "stack mode")
        assert form(256, 2) == cabi.FORM_TEAMS
        cabi.check(lib.mpl_x3_stack_mode(2 << 1), "stack mode")
        assert form(1024, 4) == cabi.FORM_PAIRS
        cabi.check(lib.mpl_x3_stack_mode(1), "stack mode")
        assert form(1024, 4) == cabi.FORM_PER_GEMM and form(1, 2) == cabi.FORM_PER_GEMM
        cabi.check(lib.mpl_x3_stack_mode(8), "stack mode")
        assert form(1, 2) == cabi.FORM_ROWS16_DIRECT
    finally:
        cabi.check(lib.mpl_x3_stack_mode(0), "stack mode")


@pytest.mark.parametrize("name,B,prec", [("chosen_v4_b8_l2", 1, "fp32"), ("chosen_v4_b8_l2", 8, "fp32"), ("chosen_v4_b8_l2", 9, "fp32"),
                                         ("chosen_v2_b1_l12", 256, "fp32"), ("chosen_v4_b8_l12", 1024, "fp32"), ("chosen_v8_b4_l2", 1024, "fp32"),
                                         ("chosen_v8_b4_l2", 1024, "bf16"), ("chosen_v4_b8_l2", 4, "bf16"), ("full_v4_b8_l2", 64, "fp32"),
                                         ("kptok_v3_b3_l2", 3, "fp32"), ("chosen_v4_b8_l2", 2, "fp32_mfma")])
def test_the_reported_form_is_the_form_that_was_launched(name, B, prec):
    """the query (mpl_block_stack_form_ex) and the launch (block_stack_impl) go through ONE predicate, and
    mpl_block_stack_last_form() says what the last forward of this thread actually launched: they agree on every engine, also for
    the small-batch engine (n_blocks from the schedule, raw tensors present) and with the per-call MPL_F_NO_SMALL_STACK flag."""
    lib = cabi.load()
    m, g = _model(name)
    m.set_matmul_precision(prec)
    V, L = g["flags"]["num_views"], g["flags"]["depth"]
    kp = bool(g["flags"].get("FPT_blocks_view_keypoint_tokens"))
    D = 32 if kp else 17 * 32 * (2 if g["flags"].get("input_rays_as_token") else 1)
    n_tok = 17 * V if kp else V
    parts = {"fp32": 2, "bf16": 1, "fp32_mfma": 0}[prec] if not kp else 0
    p, r, c = detrng.make_inputs(B, V, seed=5)
    dev = lambda lst: [torch.from_numpy(x).to(DEV) for x in lst]
    for small in ("auto", False):
        m.set_small_batch_engine(small)
        with torch.no_grad():
            out = m(dev(p), rays=dev(r), centers=dev(c))
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        launched = lib.mpl_block_stack_last_form()
        flags = 0 if small == "auto" else cabi.F_NO_SMALL_STACK
        asked = lib.mpl_block_stack_form_ex(B, n_tok, D, 8, L + 1, L, 1, parts, flags)
        assert launched == asked and launched >= 0, (name, B, prec, small, launched, asked)
        assert lib.mpl_block_stack_form(B, n_tok, D, 8, L + 1, parts, flags) == asked          # the short form: the reference's schedule
    m.set_small_batch_engine("auto")
    # a caller that hands over packed operands only (no nn.Linear tensors) never gets the small-batch engine
    assert lib.mpl_block_stack_form_ex(1, 2, 544, 8, 13, 12, 0, 2, 0) != cabi.FORM_SMALL
    assert lib.mpl_block_stack_form_ex(1, 2, 544, 8, 13, 14, 1, 2, 0) < 0                     # more blocks than applications


@pytest.mark.parametrize("name,B", [("chosen_v4_b8_l12", 512), ("chosen_v4_b8_l2", 48), ("chosen_v4_b8_l2", 1), ("full_v4_b8_l2", 200),
                                    ("chosen_v5_b19_l2", 100), ("chosen_v31_b2_l12", 9), ("chosen_v8_b4_l2", 37), ("chosen_v2_b1_l12", 97)])
def test_two_tile_stage_is_bitwise_the_one_tile_stage(name, B):
    """Teams that own two or more row tiles walk PAIRS of tiles with the two-tile stage (h2_stack2_kernel: proj / fc1 / fc2 fetch
    every W k-tile once for both tiles).  The library picks the form by the shape of the launch, so the arithmetic of an output
    element must not depend on it: forced either way (mpl_x3_stack_mode bits 1, 2) the poses are bitwise equal -- odd tile
    counts (a pair with an absent partner), ragged last tiles, 60- and 62-row tiles and the LDS attention of V = 8 / 31 included."""
    lib = cabi.load()
    m, g = _model(name)
    V = g["flags"]["num_views"]
    P, R, Cn = _big_inputs(B, V, 123)
    outs = {}
    try:
        for rt in (1, 2):
            cabi.check(lib.mpl_x3_stack_mode(rt << 1), "stack mode")
            with torch.no_grad():
                outs[rt] = m(P, rays=R, centers=Cn)
            torch.cuda.synchronize()
    finally:
        cabi.check(lib.mpl_x3_stack_mode(0), "stack mode")
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(outs[1], outs[2]), "the two-tile stage changed results: max |d| = %.3e" % float((outs[1] - outs[2]).abs().max())
    if B <= 100:
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        ref = mpl_oracle.forward(sd, g["flags"], [x.cpu() for x in P], [x.cpu() for x in R], [x.cpu() for x in Cn], dtype=torch.float64)
        _assert_close(outs[2], ref, name + " two-tile stage vs fp64 oracle")


@pytest.mark.parametrize("name,B", [("chosen_v2_b1_l12", 1), ("chosen_v2_b1_l12", 8), ("full_v2_b1_l12", 1), ("full_v2_b1_l12", 7),
                                    ("chosen_v4_b8_l12", 4), ("full_v4_b8_l2", 3), ("chosen_v5_b19_l2", 3), ("chosen_v8_b4_l2", 2),
                                    ("chosen_v4_b8_l2", 1), ("chosen_v4_b8_l12", 2), ("chosen_v4_b8_l2", 3), ("chosen_v2_b1_l12", 2),
                                    ("chosen_v2_b1_l12", 3), ("full_v4_b8_l2", 4), ("full_v2_b1_l12", 2), ("chosen_v5_b19_l2", 1),
                                    ("chosen_v8_b4_l2", 1), ("chosen_v2_b1_l12", 5), ("chosen_v4_b8_l12", 8), ("chosen_v2_b1_l12", 13),
                                    ("chosen_v2_b1_l12", 16), ("chosen_v5_b19_l2", 6), ("chosen_v8_b4_l2", 4),
                                    ("chosen_v8_b4_l2", 3), ("chosen_v2_b1_l12", 32), ("chosen_v4_b8_l12", 16), ("chosen_v5_b19_l2", 13),
                                    ("chosen_v4_b8_l2", 9), ("chosen_v8_b4_l2", 5)])
def test_small_batch_engine(name, B):
    """Up to 80 token rows (a single frame, a few frames / persons; from two sequences on as independent groups of sequences of at most
    16 rows each, side by side on disjoint workgroups: two groups of 102, or three to five of 51 with two column tiles each) run
    sm_stack.hip: every GEMM of the block stack on the whole chip
    (one 16-column tile per workgroup, the nn.Linear weights read in place, exact fp32 on the matrix cores), the activations handed
    from step to step as {value, tag} pairs, instead of one team of D / 136 workgroups.  Checked against the fp64 oracle, against
    the team kernels on the same inputs (two fp32 engines: rounding noise apart), and for batch invariance inside the engine
    (bitwise -- the split crosses the switch between the attention inside the proj workgroups, at most 4 rows, and the separate
    attention step)."""
    lib = cabi.load()
    m, g = _model(name)
    V = g["flags"]["num_views"]
    assert B * V <= 80
    P, R, Cn = _big_inputs(B, V, 321)
    with torch.no_grad():
        out, k = _kinds(lambda: m(P, rays=R, centers=Cn))
        assert k["gemm"] == 1 and k["row_stats"] == 0, k         # one persistent launch, no entry kernel
        try:
            cabi.check(lib.mpl_x3_stack_mode(8), "stack mode")
            team, k2 = _kinds(lambda: m(P, rays=R, centers=Cn))
        finally:
            cabi.check(lib.mpl_x3_stack_mode(0), "stack mode")
        assert k2["row_stats"] == 1, k2
        if B > 1:
            lo = m([x[:1].contiguous() for x in P], rays=[x[:1].contiguous() for x in R], centers=[x[:1].contiguous() for x in Cn])
            hi = m([x[1:].contiguous() for x in P], rays=[x[1:].contiguous() for x in R], centers=[x[1:].contiguous() for x in Cn])
            assert torch.equal(out, torch.cat([lo, hi], 0)), "the small-batch engine depends on the batch size"
    sd = {k_: v.detach().cpu() for k_, v in m.state_dict().items()}
    ref = mpl_oracle.forward(sd, g["flags"], [x.cpu() for x in P], [x.cpu() for x in R], [x.cpu() for x in Cn], dtype=torch.float64)
    e_sm = _assert_close(out, ref, name + " small-batch engine vs fp64 oracle")
    e_tm = _assert_close(team, ref, name + " team kernels vs fp64 oracle")
    mx, nw = mpl_oracle.rel_errors(out.cpu(), team.cpu())
    print("%s B=%d: small-batch engine %.2e/%.2e, team kernels %.2e/%.2e from fp64; apart %.2e" % ((name, B) + e_sm + e_tm + (mx,)))
    assert mx < 5e-6 and nw < 5e-6


@pytest.mark.parametrize("name,B", [("chosen_v2_b1_l12", 1), ("chosen_v4_b8_l12", 4), ("full_v2_b1_l12", 3), ("chosen_v2_b1_l12", 20),
                                    ("chosen_v4_b8_l12", 20)])
def test_small_batch_engine_hands_off_correctly_beside_other_work(name, B):
    """The steps of sm_stack.hip hand their activations over as {value, tag} pairs that the consumers poll -- no barrier, no fence.
    The guide's rule for such hand-offs: test them under UNEVEN load, not on an idle chip.  A second stream keeps the GPU busy with
    GEMMs of changing sizes (they take compute units away from the launch in bursts: workgroups of a step start late, finish at
    different times, lines are evicted between polls); every forward beside them must carry the bits of the quiet one, and no
    hand-off may be reported lost.  One group of sequences, two, and five groups with two column tiles per workgroup (40 / 80 rows:
    255 of the 256 compute units in one launch)."""
    m, g = _model(name)
    V = g["flags"]["num_views"]
    P, R, Cn = _big_inputs(B, V, 11)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    small = torch.randn(512, 512, device=DEV)
    with torch.no_grad():
        quiet = m(P, rays=R, centers=Cn).clone()
        torch.cuda.synchronize()
        bad = 0
        for it in range(150):
            with torch.cuda.stream(side):
                for k in range(1 + it % 3):
                    (a @ a) if (it + k) % 2 == 0 else (small @ small)
            out = m(P, rays=R, centers=Cn)
            if it % 10 == 9:
                torch.cuda.synchronize()
            bad += int(not torch.equal(out, quiet))
        torch.cuda.synchronize()
    assert not cabi.device_error(), "a hand-off was reported lost beside other work"
    assert bad == 0, "%d of 150 forwards beside other work differ from the quiet one" % bad
