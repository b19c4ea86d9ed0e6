"""GPU tests of the failure and concurrency behaviour of the persistent block-stack kernel (x3_stack_kernel).

The reference's convention for a forward that cannot complete is a Python exception (SURVEY.md 8b "Error convention").
Here the whole FPT block stack is ONE launch whose workgroups hand operands to each other; these tests pin down that
  * a lost hand-off is REPORTED (NaN poses for that call, RuntimeError on every later call until cleared) and never
    turns into plausible-looking poses computed from stale operands;
  * two forwards enqueued on two streams of one device are serialised by the library and both equal the serial result
    bit for bit;
  * unrelated work that occupies the CUs on another stream only delays the forward."""
import os

import numpy as np
import pytest
import torch

os.environ.setdefault("MPL_FAULT_INJECT", "1")      # the injection hook of mpl_x3_spin_limit is inert unless the process opts in

from openmpl_amd import cabi, detrng
from openmpl_amd.multiview_mpl import MultiView_MPL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FLAGS = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=2, num_views=4, pose_3d_emb_learnable=True)


def _model():
    m = MultiView_MPL(**FLAGS)
    detrng.fill_module_(m, seed=21)
    return m.to(DEV).eval()


def _inputs(B, seed):
    p, r, c = detrng.make_inputs(B, 4, seed=seed)
    mk = lambda lst: [torch.from_numpy(x).to(DEV) for x in lst]
    return mk(p), mk(r), mk(c)


def test_confidence_weights_beyond_the_fp16_window_are_reported_not_saturated():
    """confidence_as_attention_uncertainty_weight multiplies the softmax rows of the SPT attention by the caller's `conf`
    (reference multiview_mpl.py:61-62), which is data: the static scales of the split-operand engine cannot bound it.  Rounds 3-5
    clamped the weighted attention output to +-65000 -- saturated operands, plausible-looking poses.  Now an out-of-window row
    poisons ITS sequence (NaN poses), sets bit 1 of the device error word and every later call raises until cleared; the
    native-fp32 engine has no window and gives the reference's result for the same input.  Ordinary confidences never get there."""
    from oracle import mpl_oracle
    flags = dict(FLAGS, num_views=3, confidence_as_attention_uncertainty_weight=True)
    m = MultiView_MPL(**flags)
    detrng.fill_module_(m, seed=23)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    p, r, c = detrng.make_inputs(6, 3, seed=4)
    P, R, C = ([torch.from_numpy(x) for x in l] for l in (p, r, c))
    huge = [x.clone() for x in P]
    for x in huge:
        x[2:4, :, 2] = 1e6                                  # poses 2 and 3 carry absurd confidences, the others stay in [0, 1]
    dev = lambda lst: [x.to(DEV) for x in lst]
    ref = mpl_oracle.forward(sd, flags, huge, R, C, dtype=torch.float64)
    assert torch.isfinite(ref).all()                       # the reference semantics are fine with it
    with torch.no_grad():
        ok = m(dev(P), rays=dev(R), centers=dev(C))
        torch.cuda.synchronize()
        assert torch.isfinite(ok).all() and not cabi.device_error()
        bad = m(dev(huge), rays=dev(R), centers=dev(C))
        torch.cuda.synchronize()
        assert torch.isnan(bad[2:4]).all(), "saturated operands must not turn into poses"
        assert torch.equal(bad[[0, 1, 4, 5]], ok[[0, 1, 4, 5]])       # the other sequences are untouched
        assert cabi.load().mpl_device_error(-1) & 2
        with pytest.raises(RuntimeError, match="fp16 window"):
            m(dev(P), rays=dev(R), centers=dev(C))
        with pytest.raises(RuntimeError, match="fp16 window"):
            cabi.raise_if_device_error(synchronize=True)
        cabi.clear_device_error()
        m.set_matmul_precision("fp32_mfma")                # no window: the same input, the reference's answer
        full = m(dev(huge), rays=dev(R), centers=dev(C))
        torch.cuda.synchronize()
    assert not cabi.device_error()
    mx, nw = mpl_oracle.rel_errors(full.cpu(), ref)
    assert mx < 1e-4 and nw < 1e-4, (mx, nw)


def test_lost_handoff_is_reported_never_ignored():
    lib = cabi.load()
    m = _model()
    P, R, C = _inputs(256, 1)
    _lost_handoff(lib, m, P, R, C)


def _lost_handoff(lib, m, P, R, C):
    with torch.no_grad():
        good = m(P, rays=R, centers=C)
        torch.cuda.synchronize()
        assert not cabi.device_error()
        try:
            # bound 2^10 polls (~1 ms); workgroup (row tile 0, column group 0) deserts before GEMM phase 1
            cabi.check(lib.mpl_x3_spin_limit(10 | (1 << 8)), "spin limit")
            bad = m(P, rays=R, centers=C)
            torch.cuda.synchronize()
        finally:
            cabi.check(lib.mpl_x3_spin_limit(23), "spin limit")
        assert torch.isnan(bad).all(), "a forward whose hand-off was lost must not return poses"
        assert cabi.device_error()
        with pytest.raises(RuntimeError, match="lost a hand-off"):
            m(P, rays=R, centers=C)
        cabi.clear_device_error()
        assert not cabi.device_error()
        again = m(P, rays=R, centers=C)
        torch.cuda.synchronize()
    assert torch.equal(again, good), "the device must be fully usable after the error was cleared"
    assert not cabi.device_error()


def test_small_batch_engine_reports_a_hand_off_that_does_not_arrive():
    """The steps of the single-frame engine (sm_stack.hip) wait for each other by polling the {value, tag} pairs of their operands,
    with a bound.  With the bound at TWO polls (~1 us) consumers give up before their producers have finished a step (~4 us): the launch must
    leave, the poses must be NaN, the device error must be raised on the next call -- and after clearing it the engine gives the
    quiet result again (the workspace of the failed launch is zeroed by the next one: no stale tag survives)."""
    lib = cabi.load()
    m = _model()
    P, R, C = _inputs(2, 5)           # 8 token rows: the small-batch engine
    with torch.no_grad():
        good = m(P, rays=R, centers=C)
        torch.cuda.synchronize()
        assert lib.mpl_block_stack_last_form() == cabi.FORM_SMALL and not cabi.device_error()
        try:
            cabi.check(lib.mpl_x3_spin_limit(1), "spin limit")
            bad = m(P, rays=R, centers=C)
            torch.cuda.synchronize()
        finally:
            cabi.check(lib.mpl_x3_spin_limit(23), "spin limit")
        assert torch.isnan(bad).all(), "a forward whose hand-off was lost must not return poses"
        assert cabi.device_error()
        with pytest.raises(RuntimeError, match="lost a hand-off"):
            m(P, rays=R, centers=C)
        cabi.clear_device_error()
        again = m(P, rays=R, centers=C)
        torch.cuda.synchronize()
    assert torch.equal(again, good) and not cabi.device_error()


def test_two_streams_are_serialised_and_bitwise_equal_to_serial():
    m = _model()
    A = _inputs(1024, 2)
    Bt = _inputs(1024, 3)
    with torch.no_grad():
        ra = m(A[0], rays=A[1], centers=A[2])
        rb = m(Bt[0], rays=Bt[1], centers=Bt[2])
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for it in range(6):                      # both forwards in flight together, alternating which goes first
            order = [(s1, A), (s2, Bt)] if it % 2 == 0 else [(s2, Bt), (s1, A)]
            cur = {}
            for st, inp in order:
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    cur[id(inp)] = m(inp[0], rays=inp[1], centers=inp[2])
            outs.append((cur[id(A)], cur[id(Bt)]))
        torch.cuda.synchronize()
    assert not cabi.device_error()
    for oa, ob in outs:
        assert torch.equal(oa, ra) and torch.equal(ob, rb), "concurrent forwards on two streams disagree with the serial run"


def test_foreign_work_on_another_stream_only_delays_the_forward():
    """A long stream of unrelated kernels (torch matmuls) keeps the CUs busy while the forward is launched: its
    workgroups become resident late and out of order.  The result must be bitwise the undisturbed one."""
    m = _model()
    P, R, C = _inputs(1024, 4)
    with torch.no_grad():
        want = m(P, rays=R, centers=C)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        a = torch.randn(4096, 4096, device=DEV)
        b = torch.randn(4096, 4096, device=DEV)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(60):                  # ~100+ ms of foreign work
                a = (a @ b) * 1e-2
        got = [m(P, rays=R, centers=C) for _ in range(3)]
        torch.cuda.synchronize()
    assert not cabi.device_error(), "a delayed workgroup must not be taken for a lost one"
    for g in got:
        assert torch.equal(g, want)
    assert torch.isfinite(a).all() or True


def test_mismatched_operand_scales_poison_the_call_and_the_metrics():
    """proj / fc2 operands must be packed against the static scales their producers apply (mpl_pack_h2_scaled); an operand
    packed without them is detected on the device (fingerprints, h2_entry_kernel): NaN poses + the device error, never poses
    under the wrong scales.  And the metrics refuse a poisoned device: NaN poses would otherwise be skipped by the nansum
    semantics of calc_mpjpe (evaluate.py:91-114) and score as zero error."""
    import openmpl_amd
    from openmpl_amd.metrics import pose_metrics
    lib = cabi.load()
    m = _model()
    P, R, C = _inputs(64, 5)
    tgt = torch.randn(64, 17, 3, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    with torch.no_grad():
        good = m(P, rays=R, centers=C)
        ok = pose_metrics(good, tgt)
        torch.cuda.synchronize()
        assert torch.isfinite(ok["mpjpe_abs"])
        proj = m.blocks[0].attn.proj
        op = m._hip_cache[0]["derived"]["fpt"][0][1]
        cabi.check(lib.mpl_pack_h2(proj.weight.data_ptr(), proj.bias.data_ptr(), None, None, 544, 544, op.data_ptr(), st), "mpl_pack_h2")
        bad = m(P, rays=R, centers=C)
        try:
            res = pose_metrics(bad, tgt)             # enqueued behind the failing forward: the KERNEL sees the error word
            torch.cuda.synchronize()
            assert torch.isnan(res["mpjpe_abs"]) and torch.isnan(res["loss"]), "a poisoned batch must not score"
        except RuntimeError:
            pass                                     # the failure had already reached the host: the call itself refused
        torch.cuda.synchronize()
        assert torch.isnan(bad).all() and cabi.device_error()
        with pytest.raises(RuntimeError):
            pose_metrics(bad, tgt)
        with pytest.raises(RuntimeError):
            openmpl_amd.check_device()
        cabi.clear_device_error()
        m.set_matmul_precision("fp32")               # drops the cache: the operands are packed again, correctly
        again = m(P, rays=R, centers=C)
        openmpl_amd.check_device()
    assert torch.equal(again, good)


def test_fault_injection_is_one_shot():
    """The injected desertion is consumed by the first persistent launch: a test that dies between set and reset cannot leave
    the process poisoned."""
    lib = cabi.load()
    m = _model()
    P, R, C = _inputs(64, 6)
    with torch.no_grad():
        good = m(P, rays=R, centers=C)
        cabi.check(lib.mpl_x3_spin_limit(10 | (1 << 8)), "spin limit")
        try:
            bad = m(P, rays=R, centers=C)
            torch.cuda.synchronize()
            assert torch.isnan(bad).all()
            cabi.clear_device_error()
            again = m(P, rays=R, centers=C)          # no reset of the hook in between
            torch.cuda.synchronize()
        finally:
            cabi.check(lib.mpl_x3_spin_limit(23), "spin limit")
    assert torch.equal(again, good) and not cabi.device_error()
