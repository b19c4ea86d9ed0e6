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
