"""Persistent row-tile chain kernel vs one launch per GEMM (MPL_X3_LAUNCHES=1 in a child process): same goldens, timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmpl_amd.multiview_mpl import MultiView_MPL
from oracle import mpl_oracle
from tests.util import golden_inputs, golden_state_dict, load_golden

for name in sys.argv[1:] or ["chosen_v2_b1_l12", "chosen_v4_b8_l2", "chosen_v4_b8_l12", "full_v4_b8_l2", "chosen_v5_b19_l2", "chosen_v8_b4_l2"]:
    g = load_golden(name)
    m = MultiView_MPL(**g["flags"])
    m.load_state_dict(golden_state_dict(name, g), strict=True)
    m = m.cuda().eval()
    P, R, C = golden_inputs(g, "cuda")
    from openmpl_amd import cabi
    with torch.no_grad():
        cabi.load().mpl_x3_stack_mode(1)
        ref = m(P, rays=R, centers=C).clone(); torch.cuda.synchronize()
        cabi.load().mpl_x3_stack_mode(0)
        out = m(P, rays=R, centers=C); torch.cuda.synchronize()
        dmax = float((out - ref).abs().max())   # the two launch modes agree to rounding (the fc2 phase differs by <= 4 ulp)
        assert dmax <= 1e-5 * float(ref.abs().max()), "persistent chains differ from one launch per GEMM: %g" % dmax
        t0 = time.perf_counter(); out = m(P, rays=R, centers=C); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    mx, nw = mpl_oracle.rel_errors(out.cpu(), torch.from_numpy(g["out"]))
    print("%-22s %.2e %.2e  %.3f ms  finite=%s" % (name, mx, nw, dt * 1e3, bool(torch.isfinite(out).all())), flush=True)
