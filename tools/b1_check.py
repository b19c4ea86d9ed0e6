"""bf16 engine (b1_gemm.hip) check + timing beside the fp32 engine, one process:
    python tools/b1_check.py [--fast]
 * parity: new engine vs the oracle's bf16 emulation (fp64) on goldens' models at several batch sizes (one-tile form, pair
   form, ragged tiles, generic / in-register attention), persistent launch vs one launch per GEMM, pair form vs one-tile form
   (bitwise);
 * speed: ms per forward and per stack launch, bf16 vs fp32 engine, V = 8 B = 1024 depth 2 / 12 (BASELINE configs[2])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, make_batch, model_flags  # noqa: E402
from openmpl_amd import cabi, detrng  # noqa: E402
from openmpl_amd.multiview_mpl import MultiView_MPL  # noqa: E402
from oracle import mpl_oracle  # noqa: E402
from tests.util import golden_state_dict, load_golden  # noqa: E402

dev = torch.device("cuda", 0)
lib = cabi.load()
fast = "--fast" in sys.argv
OLD = 1 << 4


def mode(bits):
    cabi.check(lib.mpl_x3_stack_mode(bits), "mode")


def run(m, P, R, Cn):
    with torch.no_grad():
        return m([x.to(dev) for x in P], rays=[x.to(dev) for x in R], centers=[x.to(dev) for x in Cn]).cpu()


bad = 0
for name, B in [("chosen_v8_b4_l2", 64), ("chosen_v8_b4_l2", 1024), ("chosen_v4_b8_l12", 64), ("chosen_v5_b19_l2", 19),
                ("full_v8_b4_l2", 64), ("chosen_v4_b8_l2", 1030), ("chosen_v2_b1_l12", 130)]:
    if fast and B > 100:
        continue
    g = load_golden(name)
    sd = golden_state_dict(name, g)
    m = MultiView_MPL(**g["flags"])
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    V = g["flags"]["num_views"]
    p, r, c = detrng.make_inputs(B, V, seed=77)
    P, R, Cn = ([torch.from_numpy(x) for x in l] for l in (p, r, c))
    emu = mpl_oracle.forward(sd, g["flags"], P, R, Cn, dtype=torch.float64, fpt_matmul_bf16=True)
    ref = mpl_oracle.forward(sd, g["flags"], P, R, Cn, dtype=torch.float64)
    mode(8)                                         # no small-batch engine
    m.set_matmul_precision("bf16")
    out = run(m, P, R, Cn)
    mode(8 | 1)
    out_g = run(m, P, R, Cn)                        # one launch per GEMM
    mode(8 | (1 << 1))
    out_1 = run(m, P, R, Cn)                        # forced one-tile form
    mode(8 | (2 << 1))
    out_2 = run(m, P, R, Cn)                        # forced pair form
    out_o = out
    mode(0)
    e = mpl_oracle.rel_errors
    deep = g["flags"]["depth"] > 2
    ok = e(out, emu)[0] < (3e-3 if deep else 1e-3) and torch.isfinite(out).all()
    bw = torch.equal(out_1, out_2)
    bad += (not ok) + (not bw)
    print("%-18s B=%4d  new vs emu %.2e/%.2e  new vs fp64 ref %.2e  per-GEMM vs chain %.1e  pair==one-tile %s  %s"
          % (name, B, *e(out, emu), e(out, ref)[0], e(out_g, out)[0], bw, "ok" if ok and bw else "FAIL"), flush=True)
    del m

print("parity failures:", bad, flush=True)
for fs, V, L, B in [("chosen", 8, 2, 1024), ("chosen", 8, 12, 1024), ("chosen", 4, 12, 1024), ("full", 8, 2, 1024), ("chosen", 2, 12, 256)]:
    m = build_model(model_flags(fs, V, L), dev)
    b = [make_batch(B, V, dev, seed=1, step=s) for s in range(2)]
    res = {}
    for rep in range(2):
        for prec, bits in (("bf16", 0), ("fp32", 0)):
            mode(bits)
            m.set_matmul_precision(prec)
            with torch.no_grad():
                for i in range(3):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                n = 10
                t0 = time.perf_counter()
                for i in range(n):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e3
                cabi.profile_start()
                for i in range(4):
                    m(b[i % 2][0], rays=b[i % 2][1], centers=b[i % 2][2])
                torch.cuda.synchronize()
                pr = cabi.profile_stop()
            res.setdefault(prec, []).append((dt, pr["gemm"][0] / 4))
    mode(0)
    f = lambda k: "%.3f ms / stack %.3f ms (%.0f poses/s)" % (min(x[0] for x in res[k]), min(x[1] for x in res[k]), B / min(x[0] for x in res[k]) * 1e3)
    print("%-6s V=%d L=%2d B=%4d | bf16: %s | fp32: %s" % (fs, V, L, B, f("bf16"), f("fp32")), flush=True)
    del m
sys.exit(1 if bad else 0)
