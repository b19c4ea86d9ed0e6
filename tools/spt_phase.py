"""Phase stamps of the fused SPT kernel (MPL_SPT_ABL=16) and ablation timings: python tools/spt_phase.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import ctypes as C, torch, time
    from bench import build_model, make_batch, model_flags
    from openmpl_amd import cabi
    m = build_model(model_flags("chosen", 4, 12), torch.device("cuda"))
    P, R, Cn = make_batch(1024, 4, "cuda", 1)
    lib = cabi.load()
    dev, B, P, R, Cn = m._check_inputs(P, R, Cn)
    ent = m._marshal(dev)
    inp = cabi.Inputs(); inp.batch = B
    for v in range(4):
        inp.poses[v], inp.rays[v], inp.centers[v] = P[v].data_ptr(), R[v].data_ptr(), Cn[v].data_ptr()
    xs = torch.zeros(B * 4, 544, device="cuda")
    run = lambda: cabi.check(lib.mpl_spt_tokens(C.byref(ent["cfg"]), C.byref(ent["weights"]), C.byref(inp), xs.data_ptr(), torch.cuda.current_stream().cuda_stream), "spt")
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    msg = "ABL=%s  %.1f us" % (os.environ.get("MPL_SPT_ABL", "0"), dt * 1e6)
    if int(os.environ.get("MPL_SPT_ABL", "0")) & 16:
        t = xs.cpu().reshape(-1)[:32 * 8 * 8].reshape(-1, 8)[:, :5]
        msg += "  per-application phase cycles (mean over waves): qkv %.0f | attention %.0f | proj %.0f | fc1+gelu %.0f | fc2 %.0f" % tuple((t.mean(0) / 13).tolist())
    print(msg)
else:
    for abl in (0, 16, 1, 2, 4, 5, 7):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MPL_SPT_ABL=str(abl)))
