"""Where the SPT kernel's time goes at a single frame (spt_kernel<true>, one sequence per workgroup): launch time with parts switched
off (MPL_SPT_ABL: 1 attention, 2 GELU, 4 MFMA phases, 8 epilogue math; results garbage) and the per-phase stamps (16).
    python tools/spt_small_phase.py [B V]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import ctypes as C, torch, time
    from bench import build_model, make_batch, model_flags
    from openmpl_amd import cabi
    B, V = int(sys.argv[2]), int(sys.argv[3])
    m = build_model(model_flags("chosen", V, 12), torch.device("cuda"))
    m.set_matmul_precision(os.environ.get("PREC", "fp32"))      # at most 32 sequences: the staged fp32-MFMA kernel either way
    P, R, Cn = make_batch(B, V, "cuda", 1)
    lib = cabi.load()
    dev, B, P, R, Cn = m._check_inputs(P, R, Cn)
    ent = m._marshal(dev)
    inp = cabi.Inputs(); inp.batch = B
    for v in range(V):
        inp.poses[v], inp.rays[v], inp.centers[v] = P[v].data_ptr(), R[v].data_ptr(), Cn[v].data_ptr()
    xs = torch.zeros(max(B * V, 64), 544, device="cuda")
    run = lambda: cabi.check(lib.mpl_spt_tokens(C.byref(ent["cfg"]), C.byref(ent["weights"]), C.byref(inp), xs.data_ptr(), torch.cuda.current_stream().cuda_stream), "spt")
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    abl = int(os.environ.get("MPL_SPT_ABL", "0"))
    print("ABL=%3d  %.1f us" % (abl, dt * 1e6))
    if abl & 16:
        st = xs.flatten()[:8 * 8].reshape(8, 8).cpu()
        for w in range(8):
            print("  wave %d: qkv %6.0f att %6.0f proj %6.0f fc1 %6.0f fc2 %6.0f | staging requests %6.0f (s_memtime ticks, all applications)" % ((w,) + tuple(st[w, :6].tolist())))
else:
    B, V = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("1", "2")
    for abl in (0, 1, 2, 4, 8, 7, 15, 16, 0):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", B, V], env=dict(os.environ, MPL_SPT_ABL=str(abl)))
