"""(needs a laboratory build of spt.hip: bash tools/build_variants.sh -f spt.hip lab="" -> build_tmp/lib_lab.so; the product library ignores MPL_SPT_ABL)
Time of the fused SPT kernel with one phase compiled... switched off at a time (MPL_SPT_ABL bits: 8 qkv, 1 attention, 32 proj,
64 fc1 + GELU, 128 fc2; results are garbage): [SPT_V=2 SPT_B=256] python tools/spt_abl.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import ctypes as C, torch, time
    from bench import build_model, make_batch, model_flags
    from openmpl_amd import cabi
    V, B0 = int(os.environ.get("SPT_V", "4")), int(os.environ.get("SPT_B", "1024"))
    m = build_model(model_flags("chosen", V, 12), torch.device("cuda"))
    P, R, Cn = make_batch(B0, V, "cuda", 1)
    lib = cabi.load()
    dev, B, P, R, Cn = m._check_inputs(P, R, Cn)
    ent = m._marshal(dev)
    inp = cabi.Inputs(); inp.batch = B
    for v in range(V):
        inp.poses[v], inp.rays[v], inp.centers[v] = P[v].data_ptr(), R[v].data_ptr(), Cn[v].data_ptr()
    xs = torch.zeros(B * V, 544, device="cuda")
    run = lambda: cabi.check(lib.mpl_spt_tokens(C.byref(ent["cfg"]), C.byref(ent["weights"]), C.byref(inp), xs.data_ptr(), torch.cuda.current_stream().cuda_stream), "spt")
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("ABL=%3s  %.1f us" % (os.environ.get("MPL_SPT_ABL", "0"), dt * 1e6))
else:
    for abl in (0, 8, 1, 32, 64, 128, 9, 233, 0):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MPL_SPT_ABL=str(abl)))
