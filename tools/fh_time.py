import sys, ctypes as C, torch, time
sys.path.insert(0, "/root/repo")
from bench import build_model, make_batch, model_flags
from openmpl_amd import cabi
m = build_model(model_flags("chosen", 4, 12), torch.device("cuda"))
lib = cabi.load()
P, R, Cn = make_batch(1024, 4, "cuda", 1)
dev, B, P, R, Cn = m._check_inputs(P, R, Cn)
ent = m._marshal(dev)
x = torch.randn(4096, 544, device="cuda"); out = torch.empty(1024, 51, device="cuda")
st = torch.cuda.current_stream().cuda_stream
f = lambda: lib.mpl_fuse_head(C.byref(ent["cfg"]), C.byref(ent["weights"]), x.data_ptr(), 1024, out.data_ptr(), st)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("fuse_head alone: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
