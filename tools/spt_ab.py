"""Time the fused SPT stage alone: MPL_SPT_ABL=mask python tools/spt_ab.py [batch] [views] [depth]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmpl_amd import cabi, detrng
from openmpl_amd.multiview_mpl import MultiView_MPL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
V = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lib = cabi.load()
m = MultiView_MPL(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=L, num_views=V, pose_3d_emb_learnable=True,
                  no_transformer_spt=bool(os.environ.get("SPT_NOBLOCKS")))
detrng.fill_module_(m, seed=11)
m = m.cuda().eval()
p, r, c = detrng.make_inputs(B, V, seed=1)
P = [torch.from_numpy(t).cuda() for t in p]; R = [torch.from_numpy(t).cuda() for t in r]; Cn = [torch.from_numpy(t).cuda() for t in c]
dev, B, P, R, Cn = m._check_inputs(P, R, Cn)
ent = m._marshal(dev)
inp = cabi.Inputs(); inp.batch = B
for v in range(V):
    inp.poses[v], inp.rays[v], inp.centers[v] = P[v].data_ptr(), R[v].data_ptr(), Cn[v].data_ptr()
xs = torch.empty(B * V, 544, device="cuda")
st = torch.cuda.current_stream().cuda_stream
fn = lambda: lib.mpl_spt_tokens(C.byref(ent["cfg"]), C.byref(ent["weights"]), C.byref(inp), xs.data_ptr(), st)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("SPT abl=%s B=%d V=%d L=%d: %.1f us  (%.1f us per block application, %.1f TFLOP/s)" % (os.environ.get("MPL_SPT_ABL", "0"), B, V, L, ms * 1e3, ms * 1e3 / (L + 1), (L + 1) * V * B * (16 * 17 * 32 * 32 + 4 * 17 * 17 * 32) / ms / 1e9))

if int(os.environ.get("MPL_SPT_ABL", "0")) & 16:
    torch.cuda.synchronize()
    r = xs.flatten()[: 32 * 8 * 8].reshape(32 * 8, 8).cpu()
    m = r.mean(0) / (L + 1)
    print("per block application (shader cycles): qkv %.0f  attention %.0f  proj %.0f  fc1+gelu %.0f  fc2 %.0f  | sum %.0f = %.1f us @2.4GHz"
          % (m[0], m[1], m[2], m[3], m[4], m[:5].sum(), m[:5].sum() / 2400.0))
