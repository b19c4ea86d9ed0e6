"""Time of the long-sequence attention (joints x views grid, KPTOK): n_seq x (n_tok = 17 V) tokens, D = 32, 8 heads of 4.
python tools/att_long_time.py [V] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmpl_amd import cabi
lib = cabi.load()
V, B = int(sys.argv[1]) if len(sys.argv) > 1 else 31, int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_tok, D, H = 17 * V, 32, 8
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * n_tok, 3 * D, generator=g).cuda()
out = torch.empty(B * n_tok, D, device="cuda")
st = torch.cuda.current_stream().cuda_stream
run = lambda: cabi.check(lib.mpl_token_attention(qkv.data_ptr(), B, n_tok, D, H, out.data_ptr(), st), "att")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
pairs = B * H * n_tok * n_tok
t = qkv.double().cpu().reshape(B, n_tok, 3, H, D // H).permute(2, 0, 3, 1, 4)[:, :2]
ref = (((t[0] @ t[1].transpose(-2, -1)) * (D // H) ** -0.5).softmax(-1) @ t[2]).transpose(1, 2).reshape(2 * n_tok, D)
err = (out[:2 * n_tok].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
print("V=%d B=%d: %.1f us per launch, %.2f cycles @2.4 GHz per (query, key) pair and SIMD lane-slot, max-scaled error %.2e"
      % (V, B, ms * 1e3, ms * 1e-3 * 2.4e9 * 1024 * 32 / pairs / 1, err))
