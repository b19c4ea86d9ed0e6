"""Shared fixture of the block-stack tools: one FPT block with packed operands of the chosen engine.
ENGINE=h2 (default: fp16x2, h2_gemm.hip) | b1 (bf16, b1_gemm.hip) in the environment."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmpl_amd import cabi

ENGINE = os.environ.get("ENGINE", "h2")
lib = cabi.load()
dev = "cuda"
st = lambda: torch.cuda.current_stream().cuda_stream


def make_block(D, seed=0):
    g = torch.Generator().manual_seed(seed)
    nbytes, pack = (lib.mpl_pack_h2_bytes, lib.mpl_pack_h2) if ENGINE == "h2" else (lib.mpl_pack_bf16_bytes, lib.mpl_pack_bf16)

    def operand(N, K, ln, in_scale=None):
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
        gam = (torch.rand(K, generator=g) + 0.5).to(dev); bet = (torch.randn(K, generator=g) * 0.1).to(dev)
        o = torch.empty(nbytes(N, K), dtype=torch.uint8, device=dev)
        if in_scale:        # h2: proj / fc2 are packed against the static output scales of their producers
            cabi.check(lib.mpl_pack_h2_scaled(W.data_ptr(), b.data_ptr(), in_scale, N, K, o.data_ptr(), st()), "pack")
        else:
            cabi.check(pack(W.data_ptr(), b.data_ptr(), gam.data_ptr() if ln else None, bet.data_ptr() if ln else None, N, K, o.data_ptr(), st()), "pack")
        return o
    blk = cabi.BlockWeights()
    h2 = ENGINE == "h2"
    qkv = operand(3 * D, D, True)
    proj = operand(D, D, False, lib.mpl_pack_h2_out_scale(qkv.data_ptr(), 3 * D, D) + 8 * D if h2 else None)
    fc1 = operand(2 * D, D, True)
    fc2 = operand(D, 2 * D, False, lib.mpl_pack_h2_out_scale(fc1.data_ptr(), 2 * D, D) if h2 else None)
    keep = [qkv, proj, fc1, fc2]
    if ENGINE == "h2":
        blk.qkv_h2, blk.proj_h2, blk.fc1_h2, blk.fc2_h2 = (k.data_ptr() for k in keep)
    else:
        blk.qkv_w16, blk.proj_w16, blk.fc1_w16, blk.fc2_w16 = (k.data_ptr() for k in keep)
    return (cabi.BlockWeights * 1)(blk), keep, g
