"""CPU restatement of the split-operand layout and arithmetic of openmpl_amd/csrc/x3_gemm.hip.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ to check (a) that the three bf16 parts written by
mpl_split_bf16x3 are byte-for-byte what the definition says, in MFMA fragment order, and (b) that the six-product
sum the kernels accumulate is at least as accurate as an fp32 product.  This is not a restatement of the reference
(the reference is plain fp32 PyTorch, oracle/mpl_oracle.py); it pins the build's own derived operand.

    x = hi + mid + lo,  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (round to nearest even)
    W3[N/136 groups][K/32 k-tiles][9 column tiles][3 parts][64 lanes][8 bf16];
    lane = 16 * kq + li holds W[g*136 + tile*16 + li][kt*32 + 8*kq + j], j = 0..7 (zero where tile*16 + li >= 136)
"""
import numpy as np
import torch


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (RNE) -> fp32."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def split3(x: np.ndarray):
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_round(x)
    r = (x - hi).astype(np.float32)
    mid = bf16_round(r)
    lo = bf16_round((r - mid).astype(np.float32))
    return hi, mid, lo


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """Upper 16 bits of fp32 values that are exactly representable in bf16."""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def split_operand(W: np.ndarray) -> np.ndarray:
    """uint16 array of the bytes mpl_split_bf16x3 must produce for an nn.Linear weight W[N][K]."""
    N, K = W.shape
    assert N % 136 == 0 and K % 32 == 0
    G, KT = N // 136, K // 32
    Wp = np.zeros((G, 144, K), dtype=np.float32)
    Wp[:, :136] = W.reshape(G, 136, K)
    parts = split3(Wp)                                   # each [G][144][K]
    out = np.zeros((G, KT, 9, 3, 64, 8), dtype=np.uint16)
    for p, part in enumerate(parts):
        b = bf16_bits(part).reshape(G, 9, 16, KT, 4, 8)  # [g][tile][li][kt][kq][j]
        out[:, :, :, p] = b.transpose(0, 3, 1, 4, 2, 5).reshape(G, KT, 9, 64, 8)   # lane = kq*16 + li
    return out


def six_product_matmul(A: np.ndarray, W: np.ndarray) -> np.ndarray:
    """A . W^T from the six partial products the kernels accumulate (here in fp64, to isolate the split error)."""
    ah, am, al = (t.astype(np.float64) for t in split3(A))
    wh, wm, wl = (t.astype(np.float64) for t in split3(W))
    return al @ wh.T + ah @ wl.T + am @ wm.T + am @ wh.T + ah @ wm.T + ah @ wh.T
