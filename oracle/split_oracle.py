"""CPU restatement of the packed-operand layouts and arithmetic of the FPT GEMM engines (openmpl_amd/csrc/h2_phase.hpp):
the fp16x2 operands of the default fp32 engine (h2_gemm.hip, second half of this file) and the bf16 operands of the bf16
engine (b1_gemm.hip, first half).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ to check that the packed operands written by
mpl_pack_h2 / mpl_pack_bf16 are byte-for-byte what the definition says, in MFMA fragment order, with their trailer vectors
behind them, and that the three-product sum of the fp16x2 engine is as accurate as an fp32 product.  This is not a
restatement of the reference (the reference is plain fp32 PyTorch, oracle/mpl_oracle.py); it pins the build's own
derived operands.

    K = 136 G columns (G a multiple of 4), KT = K / 32 k-tiles, k-permuted so that a producer's two adjacent 16-column
    output tiles are one consumer fragment:
        k-tile t < 4G:        lane quarter kq, element j  <->  column 136 (t//4) + 32 (t%4) + 16 (j//4) + 4 kq + j%4
        k-tile t = 4G + u:    lane quarter kq, element j  <->  column 136 (4u + kq) + 128 + j     (the 8-column tails)
    bf16 operand: W1[N/136 groups][KS = ceil(KT / 2)][9 slots][2 k-tiles][64 lanes][8 bf16], slot s = column tile
    (0,1,2,3,8,4,5,6,7)[s]; lane = 16 kq + li holds bf16(gamma o W)[g*136 + tile*16 + li][col(2 ks + par, kq, j)] (zero where
    tile*16 + li >= 136 or the k-tile pads an odd KT); then fp32 c[N] = bias + W . beta and s[N] = sum_k bf16(gamma_k W_nk)
    (both from fp64 sums), and unused space up to the trailer size of an fp16x2 operand (5 N + 8 floats).
"""
import numpy as np
import torch


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (RNE) -> fp32."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """Upper 16 bits of fp32 values that are exactly representable in bf16."""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


SLOT_TILE = (0, 1, 2, 3, 8, 4, 5, 6, 7)


def x3_col(t: int, kq: int, j: int, G: int) -> int:
    if t < 4 * G:
        return 136 * (t // 4) + 32 * (t % 4) + 16 * (j // 4) + 4 * kq + (j % 4)
    return 136 * (4 * (t - 4 * G) + kq) + 128 + j


def k_permutation(K: int) -> np.ndarray:
    """cols[t, kq, j] = source column of element j of lane quarter kq in k-tile t; a permutation of range(K)."""
    assert K % 544 == 0
    G, KT = K // 136, K // 32
    cols = np.array([[[x3_col(t, kq, j, G) for j in range(8)] for kq in range(4)] for t in range(KT)])
    assert sorted(cols.reshape(-1).tolist()) == list(range(K))
    return cols


def b1_operand(W: np.ndarray, gamma: np.ndarray = None) -> np.ndarray:
    """uint16 array of the fragment bytes mpl_pack_bf16 must produce for an nn.Linear weight W[N][K] (gamma folded)."""
    N, K = W.shape
    assert N % 136 == 0 and K % 544 == 0
    Gn, KT = N // 136, K // 32
    KS = (KT + 1) // 2
    Wf = np.asarray(W, dtype=np.float32)
    if gamma is not None:
        Wf = (Wf * np.asarray(gamma, dtype=np.float32)[None, :]).astype(np.float32)
    Wp = np.zeros((Gn, 144, K), dtype=np.float32)
    Wp[:, :136] = Wf.reshape(Gn, 136, K)
    cols = k_permutation(K)                              # [KT][4][8]
    b = bf16_bits(bf16_round(Wp))[:, :, cols]            # [g][c][kt][kq][j]
    b = b.reshape(Gn, 9, 16, KT, 4, 8)                   # [g][tile][li][kt][kq][j]
    b = b.transpose(0, 3, 1, 4, 2, 5).reshape(Gn, KT, 9, 64, 8)[:, :, list(SLOT_TILE)]   # [g][kt][slot][lane][j]
    out = np.zeros((Gn, 2 * KS, 9, 64, 8), dtype=np.uint16)
    out[:, :KT] = b
    return out.reshape(Gn, KS, 2, 9, 64, 8).transpose(0, 1, 3, 2, 4, 5)      # [g][ks][slot][par][lane][j]


def b1_fold_vectors(W: np.ndarray, bias: np.ndarray, gamma: np.ndarray = None, beta: np.ndarray = None):
    """(c, s) fp32 vectors stored behind the bf16 fragments: s sums the ROUNDED weights the operand holds."""
    W = np.asarray(W, dtype=np.float32)
    c = np.asarray(bias, dtype=np.float64).copy()
    s = np.zeros(W.shape[0], dtype=np.float64)
    if gamma is not None:
        c += W.astype(np.float64) @ np.asarray(beta, dtype=np.float64)
        s = bf16_round((W * np.asarray(gamma, dtype=np.float32)[None, :]).astype(np.float32)).astype(np.float64).sum(1)
    return c.astype(np.float32), s.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# fp16x2 engine (openmpl_amd/csrc/h2_gemm.hip): two fp16 parts per operand under exact power-of-two scales, three
# partial products.  Same k permutation, 2 parts instead of 3:
#     W2[N/136][KT][9 slots][2 parts][64 lanes][8 fp16], then fp32 c[N], sc[N], sw[N], bound[N], so[N], meta[8]
#     sw_n = 2^(14 - e), max_k |gamma_k W_nk| = m 2^e with m in [0.5, 1)      (the scaled column maximum is in [2^13, 2^14))
#     sc_n = 1 / (sa sw_n), sa = 1024 for an operand with LayerNorm folded in, else 1
#     bound_n = sqrt(K) |gamma o W_n|_2 + |c_n|  (LayerNorm operands; 0 otherwise);
#     so_n = scale(bound_n): static scale of output column n as an operand of the next GEMM (round 4: per column; round 3 used
#     one per layer); a consumer packed against them stores W_nk / so_k;
#     meta = {scale(max bound), scale(max bound of the last third of the columns), their reciprocals, the two maxima, two fingerprints},
#     scale(v) = largest power of two p with p v <= 2^15.
H2_SA = 1024.0


def split2(x: np.ndarray):
    """fp32 -> (hi, lo) fp16 parts as fp32 arrays: hi = fp16(x), lo = fp16(x - hi) (RNE, subnormals kept)."""
    x = np.asarray(x, dtype=np.float32)
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float32).astype(np.float16).astype(np.float32)
    return hi, lo


def f16_bits(x: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)


def h2_window_scale(v: float) -> float:
    if not v > 0:
        return 1.0
    m, e = np.frexp(np.float32(32768.0) / np.float32(v))
    return float(np.ldexp(np.float32(1.0), int(min(120, max(-120, e - 1)))))


def _h2_factor(W, gamma, in_scale):
    """Per-k factor folded into the weights before their column scale: the LayerNorm gain, or the reciprocal of the static scales
    the A operand's columns arrive with (powers of two: exact)."""
    W = np.asarray(W, dtype=np.float32)
    if gamma is not None:
        return (W * np.asarray(gamma, dtype=np.float32)[None, :]).astype(np.float32)
    if in_scale is not None:
        return (W * (np.float32(1.0) / np.asarray(in_scale, dtype=np.float32))[None, :]).astype(np.float32)
    return W


def h2_trailer(W: np.ndarray, bias: np.ndarray, gamma: np.ndarray = None, beta: np.ndarray = None, in_scale: np.ndarray = None):
    """(c, sc, sw, bound, so, meta) fp32 arrays stored behind the fragments of an fp16x2 weight operand.
    so_n = scale(bound_n): the static scale of output column n when it travels on as a packed operand (LayerNorm operands; 1
    otherwise).  meta[6], meta[7] = 0.5 + sum of the binary exponents of so over all columns / over the last third (LayerNorm
    operands); meta[6] = the same over in_scale for a plain operand packed against one (0 without)."""
    W = np.asarray(W, dtype=np.float32)
    N, K = W.shape
    Wg = _h2_factor(W, gamma, in_scale)
    amax = np.abs(Wg).max(1)
    m, e = np.frexp(amax)
    sw = np.where(amax > 0, np.ldexp(np.float32(1.0), 14 - e), np.float32(1.0)).astype(np.float32)
    c = np.asarray(bias, dtype=np.float64).copy()
    if gamma is not None:
        c += W.astype(np.float64) @ np.asarray(beta, dtype=np.float64)
    c = c.astype(np.float32)
    sa = np.float32(H2_SA if gamma is not None else 1.0)
    sc = (np.float32(1.0) / (sa * sw)).astype(np.float32)
    if gamma is not None:
        nrm = np.sqrt((Wg.astype(np.float64) ** 2).sum(1))
        bound = ((np.sqrt(np.float64(K)) * nrm).astype(np.float32) + np.abs(c)).astype(np.float32)
        so = np.array([h2_window_scale(float(b)) for b in bound], dtype=np.float32)
    else:
        bound = np.zeros(N, dtype=np.float32)
        so = np.ones(N, dtype=np.float32)
    ball = float(bound.max())
    last = 3 * np.arange(N) >= 2 * N
    bv = float(bound[last].max())
    s_all, s_v = h2_window_scale(ball), h2_window_scale(bv)
    ex = lambda v: float(np.frexp(np.asarray(v, dtype=np.float32))[1].astype(np.int64).sum() - np.asarray(v).size)   # ilogb = frexp exponent - 1
    if gamma is not None:
        f6, f7 = ex(so) + 0.5, ex(so[last]) + 0.5
    else:
        f6, f7 = (ex(in_scale) + 0.5 if in_scale is not None else 0.0), 0.0
    meta = np.array([s_all, s_v, 1.0 / s_all, 1.0 / s_v, ball, bv, f6, f7], dtype=np.float32)
    return c, sc, sw, bound, so, meta


def h2_operand(W: np.ndarray, gamma: np.ndarray = None, in_scale: np.ndarray = None) -> np.ndarray:
    """uint16 array [N/136][KT][9][2][64][8] of the fragment bytes mpl_pack_h2 (gamma folded) / mpl_pack_h2_scaled (W_nk /
    in_scale_k) must produce for W[N][K]."""
    W = np.asarray(W, dtype=np.float32)
    N, K = W.shape
    assert N % 136 == 0 and K % 544 == 0
    Gn, KT = N // 136, K // 32
    Wg = _h2_factor(W, gamma, in_scale)
    amax = np.abs(Wg).max(1)
    m, e = np.frexp(amax)
    sw = np.where(amax > 0, np.ldexp(np.float32(1.0), 14 - e), np.float32(1.0)).astype(np.float32)
    Ws = (Wg * sw[:, None]).astype(np.float32)
    Wp = np.zeros((Gn, 144, K), dtype=np.float32)
    Wp[:, :136] = Ws.reshape(Gn, 136, K)
    cols = k_permutation(K)
    out = np.zeros((Gn, KT, 9, 2, 64, 8), dtype=np.uint16)
    for p, part in enumerate(split2(Wp)):
        b = f16_bits(part)[:, :, cols]                   # [g][c][kt][kq][j]
        b = b.reshape(Gn, 9, 16, KT, 4, 8).transpose(0, 3, 1, 4, 2, 5).reshape(Gn, KT, 9, 64, 8)
        out[:, :, :, p] = b[:, :, list(SLOT_TILE)]
    return out


def three_product_matmul(A: np.ndarray, W: np.ndarray) -> np.ndarray:
    """A . W^T from the three partial products of the fp16x2 engine (fp64 accumulation: isolates the split error); A and W
    are scaled into the fp16 window like the kernels do (one scale for A, one per row of W), the scales taken out again."""
    A = np.asarray(A, dtype=np.float32)
    W = np.asarray(W, dtype=np.float32)
    sa = np.float32(h2_window_scale(float(np.abs(A).max())))
    amax = np.abs(W).max(1)
    m, e = np.frexp(amax)
    sw = np.where(amax > 0, np.ldexp(np.float32(1.0), 14 - e), np.float32(1.0)).astype(np.float32)
    ah, al = (t.astype(np.float64) for t in split2(A * sa))
    wh, wl = (t.astype(np.float64) for t in split2(W * sw[:, None]))
    return (al @ wh.T + ah @ wl.T + ah @ wh.T) / (np.float64(sa) * sw.astype(np.float64))[None, :]
