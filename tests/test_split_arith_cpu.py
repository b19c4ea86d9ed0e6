"""The arithmetic behind the default fp32 path (operands split exactly into three bf16 terms, six partial products):
properties that hold on any machine, checked with the numpy restatement in oracle/split_oracle.py."""
import numpy as np
import torch

from oracle import split_oracle as so


def _samples():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-6, 1e-2, 1.0, 37.0, 1e8)])
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 3.0e38, -3.0e38, 1.1754944e-38, 65504.0,
                     0.1, 1 / 3, 2 ** -100, 255.99998], dtype=np.float32)
    return np.concatenate([x, edge])


def test_three_bf16_terms_reproduce_fp32_exactly():
    x = _samples()
    hi, mid, lo = so.split3(x)
    for part in (hi, mid, lo):      # every part is a bf16 number
        assert np.array_equal(so.bf16_round(part), part)
    total = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64)), "hi + mid + lo must equal the fp32 value exactly"
    nz = x != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_products_are_more_accurate_than_an_fp32_gemm():
    rng = np.random.default_rng(1)
    for M, K, N in ((64, 544, 136), (32, 1088, 272)):
        A = (rng.standard_normal((M, K)) * 1.3 + 0.2).astype(np.float32)
        W = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        six = so.six_product_matmul(A, W)
        fp32 = (torch.from_numpy(A) @ torch.from_numpy(W).T).numpy().astype(np.float64)
        scale = np.abs(ref).max()
        e6, e32 = np.abs(six - ref).max() / scale, np.abs(fp32 - ref).max() / scale
        assert e6 < 5e-8 and e6 < 0.2 * e32, (e6, e32)     # dropped terms (mid.lo, lo.mid, lo.lo) are below fp32 rounding


def test_fragment_layout_of_the_split_operand():
    rng = np.random.default_rng(2)
    W = rng.standard_normal((272, 64)).astype(np.float32)
    w3 = so.split_operand(W)
    assert w3.shape == (2, 2, 9, 3, 64, 8) and w3.nbytes == 2 * 2 * 27 * 1024
    hi, mid, lo = so.split3(W)
    g, kt, tile, lane, j = 1, 1, 3, 37, 5
    li, kq = lane & 15, lane >> 4
    n, k = g * 136 + tile * 16 + li, kt * 32 + 8 * kq + j
    for p, part in enumerate((hi, mid, lo)):
        assert w3[g, kt, tile, p, lane, j] == so.bf16_bits(part[n:n + 1, k:k + 1])[0, 0]
    assert not w3[:, :, 8, :, [l for l in range(64) if (l & 15) >= 8]].any(), "columns 136..143 of a group are padding"
