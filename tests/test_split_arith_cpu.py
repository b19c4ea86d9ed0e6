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
    W = rng.standard_normal((272, 544)).astype(np.float32)
    gam = (rng.random(544) + 0.5).astype(np.float32)
    w3 = so.split_operand(W, gam)
    assert w3.shape == (2, 17, 9, 3, 64, 8) and w3.nbytes == 2 * 17 * 27 * 1024
    hi, mid, lo = so.split3((W * gam[None, :]).astype(np.float32))
    # a full k-tile (group 1 of the K columns, quarter 2) and the tail k-tile (the 8 last columns of every group)
    for g, kt, slot, lane, j in ((1, 6, 3, 37, 5), (0, 16, 4, 23, 2), (1, 16, 7, 60, 7)):
        li, kq = lane & 15, lane >> 4
        tile = so.SLOT_TILE[slot]
        n = g * 136 + tile * 16 + li
        if kt < 16:
            k = 136 * (kt // 4) + 32 * (kt % 4) + 16 * (j // 4) + 4 * kq + j % 4
        else:
            k = 136 * (4 * (kt - 16) + kq) + 128 + j
        if tile * 16 + li >= 136:
            assert not w3[g, kt, slot, :, lane, j].any()
            continue
        for p, part in enumerate((hi, mid, lo)):
            assert w3[g, kt, slot, p, lane, j] == so.bf16_bits(part[n:n + 1, k:k + 1])[0, 0]
    assert not w3[:, :, 4, :, [l for l in range(64) if (l & 15) >= 8]].any(), "columns 136..143 of a group are padding"
    assert so.SLOT_TILE[4] == 8


def test_activation_operand_round_trips_and_pads():
    """A3: every row of X appears once, exactly (hi + mid + lo), at its fragment position; padding rows are zero."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((70, 544)).astype(np.float32)
    for rpt in (64, 60):
        a3 = so.split_rows(X, rpt)
        tiles = -(-70 // rpt)
        assert a3.shape == (tiles, 4, 17, 3, 64, 8)
        cols = so.k_permutation(544)
        f = lambda bits: (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        tot = f(a3[:, :, :, 0]) + f(a3[:, :, :, 1]) + f(a3[:, :, :, 2])        # [tile][rg][kt][lane][j]
        for row in (0, 17, rpt - 1, rpt, 69):
            t, rl = divmod(row, rpt)
            rg, li = divmod(rl, 16)
            got = np.zeros(544)
            for kt in range(17):
                for kq in range(4):
                    got[cols[kt, kq]] = tot[t, rg, kt, kq * 16 + li]
            assert np.array_equal(got, X[row].astype(np.float64))
        if rpt == 60:
            assert not a3[0, 3, :, :, [l for l in range(64) if (l & 15) >= 12]].any()     # rows 60..63 of tile 0: padding


def test_folded_layernorm_identity():
    """rstd (x.(gamma o W)^T - mean s) + c == LN(x).W^T + b in exact arithmetic (checked in fp64)."""
    rng = np.random.default_rng(4)
    x = rng.standard_normal((5, 544)) * 2.0 + 0.7
    W, b = rng.standard_normal((136, 544)) * 0.04, rng.standard_normal(136)
    g, e = rng.random(544) + 0.5, rng.standard_normal(544) * 0.1
    mu, var = x.mean(1, keepdims=True), x.var(1, keepdims=True)
    rs = 1.0 / np.sqrt(var + 1e-6)
    ref = ((x - mu) * rs * g + e) @ W.T + b
    c, s = so.fold_vectors(W.astype(np.float32), b.astype(np.float32), g.astype(np.float32), e.astype(np.float32))
    W32, g32 = W.astype(np.float32), g.astype(np.float32)
    fold = rs * (x @ (W32 * g32).astype(np.float64).T - mu * s.astype(np.float64)) + c.astype(np.float64)
    assert np.abs(fold - ref).max() < 2e-6 * np.abs(ref).max()
