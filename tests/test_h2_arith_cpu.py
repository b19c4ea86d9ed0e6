"""The arithmetic behind the default fp32 path (openmpl_amd/csrc/h2_gemm.hip: operands split into two fp16 terms under
exact power-of-two scales, three partial products): properties that hold on any machine, checked with the numpy
restatement in oracle/split_oracle.py."""
import numpy as np
import torch

from oracle import split_oracle as so


def test_two_fp16_terms_capture_22_bits():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1.0, 37.0, 1e3, 2.0e4)])
    x = np.concatenate([x, np.array([0.0, -0.0, 1.0, -1.0, 65504.0, -65504.0, 1 / 3, 2.0 ** -14, 2.0 ** -24, 3e-6], dtype=np.float32)])
    x = x[np.abs(x) <= 65504.0]
    hi, lo = so.split2(x)
    for part in (hi, lo):           # every part is an fp16 number
        assert np.array_equal(part.astype(np.float16).astype(np.float32), part)
    err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - x.astype(np.float64))
    # |x - hi - lo| <= 2^-22 |x| down to the subnormal grid of fp16 (2^-24 absolute, half of it by rounding)
    assert np.all(err <= np.maximum(np.abs(x).astype(np.float64) * 2.0 ** -22, 2.0 ** -25))
    nz = np.abs(x) >= 2.0 ** -14
    assert np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -11)


def test_three_products_are_as_accurate_as_an_fp32_gemm():
    rng = np.random.default_rng(1)
    for M, K, N in ((64, 544, 136), (32, 1088, 272), (16, 2176, 136)):
        A = (rng.standard_normal((M, K)) * 1.3 + 0.2).astype(np.float32)
        W = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        three = so.three_product_matmul(A, W)
        fp32 = (torch.from_numpy(A) @ torch.from_numpy(W).T).numpy().astype(np.float64)
        scale = np.abs(ref).max()
        e3, e32 = np.abs(three - ref).max() / scale, np.abs(fp32 - ref).max() / scale
        # the representation error of the operands + the dropped lo.lo term stay below the rounding of an fp32 accumulation
        assert e3 < 1.5e-7 and e3 < 0.6 * e32, (e3, e32)


def test_scales_are_exact_powers_of_two_and_keep_every_operand_in_the_window():
    rng = np.random.default_rng(2)
    for scale in (1e-12, 3e-3, 1.0, 7e5):
        W = (rng.standard_normal((272, 544)) * scale).astype(np.float32)
        W[5] = 0.0                                            # an all-zero output column
        b = rng.standard_normal(272).astype(np.float32)
        g = (rng.random(544) + 0.5).astype(np.float32)
        e = (rng.standard_normal(544) * 0.1).astype(np.float32)
        c, sc, sw, bound, oscale, meta = so.h2_trailer(W, b, g, e)
        m, ex = np.frexp(sw)
        assert np.all(m == 0.5), "column scales must be powers of two"
        amax = np.abs(W * g[None, :]).max(1) * sw
        live = amax > 0
        assert np.all(amax[live] >= 2.0 ** 13) and np.all(amax[live] < 2.0 ** 14) and sw[5] == 1.0
        assert np.array_equal(sc, (1.0 / (np.float32(1024.0) * sw)).astype(np.float32))
        # the bound really bounds |LN(x) . W_n + b_n| for any x (here: adversarial x aligned with a row of gamma o W)
        Wg = (W * g[None, :]).astype(np.float64)
        n = int(np.argmax(bound))
        z = Wg[n] - Wg[n].mean()
        z = z / np.sqrt((z ** 2).mean())                       # a LayerNorm output (gain 1, mean 0, variance 1)
        out = z @ Wg.T + c.astype(np.float64)
        assert np.all(np.abs(out) <= bound.astype(np.float64) * (1 + 1e-6))
        # one static scale per output column: a power of two that brings the column's bound to the top of the fp16 window
        mo, _ = np.frexp(oscale)
        assert np.all(mo == 0.5) and np.all(oscale * bound <= 2.0 ** 15) and np.all(oscale * bound * 2 > 2.0 ** 15)
        assert meta[6] == np.float32(np.log2(oscale.astype(np.float64)).sum() + 0.5)
        for k, rng_cols in ((0, slice(None)), (1, slice(2 * 272 // 3 + (1 if (2 * 272) % 3 else 0), None))):
            bm = float(bound[rng_cols].max())
            mm, _ = np.frexp(np.float32(meta[k]))
            assert mm == 0.5 and meta[k] * bm <= 2.0 ** 15 and meta[k] * bm * 2 > 2.0 ** 15
            assert meta[2 + k] == np.float32(1.0) / np.float32(meta[k])


def test_fragment_layout_of_the_fp16x2_operand():
    rng = np.random.default_rng(3)
    W = rng.standard_normal((272, 544)).astype(np.float32)
    gam = (rng.random(544) + 0.5).astype(np.float32)
    w2 = so.h2_operand(W, gam)
    assert w2.shape == (2, 17, 9, 2, 64, 8) and w2.nbytes == 2 * 17 * 18 * 1024
    _, _, sw, _, _, _ = so.h2_trailer(W, np.zeros(272, np.float32), gam, np.zeros(544, np.float32))
    hi, lo = so.split2(((W * gam[None, :]).astype(np.float32) * sw[:, None]).astype(np.float32))
    for g, kt, slot, lane, j in ((1, 6, 3, 37, 5), (0, 16, 4, 23, 2), (1, 16, 7, 60, 7), (0, 0, 0, 0, 0)):
        li, kq = lane & 15, lane >> 4
        tile = so.SLOT_TILE[slot]
        n = g * 136 + tile * 16 + li
        k = so.x3_col(kt, kq, j, 4)
        if tile * 16 + li >= 136:
            assert not w2[g, kt, slot, :, lane, j].any()
            continue
        for p, part in enumerate((hi, lo)):
            assert w2[g, kt, slot, p, lane, j] == so.f16_bits(part[n:n + 1, k:k + 1])[0, 0]
    assert not w2[:, :, 4, :, [l for l in range(64) if (l & 15) >= 8]].any(), "columns 136..143 of a group are padding"


def test_normalised_layernorm_input_fits_the_fp16_window():
    """|z| = |(x - mean) rstd| 2^10 <= sqrt(K) 2^10 < 65504 for every row, whatever its scale and offset (K <= 2048)."""
    rng = np.random.default_rng(4)
    for K in (544, 1088, 2048):
        x = rng.standard_normal((64, K))
        x[0] = 0.0
        x[0, 3] = 1.0                                          # all the energy in one element: the extreme case
        x[1] = 1e15 * x[1]
        x[2] = 5.0 + 1e-3 * x[2]
        mu, var = x.mean(1, keepdims=True), x.var(1, keepdims=True)
        z = (x - mu) / np.sqrt(var + 1e-6) * so.H2_SA
        assert np.abs(z).max() <= np.sqrt(K) * so.H2_SA * (1 + 1e-9) < 65504.0


def test_per_column_static_scales_equilibrate_an_outlier_channel():
    """The operand pair (A_k so_k, W_nk / so_k): one producer column 1e5 x the others (a huge v bias, one fc1 row far outside
    the rest) must not cost the OTHER channels resolution.  With one scale per layer (round 3) the outlier set the fp16 window
    of every column; per column the three-product sum stays at the accuracy of the benign case."""
    rng = np.random.default_rng(7)
    K, N, M = 544, 136, 64
    bound = np.full(K, 14.0, np.float32)                       # sqrt(K) |gamma o W_k|_2 + |c_k| of ordinary producer columns
    bound[37] = 3e6                                            # the outlier's bound
    A = (rng.standard_normal((M, K)) * 0.6).astype(np.float32)  # typical activations sit sqrt(K) below their bound
    A[:, 37] = 1e5
    W = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
    W[:, 37] *= 1e-5                                           # the consumer's column for it keeps the output finite
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    so_k = np.array([so.h2_window_scale(float(b)) for b in bound], np.float32)

    def three(a_scale, w_factor):
        amax = np.abs(W * w_factor[None, :]).max(1)
        _, e = np.frexp(amax)
        sw = np.ldexp(np.float32(1.0), 14 - e).astype(np.float32)
        ah, al = (t.astype(np.float64) for t in so.split2((A * a_scale[None, :]).astype(np.float32)))
        wh, wl = (t.astype(np.float64) for t in so.split2((W * w_factor[None, :] * sw[:, None]).astype(np.float32)))
        return (al @ wh.T + ah @ wl.T + ah @ wh.T) / sw.astype(np.float64)[None, :]

    per_col = three(so_k, (np.float32(1.0) / so_k))
    layer = np.float32(so.h2_window_scale(float(bound.max())))
    per_layer = three(np.full(K, layer, np.float32), np.ones(K, np.float32)) / np.float64(layer)
    scale = np.abs(ref).max()
    e_col, e_layer = np.abs(per_col - ref).max() / scale, np.abs(per_layer - ref).max() / scale
    assert e_col < 2e-7, e_col
    assert e_layer > 10 * e_col, (e_layer, e_col)              # what the per-layer scale lost


def test_consumer_operand_packed_against_input_scales():
    rng = np.random.default_rng(8)
    W = rng.standard_normal((136, 544)).astype(np.float32)
    ins = np.ldexp(np.float32(1.0), rng.integers(-6, 12, 544)).astype(np.float32)
    c, sc, sw, bound, oscale, meta = so.h2_trailer(W, np.zeros(136, np.float32), None, None, ins)
    amax = np.abs(W / ins[None, :]).max(1) * sw
    assert np.all(amax >= 2.0 ** 13) and np.all(amax < 2.0 ** 14) and np.all(oscale == 1.0) and np.all(bound == 0.0)
    assert np.array_equal(sc, (np.float32(1.0) / sw).astype(np.float32))
    assert meta[6] == np.float32(np.log2(ins.astype(np.float64)).sum() + 0.5) and meta[7] == 0.0
    w2 = so.h2_operand(W, None, ins)
    hi, _ = so.split2(((W / ins[None, :]).astype(np.float32) * sw[:, None]).astype(np.float32))
    assert w2[0, 3, 2, 0, 21, 4] == so.f16_bits(hi[2 * 16 + 5:2 * 16 + 6, so.x3_col(3, 1, 4, 4):so.x3_col(3, 1, 4, 4) + 1])[0, 0]
