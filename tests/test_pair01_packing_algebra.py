"""CPU: the arithmetic design of the fused blocks-0+1 kernel (liftreg_amd/csrc/conv01_fused.hip), restated in numpy.

* split3: an fp32 value is the exact sum of three bf16 values (round to nearest even, residual, residual);
* six of the nine partial products w_i * d_j (i + j <= 2) reproduce the fp32 product to 2^-22 of its size, each product exact in fp32;
* the DENSE K packing of block 0: a voxel's nine bf16 (three channels x three splits) as the records E / F / G and FIVE weight
  patterns per tap carry exactly those 18 products (no product twice, none missing); the five-channel packing (records A1..A4,
  EIGHT patterns) exactly its 30.
The GPU tests compare the kernel with an fp64 convolution; this file pins WHY the kernel may claim fp32 arithmetic
(reference semantics: Conv3d in fp32, layers/layers.py:365-369)."""
import numpy as np


def bf16_rne(x):
    """fp32 -> the nearest bf16 (ties to even), returned as fp32 (v_cvt_pk_bf16_f32)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    d0 = bf16_rne(x)
    r = (x - d0).astype(np.float32)          # exact
    d1 = bf16_rne(r)
    r = (r - d1).astype(np.float32)          # exact
    d2 = bf16_rne(r)
    return d0, d1, d2


def test_split3_is_exact_and_products_are_exact_in_fp32():
    rs = np.random.RandomState(0)
    x = (rs.standard_normal(20000) * np.exp(rs.uniform(-8, 8, 20000))).astype(np.float32)
    w = (rs.standard_normal(20000) * np.exp(rs.uniform(-8, 8, 20000))).astype(np.float32)
    xs, ws = split3(x), split3(w)
    assert np.array_equal(xs[0].astype(np.float64) + xs[1] + xs[2], x.astype(np.float64))       # the three parts ARE the value
    assert np.array_equal(ws[0].astype(np.float64) + ws[1] + ws[2], w.astype(np.float64))
    six = np.zeros(x.shape, np.float64)
    for i in range(3):
        for j in range(3 - i):
            p64 = ws[i].astype(np.float64) * xs[j].astype(np.float64)
            assert np.array_equal(p64.astype(np.float32).astype(np.float64), p64)                # 8 x 8 significant bits: exact in fp32
            six += p64
    exact = w.astype(np.float64) * x.astype(np.float64)
    assert np.all(np.abs(six - exact) <= 2.0 ** -22 * np.abs(exact))                              # the three dropped products


def _wanted(nc):
    return sorted((c, i, j) for c in range(nc) for i in range(3) for j in range(3 - i))


def test_dense_records_of_three_channels_carry_the_18_products_once():
    # record element = (channel, data split j); weight element = (channel, weight split i) or None (a zero)
    E = [(0, 0), (1, 0), (2, 0), (0, 1)]
    F = [(1, 1), (2, 1), (0, 2), (1, 2)]
    G = [(2, 2), (1, 1), (2, 1), None]
    patterns = [(E, [(0, 0), (1, 0), (2, 0), (0, 0)]), (E, [(0, 1), (1, 1), (2, 1), (0, 1)]), (E, [(0, 2), (1, 2), (2, 2), None]),
                (F, [(1, 0), (2, 0), (0, 0), (1, 0)]), (G, [(2, 0), (1, 1), (2, 1), None])]
    got = []
    for rec, wts in patterns:
        for d, w in zip(rec, wts):
            if d is None or w is None:
                continue
            assert d[0] == w[0], "a slot multiplies a weight and a datum of the same channel"
            got.append((d[0], w[1], d[1]))
    assert sorted(got) == _wanted(3) and len(got) == 18
    assert 16 * 8 < 27 * len(patterns) <= 17 * 8      # 135 record slots: 17 MFMAs of 8 slots (16 would hold 128)


def test_dense_records_of_five_channels_carry_the_30_products_once():
    A1 = [(0, 0), (1, 0), (2, 0), (3, 0)]
    A2 = [(4, 0), (0, 1), (1, 1), (2, 1)]
    A3 = [(3, 1), (4, 1), (0, 2), (1, 2)]
    A4 = [(2, 2), (3, 2), (4, 2), (4, 0)]
    patterns = [(A1, [(0, 0), (1, 0), (2, 0), (3, 0)]), (A1, [(0, 1), (1, 1), (2, 1), (3, 1)]), (A1, [(0, 2), (1, 2), (2, 2), (3, 2)]),
                (A2, [(4, 0), (0, 0), (1, 0), (2, 0)]), (A2, [(4, 1), (0, 1), (1, 1), (2, 1)]),
                (A3, [(3, 0), (4, 0), (0, 0), (1, 0)]), (A3, [(3, 1), (4, 1), None, None]),
                (A4, [(2, 0), (3, 0), (4, 0), (4, 2)])]
    got = []
    for rec, wts in patterns:
        for d, w in zip(rec, wts):
            if w is None:
                continue
            assert d[0] == w[0]
            got.append((d[0], w[1], d[1]))
    assert sorted(got) == _wanted(5) and len(got) == 30
    assert 27 * len(patterns) == 27 * 8                                       # 216 record slots = 27 MFMAs of 8, no padding


def test_dense_convolution_sum_equals_the_six_product_sum():
    """One output of block 0 (27 taps x 3 channels) accumulated through the records and patterns above equals the plain six-product
    accumulation term by term (fp64 bookkeeping of exact fp32 products) and lies within 27 * 3 * 2^-22 of the fp64 convolution."""
    rs = np.random.RandomState(1)
    x = rs.uniform(-1, 1, (27, 3)).astype(np.float32)
    w = (rs.standard_normal((27, 3)) / 9).astype(np.float32)
    xs, ws = split3(x), split3(w)
    plain = sum(float(ws[i][t, c]) * float(xs[j][t, c]) for t in range(27) for c in range(3) for i in range(3) for j in range(3 - i))
    E = lambda t: [xs[0][t, 0], xs[0][t, 1], xs[0][t, 2], xs[1][t, 0]]
    F = lambda t: [xs[1][t, 1], xs[1][t, 2], xs[2][t, 0], xs[2][t, 1]]
    G = lambda t: [xs[2][t, 2], xs[1][t, 1], xs[1][t, 2], 0.0]
    dense = 0.0
    for t in range(27):
        pats = [(E(t), [ws[0][t, 0], ws[0][t, 1], ws[0][t, 2], ws[0][t, 0]]), (E(t), [ws[1][t, 0], ws[1][t, 1], ws[1][t, 2], ws[1][t, 0]]),
                (E(t), [ws[2][t, 0], ws[2][t, 1], ws[2][t, 2], 0.0]), (F(t), [ws[0][t, 1], ws[0][t, 2], ws[0][t, 0], ws[0][t, 1]]),
                (G(t), [ws[0][t, 2], ws[1][t, 1], ws[1][t, 2], 0.0])]
        dense += sum(float(a) * float(b) for rec, wt in pats for a, b in zip(rec, wt))
    exact = float(np.sum(w.astype(np.float64) * x.astype(np.float64)))
    assert abs(dense - plain) <= 1e-15 * max(1.0, abs(plain))
    assert abs(dense - exact) <= 81 * 2.0 ** -22 * float(np.sum(np.abs(w.astype(np.float64) * x)))


def test_staged_quads_cover_what_a_column_reads_and_half_quads_never_straddle_a_row():
    """Ring 0 of a column of 4 x 8 block-1 outputs (TX = 8): record p of a row = volume x - (2 ox0 - 4).  Block 0's region needs
    x = 2 ox0 - 2 .. 2 ox0 + 16 = records 2 .. 20; the staging fetches FIVE quads from record 2 on (records 2 .. 21) as 8-byte
    halves.  For every column and every H the launcher accepts (H % 4 == 0) a half is wholly inside or wholly outside the row, so
    one bounds flag per half is the convolution's zero padding."""
    QS0, NQS = 2, 5
    for H in (4, 8, 12, 16, 24, 32, 100, 160, 256):
        for ox0 in range(0, (H // 2 + 7) // 8 * 8, 8):
            X0a = 2 * ox0 - 4
            need = {rx + tx + 2 for rx in range(17) for tx in range(3)}            # region voxel rx, tap tx -> record
            staged = {QS0 + 4 * iq + e for iq in range(NQS) for e in range(4)}
            assert need <= staged and min(need) == 2 and max(need) == 20
            for iq in range(NQS):
                xi = X0a + QS0 + 4 * iq
                assert xi % 2 == 0                                               # 8-byte aligned halves (rows are 16-byte aligned)
                for half in (xi, xi + 2):
                    inside = [0 <= half + e < H for e in range(2)]
                    assert inside[0] == inside[1], (H, ox0, iq, half)
