"""The arithmetic behind ops.set_conv_arithmetic('split') (csrc/common.h: syn_split3, csrc/conv_split.hip), restated in numpy and
checked without a GPU: an fp32 number is the EXACT sum of three bfloat16 numbers obtained by rounding to nearest even what the
previous pieces left, each of the six partial products kept by the kernels is exact in fp32, and the three products left out
are bounded by 2^-23 of the product (2^-24 at most and 2^-27 rms over random operands) -- the claims the design rests on.  (The kernels themselves are compared with float64 and with
the fp32 matrix instructions on the GPU: tests/test_split_gpu.py.)"""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 computes for finite inputs)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def split3(a):
    a = np.asarray(a, dtype=np.float32)
    a0 = bf16_rne(a)
    r1 = a - a0                      # float32 subtraction: exact (checked below)
    a1 = bf16_rne(r1)
    r2 = r1 - a1
    a2 = bf16_rne(r2)
    return a0, a1, a2


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 7, n).astype(np.float32)
    edge = np.array([1.0, -1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 255.0 / 256, 1.99999988, 3.0e-30, -7.0e30, 0.0, 100.01, -37.004],
                    dtype=np.float32)
    return np.concatenate([x, edge])


def test_three_bf16_pieces_sum_to_the_fp32_number_exactly():
    a = _samples(200000, 0)
    a0, a1, a2 = split3(a)
    d = a.astype(np.float64)
    # the subtractions are exact, the third piece takes all that is left, the pieces shrink by 2^-8 each (2^-9 for RNE)
    assert np.array_equal((d - a0) , (a - a0).astype(np.float64))
    assert np.array_equal(a0.astype(np.float64) + a1.astype(np.float64) + a2.astype(np.float64), d)
    nz = a != 0
    assert np.all(np.abs(a1[nz].astype(np.float64)) <= 2.0 ** -8 * np.abs(d[nz]))
    assert np.all(np.abs(a2[nz].astype(np.float64)) <= 2.0 ** -16 * np.abs(d[nz]))
    # every piece is a bfloat16 number: 16 low bits clear
    for p in (a0, a1, a2):
        assert not np.any(p.view(np.uint32) & 0xffff)


def test_six_partial_products_reproduce_the_product_within_fp32_rounding():
    a, b = _samples(200000, 1), _samples(200000, 2)
    A, B = split3(a), split3(b)
    kept = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]      # conv_split.hip: the products the kernels accumulate
    exact = a.astype(np.float64) * b.astype(np.float64)
    s = np.zeros_like(exact)
    for i, j in kept:
        p64 = A[i].astype(np.float64) * B[j].astype(np.float64)
        # 8 x 8 significand bits: the product of two bfloat16 numbers is exact in fp32 (the matrix core's accumulator type)
        with np.errstate(over='ignore', under='ignore'):
            p32 = (A[i] * B[j]).astype(np.float64)
        ok = np.isfinite(p32) & (np.abs(p64) > 1e-35)
        assert np.array_equal(p32[ok], p64[ok])
        s += p64
    ok = np.abs(exact) > 1e-30
    rel = np.abs(s[ok] - exact[ok]) / np.abs(exact[ok])
    # |a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|  =>  |a1 b2 + a2 b1 + a2 b2| < 2^-23 |a b|; over random operands: 2^-24.2 at most,
    # 2^-27.4 rms -- within the 2^-24 an fp32 multiply-add may lose on the product
    assert rel.max() < 2.0 ** -23.9, np.log2(rel.max())
    assert np.sqrt((rel ** 2).mean()) < 2.0 ** -27, np.log2(np.sqrt((rel ** 2).mean()))
    nine = sum(A[i].astype(np.float64) * B[j].astype(np.float64) for i in range(3) for j in range(3))
    assert np.array_equal(nine[ok], exact[ok])        # all nine partial products: the exact product
    # what a bf16-only product would give, for scale: 2^-9 .. 2^-8
    rel_bf16 = np.abs(A[0][ok].astype(np.float64) * B[0][ok].astype(np.float64) - exact[ok]) / np.abs(exact[ok])
    assert rel_bf16.max() > 2.0 ** -9


def test_split_dot_product_matches_fp32_accumulation():
    """a 648-term dot product (27 taps x 24 channels, one output of the U-Net's 24-channel layers) against float64.  A matrix
    instruction is modelled as an exact dot product over its K extent followed by ONE rounding into the fp32 accumulator:
    v_mfma_f32_16x16x32_bf16 -> 32 terms per instruction, six instructions (partial products) per K block;
    v_mfma_f32_16x16x4_f32 (the fp32_mfma arithmetic) -> 4 terms per instruction.  (What the hardware does inside an
    instruction is not specified; the GPU tests compare the real kernels with float64.)"""
    rng = np.random.default_rng(3)
    n, k = 4000, 648
    x = rng.standard_normal((n, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    exact = (x.astype(np.float64) * w.astype(np.float64)).sum(1)
    X, W = [p.astype(np.float64) for p in split3(x)], [p.astype(np.float64) for p in split3(w)]
    acc = np.zeros(n, dtype=np.float32)
    for t in range(0, k, 32):                            # one K block: six MFMAs, smallest partial products first
        for i, j in [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]:
            acc = (acc.astype(np.float64) + (W[i][:, t:t + 32] * X[j][:, t:t + 32]).sum(1)).astype(np.float32)
    plain = np.zeros(n, dtype=np.float32)
    for t in range(0, k, 4):
        plain = (plain.astype(np.float64) + (w[:, t:t + 4].astype(np.float64) * x[:, t:t + 4].astype(np.float64)).sum(1)).astype(np.float32)
    scale = np.sqrt((exact ** 2).mean())
    e_split = np.sqrt(((acc - exact) ** 2).mean()) / scale
    e_plain = np.sqrt(((plain - exact) ** 2).mean()) / scale
    # 126 roundings (+ the omitted terms) against 162: the same accuracy class
    assert e_split < 1.5 * e_plain + 1e-9 and e_split < 1e-6, (e_split, e_plain)
