"""The arithmetic claim behind the fp32 tile loops (corenav_gp_amd/csrc/cgp_kernels_fused.hpp, "fp32 products on the bf16
matrix cores"), restated in numpy and checked on the CPU: a float splits EXACTLY into three bf16 values by truncation, and
the six products the kernels keep -- accumulated in fp32, smallest first, two terms per K = 32 MFMA -- reproduce an fp32
inner product of 16 terms to fp32's own rounding level.  (The GPU side is covered by every fp32 parity test and by
tests/fuzz/fuzz_parity.py; this file pins the reasoning, so that a change of the split -- rounding instead of truncation,
another term order, a dropped term -- has a test that says what it costs.)"""
import numpy as np


def bf16_trunc(x):
    """Top 16 bits of an fp32 value (what `v_and_b32 0xffff0000` keeps), as fp32."""
    return (np.asarray(x, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    x0 = bf16_trunc(x)
    r1 = (x - x0).astype(np.float32)
    x1 = bf16_trunc(r1)
    r2 = (r1 - x1).astype(np.float32)
    x2 = bf16_trunc(r2)
    return x0, x1, x2, r1, r2


def wide_floats(rng, n):
    """Random fp32 values with exponents spread over the range a Cholesky factor holds, both signs, full mantissas."""
    m = rng.uniform(1.0, 2.0, n)
    e = rng.integers(-30, 31, n)
    return (np.where(rng.random(n) < 0.5, -1.0, 1.0) * np.ldexp(m, e)).astype(np.float32)


def test_split_is_exact_and_three_planes_suffice():
    rng = np.random.default_rng(0)
    x = wide_floats(rng, 200_000)
    x0, x1, x2, r1, r2 = split3(x)
    # the two differences are exact in fp32 (a prefix of the mantissa is removed each time) ...
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - x0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - x1.astype(np.float64))
    # ... what is left after two planes has at most 8 significant bits, so the third plane takes all of it
    assert np.array_equal(x2, r2)
    assert np.array_equal(x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64), x.astype(np.float64))
    # plane magnitudes: 2^-8 and 2^-16 of the value at most (what makes the dropped terms small)
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(x1) <= ax * 2.0 ** -7) and np.all(np.abs(x2) <= ax * 2.0 ** -15)


def bx6_dot16(a, b):
    """One 16-column chunk of one accumulator entry as bx6_compute does it: three K = 32 MFMAs, each the fp32 sum of 32 exact
    products, added to the accumulator in the order [a0|a2].[b2|b0], [a0|a1].[b1|b0], [a0|a1].[b0|b1].  The MFMA's internal
    summation is modelled as exact (fp64) followed by one rounding per instruction -- an upper bound on its precision is not
    needed for the claim, a lower one is not assumed by it: the bound below leaves room for a rounding per product."""
    a0, a1, a2 = split3(a)[:3]
    b0, b1, b2 = split3(b)[:3]
    d = lambda u, v: np.sum(u.astype(np.float64) * v.astype(np.float64), axis=-1)
    acc = np.zeros(a.shape[:-1], dtype=np.float32)
    for t in (d(a0, b2) + d(a2, b0), d(a0, b1) + d(a1, b0), d(a0, b0) + d(a1, b1)):
        acc = (acc.astype(np.float64) + t).astype(np.float32)
    return acc


def test_six_terms_reproduce_the_fp32_inner_product():
    rng = np.random.default_rng(1)
    a = wide_floats(rng, 50_000 * 16).reshape(-1, 16)
    b = wide_floats(rng, 50_000 * 16).reshape(-1, 16)
    exact = np.sum(a.astype(np.float64) * b.astype(np.float64), axis=1)
    scale = np.sum(np.abs(a.astype(np.float64) * b.astype(np.float64)), axis=1)     # what any fp32 evaluation is measured against
    got = bx6_dot16(a, b).astype(np.float64)
    # the fp32-input MFMA / an fmaf chain: 16 roundings of the running sum
    ref = np.zeros(len(a), dtype=np.float32)
    for k in range(16):
        ref = (ref.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)
    err_bx, err_ref = np.abs(got - exact) / scale, np.abs(ref.astype(np.float64) - exact) / scale
    # dropped terms a1 b2 + a2 b1 + a2 b2 <= (2 * 2^-22 + 2^-30) |a b| per product, plus three roundings of the accumulator
    assert err_bx.max() <= 2.0 ** -21 + 3 * 2.0 ** -24
    # and in practice it is of the size of the chain of sixteen fp32 roundings it replaces: on inner products dominated by one
    # term (exponents spread over 2^+-30 here) the dropped cross terms show -- 1.5 x the chain's error at the 99th percentile,
    # 1.4 x on average --, on sums of comparable terms they do not
    assert np.percentile(err_bx, 99) <= 2.0 * np.percentile(err_ref, 99) and err_bx.mean() <= 1.5 * err_ref.mean()
    n = rng.standard_normal((50_000, 16)).astype(np.float32)
    m = rng.standard_normal((50_000, 16)).astype(np.float32)
    ex = np.sum(n.astype(np.float64) * m.astype(np.float64), axis=1)
    sc = np.sum(np.abs(n.astype(np.float64) * m.astype(np.float64)), axis=1)
    r = np.zeros(len(n), dtype=np.float32)
    for k in range(16):
        r = (r.astype(np.float64) + n[:, k].astype(np.float64) * m[:, k].astype(np.float64)).astype(np.float32)
    assert (np.abs(bx6_dot16(n, m).astype(np.float64) - ex) / sc).mean() <= 1.1 * (np.abs(r.astype(np.float64) - ex) / sc).mean()


def test_two_planes_would_not_do():
    """Why three planes: with x0 + x1 only (16 mantissa bits) the same inner product is 2^-16-accurate, two orders of magnitude
    short of the fp32 contract (1e-3 after the window's conditioning has amplified it)."""
    rng = np.random.default_rng(2)
    a = wide_floats(rng, 20_000 * 16).reshape(-1, 16)
    b = wide_floats(rng, 20_000 * 16).reshape(-1, 16)
    a0, a1 = split3(a)[:2]
    b0, b1 = split3(b)[:2]
    exact = np.sum(a.astype(np.float64) * b.astype(np.float64), axis=1)
    scale = np.sum(np.abs(a.astype(np.float64) * b.astype(np.float64)), axis=1)
    two = np.sum((a0.astype(np.float64) + a1) * (b0.astype(np.float64) + b1), axis=1)
    assert (np.abs(two - exact) / scale).max() > 2.0 ** -19
