"""Arithmetic identities the HIP path relies on where it does NOT run the reference's instruction sequence, checked on the
CPU against numpy's correctly rounded binary32 operations (no GPU, no oracle): the proofs by exhaustion that
`csrc/pt_device.hpp` cites.  The device repeats them through `pt_probe_sqrt` (tests/test_gpu_pins.py)."""
import numpy as np

F = np.float32
ONE_BITS = int(F(1.0).view(np.uint32))


def chain(x):
    """glm::normalize's factor as the reference computes it: fl(1 / fl(sqrt x))"""
    return (F(1.0) / np.sqrt(x)).astype(np.float32)


def rsqrt_near_one(x):
    """ptd::rsqrt_near_one, operation for operation (the fma's product e * -0.5 is exact, so multiply-then-add rounds once too)"""
    e = (x - F(1.0)).astype(np.float32)
    w = (F(-0.5) * e + F(3 * 2.0 ** -26)).astype(np.float32)
    t = (w + (F(1.0) + F(2.0 ** -10))).astype(np.float32)
    return (t - F(2.0 ** -10)).astype(np.float32)


def test_near_one_reciprocal_root_every_argument():
    """Every binary32 from 1 - 8190 * 2^-24 to 1 + 2897 * 2^-23: four additions reproduce the root-then-reciprocal chain bit for
    bit; one float further on either side they do not (the first second-order deviations); the gate the device uses,
    [1 - 2^-12, 1 + 2^-12], lies inside with a margin of a factor two / 1.4."""
    xb = np.arange(ONE_BITS - 8190, ONE_BITS + 2897 + 1, dtype=np.int64).astype(np.uint32)
    x = xb.view(np.float32)
    assert np.array_equal(rsqrt_near_one(x).view(np.uint32), chain(x).view(np.uint32))
    # the intermediate values are exact (no hidden rounding before the one that matters)
    e = x.astype(np.float64) - 1.0
    assert np.array_equal((x - F(1.0)).astype(np.float64), e)
    assert np.array_equal((F(-0.5) * (x - F(1.0)) + F(3 * 2.0 ** -26)).astype(np.float64), -0.5 * e + 3 * 2.0 ** -26)
    # the closed form the comment states: 1 - e/2 rounded up to a multiple of 2^-23
    k = xb.astype(np.int64) - ONE_BITS
    want = np.where(k >= 0, ONE_BITS - (k & ~1), ONE_BITS + ((3 - k) >> 2))
    assert np.array_equal(chain(x).view(np.uint32).astype(np.int64), want)
    # just outside the range the identity fails: the range is tight, not a guess
    for bits in (ONE_BITS - 8191, ONE_BITS + 2898):
        v = np.array([bits], dtype=np.uint32).view(np.float32)
        assert rsqrt_near_one(v).view(np.uint32)[0] != chain(v).view(np.uint32)[0]
    lo, hi = F(1 - 2.0 ** -12), F(1 + 2.0 ** -12)
    assert ONE_BITS - int(lo.view(np.uint32)) == 4096 and int(hi.view(np.uint32)) - ONE_BITS == 2048
    assert float(lo) == float.fromhex("0x1.ffep-1") and float(hi) == float.fromhex("0x1.001p+0")      # the constants in pt_device.hpp


def test_zero_product_folds_into_the_addition():
    """mv_dir keeps the reference's m3 * 0.0f term as the multiplier pair of an fma whose addend is the rounded m2 * v.z:
    because m3 * 0 is exact (+-0, or NaN for a non-finite m3) the sum has ONE rounding either way -- compared here in
    binary64, where the exact product plus the float addend is representable, for finite, zero, infinite and NaN entries."""
    rng = np.random.default_rng(5)
    m3 = np.concatenate([rng.standard_normal(4000).astype(np.float32) * F(1e3), np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45], dtype=np.float32)])
    p = np.concatenate([rng.standard_normal(4000).astype(np.float32), np.array([0.0, -0.0, 0.0, -0.0, 1.0, -0.0, 0.0], dtype=np.float32)])
    for a in (m3, -m3):
        for b in (p, -p, np.zeros_like(p), -np.zeros_like(p)):
            with np.errstate(invalid="ignore"):
                ref = (b + (a * F(0.0)).astype(np.float32)).astype(np.float32)                # multiply, then add
                fused = (a.astype(np.float64) * 0.0 + b.astype(np.float64)).astype(np.float32)   # exact product, one rounding
            same = np.where(np.isnan(ref), np.isnan(fused), ref.view(np.uint32) == fused.view(np.uint32))
            assert same.all()


def test_exact_products_fold_into_one_fma():
    """reflect's doubling and the sampler's cross product with an axis: a product that is EXACT (by 2, by 0 or by 1) may be the
    multiplier pair of an fma without changing the value -- the reference's multiply-then-subtract rounds once as well.
    Modelled in binary64 (exact products and differences of binary32 values are representable), including subnormal and
    zero operands of either sign."""
    rng = np.random.default_rng(11)
    edge = np.array([0.0, -0.0, 1e-45, -1e-45, 1.1754944e-38, -1.1754942e-38, 3e-39, 1.0, -1.0], dtype=np.float32)
    p = np.concatenate([rng.standard_normal(5000).astype(np.float32), (rng.standard_normal(2000) * 1e-38).astype(np.float32), edge])
    i = np.concatenate([rng.standard_normal(5000).astype(np.float32), (rng.standard_normal(2000) * 1e-38).astype(np.float32), edge[::-1]])
    ref = (i - (p * F(2.0)).astype(np.float32)).astype(np.float32)                      # I - fl(2 p)
    fused = (i.astype(np.float64) - 2.0 * p.astype(np.float64)).astype(np.float32)      # fma(-2, p, I)
    assert np.array_equal(ref.view(np.uint32), fused.view(np.uint32))
    a, c = p, i
    for b in (F(0.0), F(1.0)):
        for d in (F(0.0), F(1.0)):
            ref = ((a * b).astype(np.float32) - (d * c).astype(np.float32)).astype(np.float32)          # x.y * y.z - y.y * x.z
            fused = (a.astype(np.float64) * float(b) + -((d * c).astype(np.float32)).astype(np.float64)).astype(np.float32)
            assert np.array_equal(ref.view(np.uint32), fused.view(np.uint32)), (b, d)
