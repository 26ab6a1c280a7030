"""The arithmetic of the exact scatter-add (csrc/index.hip: scatter_fixed / scatter_add128 / scatter_exact_finish_kernel)
restated with Python integers — the device code is held against math.fsum on the GPU
(tests/test_gpu_misc.py::test_scatter_add_many_bins_is_exact_and_reproducible); this pins the scheme itself on the CPU:
addends as 128-bit two's-complement integers in units of 2^(Ef - 1075 - 43), sums mod 2^128 in ANY order, one
round-to-nearest-even at the end."""
import math
import struct

import numpy as np
import pytest

M128 = (1 << 128) - 1


def _bits(v: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", v))[0]


def fixed(v: float, Ef: int):
    """scatter_fixed: (q as an unsigned 128-bit two's-complement integer or None, special flag)"""
    b = _bits(v)
    ef, m, neg = (b >> 52) & 0x7FF, b & ((1 << 52) - 1), b >> 63
    if ef == 0x7FF:
        return None, (1 if m else (4 if neg else 2))
    if ef:
        m |= 1 << 52
    else:
        ef = 1
    if m == 0:
        return None, 0
    sh = ef - Ef + 43
    if sh >= 0:
        q = m << sh
    elif sh > -53:
        q = m >> (-sh)
    else:
        return None, 0
    if q == 0:
        return None, 0
    return ((-q) & M128 if neg else q), 0


def finish(total: int, Ef: int) -> float:
    """scatter_exact_finish_kernel: the 128-bit sum -> double, rounded once (nearest, ties to even)"""
    if total == 0:
        return 0.0
    neg = total >> 127
    mag = ((-total) & M128) if neg else total
    p = mag.bit_length() - 1
    e2 = 0
    if p <= 52:
        mant = mag
    else:
        s = p - 52
        kept, half, rest = mag >> s, (mag >> (s - 1)) & 1, mag & ((1 << (s - 1)) - 1)
        mant = kept + (1 if half and (rest or (kept & 1)) else 0)
        e2 = s
    r = math.ldexp(float(mant), Ef - 1075 - 43 + e2)
    return -r if neg else r


def exact_sum(vals, order=None):
    fin = [v for v in vals if math.isfinite(v)]
    mx = max((abs(v) for v in fin), default=0.0)
    Ef = (_bits(mx) >> 52) & 0x7FF or 1
    total = 0
    for k in (order if order is not None else range(len(vals))):
        q, sp = fixed(vals[k], Ef)
        assert sp == 0
        if q is not None:
            # two 64-bit atomics: low word, then high word + the carry the low add produced
            lo, hi = total & ((1 << 64) - 1), total >> 64
            qlo, qhi = q & ((1 << 64) - 1), q >> 64
            nlo = (lo + qlo) & ((1 << 64) - 1)
            carry = 1 if nlo < lo else 0
            total = (((hi + qhi + carry) & ((1 << 64) - 1)) << 64) | nlo
    return finish(total, Ef)


@pytest.mark.parametrize("seed", range(8))
def test_exact_sum_equals_fsum_in_any_order(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 400))
    vals = (rng.normal(size=n) * np.exp2(rng.integers(-20, 21, size=n))).tolist()  # within the 43-bit window: exact
    want = math.fsum(vals)
    assert exact_sum(vals) == want
    for _ in range(3):
        assert exact_sum(vals, rng.permutation(n)) == want


def test_ties_round_to_even_and_cancellation_is_exact():
    one = 1.0
    ulp = 2.0**-52
    assert exact_sum([one, ulp / 2]) == 1.0  # tie -> even (1.0)
    assert exact_sum([one + ulp, ulp / 2]) == one + 2 * ulp  # tie -> even (upwards)
    assert exact_sum([one, ulp / 2, 2.0**-80]) == one + ulp  # above the tie
    assert exact_sum([1e10, 3.25, -1e10]) == 3.25  # what a sequential double sum loses
    assert exact_sum([2.0**40, 1.0, -(2.0**40), -1.0]) == 0.0
    assert exact_sum([-0.0, 0.0]) == 0.0


def test_addends_below_the_window_lose_at_most_a_unit_each():
    big = 2.0**60
    tiny = 2.0**-40  # 100 binades below: dropped (the device documents 2^-95 max|y| absolute error per addend)
    assert exact_sum([big, tiny]) == big
    small = 2.0**10 + 2.0**-34  # its last bit is 1 unit below the window's unit 2^(60-43-52) ... truncated towards zero
    got = exact_sum([big, small])
    assert abs(got - (big + small)) <= 2.0 ** (60 - 95 + 1) + math.ulp(big)


def test_subnormal_maximum_uses_the_subnormal_exponent():
    vals = [5e-324 * 3, 5e-324 * 7, -5e-324 * 2]
    assert exact_sum(vals) == 5e-324 * 8


def test_special_values_are_flags_not_integers():
    assert fixed(float("nan"), 1000) == (None, 1)
    assert fixed(float("inf"), 1000) == (None, 2)
    assert fixed(float("-inf"), 1000) == (None, 4)
