"""The kernels' division-free index arithmetic, checked exhaustively over the ranges the kernels use it on (no GPU: float32 multiplication and
truncation are IEEE operations, numpy's are the device's).  A wrong quotient here would be a wrong address on the device, so these are the
claims the comments next to quot() / quot_f() (uplink.hip, frontend.hip), the pair table of k_pdsch_demod (chain.hip) and the 24-bit
multipliers make."""
import numpy as np


def recip_up(d):
    """the float next above or equal to 1 / d (uplink.hip / frontend.hip recip_up)"""
    rd = 1.0 / float(d)
    r = np.float32(rd)
    if float(r) < rd:
        r = np.nextafter(r, np.float32(2.0), dtype=np.float32)
    return r


def quot(n, r):
    return (n.astype(np.float32) * np.float32(r)).astype(np.uint32)


def test_truncated_float_product_is_the_quotient_for_every_pusch_size():
    """k_pusch_demod: butterflies o < 12 * M / R, outputs o < 12 * M, divided by M, M / R, the product of the earlier radices (PuschShape)"""
    n = np.arange(12 * 1320 + 64, dtype=np.uint32)
    divisors = set()
    for n_prb in range(1, 111):
        m, rem, ns = 12 * n_prb, 12 * n_prb, 1
        divisors.add(m)
        while rem > 1:
            for rad in (9, 3, 5, 8, 4, 2):
                if rem % rad == 0:
                    break
            else:
                rad = 7
                while rem % rad:
                    rad += 2
            divisors.update((m // rad, ns))
            ns *= rad
            rem //= rad
    for s_par in (12, 6, 3, 2, 1):
        divisors.add(s_par)
    assert len(divisors) > 150
    for d in sorted(divisors):
        assert (quot(n, recip_up(d)) == n // d).all(), d


def test_truncated_float_product_for_the_channel_estimators_splits():
    """k_dl_ce: t < 5 * 2 N_rb split by 2 N_rb, t < 5 * 3 N_rb split by 3 N_rb"""
    for n_rb in range(6, 111):
        for d, top in ((2 * n_rb, 5 * 2 * n_rb), (3 * n_rb, 5 * 3 * n_rb)):
            n = np.arange(top + 64, dtype=np.uint32)
            assert (quot(n, recip_up(d)) == n // d).all(), (n_rb, d)


def test_pair_table_quotient_survives_a_reciprocal_that_is_one_ulp_off():
    """k_pdsch_demod: q / N_prb for q < 14 N_prb with r = v_rcp_f32(N_prb) * (1 + 3 * 2^-23); the hardware reciprocal is within 1 ulp"""
    for d in range(1, 111):
        exact = np.float32(1.0) / np.float32(d)
        q = np.arange(14 * d + 14, dtype=np.uint32)
        for r0 in (exact, np.nextafter(exact, np.float32(0), dtype=np.float32), np.nextafter(exact, np.float32(2), dtype=np.float32)):
            r = np.float32(r0) * np.float32(1.0 + 1.5 * 2.0 ** -22)
            assert ((q.astype(np.float32) * r).astype(np.uint32) == q // d).all(), d


def test_24_bit_multipliers():
    """i / 12 as (i * 10923) >> 17 for i < 1536 (k_pusch_demod's sub-carrier -> resource block), o / 12 as (o * 43691) >> 19 for o < 2^15
    (its de-mapper, k_pdsch_demod's compact loop), x / 14 as (x * 4682) >> 16 for x < 5461 (k_dl_ce); every product stays below 2^32 and every
    factor below 2^24"""
    i = np.arange(1536, dtype=np.uint64)
    assert ((i * 10923) >> 17 == i // 12).all() and int(i[-1]) * 10923 < 2 ** 24
    o = np.arange(1 << 15, dtype=np.uint64)
    assert ((o * 43691) >> 19 == o // 12).all() and int(o[-1]) * 43691 < 2 ** 32
    x = np.arange(5461, dtype=np.uint64)
    assert ((x * 4682) >> 16 == x // 14).all()


def test_qpsk_soft_value_guard_is_wide_enough():
    """soft_decision_127 (phy_dev.hpp): two square roots that differ by up to 2 ulp move 127 (1 - dist) by less than 2^-15 -- a quarter of the
    2^-13 guard inside which the kernel takes the correctly rounded root.  Sampled: for 2 M squared distances the product formed from the
    correctly rounded root and from its neighbours two floats either side truncate alike whenever the former is outside the guard."""
    rng = np.random.default_rng(5)
    d2 = rng.uniform(0.0, 1.2, 2_000_000).astype(np.float32)
    cap = np.float32(1.0 - 1.0 / 120)
    s = np.sqrt(d2, dtype=np.float32)

    def v_of(root):
        return np.float32(127) * (np.float32(1) - np.minimum(root, cap))

    v = v_of(s)
    f = v - np.floor(v)
    outside = (f > 2.0 ** -13) & (f < 1 - 2.0 ** -13)
    for steps in (-2, -1, 1, 2):
        other = s.copy()
        for _ in range(abs(steps)):
            other = np.nextafter(other, np.float32(np.inf if steps > 0 else -np.inf), dtype=np.float32)
        vo = v_of(np.maximum(other, np.float32(0)))
        assert np.abs(vo - v).max() < 2.0 ** -15
        assert (vo[outside].astype(np.int32) == v[outside].astype(np.int32)).all()
