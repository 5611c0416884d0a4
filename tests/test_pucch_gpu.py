"""GPU parity: PUCCH formats 1 / 1a / 1b (HIP) vs the compiled reference's liblte_phy_pucch_format_1_1a_1b_channel_decode.

The reference has no PUCCH transmitter, so the test builds one from the sequences liblte_phy_ul_init left in its struct (36.211
5.4.1: d * s(n_s) * w(i) * r_u,v^alpha(n) on the four data symbols of each slot, the reference symbols in between), sends it through a
per-slot complex gain and noise, and compares bits, bit count and return code -- equal also where the soft decision fails."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D_OF = {(0, 0): 1, (1, 0): 1j, (0, 1): -1j, (1, 1): -1}  # format 1b decision regions of the reference's decoder (:3120-3137)


def build(ref, phy, n_rb_ul, sf, n1, fmt, bits, snr_db, rng):
    t = np.zeros(352, np.float32)
    ref.ref_get_pucch_tables(phy, sf, n1, t)
    dm = [t[0:36] + 1j * t[36:72], t[72:108] + 1j * t[108:144]]
    ruv = (t[144:240] + 1j * t[240:336]).reshape(2, 4, 12)
    sw = (t[336:344] + 1j * t[344:352]).reshape(2, 4)
    d = 1 if fmt == 0 else (1 - 2 * bits[0]) if fmt == 1 else D_OF[tuple(bits)]
    grid = np.zeros((14, 1200), np.complex64)
    prb = [n1, n_rb_ul - n1 - 1]
    sig = 10 ** (-snr_db / 20) / np.sqrt(2)
    for m in range(2):
        h = rng.uniform(0.5, 1.5) * np.exp(1j * rng.uniform(-np.pi, np.pi))
        k = slice(prb[m] * 12, prb[m] * 12 + 12)
        for i, L in enumerate([0, 1, 5, 6]):
            grid[7 * m + L, k] = h * d * sw[m, i] * ruv[m, i]
        for i, L in enumerate([2, 3, 4]):
            grid[7 * m + L, k] = h * dm[m][12 * i:12 * i + 12]
    grid += (sig * (rng.standard_normal(grid.shape) + 1j * rng.standard_normal(grid.shape))).astype(np.complex64)
    return grid, t


@pytest.mark.parametrize("fft,n_rb,cell,hop,n_cs_an,shift", [(2048, 100, 17, 0, 0, 1), (512, 25, 301, 1, 0, 2), (128, 6, 44, 0, 2, 3)])
def test_pucch_matches_reference(ctx, ref, fft, n_rb, cell, hop, n_cs_an, shift):
    import openlte_amd as m
    from oracle import pyoracle as po
    rng = np.random.default_rng(fft + cell)
    phy = ref.ref_phy_new(po.FS_ENUM[fft], cell, 1, n_rb)
    assert ref.ref_ul_init_pucch(phy, cell, 3, hop, n_cs_an, shift) == 0
    n_rb_ul = ref.ref_get_n_rb_ul(phy)
    sfp = ref.ref_subframe_new()
    cases = []
    for sf in (0, 3, 9):
        for n1 in (0, 1, min(2, n_rb_ul // 2 - 1)):
            for fmt, bits in ((0, (0, 0)), (1, (0, 0)), (1, (1, 0)), (2, (0, 0)), (2, (1, 0)), (2, (0, 1)), (2, (1, 1))):
                for snr in (20.0, -3.0):
                    cases.append((sf, n1, fmt, bits, snr))
    grids, tabs, want = [], [], []
    for sf, n1, fmt, bits, snr in cases:
        g, t = build(ref, phy, n_rb_ul, sf, n1, fmt, bits, snr, rng)
        ref.ref_subframe_set_num(sfp, sf)
        po.ref_subframe_view(ref, sfp, 0)[:14] = g.real
        po.ref_subframe_view(ref, sfp, 1)[:14] = g.imag
        out, n = np.zeros(4, np.uint8), C.c_uint32()
        rc = ref.ref_pucch_decode(phy, sfp, fmt, cell, 1, n1, out, C.byref(n))
        want.append((rc, n.value, int(out[0]), int(out[1]) if n.value == 2 else 0))
        sub = np.zeros((2, 16, 1200), np.float32)
        sub[0, :14], sub[1, :14] = g.real, g.imag
        grids.append(sub)
        tabs.append(t)
    d_sub = ctx.to_device(np.ascontiguousarray(np.stack(grids)))
    bits, nb, rc = ctx.pucch_decode_dev(n_rb_ul, d_sub, [(u, c[2], c[1]) for u, c in enumerate(cases)], np.stack(tabs))
    d_sub.free()
    got = [(int(rc[u]), int(nb[u]), int(bits[u, 0]), int(bits[u, 1]) if nb[u] == 2 else 0) for u in range(len(cases))]
    assert got == want
    # at 20 dB what was sent comes back, with a confident soft decision
    for (sf, n1, fmt, b, snr), (r, n, b0, b1) in zip(cases, got):
        if snr >= 20:
            assert r == 0 and n == (2 if fmt == 2 else 1) and (fmt == 0 or (b0, b1) == (b if fmt == 2 else (b[0], 0)))
    assert any(r != 0 for r, *_ in got)  # and the noisy ones include rejected decisions
    ref.ref_subframe_free(sfp)
    ref.ref_phy_free(phy)
