"""The library's own transmit side (host code, tx*.cc -- the reference's liblte_phy_*_channel_encode / map_* / create_dl_subframe restated,
pinned to the compiled reference on the CPU by tests/test_cabi.py) looped back into its receive chain on the GPU: what is sent must be what
is decoded.  The two sides share nothing but the C-ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fft,nrb,cell", [(2048, 100, 17), (1024, 50, 301), (512, 25, 44), (128, 6, 503)])
def test_own_transmitter_into_own_receive_chain(ctx, fft, nrb, cell):
    """CRS + PSS / SSS + three PDSCH allocations (QPSK / 16QAM / 64QAM) per subframe through mi_lte_create_dl_subframe, the float samples
    into the downlink front end and a PDSCH plan: every transport block comes back bit for bit (a clean channel: status 0)."""
    import openlte_amd as m
    rng = np.random.default_rng(fft + cell)
    t = m.Transmitter(fft, nrb, cell)
    per = 14 * fft + (fft // 128) * (2 * 10 + 12 * 9)
    # (at 1.4 MHz every PRB lies under the PBCH / PSS / SSS window, where the reference's transmitter PRICES an allocation by PRB and its receiver
    # COUNTS the elements it finds, liblte_phy.cc:13936-14086 against :3722-3789: the two disagree there, in the reference as here -- other subframes)
    sfs, units_i, units_q, allocs, sent = ([1, 3, 6, 8] if nrb == 6 else [0, 3, 5, 8]), [], [], [], []
    cfi = 3 if nrb == 6 else 2
    for u, sf in enumerate(sfs):
        t.clear()
        t.signals(sf)
        mine = []
        first = 0
        for mod, n_prb in ((1, 2), (2, 1), (3, 2)):
            if first + n_prb > nrb:
                break
            # the largest transport block the reference's decoder takes from this many soft bits: it reads punctured parity as hard zeros, so
            # the code word must arrive whole -- E >= 3 (K + 4) (W4's 3240 bits on 12 PRB of 64QAM is such a pair)
            E = n_prb * (12 * (14 - cfi) - 6) * 2 * mod
            tbs = max(v for v in (m.load_library().mi_lte_tbs(i, n_prb) for i in range(27)) if 3 * (v + 28) <= E)
            a = m.make_alloc(u, mod, tbs, list(range(first, first + n_prb)), 0x100 + 7 * u + mod, rv_idx=int(rng.integers(0, 4)) if mod == 1 else 0)
            bits = rng.integers(0, 2, tbs).astype(np.uint8)
            mine.append((a, bits))
            first += n_prb
        t.pdsch(sf, cfi, mine)
        i, q = t.samples()
        assert len(i) == per
        units_i.append(i), units_q.append(q)
        # the estimator of symbols 12 and 13 looks at the NEXT subframe's first reference symbol (liblte_phy.cc:6119-6190): send it
        t.clear()
        t.signals((sf + 1) % 10)
        i, q = t.samples()
        units_i.append(i), units_q.append(q)
        allocs += [a for a, _ in mine]
        sent += [b for _, b in mine]
    t.close()
    i_all = np.concatenate(units_i + [np.zeros(per, np.float32)]) * np.float32(0.02)
    q_all = np.concatenate(units_q + [np.zeros(per, np.float32)]) * np.float32(0.02)
    cfg = m.DlCfg(fft, nrb, 1, 0)
    sub = ctx.dl_frontend(cfg, (i_all, q_all), [2 * u * per for u in range(len(sfs))], sfs, [cell] * len(sfs))
    d_sub = ctx.to_device(sub.astype(np.float32))
    plan = ctx.pdsch_plan(cfg, cfi, allocs)
    st, bits = plan.run(d_sub, sfs, [cell] * len(sfs))
    plan.close()
    d_sub.free()
    assert list(st) == [0] * len(allocs)
    for k in range(len(allocs)):
        assert np.array_equal(bits[k][:allocs[k].tbs], sent[k]), "allocation %d" % k
