"""GPU: MI_LTE_TURBO_BCJR (fixed-point max-log-MAP, the decoder the north star sketches) against its specification,
oracle/lte_oracle.c lo_turbo_decode_bcjr.  The reference has no such decoder (parity for the reference's
own decoder is the REF mode, tests/test_turbo_gpu.py), so the bar here is: every decoded bit equals the oracle's
(integer arithmetic, identical operation order), and the decoder actually decodes -- error-free well below the SNR
where the reference-faithful REF mode gives up."""
import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu


def llr_blocks(port, K, n, sigma, seed, spec_interleaver=False):
    """BPSK over AWGN -> int8 LLRs in the reference's interleaved layout.  The encoder is the oracle's (wrapped
    interleaver, identical to the 3GPP one except for the 20 uint32-overflow sizes)."""
    rng = np.random.default_rng(seed)
    tx = rng.integers(0, 2, (n, K)).astype(np.uint8)
    out = np.zeros((n, 3 * (K + 4)), np.int8)
    for b in range(n):
        d = np.zeros(3 * (K + 4), np.uint8)
        port.lo_turbo_encode(np.ascontiguousarray(tx[b]), K, d)
        x = 1.0 - 2.0 * d.reshape(3, K + 4)
        y = x + sigma * rng.standard_normal(x.shape)
        llr = np.clip(np.round(y * (8.0 / max(sigma, 0.5) ** 2)), -127, 127)
        out[b] = np.ascontiguousarray(llr.T).reshape(-1).astype(np.int8)
    return tx, out


def oracle_bcjr(port, soft, K, n_iter, spec):
    out = np.zeros((soft.shape[0], K), np.uint8)
    for b in range(soft.shape[0]):
        port.lo_turbo_decode_bcjr(np.ascontiguousarray(soft[b].astype(np.int16)), K, n_iter, 1 if spec else 0, out[b])
    return out


@pytest.mark.parametrize("K", [40, 104, 512, 1088, 3264, 6016, 6144])
@pytest.mark.parametrize("sigma", [0.0, 0.9, 1.3])
def test_bcjr_bit_exact_vs_oracle(ctx, port, K, sigma):
    import openlte_amd as m
    n = 70 if K <= 1088 else 66  # more than one tile, last tile ragged
    tx, soft = llr_blocks(port, K, n, sigma, seed=K + int(10 * sigma))
    for n_iter, spec in ((8, False), (3, True)):
        want = oracle_bcjr(port, soft[:12 if K > 3000 else n], K, n_iter, spec)
        got = ctx.turbo_decode(soft, K, mode=m.TURBO_BCJR, n_iter=n_iter, qpp_spec=spec)
        assert (got[:want.shape[0]] == want).all(), (K, sigma, n_iter, spec)
        if sigma <= 0.9 and not spec and K not in (6144,):
            assert (got == tx).all()  # decodes (the wrapped interleaver of K = 6144 is not a permutation: excluded)


def test_bcjr_outperforms_ref_mode(ctx, port):
    """At Eb/N0 ~ 1.8 dB (sigma = 1.0, rate 1/3) max-log-MAP with 8 iterations is error free while the reference's
    hard-metric decoder is not."""
    import openlte_amd as m
    K, n = 2048, 64
    tx, soft = llr_blocks(port, K, n, 1.0, seed=3)
    bcjr = ctx.turbo_decode(soft, K, mode=m.TURBO_BCJR, n_iter=8)
    ref = ctx.turbo_decode(soft, K, mode=m.TURBO_REF)
    assert (bcjr != tx).sum() == 0
    assert (ref != tx).any(axis=1).mean() > 0.5


@pytest.mark.parametrize("K", [3264, 6016])
def test_bcjr_agrees_with_the_reference_decoder_wherever_that_one_decodes(ctx, port, K):
    """SURVEY 8d W3's gate for the mode the reference does not have: on every code block where the reference's own decoder recovers
    the transmitted bits (REF mode here is that decoder bit for bit, tests/test_turbo_gpu.py), the max-log-MAP decoder's output
    equals it -- over a noise range that runs from where the reference decodes everything to where it decodes almost nothing."""
    import openlte_amd as m
    n_gate = 0
    for sigma in (0.0, 0.5, 0.7, 0.8, 0.9):
        tx, soft = llr_blocks(port, K, 128, sigma, seed=K + int(100 * sigma))
        ref = ctx.turbo_decode(soft, K, mode=m.TURBO_REF)
        bcjr = ctx.turbo_decode(soft, K, mode=m.TURBO_BCJR, n_iter=8)
        ok_ref = (ref == tx).all(axis=1)
        n_gate += int(ok_ref.sum())
        assert (bcjr[ok_ref] == ref[ok_ref]).all(), (K, sigma, int((bcjr[ok_ref] != ref[ok_ref]).any(axis=1).sum()))
        assert (bcjr == tx).all(), (K, sigma)  # and it decodes the blocks the reference's decoder loses as well
    assert n_gate >= 128  # the gate was exercised


def test_bcjr_full_batch_property(ctx, port):
    """BASELINE config 3 shape (K = 6144, 65536 blocks, 8 iterations): oracle-check 8 unique blocks and require every
    replica, wherever it sits in a tile, to decode to the same bits."""
    import openlte_amd as m
    K, uniq, n_cb = 6144, 8, 65536
    tx, soft = llr_blocks(port, K, uniq, 0.9, seed=11)
    want = oracle_bcjr(port, soft, K, 8, False)
    idx = (np.arange(n_cb) * 5 + np.arange(n_cb) // 64) % uniq
    d_in = ctx.to_device(soft[idx])
    d_out = ctx.alloc(n_cb * K)
    ctx.turbo_decode_dev(d_in, m.SOFT_I8, K, n_cb, d_out, mode=m.TURBO_BCJR, n_iter=8)
    got = d_out.download(np.uint8).reshape(n_cb, K)
    d_in.free(); d_out.free()
    assert (got == want[idx]).all()


def test_bcjr_rejects_float_input(ctx):
    import openlte_amd as m
    with pytest.raises(m.MiLteError):
        ctx.turbo_decode(np.zeros((1, 3 * 44), np.float32), 40, mode=m.TURBO_BCJR)


@pytest.mark.parametrize("block_mode", [False, True])
@pytest.mark.parametrize("snr", [30.0, 9.0])
def test_pdsch_chain_in_bcjr_mode(ctx, port, snr, block_mode):
    """mi_lte_pdsch_plan_set_decoder(BCJR / BCJR_BLOCK): the full chain with the max-log-MAP decoder (batch kernels, or one code block per wavefront).  Checker = the pieces composed on the CPU:
    the allocation's soft bits (already pinned to the oracle by the REF-mode tests) -> the restated rate un-matching -> saturation
    to int8, NULL -> 0 -> the plain-C model of the decoder -> filler removal + CRC24A.  Bits and verdicts must be equal."""
    import ctypes as C
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(2048, 100, 1, 0)
    sfs, cells = [1, 6], [17, 404]
    allocs = []
    for u in range(2):
        allocs += td.w4_allocs(u)
    iq, tx = synth.dl_units(cfg, sfs, cells, allocs, 9, snr_db=snr, max_delay=4, seed=321)
    got = ctx.dl_frontend(cfg, iq.reshape(-1, 2), np.arange(2) * iq.shape[1], sfs, cells)
    d_sub = ctx.to_device(np.ascontiguousarray(got, np.float32))
    plan = ctx.pdsch_plan(cfg, 2, allocs)
    plan.set_decoder(m.TURBO_BCJR_BLOCK if block_mode else m.TURBO_BCJR, 6, 0)
    model = port.lo_turbo_decode_bcjr_block if block_mode else port.lo_turbo_decode_bcjr
    st, bits = plan.run(d_sub, sfs, cells)
    n_ok = 0
    for a, al in enumerate(allocs):
        e = plan.soft_bits(a).astype(np.float32)
        K = al.tbs + 24
        d = np.zeros(3 * (K + 4), np.float32)
        n = port.lo_rate_unmatch_turbo(np.ascontiguousarray(e), len(e), K + 4, 1, al.tx_mode, 250368, 8, 0, al.rv_idx, d)
        assert n == 3 * (K + 4)
        soft = np.where(d == 10000.0, 0.0, np.clip(d, -127, 127)).astype(np.int16)
        c = np.zeros(K, np.uint8)
        model(soft, K, 6, 0, c)
        p = np.zeros(24, np.uint8)
        port.lo_crc24a(np.ascontiguousarray(c[:al.tbs]), al.tbs, p)
        ok = bool((p == c[al.tbs:]).all())
        assert (st[a] == 0) == ok, a
        assert (bits[a] == c[:al.tbs]).all(), a
        if ok:
            n_ok += 1
            assert (bits[a] == tx[al.unit, a % 9, :al.tbs]).all()
    assert n_ok == len(allocs) if snr >= 30 else n_ok > 0
    plan.close()
    d_sub.free()


@pytest.mark.parametrize("K", [40, 104, 512, 1088, 2048, 3264, 6016, 6080, 6144])
@pytest.mark.parametrize("sigma", [0.0, 0.9, 1.3])
def test_bcjr_block_mode_bit_exact_vs_its_model(ctx, port, K, sigma):
    """MI_LTE_TURBO_BCJR_BLOCK (k_bcjr_block: one code block per wavefront, 64 lanes = 64 segments, everything in LDS, one launch) against
    ITS model, lo_turbo_decode_bcjr_block -- the batch model with the alpha recursion restarting every lo_bcjr_block_seg_len(K) steps.
    Every block size class: one segment (K = 40), ragged last segments, 32 / 64 / 96 steps per lane, the wrapped interleaver (6144)."""
    import openlte_amd as m
    n = 5
    tx, soft = llr_blocks(port, K, n, sigma, seed=2 * K + int(10 * sigma))
    for n_iter, spec in ((8, False), (2, True)):
        want = np.zeros((n, K), np.uint8)
        for b in range(n):
            port.lo_turbo_decode_bcjr_block(np.ascontiguousarray(soft[b].astype(np.int16)), K, n_iter, 1 if spec else 0, want[b])
        got = ctx.turbo_decode(soft, K, mode=m.TURBO_BCJR_BLOCK, n_iter=n_iter, qpp_spec=spec)
        assert (got == want).all(), (K, sigma, n_iter, spec)
        if sigma <= 0.9 and not spec and K not in td.OVERFLOW_K:  # (the wrapped interleaver of those sizes is not a permutation)
            assert (got == tx).all()


def test_bcjr_block_mode_latency_of_one_block(ctx, port):
    """What the mode is for: one K = 6144 block decoded alone (8 iterations).  Prints both modes' wall time per decode."""
    import time
    import openlte_amd as m
    K = 6144
    tx, soft = llr_blocks(port, K, 1, 0.9, seed=5)
    res = {}
    for name, mode in (("batch kernels", m.TURBO_BCJR), ("one block per wavefront", m.TURBO_BCJR_BLOCK)):
        ctx.turbo_decode(soft, K, mode=mode, n_iter=8)
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.turbo_decode(soft, K, mode=mode, n_iter=8)
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
    print("BCJR, one K = 6144 block, 8 iterations, host round trip included:", {k: "%.2f ms" % v for k, v in res.items()})
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bcjr_block_latency.txt", "w") as f:
        f.write(repr(res) + "\n")
    assert res["one block per wavefront"] < res["batch kernels"]


def test_bcjr_block_mode_more_blocks_than_compute_units(ctx, port):
    """More workgroups than the chip runs at once (one wavefront per code block, ~100 KB of LDS each at K = 6144): every replica, wherever and
    whenever its workgroup runs, decodes to its model's bits."""
    import openlte_amd as m
    for K, n_cb in ((6144, 700), (1088, 3000)):
        tx, soft = llr_blocks(port, K, 4, 0.9, seed=K)
        want = np.zeros((4, K), np.uint8)
        for b in range(4):
            port.lo_turbo_decode_bcjr_block(np.ascontiguousarray(soft[b].astype(np.int16)), K, 8, 0, want[b])
        idx = (np.arange(n_cb) * 3 + np.arange(n_cb) // 7) % 4
        d_in, d_out = ctx.to_device(soft[idx]), ctx.alloc(n_cb * K)
        ctx.turbo_decode_dev(d_in, m.SOFT_I8, K, n_cb, d_out, mode=m.TURBO_BCJR_BLOCK, n_iter=8)
        got = d_out.download(np.uint8).reshape(n_cb, K)
        d_in.free(); d_out.free()
        assert (got == want[idx]).all(), K


@pytest.mark.parametrize("K,spec", [(1088, False), (512, True), (3264, False)])
def test_bcjr_early_termination_is_the_model_at_the_iterations_each_pair_ran(ctx, port, K, spec):
    """MI_LTE_TURBO_BCJR_EARLY: a tile pair (128 code blocks) stops once an iteration (from the second on) changes none of its hard
    decisions.  Specification = the plain-C model run for exactly the iterations the pair ran; the stopping rule itself is checked on the
    model too: the pair's decisions after its last iteration equal those after the one before, and no earlier iteration had that property.
    Pairs of different quality: clean (stop after 2), moderate noise, one pair with a block below threshold (runs all 8)."""
    import openlte_amd as m
    n_pairs, n = 4, 4 * 128 - 37  # the last pair is ragged
    sig = np.concatenate([np.full(128, 0.3), np.full(128, 0.85), np.full(128, 1.05), np.full(n - 384, 0.6)])
    sig[2 * 128 + 5] = 2.5  # one hopeless block keeps its pair iterating
    tx = np.zeros((n, K), np.uint8)
    soft = np.zeros((n, 3 * (K + 4)), np.int8)
    for b in range(n):
        t, s = llr_blocks(port, K, 1, float(sig[b]), seed=1000 * K + b)
        tx[b], soft[b] = t[0], s[0]
    got = ctx.turbo_decode(soft, K, mode=m.TURBO_BCJR_EARLY, n_iter=8, qpp_spec=spec)
    iters = ctx.turbo_early_exit_iterations()
    assert len(iters) == n_pairs and iters.min() >= 2 and iters.max() <= 8
    assert iters[0] == 2 and iters[2] == 8 and iters[1] <= 6, iters
    full = ctx.turbo_decode(soft, K, mode=m.TURBO_BCJR, n_iter=8, qpp_spec=spec)
    for p in range(n_pairs):
        blk = np.arange(128 * p, min(n, 128 * p + 128))
        sample = blk if K <= 1088 else blk[::8]
        dec = {k: oracle_bcjr(port, soft[sample], K, k, spec) for k in range(max(1, int(iters[p]) - 2), int(iters[p]) + 1)}
        assert (got[sample] == dec[int(iters[p])]).all(), (p, iters[p])
        if K <= 1088:  # the rule on the model, every block of the pair
            if iters[p] < 8:
                assert (dec[int(iters[p])] == dec[int(iters[p]) - 1]).all(), p
            if iters[p] > 2:  # the iteration before did change something (else the pair would have stopped there)
                assert (dec[int(iters[p]) - 1] != dec[int(iters[p]) - 2]).any(), p
        ok = sig[blk] < 2
        assert (got[blk][ok] == tx[blk][ok]).all() and (got[blk][ok] == full[blk][ok]).all()  # stopping early costs nothing on blocks that decode
