"""GPU parity: REF-mode turbo decode (HIP) vs the oracle, bit-exact (integer path)."""
import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu

# small / not multiple of 64 / power of two / W4 sizes / largest overflow-free / uint32-overflow sizes
KS = [40, 104, 512, 1088, 3264, 3584, 6016, 6144]


@pytest.fixture(params=["lockstep", "state-parallel"])
def siso(ctx, request):
    """The REF decoder's two trellis kernels (include/mi_lte.h, mi_lte_set_turbo_small_batch): code blocks on the lanes for every batch
    size, or states on the lanes for the batch sizes of these tests.  Identical results are the requirement."""
    ctx.set_turbo_small_batch(0 if request.param == "lockstep" else 4096)
    yield request.param
    ctx.set_turbo_small_batch(4096)


@pytest.mark.parametrize("K", KS)
@pytest.mark.parametrize("kind", ["clean", "awgn0.5", "awgn0.8", "hard127", "int", "i16"])
def test_turbo_ref_bit_exact(ctx, port, siso, K, kind):
    n = 70 if K <= 1088 else 66  # more than one tile, last tile ragged
    tx, soft = td.turbo_blocks(port, K, n, kind, seed=1000 + K)
    want = td.oracle_turbo_ref(port, soft, K)
    got = ctx.turbo_decode(soft, K)
    assert got.shape == want.shape
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "blocks differing from the oracle: %s" % bad[:10]
    if kind == "clean" and K not in (3584, 6144):
        assert (got == tx).all()  # noise-free loopback decodes (overflow-free K)


@pytest.mark.parametrize("K", [40, 3264, 6144])
@pytest.mark.parametrize("kind", ["rand127", "noise"])
def test_turbo_ref_bit_exact_without_a_code_word(ctx, port, siso, K, kind):
    """The SISO kernel keeps its path metrics modulo 2^16 (two trellises per lane); that is exact as long as the spread of the eight
    metrics stays below 2^15.  Inputs with no code word underneath drive the spread as far as it goes."""
    tx, soft = td.turbo_blocks(port, K, 130, kind, seed=77 + K)
    want = td.oracle_turbo_ref(port, soft, K)
    got = ctx.turbo_decode(soft, K)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "blocks differing from the oracle: %s" % bad[:10]


def test_turbo_ref_all_188_block_sizes(ctx, port, siso):
    """every LTE turbo block size through the HIP decoder against the oracle -- 66 blocks each (one full tile + a ragged
    one) x {noise-free, AWGN sigma 0.8, +-127 with 2 % flips}.  The kernels have K-dependent paths: 16 / 8 / 0 valid steps in a
    block's last 16-step unit (K % 16 == 8 for 29 sizes), 1 .. 96 sixty-four-step lines, odd and even tile-line counts, and the
    20 sizes whose interleaver index overflows uint32 in the reference (holes read as zero, the oracle's scratch being fresh)."""
    assert len(td.ALL_K) == 188 and set(td.OVERFLOW_K) <= set(td.ALL_K)

    def case(K):
        blocks = [td.turbo_blocks(port, K, 66, kind, seed=3 * K + i) for i, kind in enumerate(("clean", "awgn0.8", "hard127"))]
        return K, blocks, [td.oracle_turbo_ref(port, soft, K) for _, soft in blocks]

    bad = []
    for c0 in range(0, len(td.ALL_K), 32):  # the oracle side of 32 sizes at a time on every host core, then those sizes on the GPU
        for K, blocks, want in td.parallel_map(case, td.ALL_K[c0:c0 + 32]):
            for (tx, soft), w, kind in zip(blocks, want, ("clean", "awgn0.8", "hard127")):
                got = ctx.turbo_decode(soft, K)
                nb = int((got != w).any(axis=1).sum())
                if nb:
                    bad.append((K, kind, nb))
                if kind == "clean" and K not in td.OVERFLOW_K:
                    assert (got == tx).all(), K  # noise-free loopback decodes wherever the interleaver is a permutation
    assert not bad, "block sizes with code blocks differing from the oracle (K, input kind, blocks): %s" % bad[:20]


def test_turbo_ref_single_block_and_exact_tile(ctx, port, siso):
    for n in (1, 3, 8, 9, 64, 128):  # (the state-parallel kernel takes 8 / 4 code blocks per wavefront)
        tx, soft = td.turbo_blocks(port, 256, n, "awgn0.5", seed=n)
        assert (ctx.turbo_decode(soft, 256) == td.oracle_turbo_ref(port, soft, 256)).all()


@pytest.mark.parametrize("n", [2100, 4500, 9000])
def test_turbo_ref_state_parallel_kernel_with_several_trellises_per_wavefront(ctx, port, n):
    """The state-parallel trellis kernel takes 1, 2, 4 or 8 trellises per wavefront depending on the size of the decode (turbo.hip,
    gpw_of): 2100 / 4500 / 9000 code blocks reach the three larger settings in both of its launches.  Against the lock-step kernel on every
    block and against the oracle on the unique ones."""
    K, uniq = 104, 60
    tx, soft = td.turbo_blocks(port, K, uniq, "awgn0.8", seed=n)
    want = td.oracle_turbo_ref(port, soft, K)
    idx = (np.arange(n) * 11 + np.arange(n) // 64) % uniq
    big = np.ascontiguousarray(soft[idx])
    try:
        ctx.set_turbo_small_batch(0)
        lock = ctx.turbo_decode(big, K)
        ctx.set_turbo_small_batch(1 << 30)
        par = ctx.turbo_decode(big, K)
    finally:
        ctx.set_turbo_small_batch(4096)
    assert (lock == want[idx]).all() and (par == lock).all()


@pytest.mark.parametrize("n", [4096, 4097])
def test_turbo_ref_at_the_kernel_switch(ctx, port, n):
    """With the default setting a decode of 4096 code blocks takes the state-parallel trellis kernel and one of 4097 the lock-step one:
    both sides of the switch against the oracle (unique blocks replicated over the batch)."""
    K, uniq = 40, 50
    tx, soft = td.turbo_blocks(port, K, uniq, "awgn0.8", seed=n)
    want = td.oracle_turbo_ref(port, soft, K)
    idx = (np.arange(n) * 13 + np.arange(n) // 64) % uniq
    got = ctx.turbo_decode(np.ascontiguousarray(soft[idx]), K)
    assert (got == want[idx]).all()


def test_turbo_ref_full_batch_property(ctx, port):
    """BASELINE config 3 shape: K=6144, 65536 blocks.  Oracle-check 32 unique blocks, and require
    every replica (placed in a different tile lane) to decode to the same bits."""
    import openlte_amd as m
    K, uniq, n_cb = 6144, 32, 65536
    tx, soft = td.turbo_blocks(port, K, uniq, "hard127", seed=7)
    want = td.oracle_turbo_ref(port, soft, K)
    idx = (np.arange(n_cb) * 7 + np.arange(n_cb) // 64) % uniq
    big = soft[idx]
    d_in = ctx.to_device(big)
    d_out = ctx.alloc(n_cb * K)
    ctx.turbo_decode_dev(d_in, m.SOFT_I8, K, n_cb, d_out)
    got = d_out.download(np.uint8).reshape(n_cb, K)
    d_in.free(); d_out.free()
    assert (got == want[idx]).all()


def test_rejects_bad_args(ctx):
    import openlte_amd as m
    with pytest.raises(m.MiLteError):
        ctx.turbo_decode(np.zeros((1, 3 * (41 + 4)), np.float32), 41)  # 41 is not an LTE block size


def test_turbo_ref_matches_golden_from_reference(ctx):
    """Fixtures generated by the reference itself (tools/gen_golden.py); no oracle involved."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "turbo_ref.npz"))
    keys = sorted({k.rsplit("_", 1)[0] for k in z.files})
    for key in keys:
        K = int(key.split("_")[0][1:])
        soft, want = z[key + "_soft"], z[key + "_bits"]
        got = ctx.turbo_decode(soft, K)
        assert (np.packbits(got, axis=-1) == want).all(), key
