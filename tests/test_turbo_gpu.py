"""GPU parity: REF-mode turbo decode (HIP) vs the oracle, bit-exact (integer path)."""
import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu

# small / not multiple of 64 / power of two / W4 sizes / largest overflow-free / uint32-overflow sizes
KS = [40, 104, 512, 1088, 3264, 3584, 6016, 6144]


@pytest.mark.parametrize("K", KS)
@pytest.mark.parametrize("kind", ["clean", "awgn0.5", "awgn0.8", "hard127", "int", "i16"])
def test_turbo_ref_bit_exact(ctx, port, K, kind):
    n = 70 if K <= 1088 else 66  # more than one tile, last tile ragged
    tx, soft = td.turbo_blocks(port, K, n, kind, seed=1000 + K)
    want = td.oracle_turbo_ref(port, soft, K)
    got = ctx.turbo_decode(soft, K)
    assert got.shape == want.shape
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "blocks differing from the oracle: %s" % bad[:10]
    if kind == "clean" and K not in (3584, 6144):
        assert (got == tx).all()  # noise-free loopback decodes (overflow-free K)


def test_turbo_ref_single_block_and_exact_tile(ctx, port):
    for n in (1, 64, 128):
        tx, soft = td.turbo_blocks(port, 256, n, "awgn0.5", seed=n)
        assert (ctx.turbo_decode(soft, 256) == td.oracle_turbo_ref(port, soft, 256)).all()


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
