#!/usr/bin/env python3
"""tools/mixed_soak.py [first seed] [n seeds] [subframes per seed]: the merged decode (several code-block sizes in one launch set, turbo.hip: KSeg)
against the compiled reference over further seeds of the mixed-traffic draw -- tests/test_mixed_gpu.py's first test in a loop, both shapes of the
trellis kernel.  Every allocation's verdict and decoded bits (also of the blocks whose CRC fails) must be the reference's, from the reference's own
received grid.  One line per seed, a total at the end; exit code 1 on any difference."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import openlte_amd as m
from oracle import pyoracle as po
import test_mixed_gpu as T

first, n_seeds, n_sf = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 20), (int(sys.argv[3]) if len(sys.argv) > 3 else 32)
ref = po.ref()
assert ref is not None, "oracle/_ref not built"
ctx = m.Context(0)
cfg = m.DlCfg(2048, 100, 1, m.IQ_I8)
tot = bad = sizes = 0
for seed in range(first, first + n_seeds):
    lists = T._lists(n_sf, seed)
    iq, tx, allocs = T._synth(m, cfg, lists, 7000 + seed)
    grids, want = T._ref_decode_all(ref, po, lists, iq)
    d_sub = ctx.to_device(np.concatenate(grids))
    plan = ctx.pdsch_plan(cfg, 2, allocs)
    diffs = 0
    for small in (0, 4096):  # lock-step trellis kernel / state-parallel one
        ctx.set_turbo_small_batch(small)
        st, bits = plan.run(d_sub, [l[0] for l in lists], [l[1] for l in lists])
        assert "over all block sizes" in ctx.last_kernels()
        diffs += sum(int(st[a] != rc or not (bits[a] == b).all()) for a, (rc, b) in enumerate(want))
    ctx.set_turbo_small_batch(4096)
    plan.close(); d_sub.free()
    k = len({a.tbs for a in allocs})
    ok = sum(int(rc == 0) for rc, b in want)
    print("seed %d: %d allocations, %d block sizes, %d decoded by the reference, differing (either kernel shape): %d" % (seed, len(allocs), k, ok, diffs), flush=True)
    tot += len(allocs); bad += diffs; sizes = max(sizes, k)
print("total: %d allocations over %d seeds, up to %d block sizes per batch, %d differ from the compiled reference" % (tot, n_seeds, sizes, bad))
sys.exit(1 if bad else 0)
