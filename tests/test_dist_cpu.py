"""CPU test of the N > 1 path: two processes, gloo, 127.0.0.1 -- barrier, max-over-ranks timing and
the block-cyclic unit sharding (there is no data-path collective to test: the path shards by unit)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_control_plane(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    assert [o["world"] for o in out] == [2, 2]
    assert out[0]["n"] + out[1]["n"] == 1000 and out[0]["first"] == [0, 2, 4] and out[1]["first"] == [1, 3, 5]
    assert out[0]["t"] == out[1]["t"] == 2.0  # max over ranks


def test_sharding_covers_every_unit_once():
    from openlte_amd.sharding import shard_units, shard_counts
    for n, w in ((10, 1), (10, 3), (8192, 8), (7, 8)):
        seen = sorted(u for r in range(w) for u in shard_units(n, r, w))
        assert seen == list(range(n))
        assert sum(shard_counts(n, w)) == n
