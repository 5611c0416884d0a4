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


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it must become two ranks (torch.distributed.run, 127.0.0.1) and report
    n_gpus = 2 only because two ranks really ran: the selftest workload exercises launch, sharding, barrier and the max-over-ranks
    timing on CPU (gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "selftest", "--steps", "3", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints the one JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["shard_counts"] == [500, 500]
    assert sorted(x[0] for x in out["ranks"]) == [0, 1] and sorted(x[1] for x in out["ranks"]) == [0, 1]
    assert {tuple(x[3]) for x in out["ranks"]} == {(0, 2, 4), (1, 3, 5)}


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "selftest"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in (r.stdout + r.stderr)


def test_bench_gpus_8_selftest_on_cpu():
    """The same at the width the scaling run uses: eight ranks (gloo, 127.0.0.1), every unit owned by exactly one of them, one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "selftest", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["shard_counts"] == [125] * 8 and out["scaling"] == "weak"
    assert sorted(x[0] for x in out["ranks"]) == list(range(8)) and sorted(x[1] for x in out["ranks"]) == list(range(8))
    assert {tuple(x[3]) for x in out["ranks"]} == {(k, k + 8, k + 16) for k in range(8)}


def test_bench_gpus_8_chain_setup_on_cpu():
    """The host side of the CHAIN workload's set-up at the width and size the scaling run uses (`--gpus 8`, 65 536 subframes per rank), no
    GPU: eight ranks synthesise their 96 unique subframes (rank-dependent cells and seeds) and build their 589 824 allocation
    descriptors; the 4.6 GB batch per rank is laid down in HBM by repeated uploads and never exists on the host.  Must stay a matter of
    seconds and of well under 2 GB per rank (it was 29 s and 5 GB per rank when the batch was built on the host first)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "chain-setup"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["subframes_per_rank"] == 65536
    assert out["slowest_rank_s"] < 60 and out["peak_resident_GB_per_rank"] < 2.0, out
    assert out["batch_bytes_in_hbm_per_rank"] == 65536 * 70240 and out["descriptor_bytes_per_rank"] == 65536 * 9 * 260
