"""Worker for tests/test_dist_cpu.py: the control plane bench.py uses for N > 1 (gloo on CPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from openlte_amd.sharding import shard_units  # noqa: E402

rank, world, barrier, max_reduce = bench.dist_setup(int(os.environ["WORLD_SIZE"]))
barrier()
mine = list(shard_units(1000, rank, world))
t = max_reduce(1.0 + rank)  # slowest rank defines the step time
barrier()
with open(os.path.join(sys.argv[1], "rank%d.json" % rank), "w") as f:
    json.dump({"rank": rank, "world": world, "n": len(mine), "first": mine[:3], "t": t}, f)
