#!/usr/bin/env python3
"""One line per run: workload value, ms per step, and the per-kernel milliseconds bench.py measures with HIP events."""
import json
import os
import subprocess
import sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", sys.argv[1], "--no-turbo-leg", "--no-host-leg"] + sys.argv[2:], capture_output=True, text=True, cwd=root).stdout.strip().splitlines()[-1]
d = json.loads(out)
print(sys.argv[1], d["value"], d["unit"], "%.3f ms/step" % d["ms_per_step"], " ".join("%s %.3f" % (k, v["ms_per_step"]) for k, v in sorted(d.get("kernels", {}).items())))
