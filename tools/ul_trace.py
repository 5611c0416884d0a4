#!/usr/bin/env python3
"""Kernel and HIP API time of the per-call uplink loop (dropin_ul_gpu with UL_DEMO_REPEAT) under rocprofv3."""
import csv, glob, os, subprocess, sys, tempfile, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_dropin_gpu as t
reps = 100
with tempfile.TemporaryDirectory() as d:
    args = t._ul_demo_args(d)
    env = dict(os.environ, UL_DEMO_REPEAT=str(reps), TMPDIR="/tmp")
    out = os.path.join(d, "tr")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--hip-trace", "--output-format", "csv", "-d", out, "-o", "t", "--",
                        os.path.join(root, "shim", "_build", "dropin_ul_gpu")] + args, capture_output=True, text=True, env=env, timeout=600, cwd="/tmp")
    print([l for l in r.stderr.splitlines() if l.startswith("timing")])
    def rows(pat):
        f = glob.glob(out + "/**/" + pat, recursive=True)
        return list(csv.DictReader(open(f[0]))) if f else []
    k, a = rows("*kernel_trace.csv"), rows("*hip_api_trace.csv")
    agg = collections.defaultdict(lambda: [0, 0])
    for x in k:
        n = x["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        agg[n][0] += 1; agg[n][1] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-24s %6.2f per subframe, avg %7.1f us, %7.1f us per subframe" % (n, v[0] / reps, v[1] / v[0] / 1e3, v[1] / reps / 1e3))
    ag = collections.defaultdict(lambda: [0, 0])
    for x in a:
        ag[x["Function"]][0] += 1; ag[x["Function"]][1] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
    for n, v in sorted(ag.items(), key=lambda kv: -kv[1][1])[:8]:
        print("  API %-24s %6.1f per subframe, %7.1f us per subframe (traced)" % (n, v[0] / reps, v[1] / reps / 1e3))
