#!/usr/bin/env python3
"""tools/sq_json.py <tools/pmc_sq.sh text output> <workload> <bench json of the same run> [steps in that run = 3]:
profiles/sq_counters_<workload>.json -- per kernel and STEP of the workload the summed SQ counters (wave64 vector instructions issued, wavefronts,
busy cycles), stamped with the library's build id (the bench line of the profiled run carries it).  bench.py turns SQ_INSTS_VALU into the
`roofline.valu` object: the vector ALUs' issue time per step next to the HBM figure."""
import ast, collections, json, os, re, sys
src, workload, bench = sys.argv[1], sys.argv[2], sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for l in open(src):
    m = re.match(r"\('(k_\w+)', '(\d+)'\) launches (\d+) (\{.*\})", l.strip())
    if not m:
        continue
    n = int(m.group(3))
    for c, v in ast.literal_eval(m.group(4)).items():
        acc[(m.group(1), m.group(2))][c] = v * n  # per-launch average x launches (a counter that two passes collected: the later pass's value stands)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for (k, grid), d in acc.items():
    if k.startswith(("k_copy", "k_done", "k_rm_rank")):  # set-up and the copy-rate measurement: outside the timed steps
        continue
    for c, v in d.items():
        per[k][c] += v / steps
bid, units = None, None
for l in open(bench):
    if l.startswith("{"):
        d = json.loads(l)
        bid, units = d.get("build_id"), d.get("config", {}).get("units_per_gpu_per_step")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"workload": workload, "build_id": bid, "steps_profiled": steps, "units_per_step": units,
       "note": "rocprofv3 --pmc SQ_* passes of bench.py (tools/pmc_sq.sh), summed over each kernel's launches and divided by the steps of the profiled run",
       "per_step": {k: {c: int(v) for c, v in d.items()} for k, d in sorted(per.items())}}
with open(os.path.join(root, "profiles", "sq_counters_%s.json" % workload), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out)[:400])
