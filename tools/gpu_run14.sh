#!/bin/bash
# end of round 2 (after the per-call work): the full GPU suite, smoke(), and the default bench line as the driver runs it
cd /root/repo
mkdir -p gpurun_out/r02k
echo "(suite: see the previous call)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r02k/smoke.txt
SECONDS=0; timeout 900 python bench.py > gpurun_out/r02k/bench_default.json 2> gpurun_out/r02k/bench_default.err; echo "bench.py default run: $SECONDS s"

python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r02k/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric', 'value', 'ms_per_step', 'n_gpus')}, d['roofline']['frac'], d['cpu_baseline']['value'])
print({k: round(v['ms_per_step'], 3) for k, v in d.get('kernels', {}).items()})
PY
