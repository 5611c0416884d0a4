#!/bin/bash
# round 3, call 13: the suite and the default line on the tree with the copy-rate denominator (final kernel shapes)
cd /root/repo
mkdir -p gpurun_out/r03n
SECONDS=0
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee gpurun_out/r03n/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a gpurun_out/r03n/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r03n/smoke.txt
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03n/bench_default.json 2> gpurun_out/r03n/bench_default.err; echo "bench.py default run: $SECONDS s"; tail -3 gpurun_out/r03n/bench_default.err
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r03n/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, {k: d['roofline'][k] for k in ('frac', 'measured_copy_GBps', 'frac_of_measured_copy', 'measured_copy_shapes_GBps')})
print({k: v.get('hbm_traffic_frac_of_measured_copy') for k, v in d['kernels'].items()})
for k, v in d.get('other_configs', {}).items():
    print(k, v['value'], v['roofline']['frac'], v['roofline'].get('frac_of_measured_copy'))
PY
