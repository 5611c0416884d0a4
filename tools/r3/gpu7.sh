#!/bin/bash
# round 3, call 7: the suite as the driver runs it, smoke(), and the default bench line as the driver runs it
cd /root/repo
mkdir -p gpurun_out/r03g
SECONDS=0
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -16 | tee gpurun_out/r03g/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a gpurun_out/r03g/pytest_gpu.txt
cp gpurun_out/fuzz_report.json gpurun_out/r03g/
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r03g/smoke.txt
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03g/bench_default.json 2> gpurun_out/r03g/bench_default.err; echo "bench.py default run: $SECONDS s"
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r03g/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric', 'value', 'ms_per_step', 'n_gpus')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'])
print({k: round(v['ms_per_step'], 3) for k, v in d.get('kernels', {}).items()})
print({k: (v['ms_per_step'], v['frac_of_peak'], v.get('hbm_traffic_over_algorithmic')) for k, v in d['stages'].items()})
for k, v in d.get('other_configs', {}).items():
    print(k, v['value'], v['roofline']['kernel'], v['roofline']['frac'], v['roofline']['stage_frac'], v.get('cpu_baseline', {}).get('value'))
for k, v in d['turbo_decode'].items():
    if isinstance(v, dict): print(k, v['mbit_per_s'], v['ms_per_decode'], v.get('iterations_per_tile_pair'))
print(d.get('from_host_buffers'))
PY
