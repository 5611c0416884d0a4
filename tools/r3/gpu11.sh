#!/bin/bash
# round 3, call 11: the copy-rate denominator in the default line; new argument test
cd /root/repo
mkdir -p gpurun_out/r03k
timeout 300 python -m pytest tests/test_args_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03k/bench_default.json 2> gpurun_out/r03k/bench_default.err; echo "bench.py default run: $SECONDS s"; tail -3 gpurun_out/r03k/bench_default.err
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r03k/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, {k: d['roofline'][k] for k in ('frac', 'measured_copy_GBps', 'frac_of_measured_copy')})
print({k: v.get('hbm_traffic_frac_of_measured_copy') for k, v in d['kernels'].items()})
for k, v in d.get('other_configs', {}).items():
    print(k, v['value'], v['roofline']['frac'], v['roofline'].get('frac_of_measured_copy'))
PY
