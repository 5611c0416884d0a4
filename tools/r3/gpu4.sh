#!/bin/bash
# round 3, call 4: pipeline tests again (capture split), the default bench line with the new config legs, the strong-scaling capture run
cd /root/repo
mkdir -p gpurun_out/r03d
timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r03d/pytest_pipeline.txt
SECONDS=0; timeout 900 python bench.py > gpurun_out/r03d/bench_default.json 2> gpurun_out/r03d/bench_default.err; echo "bench.py default run: $SECONDS s"; tail -3 gpurun_out/r03d/bench_default.err
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r03d/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric', 'value', 'ms_per_step', 'n_gpus')}, d['roofline']['frac'], d['cpu_baseline']['value'])
print({k: round(v['ms_per_step'], 3) for k, v in d.get('kernels', {}).items()})
print(d['stages'])
for k, v in d.get('other_configs', {}).items():
    print(k, v['value'], v['unit'], v['ms_per_step'], v['roofline']['kernel'], v['roofline']['frac'], v.get('cpu_baseline', {}).get('value'), v['kernels_ms_per_step'])
print(d.get('from_host_buffers'))
PY
for g in 1 2; do timeout 600 python bench.py --strong --gpus $g --oversubscribe --steps 3 --warmup 1 --units 16384 2>&1 | tail -1 | tee gpurun_out/r03d/bench_strong_$g.json; done
