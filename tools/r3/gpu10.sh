#!/bin/bash
# round 3, call 10: the default bench line with the cpu_baseline legs on the single-precision FFT stand-in; the strong run after the counting-sort re-plan
cd /root/repo
mkdir -p gpurun_out/r03j
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03j/bench_default.json 2> gpurun_out/r03j/bench_default.err; echo "bench.py default run: $SECONDS s"
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r03j/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric', 'value', 'ms_per_step', 'n_gpus')}, d['roofline']['frac'])
print(d['cpu_baseline']); print(d['cpu_baseline_all_cores'])
for k, v in d.get('other_configs', {}).items():
    print(k, v['value'], v['roofline']['frac'], v.get('cpu_baseline'))
PY
for g in 1 2; do timeout 600 python bench.py --strong --gpus $g --oversubscribe --steps 3 --warmup 1 --units 16384 2>&1 | tail -1 > gpurun_out/r03j/bench_strong_$g.json; python -c "
import json; d=json.loads(open('/root/repo/gpurun_out/r03j/bench_strong_$g.json').read().strip().splitlines()[-1]); print($g, d['value'], d['ms_per_step'])"; done
timeout 300 python -m pytest tests/test_oracle.py -q -x -p no:cacheprovider 2>&1 | tail -2
