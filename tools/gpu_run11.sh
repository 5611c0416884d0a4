#!/bin/bash
# full GPU suite + the default bench line + smoke, after the day's kernel changes
cd /root/repo
mkdir -p gpurun_out/r02j
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee gpurun_out/r02j/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "^ERROR: DCI" | tail -2
timeout 600 python bench.py > gpurun_out/r02j/bench_chain.json 2> gpurun_out/r02j/bench_chain.err; tail -c 1500 gpurun_out/r02j/bench_chain.json | head -c 600; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02j/bench_chain.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
print({k:v['ms_per_step'] for k,v in d['kernels'].items()})
print(d.get('turbo_decode')); print(d.get('from_host_buffers') or d.get('extra',{}).get('from_host_buffers'))
PY
