#!/bin/bash
# round 3, call 12: copy-rate shapes and sizes on this box; default bench line with them
cd /root/repo
mkdir -p gpurun_out/r03l
python - <<'PY' 2>&1 | tee gpurun_out/r03l/copy_rates.txt
import openlte_amd as m
ctx = m.Context(0)
for mb in (64, 256, 1024, 4096):
    r = ctx.device_copy_rate(mb << 20, 10)
    print("%5d MiB: best %.1f GB/s  shapes (grid-stride, 4 in flight, 4 in flight nt) %s" % (mb, r, ctx.device_copy_rates()))
PY
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03l/bench_default.json 2> gpurun_out/r03l/bench_default.err; echo "bench.py default run: $SECONDS s"; tail -3 gpurun_out/r03l/bench_default.err
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r03l/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, {k: d['roofline'][k] for k in ('frac', 'measured_copy_GBps', 'frac_of_measured_copy', 'measured_copy_shapes_GBps')})
print({k: v.get('hbm_traffic_frac_of_measured_copy') for k, v in d['kernels'].items()})
PY
