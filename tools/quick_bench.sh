#!/bin/bash
# quick GPU check: parity tests + saturated single-stream chain/turbo benches (per-kernel ms)
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
show() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()}, d.get('crc_pass'))
    else: print(l.strip()[:300])
"; }
python bench.py --units 32768 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | show chain1s
python bench.py --units 32768 --streams 2 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | show chain2s
python bench.py --workload turbo --units 131072 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | show turbo1s
python bench.py --workload frontend --units 32768 --streams 1 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | show fe1s
