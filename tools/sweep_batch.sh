for cfg in "8192 2" "16384 2" "32768 2" "32768 4" "65536 4"; do set -- $cfg
 python bench.py --units $1 --streams $2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 $2', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
    else: print(l.strip()[:200])
"
done
for cfg in "65536 2" "131072 2" "262144 4"; do set -- $cfg
 python bench.py --workload turbo --units $1 --streams $2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('turbo $1 $2', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
    else: print(l.strip()[:200])
"
done
