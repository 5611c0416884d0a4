show() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['value'], d['ms_per_step'])
    else: print(l.strip()[:200])
"; }
for cfg in "8192 1" "8192 2" "16384 1" "16384 2" "32768 1" "65536 1" "65536 2"; do set -- $cfg
 python bench.py --units $1 --streams $2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | show "chain $1 $2"
done
for cfg in "65536 1" "65536 2" "65536 4" "131072 1" "262144 1"; do set -- $cfg
 python bench.py --workload turbo --units $1 --streams $2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | show "turbo $1 $2"
done
for cfg in "10000 1" "10000 2" "32768 1"; do set -- $cfg
 python bench.py --workload frontend --units $1 --streams $2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | show "fe $1 $2"
done
