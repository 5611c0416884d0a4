#!/bin/bash
# what the GPU's clocks and power are while the chain bench runs (is the chip power-limited?): samples rocm-smi once a second
cd "$(dirname "$0")/.."
(timeout 120 python bench.py --workload chain --steps 60 --no-turbo-leg --no-host-leg --no-cpu-baseline --no-kernel-events > /tmp/b.json 2>/dev/null) &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showuse -d 0 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ';'; echo
  sleep 1
done
wait $BP
tail -1 /tmp/b.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
