#!/bin/bash
# round 3, call 5: BCJR early termination (test + numbers), the chain against the round-2 library on the same box
cd /root/repo
mkdir -p gpurun_out/r03e
timeout 600 python -m pytest tests/test_bcjr_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r03e/pytest_bcjr.txt
for d in bcjr bcjr_early; do
  timeout 300 python bench.py --workload turbo --decoder $d --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03e/bench_turbo_$d.json
  timeout 300 python bench.py --decoder $d --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-turbo-leg 2>&1 | tail -1 > gpurun_out/r03e/bench_chain_$d.json
done
python - <<'PY'
import json
for n in ("turbo_bcjr", "turbo_bcjr_early", "chain_bcjr", "chain_bcjr_early"):
    d = json.loads(open('/root/repo/gpurun_out/r03e/bench_%s.json' % n).read().strip().splitlines()[-1])
    print(n, d['value'], d['unit'], d['ms_per_step'], {k: round(v['ms_per_step'], 3) for k, v in d.get('kernels', {}).items()}, d.get('crc_pass'), d.get('sampled_blocks_equal_tx_bits'))
PY
AB_TIMEOUT=150 bash tools/ab/run_variants.sh chain --steps 10 --warmup 2 2>&1 | tee gpurun_out/r03e/variants_chain_r02_vs_head.txt
