#!/bin/bash
# round 3, call 9: the one-call-per-subframe host form + explicit cache contract, the counting-sort re-plan (--strong), then the whole suite
cd /root/repo
mkdir -p gpurun_out/r03i
timeout 600 python -m pytest tests/test_dropin_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -12 | tee gpurun_out/r03i/pytest_new.txt
cp gpurun_out/scan_timing_cache_modes.txt gpurun_out/r03i/ 2>/dev/null
timeout 600 python bench.py --strong --gpus 2 --oversubscribe --steps 3 --warmup 1 --units 16384 > gpurun_out/r03i/bench_strong_2.json 2> gpurun_out/r03i/bench_strong_2.err; tail -c 600 gpurun_out/r03i/bench_strong_2.json
timeout 600 python tools/r3/subframe_call_timing.py 2>&1 | tail -12 | tee gpurun_out/r03i/subframe_call_timing.txt
SECONDS=0
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -6 | tee gpurun_out/r03i/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a gpurun_out/r03i/pytest_gpu.txt
