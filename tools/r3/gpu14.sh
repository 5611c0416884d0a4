#!/bin/bash
# round 3, call 14: PBCH / PRACH fuzz against the compiled reference, PRACH argument refusals, pipeline tests after the error-path change
cd /root/repo
mkdir -p gpurun_out/r03p
timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x -k "pbch or prach" --durations=4 -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/r03p/pytest_fuzz_next_rows.txt
timeout 600 python -m pytest tests/test_prach_gpu.py tests/test_pipeline_gpu.py tests/test_args_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
cp gpurun_out/fuzz_report.json gpurun_out/r03p/ 2>/dev/null
