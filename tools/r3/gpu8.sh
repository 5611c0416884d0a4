#!/bin/bash
# round 3, call 8: vote reads C1 from perm -- parity, then per-kernel times against the previous library (variant OLDVOTE), base first and last
cd /root/repo
mkdir -p gpurun_out/r03h
timeout 900 python -m pytest tests/test_turbo_gpu.py tests/test_chain_gpu.py tests/test_fuzz_gpu.py tests/test_uplink_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r03h/pytest.txt
AB_TIMEOUT=150 bash tools/ab/run_variants.sh chain --steps 10 --warmup 2 2>&1 | tee gpurun_out/r03h/variants_vote_c1.txt
