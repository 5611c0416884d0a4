#!/bin/bash
# the default line on the shipped defaults (one stream) + the chain test file
cd /root/repo
o=gpurun_out/r04w; mkdir -p $o; rm -f $o/bench_kernels.txt
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -3 | tee $o/pytest_chain.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; tail -c 200 $o/bench_default.json; tail -3 $o/bench_default.err
for i in 1 2; do timeout 300 python tools/ab/bench_kernels.py chain --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
