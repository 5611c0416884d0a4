#!/bin/bash
cd /root/repo
o=gpurun_out/r04z; mkdir -p $o; rm -f $o/*.txt
timeout 900 python -m pytest tests/test_bcjr_gpu.py tests/test_chain_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee $o/pytest_bcjr.txt
for i in 1 2; do timeout 300 python tools/ab/bench_kernels.py chain --decoder bcjr_early --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_chain_bcjr.txt; done
timeout 400 python tools/ab/bench_kernels.py chain --decoder bcjr --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_chain_bcjr.txt
