#!/bin/bash
# BCJR: no fill / no read of the first half-iteration's a-priori values, half of the boundary fill
cd /root/repo
o=gpurun_out/r04z; mkdir -p $o; rm -f $o/*.txt
timeout 900 python -m pytest tests/test_bcjr_gpu.py tests/test_chain_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee $o/pytest_bcjr.txt
for d in bcjr_early bcjr bcjr_early bcjr; do timeout 200 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_turbo_bcjr.txt; done
timeout 300 python tools/ab/bench_kernels.py chain --decoder bcjr_early --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_turbo_bcjr.txt
