#!/bin/bash
cd /root/repo
o=gpurun_out/r04n; mkdir -p $o
timeout 200 python tools/ab/bench_kernels.py turbo --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt
for d in bcjr bcjr_early; do MI_LTE_BCJR_LAUNCH_PER_HALF=1 timeout 200 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | sed 's/^/launch-per-half: /' | tee -a $o/bench_kernels.txt; done
for d in bcjr bcjr_early; do timeout 200 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
rocm-smi --showclocks 2>/dev/null | head -20
