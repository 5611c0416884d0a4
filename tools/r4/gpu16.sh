#!/bin/bash
cd /root/repo
for d in bcjr; do MI_LTE_BCJR_DEBUG_NO_BARRIER=1 timeout 200 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | sed 's/^/no barrier (timing only): /'; done
for d in bcjr; do timeout 200 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1; done
