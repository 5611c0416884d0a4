#!/bin/bash
# round 4, call 1: the VALU issue-rate micro-benchmarks (review item 1) + per-kernel baseline of this round's box
cd /root/repo
o=gpurun_out/r04a; mkdir -p $o
timeout 300 tools/ubench/issue_rate 2>&1 | tee $o/ubench_issue_rate.txt
for b in valu_rate pk_rate cmp_rate; do echo "== $b"; timeout 120 tools/ubench/$b 2>&1; done | tee $o/ubench_r3_programs.txt
for w in chain turbo frontend uplink; do
timeout 300 python tools/ab/bench_kernels.py $w --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt
done
timeout 200 python tools/ab/bench_kernels.py turbo --decoder bcjr --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt
