#!/bin/bash
# round 4, call 26: whole GPU suite + smoke + the default bench line with the 8 x 16 x 16 transform
cd /root/repo
o=gpurun_out/r04w; mkdir -p $o
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -6 | tee $o/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a $o/pytest_gpu.txt
cp gpurun_out/fuzz_report.json $o/fuzz_report.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^ERROR: DCI" | tail -3 | tee $o/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; tail -c 300 $o/bench_default.json; tail -3 $o/bench_default.err
for i in 1 2; do timeout 300 python tools/ab/bench_kernels.py chain --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
timeout 120 tools/ubench/store_rate 2>&1 | tee $o/store_rate.txt
