#!/bin/bash
# round 4, call 32: whole GPU suite + smoke + default bench line + BCJR lines and profiles on the round's last kernels
cd /root/repo
o=gpurun_out/r04w; mkdir -p $o; rm -f $o/bench_kernels.txt
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -6 | tee $o/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a $o/pytest_gpu.txt
cp gpurun_out/fuzz_report.json $o/fuzz_report.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^ERROR: DCI" | tail -3 | tee $o/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; tail -c 200 $o/bench_default.json; tail -3 $o/bench_default.err
for i in 1 2; do timeout 300 python tools/ab/bench_kernels.py chain --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
TAG=r04
for w in "turbo_bcjr --workload turbo --decoder bcjr" "turbo_bcjr_early --workload turbo --decoder bcjr_early"; do
  set -- $w; t=$1; shift
  rm -rf gpurun_out/prof_${TAG}_$t
  timeout 600 bash tools/profile_bench.sh ${TAG}_$t "$@" > /dev/null 2>&1
done
for d in bcjr bcjr_early; do timeout 200 python bench.py --workload turbo --decoder $d --no-cpu-baseline > gpurun_out/r04_bench_turbo_$d.json 2>/dev/null; done
for d in bcjr bcjr_early; do timeout 300 python bench.py --workload chain --decoder $d --no-cpu-baseline --no-host-leg > gpurun_out/r04_bench_chain_$d.json 2>/dev/null; done
